"""Steady state of the rearrange envs WITH episode ends: B envs x T env.steps of random actions with pipelined resets (an ended episode runs the reset
recipe -- 100 stabilisation steps, 10 steps of one random action, 100 zero-action steps -- inside the following step calls), goal time-outs shortened so that
every env goes through the recipe several times.  Reports env-steps/s (recipe steps included), episode counts, status bits.

    python tools/soak_rearrange.py [B] [T] [blocks|ycb] [max_timesteps_per_goal_per_obj]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.rearrange.blocks import BatchedBlockRearrangeEnv      # noqa: E402
from robogym_amd.envs.rearrange.ycb import BatchedYcbRearrangeEnv         # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ycb = len(sys.argv) > 3 and sys.argv[3] == "ycb"
per_obj = int(sys.argv[4]) if len(sys.argv) > 4 else 30
env = (BatchedYcbRearrangeEnv if ycb else BatchedBlockRearrangeEnv)(B, pipelined_reset=True, max_timesteps_per_goal_per_obj=per_obj, starting_seed=5)
env.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(9)
ended = started = 0
inside = 0.0
seen = torch.zeros(B, dtype=torch.int32, device=env.device)
bad = 0
torch.cuda.synchronize()
t0 = time.time()
for t in range(T):
    obs, reward, done, info = env.step(torch.rand((B, 6), generator=gen, device=env.device) * 2 - 1)
    ended += int(done.sum()); started += int(info["episode_started"].sum()); inside += float(info["resetting"].float().mean())
    seen |= env.sim.status.reshape(-1).to(torch.int32) | env.solver_sim.status.reshape(-1).to(torch.int32)
    bad += int((~torch.isfinite(env.packed).all(1)).sum())
torch.cuda.synchronize()
el = time.time() - t0
print("rearrange/%s, %d envs x %d steps with pipelined resets (goal time-out %d steps): %.1f s = %.0f env-steps/s (recipe steps included; %.1f %% of the env-steps were inside the recipe)"
      % ("ycb" if ycb else "blocks", B, T, per_obj * env.N, el, B * T / el, 100 * inside / T))
bits = {int(b): int(((seen & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 32, 64) if int(((seen & b) != 0).sum())}
print("episodes ended %d, started %d; envs with a status bit ever %d %s; non-finite observation rows %d" % (ended, started, int((seen != 0).sum()), bits, bad))
