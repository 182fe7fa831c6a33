"""Steady state of the rearrange envs WITH episode ends: B envs x T env.steps of random actions with pipelined resets (an ended episode runs the reset
recipe -- 100 stabilisation steps, 10 steps of one random action, 100 zero-action steps -- inside the following step calls), goal time-outs shortened so that
every env goes through the recipe several times.  Reports env-steps/s (recipe steps included), episode counts, status bits.

    python tools/soak_rearrange.py [B] [T] [blocks|ycb] [max_timesteps_per_goal_per_obj] [host|device] [tcp+roll+yaw|tcp+wrist|joint]

`device`: the recipe in ra_recipe_kernel (device_reset=True), counters accumulated on the device and read after the run -- no readback inside the loop."""
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
device_reset = len(sys.argv) > 5 and sys.argv[5] == "device"
mode = sys.argv[6] if len(sys.argv) > 6 else "tcp+roll+yaw"
env = (BatchedYcbRearrangeEnv if ycb else BatchedBlockRearrangeEnv)(B, pipelined_reset=True, max_timesteps_per_goal_per_obj=per_obj, starting_seed=5, device_reset=device_reset, control_mode=mode)
env.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(9)
ended = started = bad = torch.zeros((), device=env.device, dtype=torch.int64)
inside = torch.zeros((), device=env.device)
seen = torch.zeros(B, dtype=torch.int32, device=env.device)
torch.cuda.synchronize()
t0 = time.time()
for t in range(T):
    obs, reward, done, info = env.step(torch.rand((B, env.action_dim), generator=gen, device=env.device) * 2 - 1)
    ended = ended + done.sum(); started = started + info["episode_started"].sum(); inside = inside + info["resetting"].float().mean()
    seen |= env.sim.status.reshape(-1).to(torch.int32)
    if env.solver_sim is not None:
        seen |= env.solver_sim.status.reshape(-1).to(torch.int32)
    bad = bad + (~torch.isfinite(env.packed).all(1)).sum()
torch.cuda.synchronize()
el = time.time() - t0
ended, started, bad, inside = int(ended), int(started), int(bad), float(inside)
print("control_mode %s" % mode)
print("rearrange/%s, %d envs x %d steps with pipelined resets (goal time-out %d steps): %.1f s = %.0f env-steps/s (recipe steps included; %.1f %% of the env-steps were inside the recipe)"
      % ("ycb" if ycb else "blocks", B, T, per_obj * env.N, el, B * T / el, 100 * inside / T))
bits = {int(b): int(((seen & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 32, 64) if int(((seen & b) != 0).sum())}
if device_reset:
    print("device recipe: placements that ran out of trials %d" % int(env.placement_failed.sum()))
print("episodes ended %d, started %d; envs with a status bit ever %d %s; non-finite observation rows %d" % (ended, started, int((seen != 0).sum()), bits, bad))
