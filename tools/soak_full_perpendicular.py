"""Soak of BatchedFullPerpendicularEnv (dactyl/full_perpendicular, B envs, iid random actions): env.step for N steps, envs that report
`done` (dropped cube with stop_on_fall, 1600 steps without reaching the goal, crash) are reset with the reference's recipe every
RESET_EVERY steps; counts status bits, dones, goals reached.
    python tools/soak_full_perpendicular.py [B] [steps] [pipelined]
With `pipelined` the env restarts finished episodes by itself inside the step calls (no reset() in the loop)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.full_perpendicular import BatchedFullPerpendicularEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
RESET_EVERY = 25
PIPE = len(sys.argv) > 3 and sys.argv[3] == "pipelined"
env = BatchedFullPerpendicularEnv(B, device="cuda:0", starting_seed=3, pipelined_reset=PIPE)
env.stop_on_fall = True
env.constants.max_pose_resets = 4
t0 = time.perf_counter()
env.reset()
torch.cuda.synchronize()
print("reset of %d envs (recipe: 30 env.steps + scramble, up to 4 passes): %.1f s; on palm %.3f" % (
    B, time.perf_counter() - t0, float((env.sim.scratch("site_xpos")[:, 3 * env.sim.center_site + 2] > 0.04).float().mean())))
gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
pending = torch.zeros(B, dtype=torch.bool, device="cuda:0")
status_or = torch.zeros(B, dtype=torch.int32, device="cuda:0")
ndone = nreset = 0
succ = torch.zeros(B, dtype=torch.int64, device="cuda:0")
t0 = time.perf_counter()
for k in range(N):
    obs, rew, done, info = env.step(torch.rand((B, 20), generator=gen, device="cuda:0") * 2 - 1)
    status_or |= env.sim.status
    pending |= done
    ended = done.to(torch.int64) if k == 0 else ended + done.to(torch.int64)
    succ += info["sub_goal_is_successful"].to(torch.int64)
    if PIPE:
        started = info["episode_started"].to(torch.int64) if k == 0 else started + info["episode_started"].to(torch.int64)
    elif (k + 1) % RESET_EVERY == 0:
        n = int(pending.sum())
        if n:
            ndone += n
            env.reset(pending)
            nreset += 1
            pending.zero_()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bits = int(torch.bitwise_or(status_or, torch.zeros_like(status_or)).max())
allbits = 0
for v in torch.unique(status_or).tolist():
    allbits |= int(v)
if PIPE:
    print("pipelined resets: episodes ended %d, started %d, envs inside the recipe now %d" % (int(ended.sum()), int(started.sum()), int(env._flags["resetting"].sum())))
print("%d env.steps x %d envs in %.1f s (%.0f env-steps/s incl. %d masked resets of %d envs in total); status bits seen: 0x%x; goals reached %d; max |qpos| %.2e, NaN %d" % (
    N, B, dt, N * B / dt, nreset, ndone, allbits, int(succ.sum()), float(env.sim.qpos.abs().max()), int(torch.isnan(env.sim.qpos).sum())))
