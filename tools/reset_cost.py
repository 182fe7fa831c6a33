"""Cost of resets at B = 8192: synchronous masked `reset(mask)` vs pipelined resets (episodes of 60 steps)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import BatchedLockedEnv, LockedEnvConstants, make_simple_env  # noqa: E402

B = 8192; dev = torch.device("cuda:0")
env = make_simple_env(batch_size=B, device=dev, starting_seed=1); env.reset()
for n in (1, 80, 800, 8192):
    m = torch.zeros(B, dtype=torch.bool, device=dev); m[:n] = True
    torch.cuda.synchronize(); t = time.time(); env.reset(mask=m); torch.cuda.synchronize()
    print("synchronous masked reset of %4d envs: %.3f s" % (n, time.time() - t))
gen = torch.Generator(device=dev); gen.manual_seed(0)
for name, kw, cst in (("no resets (goal timeout 400)", {}, LockedEnvConstants()),
                      ("pipelined resets, goal timeout 30 steps", {"pipelined_reset": True}, LockedEnvConstants(max_timesteps_per_goal=30))):
    env = BatchedLockedEnv(B, device=dev, constants=cst, starting_seed=1, **kw); env.reset()
    for _ in range(5):
        env.step(torch.rand((B, 20), generator=gen, device=dev) * 2 - 1)
    torch.cuda.synchronize(); t = time.time(); nd = 0
    for _ in range(120):
        obs, r, done, info = env.step(torch.rand((B, 20), generator=gen, device=dev) * 2 - 1)
        nd = nd + done.sum()
    torch.cuda.synchronize(); el = time.time() - t
    print("%-42s 120 steps in %.2f s = %.0f env-steps/s (incl. recipe steps); %d episodes ended" % (name, el, 120 * B / el, int(nd)))
