"""dactyl/full_perpendicular with the reference's wrapper stack (make_env(constants={"randomize": False})): env-steps/s at B = 4096 over 10 steps, synchronous and
with the wrappers' auto_reset around pipelined resets.      python tools/full_cube_wrapped_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
from robogym_amd.envs.dactyl.full_perpendicular import make_env
for pipe in (False, True):
    env = make_env(constants={"randomize": False}, batch_size=4096, starting_seed=3, pipelined_reset=pipe)
    obs = env.reset()
    gen = torch.Generator(device="cpu").manual_seed(0)
    acts = [torch.randint(0, 11, (4096, 20), generator=gen).to(env.device) for _ in range(12)]
    for a in acts[:2]:
        env.step(a)
    torch.cuda.synchronize(); t0 = time.time()
    for a in acts[2:]:
        obs, reward, done, info = env.step(a)
    torch.cuda.synchronize(); el = time.time() - t0
    print("pipelined_reset", pipe, "%.0f env-steps/s" % (4096 * 10 / el), "finite", bool(torch.isfinite(obs["relative_goal"]).all()), "status", int(env.unwrapped.mujoco_simulation.status.max()), "done", int(done.sum()))
    del env; torch.cuda.empty_cache()
