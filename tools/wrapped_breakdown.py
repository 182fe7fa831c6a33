"""Where the default make_env() (randomize=True) loses against the unwrapped env: (a) the per-env parameter rows alone (unwrapped env,
rows = the model's own values), (b) the wrapper stack's tensor kernels (time of the two fused launches inside a wrapped step, by events),
(c) the randomized physics itself (same launches, randomized rows).
    python tools/wrapped_breakdown.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import make_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)


def run(label, env, act, inner=None):
    for _ in range(5):
        env.step(act())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    sim = inner.mujoco_simulation
    orig = sim.env_step
    cur = [None]

    def timed(*a, **k):
        cur[0][0].record(); r = orig(*a, **k); cur[0][1].record(); return r
    sim.env_step = timed
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in ev:
        cur[0] = p
        env.step(act())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sim.env_step = orig
    phys = sum(a.elapsed_time(b) for a, b in ev) / steps
    print("%-44s %.2f ms per step, physics launches %.2f ms, everything else %.2f ms" % (label, 1e3 * dt / steps, phys, 1e3 * dt / steps - phys))


cont = lambda: torch.rand((B, 20), generator=gen, device="cuda:0") * 2 - 1
disc = lambda: torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0")
env = make_env(batch_size=B, device="cuda:0", starting_seed=1, apply_wrappers=False); env.reset()
run("make_simple_env", env, cont, env)
env.mujoco_simulation.params          # switches the per-env parameter rows on (values = the model's)
run("make_simple_env + per-env parameter rows", env, cont, env)
del env
for label, kw in (("make_env(randomize=False)", dict(constants={"randomize": False})), ("make_env() [randomize=True]", dict())):
    env = make_env(batch_size=B, device="cuda:0", starting_seed=1, **kw); env.reset()
    run(label, env, disc, env.env)
    del env
