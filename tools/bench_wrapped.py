"""Throughput of the reference's DEFAULT env (make_env(): wrapper stack, randomize=True) next to the unwrapped env that bench.py
measures: the wrapper stack is ~150 small [B, ...] tensor kernels per step around the two fused launches.
    python tools/bench_wrapped.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.locked import make_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
for label, kw in (("make_env(randomize=False)", dict(constants={"randomize": False})), ("make_env()  [randomize=True]", dict()), ("make_simple_env", dict(apply_wrappers=False))):
    env = make_env(batch_size=B, device="cuda:0", starting_seed=1, **kw)
    env.reset()
    act = (lambda: torch.randint(0, 11, (B, 20), generator=gen, device="cuda:0")) if kw.get("apply_wrappers", True) else (lambda: torch.rand((B, 20), generator=gen, device="cuda:0") * 2 - 1)
    for _ in range(5):
        env.step(act())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        env.step(act())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-30s B=%d: %.0f env-steps/s, %.2f ms per step" % (label, B, B * steps / dt, 1e3 * dt / steps))
    del env
