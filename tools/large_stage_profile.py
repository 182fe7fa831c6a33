"""Stage cycle counters of the large-model stepper (rb_step_kernel, flags bit 1): mean cycles per mj_step per workgroup.
    python tools/large_stage_profile.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.dactyl.full_perpendicular import load_full_perpendicular_model  # noqa: E402
from robogym_amd.mujoco.large_simulation import LargeModelSimulation  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sim = LargeModelSimulation(load_full_perpendicular_model(), B, device="cuda:0")
zero = torch.zeros((B, 20), device="cuda:0")
for _ in range(6):
    sim.env_step(action=zero, nforward_ticks=3)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
sim.stats.zero_()
a = torch.rand((B, 20), generator=gen, device="cuda:0") * 2 - 1
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); sim.env_step(action=a, nforward_ticks=3, flags=2); t1.record()
torch.cuda.synchronize()
nv = sim.info["nv"]
allp = sim.scratch("dbg")[:, 8 + 5 * nv:8 + 5 * nv + 16].cpu().numpy().mean(0) / sim.n_substeps
prof = allp[:8]
names = ["frames + com", "tendon + crb", "velocity", "collision", "rows", "pid + smooth", "Newton", "Euler"]
st = sim.stats.sum(0).cpu().numpy()
print("B %d: launch %.1f ms; per mj_step per workgroup (cycles): total %.0f" % (B, t0.elapsed_time(t1), prof.sum()))
for n, p in zip(names, prof):
    print("  %-14s %10.0f  %5.1f %%" % (n, p, 100 * p / prof.sum()))
print("means: ncon %.1f nefc %.0f Newton iterations %.2f" % (st[0] / st[3], st[1] / st[3], st[2] / st[3]))
print("inside Newton (cycles per mj_step): " + ", ".join("%s %.0f" % (n, v) for n, v in zip(["M x, J x, cost", "J' f, gradient", "H assembly", "Cholesky", "substitution", "M v, J v", "line search"], allp[8:15])) + ("; probe (a -DRB_HESS_PROBE build) %.0f" % allp[15] if allp[15] else ""))
if allp[15] > 0:
    print("probe slot (builds with -DRB_CHOL_PROBE: diagonal block + row solve + inverse of the blocked Cholesky; -DRB_HESS_PROBE: the per-contact loop of the Hessian assembly): %.0f cycles per mj_step" % allp[15])
