"""Stage cycle counters of rb_step_kernel on the two rearrange worlds (flags bit 1): mean cycles per mj_step per workgroup.
    python tools/rearrange_stage_profile.py [B] [ycb]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.rearrange.blocks import BatchedBlockRearrangeEnv  # noqa: E402

from robogym_amd.envs.rearrange.ycb import BatchedYcbRearrangeEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = (BatchedYcbRearrangeEnv if "ycb" in sys.argv[2:] else BatchedBlockRearrangeEnv)(B, stabilize_steps=20, n_random_initial_steps=2, settle_steps=10)
env.reset()
gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
act = lambda: torch.rand((B, 6), generator=gen, device="cuda:0") * 2 - 1
for _ in range(3):
    env.step(act())
names = ["frames + com", "tendon + crb", "velocity", "collision", "rows", "pid + smooth", "Newton", "Euler"]
a = act()
for label, sim, run in (("solver world", env.solver_sim, lambda f: env.solver_sim.step_tcp(env.sim, a, env.tcp, flags=f)), ("main world", env.sim, lambda f: env.sim.env_step(nforward_ticks=2, flags=32 | f))):
    sim.stats.zero_()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); run(2); t1.record(); torch.cuda.synchronize()
    nv = sim.info["nv"]
    allp = sim.scratch("dbg")[:, 8 + 5 * nv:8 + 5 * nv + 16].cpu().numpy().mean(0) / sim.n_substeps
    prof = allp[:8]
    st = sim.stats.sum(0).cpu().numpy()
    print("%s, B %d: launch %.1f ms; per mj_step per workgroup (cycles): total %.0f" % (label, B, t0.elapsed_time(t1), prof.sum()))
    for n, p in zip(names, prof):
        print("  %-14s %10.0f  %5.1f %%" % (n, p, 100 * p / prof.sum()))
    print("  means: ncon %.1f nefc %.0f Newton iterations %.2f" % (st[0] / st[3], st[1] / st[3], st[2] / st[3]))
    print("  inside Newton: " + ", ".join("%s %.0f" % (n, v) for n, v in zip(["M x, J x, cost", "J' f, gradient", "H assembly", "Cholesky", "substitution", "M v, J v", "line search"], allp[8:15])) + ("; probe (a -DRB_HESS_PROBE build) %.0f" % allp[15] if allp[15] else ""))
