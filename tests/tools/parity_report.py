"""SURVEY §8(d) parity protocol on the MI355X: free-running qpos L-inf (non-target joints) of the HIP path vs the
CPU oracle at env-steps 1, 10, 100, 1000 and the first step beyond 1e-4, plus the re-synchronised one-env-step
errors over the same action stream.  Oracle = test infrastructure (oracle/): this tool is a checker, not product.

    python tests/tools/parity_report.py [n_streams] [n_steps]  > profiles/rNN_parity.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.env_oracle import OracleLockedEnvPhysics  # noqa: E402
from robogym_amd.envs.dactyl.locked import LockedSimulation, load_locked_model  # noqa: E402
from tests.helpers import NON_TARGET_QPOS, resync_errors, sync_state_from_oracle  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
from oracle import rg_oracle  # noqa: E402
from robogym_amd.mujoco import simulation_interface  # noqa: E402

model = load_locked_model()
for kernel_variant in (True, False):
    rg_oracle.set_kernel_variant(kernel_variant)
    simulation_interface.MPR_PLANE_DEPTH = kernel_variant
    print("=" * 110)
    print("ORACLE = %s" % ("PORTAL-PLANE contact depth on both sides (kernel flags bit 4; oracle: set_kernel_variant, box-box through MPR): does the kernel compute what it says, at fp32 tolerance"
                           if kernel_variant else "DEFAULT on both sides (libccd contact depth, multi-point box-box): distance of the product to the MuJoCo restatement"))
    marks = [1, 10, 100, 1000]
    print("free-running drift, kernel (fp32, MI355X) vs oracle (fp64, CPU), dactyl/locked, iid U(-1,1) relative actions, same bytes at step 0")
    print("stream  " + "  ".join("Linf@%-5d" % m for m in marks) + "  first step with Linf > 1e-4   contacts/substep")
    for sidx in range(n_streams):
        rng = np.random.RandomState(20200901 + 1 + sidx)
        ora = OracleLockedEnvPhysics(model)
        ora.sim.reset(); ora.settle(30)
        sim = LockedSimulation(model, 1, device="cuda:0")
        sync_state_from_oracle(sim, ora)
        first, at = None, {}
        for t in range(1, n_steps + 1):
            a = rng.uniform(-1, 1, 20)
            sim.env_step(action=torch.tensor(a[None].astype(np.float32), device="cuda:0"), nforward_ticks=3)
            ora.env_step(a)
            e = np.abs(sim.qpos.cpu().numpy()[0].astype(np.float64) - ora.sim.qpos)[NON_TARGET_QPOS].max()
            if first is None and e > 1e-4:
                first = t
            if t in marks:
                at[t] = e
        st = sim.get_field(7).cpu().numpy()[0]
        print("%-7d " % sidx + "  ".join("%-10.2e" % at.get(m, float("nan")) for m in marks) + "  %-28s %.2f" % (first, st[0] / max(st[3], 1)))

    print()
    print("re-synchronised one-env-step errors (kernel restarted from the oracle's fp32-rounded state before every env.step)")
    ora = OracleLockedEnvPhysics(model)
    ora.sim.reset(); ora.settle(30)
    sim = LockedSimulation(model, 1, device="cuda:0")
    rng = np.random.RandomState(20200901 + 1)
    errs = resync_errors(sim, ora, rng.uniform(-1, 1, (200, 20)), detail=True)
    for name, col in (("qpos", 0), ("qvel", 1), ("pid state", 2)):
        v = errs[:, col]
        print("  %-9s median %.2e   p90 %.2e   p99 %.2e   max %.2e   (200 env-steps)" % (name, np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
    print("  qpos error by the largest contact count the env.step went through (tail characterisation; compare profiles/r03_precision.txt:")
    print("  the same table between an fp32 and an fp64 build of the ORACLE):")
    for lo, hi in ((0, 0), (1, 2), (3, 4), (5, 6), (7, 9), (10, 99)):
        sel = (errs[:, 3] >= lo) & (errs[:, 3] <= hi)
        if sel.any():
            v = errs[sel, 0]
            print("    ncon %2d-%-2d  n %-4d median %.2e  p90 %.2e  max %.2e   deepest penetration (median) %.2e m" % (lo, hi, sel.sum(), np.median(v), np.percentile(v, 90), v.max(), np.median(errs[sel, 4])))
    big = errs[errs[:, 0] > 2e-3]
    print("  env-steps beyond 2e-3: %d of %d%s" % (len(big), len(errs), "" if not len(big) else "; their contact counts %s, deepest penetrations %s" % (
        [int(x) for x in big[:, 3]], ["%.1e" % x for x in big[:, 4]])))
