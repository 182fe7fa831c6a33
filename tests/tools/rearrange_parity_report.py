"""Re-synchronised env.step errors of the rearrange worlds on the MI355X against the CPU oracle, as a distribution: before every env.step (40 + 40 mj_steps,
TCP solver world + main world) both sides start from the oracle's state rounded to fp32 (protocol of tests/test_rearrange_kernel.py / DESIGN.md §5).

    python tests/tools/rearrange_parity_report.py [steps_blocks] [steps_ycb]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robogym_amd.envs.rearrange.xml import load_blocks_model, load_solver_model, load_ycb_model      # noqa: E402
from tests import test_rearrange_kernel as TK                                                          # noqa: E402
from tests import test_rearrange_ycb as TY                                                             # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def table(title, names, errs):
    print(title)
    for k, n in enumerate(names):
        e = errs[:, k]
        print("  %-18s median %.2e  p90 %.2e  p99 %.2e  max %.2e" % (n, np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))


def by_history(names, errs, same):
    """the same table split by contact history (tests/test_rearrange_kernel.py contact_history): steps whose 40 + 40 mj_steps held the same contact and row counts on both sides"""
    for label, sel in (("same contact history", same), ("differing contact history", ~same)):
        if sel.any():
            table("  -- %s: %d of %d steps" % (label, sel.sum(), len(sel)), names, errs[sel])


errs, same = TK._resync_env_steps((load_blocks_model(5), load_solver_model()), None, "cuda:0", n_substeps=40, nsteps=nb, seed=5, classify=True)
print("  steps with solver qpos error > 1e-3:", [(int(k), float("%.3g" % errs[k, 0])) for k in np.nonzero(errs[:, 0] > 1e-3)[0]])
table("rearrange/blocks, %d re-synchronised env.steps (random actions, every third pressing down), L-inf per env.step:" % nb,
      ["solver qpos", "mocap", "main ctrl", "main qpos", "main qvel", "main pid", "sensordata (rel)"], errs)
by_history(["solver qpos", "mocap", "main ctrl", "main qpos", "main qvel", "main pid", "sensordata (rel)"], errs, same)
errs, same = TY._resync((load_ycb_model(8), load_solver_model()), None, "cuda:0", 40, ny, classify=True)
table("rearrange/ycb (object set 0), %d re-synchronised env.steps:" % ny, ["main ctrl", "main qpos", "main qvel"], errs)
by_history(["main ctrl", "main qpos", "main qvel"], errs, same)
for k in (1, 2, 3, 4, 5):
    errs, same = TY._resync((load_ycb_model(8, set_index=k), load_solver_model()), None, "cuda:0", 40, max(ny // 6, 4), spread=True, classify=True)
    table("rearrange/ycb object set %d, %d re-synchronised env.steps (%d with the same contact history):" % (k, len(errs), same.sum()), ["main ctrl", "main qpos", "main qvel"], errs)
