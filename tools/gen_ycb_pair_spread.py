"""tests/golden/ycb_pair_spread.json: what fp32 arithmetic ALONE does to the re-synchronised env.step protocol of the ycb parity tests
(tests/test_rearrange_ycb.py `_resync`), measured on the oracle pair -- oracle/rg_oracle.c built in double (THE oracle) and the same source built in
float (oracle/librg_oracle_f32.so), both stepped from the double oracle's state with the same actions.  VERDICT r05 "next" 4: the GPU tests hold the
HIP kernel's qpos / qvel / ctrl errors to a multiple of THIS spread (per shipped object set, split by whether both sides went through the same
contact / row counts) instead of to bounds that were widened to what had been measured on the GPU.

    python tools/gen_ycb_pair_spread.py            # CPU only (both oracle builds); ~2 min

The protocol (placement, settle, action stream: RandomState(3)) is the test's own, imported from it.  Test infrastructure: reads oracle/ and tests/."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from robogym_amd.envs.rearrange.blocks import load_solver_model          # noqa: E402
from robogym_amd.envs.rearrange.ycb import load_ycb_model                # noqa: E402
from tests.test_rearrange_ycb import N, oracle_pair_spread               # noqa: E402

NSTEPS = 12


def main():
    out = {"protocol": "tests/test_rearrange_ycb.py _resync: 40 + 40 mj_steps per env.step, re-synchronised from the double oracle before every step, actions RandomState(3)",
           "nsteps": NSTEPS, "columns": ["main ctrl", "main qpos", "main qvel"], "sets": {}}
    for k in range(6):
        models = (load_ycb_model(N, set_index=k), load_solver_model())
        errs, same = oracle_pair_spread(models, 40, NSTEPS, spread=k > 0)
        rec = {"same_history_steps": int(same.sum()), "max_all": errs.max(axis=0).tolist(), "median_all": np.median(errs, axis=0).tolist(),
               "max_same": (errs[same].max(axis=0) if same.any() else np.zeros(3)).tolist()}
        out["sets"][str(k)] = rec
        print("set", k, rec, flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "ycb_pair_spread.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
