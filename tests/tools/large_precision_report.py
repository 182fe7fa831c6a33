"""dactyl/full_perpendicular: the oracle built in FLOAT against the same source in DOUBLE, re-synchronised env.steps (the
protocol of test_large_model_resync_env_steps_gpu) -- the level at which precision alone separates two runs of the SAME
algorithm on this model, whose neighbouring cubelets touch at 1e-8 ... 1e-6 m (contacts flicker at rounding level).
    python tests/tools/large_precision_report.py [n_steps] >> profiles/rNN_precision.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.rg_oracle import OracleSim  # noqa: E402
from robogym_amd.envs.dactyl.full_perpendicular import load_full_perpendicular_model  # noqa: E402
from robogym_amd.mujoco import setconst  # noqa: E402
from robogym_amd.mujoco.big_tables import derive_big_tables  # noqa: E402
from robogym_amd.mujoco.model_blob import pack_model  # noqa: E402
from tests.test_large_model import OracleFullCube  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
m = load_full_perpendicular_model(); setconst.set_constants(m); derive_big_tables(m)
A, names = m.arrays, m.names["joint"]
hand_j = [j for j, n in enumerate(names) if n.startswith("robot0:")]
hq = np.array([A["jnt_qposadr"][j] for j in hand_j])
P = np.zeros((20, len(hand_j)))
for u in range(20):
    if A["actuator_trntype"][u] == 0:
        P[u, hand_j.index(int(A["actuator_trnid"][u]))] = 1
    else:
        t = int(A["actuator_trnid"][u])
        for w in range(A["tendon_adr"][t], A["tendon_adr"][t] + A["tendon_num"][t]):
            P[u, hand_j.index(int(A["wrap_objid"][w]))] = 1
o64, o32 = OracleFullCube(m, P, hq), OracleFullCube(m, P, hq)
o32.sim = OracleSim(pack_model(m), f32=True)
o64.hold_pose()
for _ in range(60):
    o64.sim.step()
non_target = np.array([i for j, n in enumerate(names) if not n.startswith("target:") for i in range(A["jnt_qposadr"][j], A["jnt_qposadr"][j] + {0: 7, 1: 4, 2: 1, 3: 1}[int(A["jnt_type"][j])])])
rng = np.random.RandomState(3)
E = []
for _ in range(n_steps):
    a = rng.uniform(-1, 1, 20)
    st = o64.state_f32()
    s = o32.sim
    s.qpos[:] = st["qpos"]; s.qvel[:] = st["qvel"]; s.pid[:] = st["pid"]; s.qacc_warmstart[:] = st["warm"]; s.ctrl[:] = st["ctrl"]
    o32.env_step(a); o64.env_step(a)
    d = np.abs(o32.sim.qpos.astype(np.float64) - o64.sim.qpos)
    E.append((d[non_target].max(), d[hq].max(), np.abs(o32.sim.qvel.astype(np.float64) - o64.sim.qvel).max(), o64.sim.ncon, o32.sim.ncon))
E = np.array(E)
print("=" * 110)
print("dactyl/full_perpendicular (nv 168, touching cubelets), oracle FLOAT vs oracle DOUBLE, %d re-synchronised env.steps of iid U(-1,1) relative actions:" % n_steps)
print("  qpos (non-target) median %.2e p90 %.2e max %.2e | hand joints median %.2e max %.2e | qvel median %.2e max %.2e | contacts at the end of a step: double %.1f, float %.1f (differ on %d of %d steps)" % (
    np.median(E[:, 0]), np.percentile(E[:, 0], 90), E[:, 0].max(), np.median(E[:, 1]), E[:, 1].max(), np.median(E[:, 2]), E[:, 2].max(), E[:, 3].mean(), E[:, 4].mean(), int((E[:, 3] != E[:, 4]).sum()), n_steps))
