"""Re-synchronised one-env-step errors of the loaded librgstep build vs the oracle (kernel-variant configuration),
200 random-action env-steps, plus the mean Newton iterations: the A/B companion of tools/gpu_ab.sh (RGSTEP_LIB)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rg_oracle
from oracle.env_oracle import OracleLockedEnvPhysics
from robogym_amd.envs.dactyl.locked import LockedSimulation, load_locked_model
from tests.helpers import resync_errors
rg_oracle.set_kernel_variant(True)
from robogym_amd.mujoco import simulation_interface  # noqa: E402
simulation_interface.MPR_PLANE_DEPTH = True
model = load_locked_model()
ora = OracleLockedEnvPhysics(model); ora.sim.reset(); ora.settle(30)
ora.sim.stats_reset()
sim = LockedSimulation(model, 1, device="cuda:0")
rng = np.random.RandomState(20200901 + 1)
errs = resync_errors(sim, ora, rng.uniform(-1, 1, (200, 20)))
st = sim.get_field(7).cpu().numpy()[0]
print("%s: qpos median %.2e p90 %.2e p99 %.2e max %.2e | qvel median %.2e p90 %.2e max %.2e | newton iters/substep %.2f" % (
    os.path.basename(os.environ.get("RGSTEP_LIB", "librgstep.so")), np.median(errs[:, 0]), np.percentile(errs[:, 0], 90), np.percentile(errs[:, 0], 99), errs[:, 0].max(),
    np.median(errs[:, 1]), np.percentile(errs[:, 1], 90), errs[:, 1].max(), st[2] / max(st[3], 1)))
os_ = ora.sim.stats()
print("oracle (fp64, tolerance 1e-8) over the same steps: newton iters/substep %.2f, contacts/substep %.2f, rows/substep %.1f  (kernel: contacts %.2f, rows %.1f)" % (
    os_["iters"], os_["ncon"], os_["nefc"], st[0] / max(st[3], 1), st[1] / max(st[3], 1)))   # (ro_stats returns means per mj_step)
