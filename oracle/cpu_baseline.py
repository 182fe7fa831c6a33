"""cpu_baseline leg of bench.py: the CPU oracle (double-precision C restatement of mj_step, one env per process) stepping
dactyl/locked env.steps on every host core.  TEST INFRASTRUCTURE (oracle/): timed beside the HIP path, never part of it.

SURVEY.md §8(d): "one env per thread, N = all host cores, state N"; kind "port" (a CPU restatement, not mujoco-py: the
reference's own physics is a closed binary that is absent here).

    python -m oracle.cpu_baseline --worker SECONDS SEED      (one core; prints "<env-steps> <seconds>")
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(seconds, seed):
    from oracle.env_oracle import OracleLockedEnvPhysics
    from robogym_amd.envs.dactyl.locked import load_locked_model

    ora = OracleLockedEnvPhysics(load_locked_model())
    ora.settle(30)
    rng = np.random.RandomState(seed)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            ora.env_step(rng.uniform(-1, 1, 20))
            n += 1
            if ora.sim.qpos[2] < -0.1:  # dropped: start over from a settled pose
                ora.sim.reset(); ora.settle(30); ora.prev_dist = None
    return n, time.perf_counter() - t0


def run(seconds=12.0, cores=None):
    """One worker process per host core (separate interpreters: the caller holds a HIP context that must not be forked)."""
    cores = cores or len(os.sched_getaffinity(0))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", "--worker", str(seconds), str(20200901 + 1 + 7919 * i)],
                              stdout=subprocess.PIPE, cwd=ROOT, env=env) for i in range(cores)]
    res = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=seconds + 120)
            n, dt = out.decode().strip().split("\n")[-1].split()
            res.append((int(n), float(dt)))
        except Exception:   # a worker that failed to start is simply not counted (cores reports the ones that ran)
            p.kill()
    wall = time.perf_counter() - t0
    if not res:
        raise RuntimeError("no cpu_baseline worker finished")
    total, rate = sum(n for n, _ in res), sum(n / dt for n, dt in res)
    return {"value": rate, "unit": "env-steps/s", "cores": len(res), "kind": "port",
            "sample": "%d env-steps of dactyl/locked (10 substeps + 3 forwards each, same action distribution), one oracle env per process on %d host cores for %.0f s each "
                      "(%.1f env-steps/s per core; %.0f s wall incl. process start-up); CPU restatement of MuJoCo, not mujoco-py" % (total, len(res), seconds, rate / len(res), wall)}


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--worker":
        n, dt = _worker(float(sys.argv[2]), int(sys.argv[3]))
        print(n, dt)
    else:
        print(run(float(sys.argv[1]) if len(sys.argv) > 1 else 12.0))
