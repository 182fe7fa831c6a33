"""dactyl/reach free-running protocol of tests/test_reach.py::test_reach_free_running_1000_steps_gpu for one library build / launch-flag set: kernel vs double oracle and, beside it,
the oracle's float build vs the double oracle (what fp32 alone does).  usage: python tools/reach_free_running_probe.py [flags] ; RGSTEP_LIB selects the build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import rg_oracle
from robogym_amd.mujoco import simulation_interface
from robogym_amd.mujoco.model_blob import pack_model
from tests.test_reach import OracleReachPhysics, ReachSimulation, _smooth_actions, _sync
from robogym_amd.envs.dactyl.reach import load_reach_model

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = load_reach_model()
for variant in (True, False):
    rg_oracle.set_kernel_variant(variant); simulation_interface.MPR_PLANE_DEPTH = variant
    for seed in (11, 12, 13):
        sim = ReachSimulation(m, 2, device="cuda:0", relative_action=False)
        ora = OracleReachPhysics(m, relative_action=False); ora.zero_control_settle(20)
        _sync(sim, ora)
        o32 = OracleReachPhysics(m, relative_action=False); o32.sim = rg_oracle.OracleSim(pack_model(m), f32=True)
        for n in ("qpos", "qvel", "ctrl", "pid", "qacc_warmstart"):
            getattr(o32.sim, n)[:] = getattr(ora.sim, n)
        acts = _smooth_actions(1000, 20, 0.2, seed)
        err, e32 = np.zeros(1000), np.zeros(1000)
        for k in range(1000):
            sim.env_step(action=torch.tensor(np.repeat(acts[k][None].astype(np.float32), 2, 0), device=sim.device), nforward_ticks=3, flags=flags)
            a = acts[k].astype(np.float32).astype(np.float64)
            ora.env_step(a); o32.env_step(a)
            err[k] = np.abs(sim.qpos[0].cpu().numpy().astype(np.float64) - ora.sim.qpos).max()
            e32[k] = np.abs(o32.sim.qpos.astype(np.float64) - ora.sim.qpos).max()
        f = lambda e: "end %.1e median %.1e beyond %.3f max %.1e" % (e[-1], np.median(e), np.mean(e > 1e-4), e.max())
        print("%s seed %d flags %d | kernel: %s | float oracle: %s" % ("plane  " if variant else "default", seed, flags, f(err), f(e32)), flush=True)
