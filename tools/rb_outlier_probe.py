"""debug probe (round 6): the tcp+wrist env.step protocol of tests/test_rearrange_env.py, per (step, env): robot_joint_pos / qpos error vs oracle, contact-history flag, Newton iterations per mj_step
kernel vs oracle in both worlds.  RGSTEP_LIB selects the build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robogym_amd.envs.rearrange.blocks import BatchedBlockRearrangeEnv
from tests.test_rearrange_env import _oracle_from_kernel
from tests.test_rearrange_kernel import contact_history

B, n_substeps, nsteps = 4, 40, 12
env = BatchedBlockRearrangeEnv(B, device="cuda:0", n_substeps=n_substeps, control_mode="tcp+wrist", stabilize_steps=20, n_random_initial_steps=2, settle_steps=10)
env.reset()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
for step in range(nsteps):
    a = rng.uniform(-1, 1, (B, env.action_dim)).astype(np.float32)
    a[:, 2] = -np.abs(a[:, 2])
    oracles = [_oracle_from_kernel(env, r, n_substeps) for r in range(B)]
    km = env.sim.stats.cpu().numpy().astype(np.float64); kc = env.solver_sim.stats.cpu().numpy().astype(np.float64)
    obs, rew, done, info = env.step(torch.tensor(a, device=env.device)); env.sync()
    km2 = env.sim.stats.cpu().numpy().astype(np.float64); kc2 = env.solver_sim.stats.cpu().numpy().astype(np.float64)
    for r in range(B):
        o = oracles[r]
        o.main.sim.stats_reset(); o.solver.sim.stats_reset()
        oobs = o.env_step(a[r].astype(np.float64))[0]
        same = contact_history(env.sim, o.main, km[r], row=r) and contact_history(env.solver_sim, o.solver, kc[r], row=r)
        e = lambda k: float(np.abs(obs[k][r].cpu().numpy().astype(np.float64).reshape(np.asarray(oobs[k]).shape) - oobs[k]).max())
        sm, ss = o.main.sim.stats(), o.solver.sim.stats()
        print("step %2d env %d same %d | robot_joint_pos %.2e qpos %.2e obj_pos %.2e | main iters kernel %.2f oracle %.2f  nefc %.1f/%.1f | solver iters kernel %.2f oracle %.2f nefc %.1f/%.1f" % (
            step, r, same, e("robot_joint_pos"), e("qpos"), e("obj_pos"),
            (km2[r, 2] - km[r, 2]) / max(km2[r, 3] - km[r, 3], 1), sm["iters"] / max(sm["steps"], 1), (km2[r, 1] - km[r, 1]) / max(km2[r, 3] - km[r, 3], 1), sm["nefc"] / max(sm["steps"], 1),
            (kc2[r, 2] - kc[r, 2]) / max(kc2[r, 3] - kc[r, 3], 1), ss["iters"] / max(ss["steps"], 1), (kc2[r, 1] - kc[r, 1]) / max(kc2[r, 3] - kc[r, 3], 1), ss["nefc"] / max(ss["steps"], 1)), flush=True)
