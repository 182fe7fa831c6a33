"""Quick GPU probe of the rearrange dual simulation (physics launches only): time per env.step at batch B, stats, status bits."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from robogym_amd import _native
from robogym_amd.envs.rearrange.xml import load_blocks_model, load_solver_model
from robogym_amd.mujoco.large_simulation import LargeModelSimulation

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
main, solver = load_blocks_model(5), load_solver_model()
sm = LargeModelSimulation(main, B, n_substeps=40, hand=False)
sc = LargeModelSimulation(solver, B, n_substeps=40, hand=False)
A = main.arrays; jn = main.names["joint"]
arm0 = torch.tensor(np.deg2rad([135.0, -90, 135, -100, -240, 135]), dtype=torch.float32, device="cuda")
sm.qpos[:, :6] = arm0; sm.ctrl[:, :6] = arm0; sc.qpos[:, :6] = arm0
rng = np.random.RandomState(0)
ztop = 0.453 + 0.03324 + 0.0254
for i in range(5):
    qa = int(A["jnt_qposadr"][jn.index("object%d:joint" % i)])
    p = np.stack([1.2 + 0.11 * i + 0.02 * rng.rand(B), 0.55 + 0.1 * i + 0.02 * rng.rand(B), np.full(B, ztop)], 1)
    yaw = rng.uniform(0, 2 * np.pi, B)
    sm.qpos[:, qa:qa + 3] = torch.tensor(p, dtype=torch.float32, device="cuda")
    sm.qpos[:, qa + 3:qa + 7] = torch.tensor(np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], 1), dtype=torch.float32, device="cuda")
# weld data identity (reset_mocap_welds)
sc.eq_data[:, :7] = torch.tensor([0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
args = _native.RbTcpArgs()
sj = solver.names["joint"]; As = solver.arrays
for k in range(6):
    args.arm_qposadr[k] = int(As["jnt_qposadr"][sj.index("robot0:J%d" % (k + 1))]); args.main_arm_qposadr[k] = int(A["jnt_qposadr"][jn.index("robot0:J%d" % (k + 1))])
args.main_gripper_actuator = 6; args.tcp_body = solver.name2id("body", "robot0:gripper_tcp"); args.wrist_joint = sj.index("robot0:J6"); args.reset_controller_error = 1
args.max_position_change = 0.1; args.speed_roll = float(np.deg2rad(200)); args.speed_pitch = float(np.deg2rad(600)); args.joint_drift_threshold = float(np.deg2rad(1))
args.gripper_ctrl_lo = -0.04473; args.gripper_ctrl_hi = 0.0
act = torch.tensor(rng.uniform(-1, 1, (steps + 3, B, 6)).astype(np.float32), device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for t in range(steps + 3):
    if t == 3:
        torch.cuda.synchronize(); t0 = time.time()
    ev[0].record(); sc.step_tcp(sm, act[t], args); ev[1].record(); sm.env_step(nforward_ticks=2, flags=32); ev[2].record()
torch.cuda.synchronize(); dt = (time.time() - t0) / steps
print("B %d: %.1f ms per env.step (solver launch %.1f ms, main launch %.1f ms) -> %.0f env-steps/s" % (B, dt * 1e3, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), B / dt))
st = sm.stats.sum(0).cpu().numpy(); print("main: ncon %.2f nefc %.1f iters %.2f per mj_step; status bits main %d solver %d" % (st[0] / st[3], st[1] / st[3], st[2] / st[3], int(sm.status.max()), int(sc.status.max())))
st = sc.stats.sum(0).cpu().numpy(); print("solver: ncon %.2f nefc %.1f iters %.2f per mj_step" % (st[0] / st[3], st[1] / st[3], st[2] / st[3]))
print("finite", bool(torch.isfinite(sm.qpos).all()), "tcp force mean", sm.sensordata[:, 6:9].mean(0).cpu().numpy())
