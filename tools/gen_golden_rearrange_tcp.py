"""Golden vectors for the TCP arm's action path (tests/golden/rearrange_tcp.npz): the REAL `FreeDOFTcpArm.denormalize_position_control / constrain_quat_ctrl`
(/root/reference/robogym/robot/ur16e/mujoco/free_dof_tcp_arm.py:131-181, class attributes of `FreeRollYawTcpArm` :248-254) and `MocapSolver.get_tcp_quat` /
`Solver.get_joint_mapping` (robot/control/tcp/mocap_solver.py:17-49, solver.py:10-71), their source executed as it stands on stub objects (the modules' import
chains need gym / mujoco_py, which are absent here): normalised action, wrist joint position, gripper orientation  ->  denormalised control, constrained
angles, the quaternion difference handed to the mocap solver.  Needs /root/reference; the fixture travels.

    python tools/gen_golden_rearrange_tcp.py
"""
import ast
import enum
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
np.float = float
sys.path.insert(0, "/root/reference")
from robogym.utils import rotation  # noqa: E402

ROBOT = "/root/reference/robogym/robot"


def extract(path, cls_name, members):
    """class `cls_name` of the file reduced to `members` (methods / class-level assignments), as a code object"""
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name][0]
    keep = []
    for n in cls.body:
        name = n.name if isinstance(n, ast.FunctionDef) else (n.targets[0].id if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name) else
                                                              (n.target.id if isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name) else None))
        if name in members:
            if isinstance(n, ast.FunctionDef):
                n.decorator_list = [d for d in n.decorator_list if not (isinstance(d, ast.Name) and d.id == "abstractmethod") and not (isinstance(d, ast.Attribute) and d.attr == "abstractmethod")]
            keep.append(n)
    assert {getattr(k, "name", None) or (k.targets[0].id if isinstance(k, ast.Assign) else k.target.id) for k in keep} == set(members), (cls_name, members)
    shell = ast.ClassDef(name=cls_name, bases=[], keywords=[], body=keep, decorator_list=[])
    mod_ = ast.Module(body=[shell], type_ignores=[])
    ast.fix_missing_locations(mod_)
    return compile(mod_, path, "exec")


def main():
    ns = {"np": np, "rotation": rotation, "Enum": enum.Enum, "Dict": dict, "List": list, "Optional": None}
    # PrincipalAxis and the speed scale are module-level in the reference: take those statements too
    for path, names in ((ROBOT + "/control/tcp/solver.py", {"PrincipalAxis"}), (ROBOT + "/ur16e/mujoco/free_dof_tcp_arm.py", {"DOF_DIM_SPEED_SCALE"})):
        tree = ast.parse(open(path).read())
        body = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name in names) or
                (isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name) and n.target.id in names)]
        for n in body:
            if isinstance(n, ast.AnnAssign):
                n.annotation = ast.Name(id="dict", ctx=ast.Load())
        m_ = ast.Module(body=body, type_ignores=[]); ast.fix_missing_locations(m_)
        exec(compile(m_, path, "exec"), ns)
    exec(extract(ROBOT + "/control/tcp/solver.py", "Solver", {"get_joint_mapping"}), ns)
    exec(extract(ROBOT + "/control/tcp/mocap_solver.py", "MocapSolver", {"JOINT_MAPPING", "get_tcp_quat", "align_axis"}), ns)
    exec(extract(ROBOT + "/ur16e/mujoco/free_dof_tcp_arm.py", "FreeDOFTcpArm", {"JOINT_DRIFT_THRESHOLD", "denormalize_position_control", "constrain_quat_ctrl"}), ns)
    exec(extract(ROBOT + "/ur16e/mujoco/free_dof_tcp_arm.py", "FreeRollYawTcpArm", {"DOF_DIMS"}), ns)
    exec(extract(ROBOT + "/ur16e/mujoco/free_dof_tcp_arm.py", "FreeWristTcpArm", {"DOF_DIMS", "ALIGN_AXIS"}), ns)
    PA, Solver, Mocap, Arm, RollYaw, Wrist = ns["PrincipalAxis"], ns["Solver"], ns["MocapSolver"], ns["FreeDOFTcpArm"], ns["FreeRollYawTcpArm"], ns["FreeWristTcpArm"]

    sys.path.insert(0, os.path.join(HERE, ".."))
    from robogym_amd.envs.rearrange.xml import load_solver_model
    model = load_solver_model()
    jn = model.names["joint"]
    rng_lo = np.array([model.arrays["jnt_range"][jn.index("robot0:J%d" % k)][0] for k in range(1, 7)])
    rng_hi = np.array([model.arrays["jnt_range"][jn.index("robot0:J%d" % k)][1] for k in range(1, 7)])

    rng = np.random.RandomState(5)
    T = 96
    out = {k: [] for k in ("action", "mpc", "q", "gripper_quat", "denorm", "angles", "dquat")}
    for t in range(T):
        mpc = [0.1, 0.05, 0.165, 0.03][t % 4]
        q = rng.uniform(rng_lo, rng_hi)
        if t % 3 == 0:      # wrist joint near one end of its range: the constraint binds
            q[5] = (rng_hi[5] - rng.uniform(0, 0.05)) if t % 2 else (rng_lo[5] + rng.uniform(0, 0.05))
        gq = rng.randn(4); gq /= np.linalg.norm(gq)
        a = rng.uniform(-1, 1, 5)
        solver = types.SimpleNamespace(dof_dims=RollYaw.DOF_DIMS, dof_dims_axes=[ax.value for ax in RollYaw.DOF_DIMS], alignment_axis=None, JOINT_MAPPING=Mocap.JOINT_MAPPING,
                                       body_name="tcp", mj_sim=types.SimpleNamespace(data=types.SimpleNamespace(get_body_xquat=lambda name, gq=gq: gq.copy())))
        solver.get_joint_mapping = types.MethodType(Solver.get_joint_mapping, solver)
        solver.get_tcp_quat = types.MethodType(Mocap.get_tcp_quat, solver)
        arm = types.SimpleNamespace(is_in_joint_control_mode=False, max_position_change=mpc, solver=solver, JOINT_DRIFT_THRESHOLD=Arm.JOINT_DRIFT_THRESHOLD,
                                    speed_per_dof_dim=[ns["DOF_DIM_SPEED_SCALE"][ax] * mpc for ax in RollYaw.DOF_DIMS],
                                    observe=lambda q=q: types.SimpleNamespace(joint_positions=lambda: q.copy()),
                                    actuator_ctrl_range_lower_bound=lambda: rng_lo.copy(), actuator_ctrl_range_upper_bound=lambda: rng_hi.copy())
        den = Arm.denormalize_position_control(arm, a.copy(), relative_action=True)
        pos, angle = np.split(den, (3,))                                     # FreeDOFTcpArm.set_position_control, :193-196
        angle = Arm.constrain_quat_ctrl(arm, angle.copy())
        dq = solver.get_tcp_quat(angle)
        for k, v in (("action", a), ("mpc", mpc), ("q", q), ("gripper_quat", gq), ("denorm", den), ("angles", angle), ("dquat", dq)):
            out[k].append(np.asarray(v, dtype=np.float64))
    path = os.path.join(HERE, "..", "tests", "golden", "rearrange_tcp.npz")
    np.savez_compressed(path, wrist_range=np.array([rng_lo[5], rng_hi[5]]), **{k: np.array(v) for k, v in out.items()})
    bind = np.sum(np.abs(np.array(out["angles"])[:, 1] - np.array(out["denorm"])[:, 4]) > 1e-12)
    print("wrote", path, "; the wrist constraint binds in %d of %d samples" % (bind, T))

    # ---- control_mode tcp+wrist: FreeWristTcpArm (DOF_DIMS = [PITCH], ALIGN_AXIS = PITCH; free_dof_tcp_arm.py:238-246) -- 4 arm numbers, the commanded orientation
    # forced back onto the vertical by MocapSolver.align_axis (mocap_solver.py:58-74)
    out = {k: [] for k in ("action", "mpc", "q", "gripper_quat", "denorm", "angles", "dquat")}
    for t in range(T):
        mpc = [0.1, 0.05, 0.165, 0.03][t % 4]
        q = rng.uniform(rng_lo, rng_hi)
        if t % 3 == 0:
            q[5] = (rng_hi[5] - rng.uniform(0, 0.05)) if t % 2 else (rng_lo[5] + rng.uniform(0, 0.05))
        if t % 2:      # a gripper pointing (almost) straight down, as the arm holds it, tilted by a few degrees
            tilt = rng.uniform(-0.15, 0.15, 3)
            gq = rotation.quat_mul(rotation.euler2quat(np.array([np.pi, 0.0, rng.uniform(-np.pi, np.pi)])), rotation.euler2quat(tilt))
        else:
            gq = rng.randn(4); gq /= np.linalg.norm(gq)
        a = rng.uniform(-1, 1, 4)
        solver = types.SimpleNamespace(dof_dims=Wrist.DOF_DIMS, dof_dims_axes=[ax.value for ax in Wrist.DOF_DIMS], alignment_axis=Wrist.ALIGN_AXIS, JOINT_MAPPING=Mocap.JOINT_MAPPING,
                                       body_name="tcp", align_axis=Mocap.align_axis, mj_sim=types.SimpleNamespace(data=types.SimpleNamespace(get_body_xquat=lambda name, gq=gq: gq.copy())))
        solver.get_joint_mapping = types.MethodType(Solver.get_joint_mapping, solver)
        solver.get_tcp_quat = types.MethodType(Mocap.get_tcp_quat, solver)
        arm = types.SimpleNamespace(is_in_joint_control_mode=False, max_position_change=mpc, solver=solver, JOINT_DRIFT_THRESHOLD=Arm.JOINT_DRIFT_THRESHOLD,
                                    speed_per_dof_dim=[ns["DOF_DIM_SPEED_SCALE"][ax] * mpc for ax in Wrist.DOF_DIMS],
                                    observe=lambda q=q: types.SimpleNamespace(joint_positions=lambda: q.copy()),
                                    actuator_ctrl_range_lower_bound=lambda: rng_lo.copy(), actuator_ctrl_range_upper_bound=lambda: rng_hi.copy())
        den = Arm.denormalize_position_control(arm, a.copy(), relative_action=True)
        pos, angle = np.split(den, (3,))
        angle = Arm.constrain_quat_ctrl(arm, angle.copy())
        dq = solver.get_tcp_quat(angle)
        for k, v in (("action", a), ("mpc", mpc), ("q", q), ("gripper_quat", gq), ("denorm", den), ("angles", angle), ("dquat", dq)):
            out[k].append(np.asarray(v, dtype=np.float64))
    path = os.path.join(HERE, "..", "tests", "golden", "rearrange_tcp_wrist.npz")
    np.savez_compressed(path, wrist_range=np.array([rng_lo[5], rng_hi[5]]), **{k: np.array(v) for k, v in out.items()})
    print("wrote", path)


if __name__ == "__main__":
    main()
