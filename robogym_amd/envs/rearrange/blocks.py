"""rearrange/blocks on the MI355X (BASELINE.json configs[3]: UR16e + 2f-85 gripper, table contacts, num_objects = 5, batch 4096).

Host-side mirror of the reference's `BlockRearrangeEnv` (/root/reference/robogym/envs/rearrange/blocks.py:23-40 over
envs/rearrange/common/base.py `RearrangeEnv`), batched: `make_env` / `make_simple_env`, `step(actions[B, 6])`, `reset(mask)`, `observe()`.

`step` = THREE launches, no host synchronisation:
  1. `rb_batch_step_tcp`   the TCP solver's own world: sync to the main arm, forward, mocap target from the action, 40 x mj_step, main ctrl
                           (JointControlledTcpArm.set_position_control, robot/ur16e/mujoco/joint_controlled_tcp_arm.py:89-97)
  2. `rb_batch_step_ex`    the main world: 40 x mj_step + two state-less forwards, the last in full (sensors, final-state stage arrays)
                           (SimulationInterface.step + RobotEnv._observe_sync, simulation_interface.py:176-189, robot_env.py:672-688)
  3. `ra_env_post_step`    observation row, reward, done, goal distances, MultiGoalTracker, gripper hand-over to the solver world
                           (robogym_amd/csrc/ra_env_kernel.h)

Host work (numpy, outside the per-step path): the reset recipe's placement sampling (`place_objects_in_grid`, common/utils.py:719-829: a grid of
cells over the placement area, distinct random cells; restated for boxes, distribution-equivalent, not draw-for-draw) and goal sampling
(`ObjectStateGoal.next_goal`, goals/object_state.py:355-418: targets keep the objects' initial yaw, new grid placement).  The per-episode
"re-creation" of the simulation (common/base.py:850-856) is the same model with new object poses here: blocks have no per-episode shape
parameters at the default `object_scale_low/high = 0`, material `default`.

`control_mode = "joint"` (robot_interface.py:9-20, MujocoURJointGripperCompositeRobot = JointControlledArm + MujocoRobotiqGripper, composite/ur_gripper_arm.py): no TCP
solver world; `step(actions[B, 7])`, the action map (six arm joints relative to their positions, capped by max_position_change; the gripper relative to its control)
runs at the head of the main world's launch (rb_batch_set_action_limits) -- TWO launches per step.

`control_mode = "tcp+wrist"` (FreeWristTcpArm, free_dof_tcp_arm.py:238-246): `step(actions[B, 5])` = xyz, wrist, gripper; the solver world's launch ignores the roll number
and aligns the commanded orientation with the vertical (rb_tcp_args.wrist_only).

`tcp_solver_mode = "mocap"` (MujocoIdealURGripperCompositeRobot, composite/ur_gripper_arm.py:126-128): the main world's arm on the mocap weld, no joint actuators, no solver
world; the hook runs on the env's own world (rb_tcp_args.self_world) -- ONE physics launch per step.  Blocks, synchronous reset.

Not built for this env: vision, `teleport_to_goal`, masks of the placement area, duplicated-object groups.
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from robogym_amd import _native
from robogym_amd.envs.rearrange.xml import load_blocks_model, load_solver_model, object_bounding_boxes
from robogym_amd.mujoco.large_simulation import LargeModelSimulation

TABLETOP_EXPERIMENT_INITIAL_POS = np.deg2rad(np.array([135.0, -90, 135, -100, -240, 135]))   # robot/ur16e/arm_interface.py:27
SPEED_ROLL, SPEED_PITCH = float(np.deg2rad(200)), float(np.deg2rad(600))                      # free_dof_tcp_arm.py:13-17
JOINT_DRIFT_THRESHOLD = float(np.deg2rad(1))
FLAG_FULL_FORWARD = 32

OBS_KEYS = [("obj_pos", "N3"), ("obj_rel_pos", "N3"), ("obj_vel_pos", "N3"), ("obj_rot", "N3"), ("obj_vel_rot", "N3"), ("robot_joint_pos", 6), ("gripper_pos", 3),
            ("gripper_velp", 3), ("gripper_controls", 1), ("gripper_qpos", 1), ("gripper_vel", 1), ("qpos", "nq"), ("qpos_goal", "nq"), ("goal_obj_pos", "N3"),
            ("goal_obj_rot", "N3"), ("is_goal_achieved", 1), ("rel_goal_obj_pos", "N3"), ("rel_goal_obj_rot", "N3"), ("obj_gripper_contact", "N2"), ("obj_bbox_size", "N3"),
            ("obj_colors", "N4"), ("safety_stop", 1), ("tcp_force", 3), ("tcp_torque", 3)]


def euler2quat(e):
    """robogym/utils/rotation.py:110-126 convention: q = qx(e0) qy(e1) qz(e2) (pinned by tests/golden/rearrange_rotation.npz)"""
    e = np.asarray(e, dtype=np.float64)
    hx, hy, hz = 0.5 * e[..., 0], 0.5 * e[..., 1], 0.5 * e[..., 2]
    cx, sx, cy, sy, cz, sz = np.cos(hx), np.sin(hx), np.cos(hy), np.sin(hy), np.cos(hz), np.sin(hz)
    return np.stack([cx * cy * cz - sx * sy * sz, sx * cy * cz + cx * sy * sz, cx * sy * cz - sx * cy * sz, cx * cy * sz + sx * sy * cz], axis=-1)


class BatchedBlockRearrangeEnv:
    def __init__(self, batch_size: int, device="cuda:0", num_objects: int = 5, starting_seed: int = 0, max_position_change: float = 0.1,
                 arm_reset_controller_error: bool = True, n_random_initial_steps: int = 10, stabilize_steps: int = 100, settle_steps: int = 100,
                 success_threshold=None, penalty=None, max_timesteps_per_goal_per_obj: int = 200, successes_needed: int = 5, success_reward: float = 5.0,
                 use_goal_distance_reward: bool = True, goal_reward_per_object: float = 1.0, used_table_portion: float = 1.0, lib=None, n_substeps: int = 40,
                 main_model=None, wrappers: bool = False, n_action_bins: int = 11, smooth_alpha: float = 0.3, reward_clip: float = 100.0,
                 pipelined_reset: bool = False, action_spacing: str = "linear", per_env_parameters: bool = True, randomizer_params: Optional[dict] = None,
                 stabilize_object_damping: float = 1.0e-3, control_mode: str = "tcp+roll+yaw", device_reset: bool = False, tcp_solver_mode: str = "mocap_ik"):
        """`per_env_parameters`: every env carries its own copy of the randomisable model fields (`self.sim.params`, LargeModelSimulation(env_params=True)) -- what
        the reference's simulation randomizers and `stabilize_objects` write into `sim.model`.  On by default (measured cost: 0.7 % of the step,
        profiles/r05_ab_rb_env_params.txt); off: the model's own arrays, no randomizers, no damping change while the objects stabilise.
        `device_reset` (with `pipelined_reset`): the recipe's stage machine, the begin-of-episode state and the placement / goal sampling run in ONE more launch after the
        env kernel (ra_env_recipe_step, include/rgstep.h) instead of host numpy behind a readback of the done / goal flags: a step call never waits for the GPU.
        `randomizer_params`: name -> parameter of `build_simulation_randomizers` (the reference's ADR-controlled values; all zero by default = identity)."""
        self.B, self.N = int(batch_size), int(num_objects)
        self._L = lib if lib is not None else _native.lib()
        self.control_mode = _control_mode_name(control_mode)
        self.joint_control = self.control_mode == "joint"      # ControlMode.JOINT: no TCP solver world (RobotControlParameters.requires_solver_sim, robot_interface.py:83-91)
        self.wrist_only = self.control_mode == "tcp+wrist"      # ControlMode.TCP_WRIST: FreeWristTcpArm in the solver world, 4 arm numbers + the gripper's
        AD = 7 if self.joint_control else 6                     # the launches' action width (tcp+wrist: the six of tcp+roll+yaw with the roll number ignored)
        self.action_dim = 5 if self.wrist_only else AD          # the env's action width
        self.launch_action_dim = AD
        self.max_position_change = float(max_position_change)
        main = main_model if main_model is not None else load_blocks_model(self.N)   # (main_model: the same world with other objects, envs/rearrange/ycb.py)
        # TcpSolverMode.MOCAP (robot_interface.py:22-29): the MAIN world's arm hangs on the mocap weld itself -- MujocoIdealURGripperCompositeRobot, one world, no joint actuators
        self.tcp_solver_mode = _solver_mode_name(tcp_solver_mode)
        self.ideal_arm = self.tcp_solver_mode == "mocap" and not self.joint_control
        if self.ideal_arm:
            if main_model is not None:
                raise NotImplementedError("tcp_solver_mode mocap is built for the blocks world (its mocap-arm model is shipped; the ycb sets are compiled with joint actuators)")
            if pipelined_reset or device_reset:
                raise NotImplementedError("tcp_solver_mode mocap with pipelined resets: the synchronous reset() only (the reference's sim initialisation steps the world before the recipe)")
            main = load_blocks_model(self.N, mocap_arm=True)
        solver = None if (self.joint_control or self.ideal_arm) else load_solver_model()
        self.randomizer_params = dict(randomizer_params or {})
        self.per_env_parameters = bool(per_env_parameters)
        if self.randomizer_params and not self.per_env_parameters:
            raise ValueError("randomizer_params need per_env_parameters=True")
        self.sim = LargeModelSimulation(main, self.B, device=device, n_substeps=n_substeps, lib=lib, hand=False, env_params=self.per_env_parameters)
        self.solver_sim = None if solver is None else LargeModelSimulation(solver, self.B, device=device, n_substeps=n_substeps, lib=lib, hand=False)
        self.device = self.sim.device
        self._ext_cols = torch.tensor([0, 1, 2, 4, 5], device=self.device) if self.wrist_only else None      # tcp+wrist: the launch's columns without the roll
        self.model, self.solver_model = main, solver
        self.n_random_initial_steps, self.stabilize_steps, self.settle_steps = n_random_initial_steps, stabilize_steps, settle_steps
        self.used_table_portion = used_table_portion
        self._rng = np.random.RandomState(starting_seed)
        A, As = main.arrays, (None if solver is None else solver.arrays)
        jn, sj = main.names["joint"], (None if solver is None else solver.names["joint"])
        self.arm_q = [int(A["jnt_qposadr"][jn.index("robot0:J%d" % k)]) for k in range(1, 7)]
        self.obj_q = [int(A["jnt_qposadr"][jn.index("object%d:joint" % i)]) for i in range(self.N)]
        self.obj_v = [int(A["jnt_dofadr"][jn.index("object%d:joint" % i)]) for i in range(self.N)]
        self.grip_q = int(A["jnt_qposadr"][jn.index("robot0:r_gripper_RJ0_outer")])
        self.grip_act = main.names["actuator"].index("robot0:r_gripper_finger_joint")
        self.nq, self.nu = self.sim.nq, self.sim.nu
        self.obs_dim = 36 * self.N + 23 + 2 * self.nq
        gn = main.names["geom"]
        tb, tg = main.name2id("body", "table"), gn.index("table")
        self.table_pos, self.table_size = A["body_pos"][tb].copy(), A["geom_size"][tg].copy()
        self.table_height = float(self.table_pos[2] + self.table_size[2])
        bb = object_bounding_boxes(main, self.N)                        # per object: centre and half extents of its vertices in the body frame
        self.obj_center, self.obj_half = bb[:, :3].copy(), bb[:, 3:].copy()
        # ---- TCP hook arguments
        t = self.tcp = _native.RbTcpArgs()
        if self.joint_control:
            # JointControlledArm drives ctrl[:6] from the six arm joints (joint_controlled_arm.py:186-190), MujocoRobotiqGripper its own actuator; relative actions:
            # arm joint k around its position, +- min(ctrl range / 2, max_position_change); the gripper around its control, +- ctrl range / 2
            assert self.grip_act == 6 and self.nu == 7 and list(np.diff(self.arm_q)) == [1] * 5, "joint control: ctrl[:6] = the arm's joints, ctrl[6] = the gripper"
            assert [int(np.ravel(A["actuator_trnid"][u])[0]) for u in range(6)] == [jn.index("robot0:J%d" % k) for k in range(1, 7)], main.names["actuator"]
            P = np.zeros((7, 6), dtype=np.float32); P[:6] = np.eye(6)
            self.sim.set_action_map(self.arm_q[0], P, relative_action=True, max_position_change=float(max_position_change), ctrl_centre_mask=1 << self.grip_act)
            self.solver_arm_q, self.solver_grip_q, self.solver_grip_act = [], -1, -1
        elif self.ideal_arm:
            # the hook on the env's own world (rb_tcp_args.self_world): TCP pose + denormalised action -> mocap target, gripper target into this world's ctrl, then its mj_steps
            assert self.nu == 1 and self.grip_act == 0 and int(A["body_mocapid"][main.name2id("body", "robot0:mocap")]) == 0
            for k in range(6):
                t.arm_qposadr[k] = t.main_arm_qposadr[k] = self.arm_q[k]
            t.main_gripper_actuator = self.grip_act; t.tcp_body = main.name2id("body", "robot0:gripper_tcp"); t.wrist_joint = jn.index("robot0:J6")
            t.reset_controller_error = 0; t.self_world = 1; t.wrist_only = 1 if self.wrist_only else 0
            t.max_position_change = max_position_change; t.speed_roll = SPEED_ROLL; t.speed_pitch = SPEED_PITCH; t.joint_drift_threshold = JOINT_DRIFT_THRESHOLD
            t.gripper_ctrl_lo, t.gripper_ctrl_hi = float(A["actuator_ctrlrange"][self.grip_act, 0]), float(A["actuator_ctrlrange"][self.grip_act, 1])
            self.solver_arm_q, self.solver_grip_q, self.solver_grip_act = [], -1, -1
            self._tcp_body, self._gripper_base_body = int(t.tcp_body), main.name2id("body", "robot0:gripper_base")
        else:
            for k in range(6):
                t.arm_qposadr[k] = int(As["jnt_qposadr"][sj.index("robot0:J%d" % (k + 1))]); t.main_arm_qposadr[k] = self.arm_q[k]
            t.main_gripper_actuator = self.grip_act; t.tcp_body = solver.name2id("body", "robot0:gripper_tcp"); t.wrist_joint = sj.index("robot0:J6")
            t.reset_controller_error = 1 if arm_reset_controller_error else 0
            t.wrist_only = 1 if self.wrist_only else 0
            t.max_position_change = max_position_change; t.speed_roll = SPEED_ROLL; t.speed_pitch = SPEED_PITCH; t.joint_drift_threshold = JOINT_DRIFT_THRESHOLD
            t.gripper_ctrl_lo, t.gripper_ctrl_hi = float(A["actuator_ctrlrange"][self.grip_act, 0]), float(A["actuator_ctrlrange"][self.grip_act, 1])
            self.solver_arm_q = [int(t.arm_qposadr[k]) for k in range(6)]
            self.solver_grip_q = int(As["jnt_qposadr"][sj.index("robot0:r_gripper_RJ0_outer")])
            self.solver_grip_act = solver.names["actuator"].index("robot0:r_gripper_finger_joint")
        # ---- env-level state (device) and the post kernel's arguments
        dev, B, N = self.device, self.B, self.N
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=dev)
        self.packed = z(B, self.obs_dim + 4)
        self.t, self.steps, self.ssl, self.successes, self.consecutive = (z(B, dt=torch.int32) for _ in range(5))
        self.prev_nsucc, self.prev_valid = z(B), z(B, dt=torch.int32)
        self.goal, self.goal_rot, self.qpos_goal, self.static_obs = z(B, N, 7), z(B, N, 3), z(B, self.nq), z(B, N, 7)
        self.reward, self.goal_dist = z(B, 3), z(B, 2)
        self.done, self.goal_reset, self.trial_success, self.sub_goal_ok, self.env_crash, self.objects_off_table = (z(B, dt=torch.bool) for _ in range(6))
        self.info_ssl = z(B, dt=torch.int32)
        self.goal[:, :, 3] = 1.0
        st = dict(success_threshold or {"obj_pos": 0.04, "obj_rot": 0.2})
        pen = dict(penalty or dict(table_collision=0.0, objects_off_table=1.0, wrist_collision=0.0))
        a = self.post = _native.RaPostArgs()
        P = lambda x: ctypes.c_void_p(x.data_ptr())
        a.obs, a.obs_dim, a.num_objects = P(self.packed), self.obs_dim, N
        for name, ten in (("t", self.t), ("steps", self.steps), ("steps_since_last_goal", self.ssl), ("successes_so_far", self.successes), ("consecutive", self.consecutive),
                          ("prev_nsucc", self.prev_nsucc), ("prev_valid", self.prev_valid), ("goal", self.goal), ("goal_rot", self.goal_rot), ("qpos_goal", self.qpos_goal),
                          ("static_obs", self.static_obs), ("reward", self.reward), ("goal_dist", self.goal_dist), ("done", self.done), ("goal_reset", self.goal_reset),
                          ("trial_success", self.trial_success), ("sub_goal_ok", self.sub_goal_ok), ("env_crash", self.env_crash), ("objects_off_table", self.objects_off_table),
                          ("info_ssl", self.info_ssl)):
            setattr(a, name, P(ten))
        for i in range(N):
            a.obj_body[i] = main.name2id("body", "object%d" % i)
        a.tcp_body = main.name2id("body", "robot0:gripper_tcp")
        for k in range(6):
            a.arm_qposadr[k] = self.arm_q[k]
        a.grip_qposadr, a.grip_dofadr, a.grip_act = self.grip_q, int(A["jnt_dofadr"][jn.index("robot0:r_gripper_RJ0_outer")]), self.grip_act
        a.finger_geom[0], a.finger_geom[1] = gn.index("robot0:left_contact_v"), gn.index("robot0:right_contact_v")
        a.table_plane_geom = gn.index("table_collision_plane")
        sn = main.names["sensor"]
        a.force_adr, a.torque_adr = int(A["sensor_adr"][sn.index("toolhead_force")]), int(A["sensor_adr"][sn.index("toolhead_torque")])
        mask = 0
        for bname in ("robot0:gripper_base", "left_gripper", "left_inner_follower", "left_outer_driver", "right_gripper", "right_inner_follower", "right_outer_driver"):
            bid = main.name2id("body", bname)
            for g in range(len(gn)):
                if A["geom_bodyid"][g] == bid:
                    if g >= 64:      # (the env kernel's gripper - table scan keeps the gripper's geoms in one 64-bit mask)
                        raise ValueError("gripper geom %r has id %d: the env kernel needs the gripper's geoms below id 64" % (gn[g], g))
                    mask |= 1 << g
        a.gripper_geom_mask = mask
        lo, hi = self.table_pos - self.table_size, self.table_pos + self.table_size
        a.table_min[0], a.table_min[1], a.table_max[0], a.table_max[1], a.table_height = float(lo[0]), float(lo[1]), float(hi[0]), float(hi[1]), self.table_height
        a.pos_threshold, a.rot_threshold, a.goal_pos_offset, a.goal_rot_weight = st["obj_pos"], st["obj_rot"], 0.0, 1.0
        a.goal_reward_per_object, a.success_reward = goal_reward_per_object, success_reward
        a.penalty_table_collision, a.penalty_objects_off_table, a.penalty_safety_stop = pen.get("table_collision", 0.0), pen.get("objects_off_table", 0.0), pen.get("safety_stop", 0.0)
        a.safety_stop_force = 150.0                                      # robot/ur16e/arm_interface.py:46
        a.max_timesteps_per_goal, a.successes_needed, a.use_goal_distance_reward = max_timesteps_per_goal_per_obj * N, successes_needed, int(use_goal_distance_reward)
        a.solver_grip_qposadr, a.solver_grip_act = self.solver_grip_q, self.solver_grip_act
        self.action_shape = (self.B, self.action_dim)
        self._zero_action = z(B, AD)
        # ---- per-env model parameters: the reference's simulation randomizers (applied after _reset, robot_env.py:779-783) and stabilize_objects' damping change
        self.stabilize_object_damping = float(stabilize_object_damping)
        self.obj_dofs = torch.tensor([d for v in self.obj_v for d in range(v, v + 6)], device=dev, dtype=torch.long)
        self.randomizers = build_simulation_randomizers(main, self.randomizer_params) if self.per_env_parameters else []
        self._rand_gen = torch.Generator(device=dev); self._rand_gen.manual_seed(int(starting_seed) + 90001)
        if self.per_env_parameters:
            self._param_defaults = {k: self.sim.params[k][0].clone() for k in self.sim.params.keys()}
            self._param_block_default = self.sim.params.block[0].clone()
        # ---- RearrangeEnv.apply_wrappers (common/base.py:986-996): SmoothActionWrapper(alpha = 0.3) -> ClipRewardWrapper -> DiscretizeActionWrapper, all inside
        # the launches: the solver world's launch maps bin indices to actions and smooths them (rb_tcp_args), the post kernel clips the reward
        self.wrapped = bool(wrappers)
        self.n_action_bins = int(n_action_bins)
        self.ema_value, self.ema_t, self.action_ema = z(B, AD), z(B, dt=torch.int32), z(B, AD)
        self.bins = torch.tensor(np.tile(action_bin_array(-1.0, 1.0, self.n_action_bins, action_spacing), (AD, 1)).astype(np.float32), device=dev).contiguous()   # over Box(-1, 1)
        tw = self.tcp_wrapped = _native.RbTcpArgs()
        ctypes.memmove(ctypes.byref(tw), ctypes.byref(t), ctypes.sizeof(t))
        tw.bins, tw.nbins = self.bins.data_ptr(), self.n_action_bins
        self.ema_alpha = tw.ema_alpha = float(np.power(smooth_alpha, float(np.asarray(A["opt_timestep"]).reshape(-1)[0]) * n_substeps / 0.08))       # SmoothActionWrapper.reset (wrappers/util.py:203-211)
        tw.ema_value, tw.ema_t, tw.action_out = self.ema_value.data_ptr(), self.ema_t.data_ptr(), self.action_ema.data_ptr()
        if self.wrapped:
            a.reward_clip = float(reward_clip)
        # ---- pipelined resets: an env whose episode ended runs the reset recipe INSIDE the following step calls (the stepper's launches take everybody
        # along; a synchronous reset of a few envs costs ~210 launch pairs at the latency of a full batch).  The recipe's bookkeeping is host numpy (one
        # readback of the flags per step -- a step is 70 ms of GPU work), its physics goes through the same three launches as the live envs': `hold` /
        # `scripted` feed the recipe's actions to the solver world's launch, `frozen` keeps the env kernel from scoring those envs.
        self.pipelined = bool(pipelined_reset)
        self._stage, self._left = np.zeros(B, dtype=np.int8), np.zeros(B, dtype=np.int32)      # 0 live, 1 stabilise, 2 random action, 3 settle
        self._yaw = np.zeros((B, N))
        self.ended_rows = np.zeros(0, dtype=np.int64)          # rows whose episode ended on the last step (pipelined resets)
        self.hold, self.scripted, self.frozen, self.solver_active = z(B, dt=torch.int32), z(B, AD), z(B, dt=torch.uint8), torch.ones(B, dtype=torch.int32, device=dev)
        self.hold_ctrl = z(B, dt=torch.int32)      # joint control, pipelined resets: envs whose objects stabilise keep their stored controls (no action reaches the robot)
        self.resetting, self.episode_started = z(B, dt=torch.bool), z(B, dt=torch.bool)
        self.nticks = torch.full((B,), 2, dtype=torch.int32, device=dev)       # state-less forwards (controller ticks) per env of the main world's launch
        if self.pipelined:
            for args_ in (t, tw):
                args_.hold, args_.scripted = self.hold.data_ptr(), self.scripted.data_ptr()
            a.frozen = self.frozen.data_ptr()
        self.device_reset = bool(device_reset)
        if self.device_reset:
            if not self.pipelined:
                raise ValueError("device_reset needs pipelined_reset=True (the synchronous reset() keeps its host recipe)")
            self.stage, self.left, self.yaw, self.placement_failed = z(B, dt=torch.int32), z(B, dt=torch.int32), z(B, N), z(B, dt=torch.int32)
            self.reobserve, self.ended, self.stabilised = torch.full((B,), 2, dtype=torch.uint8, device=dev), z(B, dt=torch.bool), z(B, dt=torch.bool)
            r = self.recipe = _native.RaRecipeArgs()
            r.num_objects, r.action_dim = N, AD
            for name, ten in (("stage", self.stage), ("left", self.left), ("yaw", self.yaw), ("done", self.done), ("goal_reset", self.goal_reset), ("hold", self.hold),
                              ("hold_ctrl", self.hold_ctrl), ("solver_active", self.solver_active), ("nticks", self.nticks), ("scripted", self.scripted), ("frozen", self.frozen),
                              ("resetting", self.resetting), ("episode_started", self.episode_started), ("reobserve", self.reobserve), ("ended", self.ended),
                              ("stabilised", self.stabilised), ("placement_failed", self.placement_failed), ("t", self.t), ("steps", self.steps), ("steps_since_last_goal", self.ssl),
                              ("successes_so_far", self.successes), ("consecutive", self.consecutive), ("prev_valid", self.prev_valid), ("ema_t", self.ema_t),
                              ("ema_value", self.ema_value), ("action_ema", self.action_ema), ("goal", self.goal), ("goal_rot", self.goal_rot), ("qpos_goal", self.qpos_goal),
                              ("static_obs", self.static_obs)):
                setattr(r, name, P(ten))
            for i in range(N):
                r.obj_qposadr[i] = self.obj_q[i]
                for k in range(3):
                    r.obj_center[i][k], r.obj_half[i][k] = float(self.obj_center[i, k]), float(self.obj_half[i, k])
            for k in range(6):
                r.arm_qposadr[k], r.arm_start[k] = self.arm_q[k], float(TABLETOP_EXPERIMENT_INITIAL_POS[k])
                r.solver_arm_qposadr[k] = self.solver_arm_q[k] if self.solver_arm_q else 0
            (off_x, off_y, _), (width, height, _) = self.placement_area()
            r.area_offset[0], r.area_offset[1], r.area_size[0], r.area_size[1] = float(off_x), float(off_y), float(width), float(height)
            for k in range(3):
                r.table_pos[k], r.table_size[k] = float(self.table_pos[k]), float(self.table_size[k])
            r.stabilize_steps, r.n_random_initial_steps, r.settle_steps = int(stabilize_steps), int(n_random_initial_steps), int(settle_steps)
            r.seed, r.step = (int(starting_seed) * 2654435761 + 40503) & 0xFFFFFFFF, 0

    # ------------------------------------------------------------------ launches
    def _stream(self):
        return None if self.sim._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _post(self):
        _native.check(self._L, self._L.ra_env_post_step(self.sim._bh, None if self.solver_sim is None else self.solver_sim._bh, ctypes.byref(self.post), self._stream()), "ra_env_post_step")

    def _joint_action(self, actions, wrapped):
        """Joint control: the [B, 7] action that reaches `RobotEnv._set_action`.  With the wrapper stack: bin index -> value (DiscretizeActionWrapper.action,
        wrappers/util.py:66-70) -> exponential moving average with bias correction (SmoothActionWrapper.step / IncrementalExpAvg, util.py:142-160, 213-218) -- a few
        elementwise device ops here (in TCP mode the solver world's launch does them).  Pipelined resets: envs inside their recipe take its scripted action."""
        if wrapped:
            x = self.bins[torch.arange(self.action_dim, device=self.device)[None, :], actions.long()]
            live = (self.hold == 0) if self.pipelined else torch.ones(self.B, dtype=torch.bool, device=self.device)
            al = self.ema_alpha
            v = torch.where(live[:, None], self.ema_value * al + (1.0 - al) * x, self.ema_value)
            self.ema_value.copy_(v)
            self.ema_t += live.to(torch.int32)
            out = v / (1.0 - torch.pow(torch.full_like(v[:, :1], al), self.ema_t[:, None].to(torch.float32))).clamp_min(1.0e-30)
            self.action_ema.copy_(torch.where(live[:, None], out, self.action_ema))
            actions = self.action_ema
        if self.pipelined:
            actions = torch.where(self.hold[:, None] != 0, self.scripted, actions)
        return actions.contiguous()

    def _physics(self, actions, active=None, wrapped=False, solver_active=None, phase=None):
        """`phase`: None = both physics launches; "solver" / "main" = one of them (envs/rearrange/ycb.py GroupedYcbRearrangeEnv records the same phase of every group and
        issues it as ONE launch, rb_multi_begin / rb_multi_launch)."""
        if phase == "solver" and (self.joint_control or self.ideal_arm):
            return                        # one-world robots have no solver phase
        if self.joint_control:     # CompositeRobot.set_position_control at the head of the main world's launch; no solver world
            self._keep_act = self._joint_action(actions, wrapped)
            self.sim.env_step(action=self._keep_act, nforward_ticks=2, flags=FLAG_FULL_FORWARD, active=active, nticks=self.nticks if self.pipelined else None,
                              hold=self.hold_ctrl if self.pipelined else None)
            return
        if self.ideal_arm:         # one launch: the hook on this world, its mj_steps, the two state-less forwards (the last in full)
            args = self.tcp_wrapped if wrapped else self.tcp
            if wrapped:
                args.action_index = actions.data_ptr(); self._keep_idx = actions
            args.nforward_ticks = 2
            self.sim.step_tcp(self.sim, None if wrapped else actions, args, flags=FLAG_FULL_FORWARD, active=active)
            return
        sa = active if solver_active is None else solver_active      # (pipelined resets: envs that are settling their objects skip the solver world)
        if phase == "main":
            pass
        elif wrapped:       # `actions`: int32 bin indices through the wrapper stack's action path
            self.tcp_wrapped.action_index = actions.data_ptr()
            self._keep_idx = actions
            self.solver_sim.step_tcp(self.sim, None, self.tcp_wrapped, active=sa)
        else:
            self._keep_actions = actions
            self.solver_sim.step_tcp(self.sim, actions, self.tcp, active=sa)
        if phase == "solver":
            return
        # live envs: sim.step's forward + _observe_sync's forward = two controller ticks, the last forward in full; envs inside their reset recipe (pipelined
        # resets): `mujoco_simulation.step()` only = one tick, `self.nticks` (the recipe's last step gets the second tick: the _observe_sync that ends a reset)
        self.sim.env_step(nforward_ticks=2, flags=FLAG_FULL_FORWARD, active=active, nticks=self.nticks if self.pipelined else None)

    def _recipe_physics(self, actions, active):
        """One step of the reset recipe's robot moves, `self._set_action(action); self.mujoco_simulation.step()` (common/base.py:484-496): the TCP solver world's
        launch, the main world's mj_steps with ONE state-less forward (MjSim.step), no _observe_sync: no second forward, no gripper hand-over to the solver world."""
        if self.joint_control:
            self.sim.env_step(action=actions, nforward_ticks=1, active=active)
            return
        if self.ideal_arm:
            self.tcp.nforward_ticks = 1
            self.sim.step_tcp(self.sim, actions, self.tcp, active=active)
            return
        self.solver_sim.step_tcp(self.sim, actions, self.tcp, active=active)
        self.sim.env_step(nforward_ticks=1, active=active)

    def step(self, actions: torch.Tensor):
        """RobotEnv.step (robot_env.py:804-844): returns (obs dict of [B, ...] views, reward [B, 3], done [B], info dict of tensors).
        Unwrapped: `actions` float32 [B, 6] in [-1, 1].  With the wrapper stack (`make_env`'s default): integer bin indices [B, 6] in [0, n_action_bins)."""
        self._step_launch(actions)
        return self._step_finish()

    def _step_launch(self, actions, phase=None):
        """The step's three launches, nothing that waits for them (envs/rearrange/ycb.py GroupedYcbRearrangeEnv enqueues several groups before it finishes any; with
        `phase` = "solver" / "main" / "post" one of the three, the action prepared by the first)."""
        if phase in (None, "solver"):
            assert actions.shape == self.action_shape and actions.device == self.device
            if self.wrist_only:      # [x, y, z, wrist, gripper] -> the launch's six columns (the roll column is ignored by the hook)
                actions = torch.cat([actions[:, :3], torch.zeros_like(actions[:, :1]), actions[:, 3:]], 1)
            if self.wrapped:
                assert not actions.dtype.is_floating_point, "the wrapped env takes MultiDiscrete actions (bin indices)"
                actions = actions.to(torch.int32).contiguous()
            else:
                assert actions.dtype == torch.float32 and actions.is_contiguous()
            self._phase_actions = actions
        if phase == "post":
            return self._post()
        sa = self.solver_active if self.pipelined else None
        self._physics(self._phase_actions, wrapped=self.wrapped, solver_active=sa, phase=phase)
        if phase is None:
            self._post()

    def _step_finish(self):
        if self.pipelined and self.device_reset:
            self._advance_recipes_device()
        elif self.pipelined:
            self._advance_recipes()
        return self.observe(), self.reward, self.done, self.info()

    def info(self):
        return {"goal_dist_obj_pos": self.goal_dist[:, 0], "goal_dist_obj_rot": self.goal_dist[:, 1], "goal_reset": self.goal_reset, "trial_success": self.trial_success,
                "sub_goal_is_successful": self.sub_goal_ok, "env_crash": self.env_crash, "objects_off_table": self.objects_off_table, "successes_so_far": self.successes,
                "steps_since_last_goal": self.info_ssl, "resetting": self.resetting, "episode_started": self.episode_started}

    def observe(self, packed=None, action_ema=None):
        """Views into the packed row, keys / shapes of `RearrangeEnv._observe_simple` (common/base.py:376-421).  (`packed` / `action_ema`: rows of several envs
        put together by the caller, envs/rearrange/ycb.py.)"""
        out, o, N = {}, 0, self.N
        packed = self.packed if packed is None else packed
        for k, w in OBS_KEYS:
            if isinstance(w, str):
                n = self.nq if w == "nq" else N * int(w[1])
                v = packed[:, o:o + n]
                out[k] = v if w == "nq" else v.view(packed.shape[0], N, int(w[1]))
            else:
                n = w
                out[k] = packed[:, o:o + n]
            o += n
        assert o == self.obs_dim
        if self.wrapped:
            ema = self.action_ema if action_ema is None else action_ema
            out["action_ema"] = ema[:, self._ext_cols] if self.wrist_only else ema      # SmoothActionWrapper's observation: the smoothed action of the last step (zeros after reset)
        return out

    # ------------------------------------------------------------------ reset (host work + physics launches)
    def _grid_placement(self, yaw, rows):
        """place_objects_in_grid (common/utils.py:719-829) for boxes: AABB of each yawed block, a grid of cells sized by the largest block over the
        placement area (simulation/base.py:992-1010), distinct random cells."""
        B, N = len(rows), self.N
        half = self._aabb_half(yaw)                                      # rotate_bounding_box
        c, s_ = np.cos(yaw), np.sin(yaw)
        centre = np.stack([c * self.obj_center[:, 0] - s_ * self.obj_center[:, 1], s_ * self.obj_center[:, 0] + c * self.obj_center[:, 1], np.broadcast_to(self.obj_center[:, 2], yaw.shape)], -1)
        (off_x, off_y, _), (width, height, _) = self.placement_area()
        out = np.zeros((B, N, 3))
        xy = np.zeros((B, N, 2))
        ncol, nrow = (width // (2 * half[:, :, 0].max(1))).astype(int), (height // (2 * half[:, :, 1].max(1))).astype(int)
        crowded = ncol * nrow < N
        for r in np.nonzero(~crowded)[0]:
            cw, ch = width / ncol[r], height / nrow[r]
            cells = self._rng.permutation(ncol[r] * nrow[r])[:N]
            row_i, col_i = cells // ncol[r], cells % ncol[r]
            xy[r] = np.stack([cw * col_i + half[r, :, 0], ch * row_i + half[r, :, 1]], -1)
        if crowded.any():
            # fewer cells than objects (large mesh objects): place_objects_with_no_constraint (common/utils.py:829-880, _place_objects :623-716) --
            # uniform proposals inside the area, rejected while the object's box overlaps a placed one; a set restarts when one of its objects runs
            # out of trials (the reference gives up after max_placement_trial_count restarts and resamples the episode; this keeps trying).
            # All crowded rows are sampled together; the objects of a row are placed largest first (the reference goes in object order: with a table this
            # full that mostly runs out of trials on the last large object and starts over).
            pending = np.nonzero(crowded)[0]
            area = np.array([width, height])
            order = np.argsort(-(self.obj_half[:, 0] * self.obj_half[:, 1]), kind="stable")
            rounds = 0
            while len(pending):
                rounds += 1
                if rounds > 200:      # (the reference gives up after max_placement_trial_count restarts as well, common/utils.py:623-716)
                    raise RuntimeError("no collision-free placement for %d envs: the objects' bounding boxes cover %.0f %% of the placement area" % (
                        len(pending), 100 * float((4 * self.obj_half[:, 0] * self.obj_half[:, 1]).sum()) / (width * height)))
                h = half[pending]
                cur, alive = np.zeros((len(pending), N, 2)), np.ones(len(pending), dtype=bool)
                done_objs = []
                for i in order:
                    placed = np.zeros(len(pending), dtype=bool)
                    for _ in range(100):
                        need = np.nonzero(alive & ~placed)[0]
                        if len(need) == 0:
                            break
                        c_ = self._rng.uniform(h[need, i, :2], area - h[need, i, :2])
                        free = np.ones(len(need), dtype=bool)
                        if done_objs:
                            apart = (np.abs(c_[:, None, :] - cur[need][:, done_objs]) >= h[need, i, None, :2] + h[need][:, done_objs, :2]).any(-1)
                            free = apart.all(1)
                        cur[need[free], i] = c_[free]
                        placed[need[free]] = True
                    alive &= placed
                    done_objs.append(i)
                xy[pending[alive]] = cur[alive]
                pending = pending[~alive]
        p = np.concatenate([xy, half[:, :, 2:3] + 2 * self.table_size[2]], -1)
        out = p + [off_x, off_y, 0.0] - self.table_size + self.table_pos - centre      # (the body origin, from the centre of its bounding box)
        return out

    def placement_area(self):
        """RearrangeSimulationInterface.get_placement_area / get_table_setting (simulation/base.py:980-1010): (offset, size) of the box objects and goals are placed
        in, offset measured from the table's low corner; `used_table_portion` is clipped to at least a tenth per object."""
        tsx, tsy = 2 * self.table_size[0], 2 * self.table_size[1]
        portion = float(np.clip(self.used_table_portion, self.N * 0.1, 1.0))
        width, height = 0.5 * tsx * portion, 0.38 * tsy * portion
        return (0.5 * tsx - width / 2.0, 0.44 * tsy - height / 2.0, 2 * self.table_size[2]), (width, height, 0.26)

    def _write_goal(self, rows, goal_pos, yaw):
        dev = self.device
        idx = torch.as_tensor(rows, device=dev, dtype=torch.long)
        eul = np.zeros(goal_pos.shape); eul[..., 2] = np.mod(yaw + np.pi, 2 * np.pi) - np.pi        # mat2euler of a z-rotation, normalised as get_target_rot + normalize_angles give it
        quat = euler2quat(eul)
        g = np.concatenate([goal_pos, quat], -1).astype(np.float32)
        self.goal[idx] = torch.tensor(g, device=dev)
        self.goal_rot[idx] = torch.tensor(eul.astype(np.float32), device=dev)
        qg = self.sim.qpos[idx].clone()                                   # qpos_goal: the current qpos with the objects at their goals (object_state.py:381-389)
        for i, qa in enumerate(self.obj_q):
            qg[:, qa:qa + 7] = torch.tensor(g[:, i], device=dev)
        self.qpos_goal[idx] = qg
        self.prev_valid[idx] = 0

    def _begin_episode_state(self, rows, idx):
        """What RearrangeEnv._reset writes before anything is simulated (common/base.py:897-932): both worlds as freshly made, the arm's start pose, object
        rotations about z and their placement, bounding boxes / colours of the static observation.  Returns the drawn yaw angles [len(rows), N]."""
        dev, N = self.device, self.N
        A = self.model.arrays
        worlds = [(self.sim, A)] + ([] if self.solver_sim is None else [(self.solver_sim, self.solver_model.arrays)])
        if self.per_env_parameters:      # _recreate_sim (common/base.py:850-856): a fresh model -- the previous episode's randomised values are gone
            for k, v in self._param_defaults.items():
                self.sim.params[k][idx] = v
        # MjSim of a fresh model: qpos0, zero velocities / controller state / time; robot.reset()
        for sim, model in worlds:
            sim.qpos[idx] = torch.tensor(model["qpos0"].astype(np.float32), device=dev)
            for f in (sim.qvel, sim.ctrl, sim.pid, sim.qacc_warmstart):
                f[idx] = 0
            sim.view(_native.RG_F_TIME)[idx] = 0
            sim.view(_native.RG_F_STATUS)[idx] = 0
        arm0 = torch.tensor(TABLETOP_EXPERIMENT_INITIAL_POS.astype(np.float32), device=dev)
        if self.ideal_arm:
            self._initialize_mocap_world(idx)
        self.sim.qpos[idx[:, None], torch.tensor(self.arm_q, device=dev)] = arm0
        if self.ideal_arm:       # IdealJointControlledTcpArm.reset: joint positions, then solver.reset() = reset_mocap_welds (with its forward) + reset_mocap2body_xpos
            self._mocap_to_body(idx, self._tcp_body)
        else:
            self.sim.ctrl[idx, :6] = arm0
        if self.solver_sim is not None:
            self.solver_sim.qpos[idx[:, None], torch.tensor(self.solver_arm_q, device=dev)] = arm0
            self.solver_sim.eq_data[idx, :7] = torch.tensor([0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device=dev)     # reset_mocap_welds
        # object rotations about z and grid placement (common/base.py:613-640, 797-822)
        yaw = self._rng.uniform(0.0, 2 * np.pi, (len(rows), N))
        pos = self._grid_placement(yaw, rows)
        quat = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], -1)
        for i, qa in enumerate(self.obj_q):
            self.sim.qpos[idx, qa:qa + 7] = torch.tensor(np.concatenate([pos[:, i], quat[:, i]], -1).astype(np.float32), device=dev)
        colors = self._rng.random_sample((len(rows), N, 4)); colors[..., 3] = 1.0
        so = np.concatenate([np.broadcast_to(self._aabb_half(yaw), (len(rows), N, 3)), colors], -1)
        self.static_obs[idx] = torch.tensor(so.astype(np.float32), device=dev)
        return yaw

    def _mocap_to_body(self, idx, body):
        """reset_mocap_welds + "mocap pose <- that body's pose" for the envs `idx`: the weld's relative pose back to identity, one forward (kinematics + the controller
        tick of mj_forward), the mocap row from the body's frame (gym.envs.robotics.utils.reset_mocap_welds / reset_mocap2body_xpos)."""
        active = torch.zeros(self.B, dtype=torch.int32, device=self.device); active[idx] = 1
        self.sim.eq_data[idx, :7] = torch.tensor([0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device=self.device)
        self.sim.env_step(nsubsteps=0, nforward_ticks=1, active=active)
        xp, xq = self.sim.scratch("xpos"), self.sim.scratch("xquat")
        self.sim.mocap[idx] = torch.cat([xp[idx, 3 * body:3 * body + 3], xq[idx, 4 * body:4 * body + 4]], 1)

    def _initialize_mocap_world(self, idx):
        """RearrangeEnv._initialize_sim_state for a mocap-actuated arm (common/base.py:448-465): welds reset, forward, the mocap body put at the pose of
        robot0:gripper_base, ten simulation steps (the weld pulls the TCP there)."""
        active = torch.zeros(self.B, dtype=torch.int32, device=self.device); active[idx] = 1
        self._mocap_to_body(idx, self._gripper_base_body)
        for _ in range(10):
            self.sim.env_step(nforward_ticks=1, active=active)

    def reset(self, mask: Optional[torch.Tensor] = None):
        """RobotEnv.reset -> RearrangeEnv._reset (common/base.py:897-932): robot start pose, object rotations + grid placement, stabilisation
        (100 simulation steps), n_random_initial_steps of one random action then 100 zero-action steps, tracker reset, first goal."""
        rows = np.arange(self.B) if mask is None else np.nonzero(mask.cpu().numpy())[0]
        if len(rows) == 0:
            return self.observe()
        dev, N = self.device, self.N
        idx = torch.as_tensor(rows, device=dev, dtype=torch.long)
        active = torch.zeros(self.B, dtype=torch.int32, device=dev); active[idx] = 1
        if self.pipelined:       # a synchronous reset ends whatever recipe those envs were in
            self._stage[rows] = 0; self._left[rows] = 0
            if self.device_reset:
                self.stage[idx] = 0; self.left[idx] = 0
            self.hold[idx] = 0; self.hold_ctrl[idx] = 0; self.frozen[idx] = 0; self.solver_active[idx] = 1; self.resetting[idx] = False; self.episode_started[idx] = False
            self.nticks[idx] = 2; self._nticks_host = None
        yaw = self._begin_episode_state(rows, idx)
        # stabilize_objects (common/utils.py:76-92): the objects' dof damping is lowered to 1e-3 while they settle and restored afterwards -- with per-env parameter
        # rows; without them the model's own damping (0.01) stays (blocks at rest on the table need no help; mesh objects settle a little slower)
        self._set_object_damping(idx, self.stabilize_object_damping)
        for _ in range(self.stabilize_steps):
            self.sim.env_step(nforward_ticks=1, active=active)
        self._set_object_damping(idx, None)
        # _randomize_robot_initial_position (common/base.py:498-510)
        if self.n_random_initial_steps >= 1:
            act = torch.zeros(self.B, self.launch_action_dim, device=dev)
            act[idx] = torch.tensor(self._rng.uniform(-1, 1, (len(rows), self.launch_action_dim)).astype(np.float32), device=dev)
            for _ in range(self.n_random_initial_steps):
                self._recipe_physics(act, active)
            for _ in range(self.settle_steps):
                self._recipe_physics(self._zero_action, active)
        # tracker reset and the first goal (robot_env.py:780-792; ObjectStateGoal.next_goal with randomize_goal_rot = False)
        for f in (self.t, self.steps, self.ssl, self.successes, self.consecutive, self.ema_t):
            f[idx] = 0
        self.ema_value[idx] = 0; self.action_ema[idx] = 0          # SmoothActionWrapper.reset: a fresh filter, action_ema = 0
        self._randomize_simulation(idx)                               # simulation_randomizer.randomize AFTER _reset (robot_env.py:779-783)
        self._write_goal(rows, self._grid_placement(yaw, rows), yaw)
        self.sim.env_step(nsubsteps=0, nforward_ticks=1, flags=FLAG_FULL_FORWARD, active=active)       # the forward of _observe_sync
        # the first observation of the new episodes: the env kernel for exactly those rows (observation + gripper hand-over, zeroed reward / done, the success count
        # the first step's goal reward is measured from); the rows of everybody else -- and the reward / done / info tensors their last step() returned -- stay
        self._reobserve(rows, np.zeros(0, dtype=np.int64))
        return self.observe()

    # ------------------------------------------------------------------ pipelined resets
    def _reobserve(self, started, regoaled):
        """The observation rows of these envs again, after their goal changed, every other env skipped by the env kernel: `started` (first observation of a new
        episode: reward 0, done 0) and `regoaled` (live envs with a new goal: the observation entries only, this step's reward / done / flags stay)."""
        if len(started) + len(regoaled) == 0:
            return
        saved = self.frozen.clone()
        self.frozen.fill_(2)
        self.frozen[torch.as_tensor(started, device=self.device, dtype=torch.long)] = 1
        self.frozen[torch.as_tensor(regoaled, device=self.device, dtype=torch.long)] = 3
        had = self.post.frozen
        self.post.frozen = self.frozen.data_ptr()
        self._post()
        self.post.frozen = had
        self.frozen.copy_(saved)

    def _advance_recipes(self):
        """After the step's three launches: start the reset recipe of the envs whose episode ended, move the envs inside it one step on (stabilise ->
        random action -> settle -> episode start: tracker reset, first goal, first observation), give the live envs that reached their goal a new one."""
        dev = self.device
        done = self.done.cpu().numpy()                       # (the one host synchronisation of a step)
        newgoal = self.goal_reset.cpu().numpy()
        self.episode_started.zero_()
        st, left = self._stage, self._left
        # ---- envs inside the recipe: this step counted
        inside = st > 0
        left[inside] -= 1
        nxt = np.nonzero(inside & (left <= 0))[0]
        started = []
        was = st[nxt].copy()
        for stage in (1, 2, 3):
            rows = nxt[was == stage]
            if len(rows) == 0:
                continue
            idx = torch.as_tensor(rows, device=dev, dtype=torch.long)
            if stage == 1:
                self._set_object_damping(idx, None)                     # stabilize_objects restores the objects' damping
            if stage == 1 and self.n_random_initial_steps >= 1:         # -> one random action for n_random_initial_steps steps
                st[rows], left[rows] = 2, self.n_random_initial_steps
                self.scripted[idx] = torch.tensor(self._rng.uniform(-1, 1, (len(rows), self.launch_action_dim)).astype(np.float32), device=dev)
                self.solver_active[idx] = 1; self.hold_ctrl[idx] = 0
            elif stage == 2:                                            # -> zero action while everything settles
                st[rows], left[rows] = 3, self.settle_steps
                self.scripted[idx] = 0
            else:                                                       # -> the episode starts
                started.append(rows)
        if started:
            rows = np.concatenate(started)
            idx = torch.as_tensor(rows, device=dev, dtype=torch.long)
            st[rows], left[rows] = 0, 0
            for f in (self.t, self.steps, self.ssl, self.successes, self.consecutive, self.ema_t, self.hold):
                f[idx] = 0
            self.ema_value[idx] = 0; self.action_ema[idx] = 0; self.scripted[idx] = 0
            self.frozen[idx] = 0; self.solver_active[idx] = 1; self.hold_ctrl[idx] = 0
            self.resetting[idx] = False; self.episode_started[idx] = True
            self._randomize_simulation(idx)
            yaw = self._yaw[rows]
            self._write_goal(rows, self._grid_placement(yaw, rows), yaw)
        # ---- live envs with a reached goal: ObjectStateGoal.next_goal (what reset_goals() does on request)
        grows = np.nonzero(newgoal.astype(bool) & (st == 0) & ~done.astype(bool))[0]     # (an episode that ends on the same step keeps the reached goal's entries in its terminal observation: ra_recipe_kernel)
        if len(grows):
            yaw = self.goal_rot[torch.as_tensor(grows, device=dev, dtype=torch.long), :, 2].cpu().numpy().astype(np.float64)
            self._write_goal(grows, self._grid_placement(yaw, grows), yaw)
        self._reobserve(np.concatenate(started) if started else np.zeros(0, dtype=np.int64), grows)
        # ---- episodes that ended on this step: their recipe begins (the returned observation / reward / done are the terminal ones)
        rows = np.nonzero(done.astype(bool) & (st == 0))[0]
        self.ended_rows = rows                              # (envs/rearrange/ycb.py hands these slots out again, to other envs of the batch)
        if len(rows):
            idx = torch.as_tensor(rows, device=dev, dtype=torch.long)
            self._yaw[rows] = self._begin_episode_state(rows, idx)
            self._set_object_damping(idx, self.stabilize_object_damping)
            st[rows], left[rows] = 1, self.stabilize_steps
            self.hold[idx] = 1; self.scripted[idx] = 0; self.frozen[idx] = 4; self.solver_active[idx] = 0; self.hold_ctrl[idx] = 1
            self.resetting[idx] = True
            if self.stabilize_steps <= 0:        # (degenerate configuration: straight to the next stage on the following step)
                left[rows] = 1
        # controller ticks of the NEXT step's main-world launch: two for live envs, one inside the recipe, two on the recipe's last step (see _physics)
        last_stage = 3 if self.n_random_initial_steps >= 1 else 1
        want = np.where(st > 0, np.where((st == last_stage) & (left <= 1), 2, 1), 2).astype(np.int32)
        if not np.array_equal(want, getattr(self, "_nticks_host", None)):
            self._nticks_host = want
            self.nticks.copy_(torch.as_tensor(want, device=dev))

    def _advance_recipes_device(self):
        """`_advance_recipes` without the host: the recipe kernel (stage machine, begin-of-episode state, placement and goal sampling), tensor ops on its masks for the
        per-env parameter rows (the block restored for an episode that ended = `_recreate_sim`'s fresh model; stabilize_objects' damping change; the simulation randomizers
        for the envs whose episode starts), and the env kernel once more over the `reobserve` codes (first observation of a new episode / the new goal's entries)."""
        r = self.recipe
        r.step = (int(r.step) + 1) & 0xFFFFFFFF
        _native.check(self._L, self._L.ra_env_recipe_step(self.sim._bh, None if self.solver_sim is None else self.solver_sim._bh, ctypes.byref(r), self._stream()), "ra_env_recipe_step")
        if self.per_env_parameters:
            P = self.sim.params
            P.block.copy_(torch.where(self.ended[:, None], self._param_block_default[None, :], P.block))
            d, cols = P["dof_damping"], self.obj_dofs
            cur = d[:, cols]
            d[:, cols] = torch.where(self.ended[:, None], torch.full_like(cur, self.stabilize_object_damping),
                                     torch.where(self.stabilised[:, None], self._param_defaults["dof_damping"][cols][None, :].expand_as(cur), cur))
            for rz in self.randomizers:
                rz.randomize(self.sim, self._rand_gen, self.episode_started)
        had = self.post.frozen
        self.post.frozen = self.reobserve.data_ptr()
        self._post()
        self.post.frozen = had
        self.ended_rows = np.zeros(0, dtype=np.int64)

    def _set_object_damping(self, idx, value):
        """RearrangeSimulationInterface.set_object_damping for the envs `idx` (simulation/base.py:753-770): `value` on the objects' six dofs, None = the model's own."""
        if not self.per_env_parameters:
            return
        d = self.sim.params["dof_damping"]
        cols = self.obj_dofs
        d[idx[:, None], cols[None, :]] = self._param_defaults["dof_damping"][cols] if value is None else float(value)

    def _randomize_simulation(self, idx):
        """`randomization.simulation_randomizer.randomize(mj_sim, random_state)` for the envs `idx` (robot_env.py:779-783): every randomizer draws new values of its
        model field from the model's own (the block was restored at the start of the reset), on the device."""
        if not self.randomizers:
            return
        mask = torch.zeros(self.B, dtype=torch.bool, device=self.device); mask[idx] = True
        for r in self.randomizers:
            r.randomize(self.sim, self._rand_gen, mask)

    def _aabb_half(self, yaw):
        """half extents [.., N, 3] of the objects' bounding boxes after a rotation by `yaw` [.., N] about z"""
        sx, sy, sz = self.obj_half[:, 0], self.obj_half[:, 1], self.obj_half[:, 2]
        return np.stack([np.abs(np.cos(yaw)) * sx + np.abs(np.sin(yaw)) * sy, np.abs(np.sin(yaw)) * sx + np.abs(np.cos(yaw)) * sy, np.broadcast_to(sz, yaw.shape)], -1)

    def _observe_only(self, rows=None):
        """The observation rows of `rows` (default: everybody) from the state as it is, without a step: observation entries, gripper hand-over, zeroed
        reward / done, and -- as `_observe_sync` does through `update_goal_info` -- the success count the next step's goal reward is measured from."""
        self._reobserve(np.arange(self.B) if rows is None else np.asarray(rows), np.zeros(0, dtype=np.int64))

    def reset_goals(self):
        """`reset_goal` for the envs the tracker flagged (robot_env.py:893-909): a new ObjectStateGoal placement; host sampling."""
        rows = np.nonzero(self.goal_reset.cpu().numpy())[0]
        if len(rows) == 0:
            return
        yaw = self.goal_rot[torch.as_tensor(rows, device=self.device, dtype=torch.long), :, 2].cpu().numpy().astype(np.float64)
        self._write_goal(rows, self._grid_placement(yaw, rows), yaw)
        self._reobserve(np.zeros(0, dtype=np.int64), rows)      # robot_env.py:893-909: the observation returned after a goal reset carries the new goal

    def sync(self):
        self.sim.sync()


def build_simulation_randomizers(model, params: Optional[dict] = None):
    """`RearrangeEnv.build_simulation_randomizers` (common/base.py:1008-1092) on the batched randomizers of robogym_amd/randomization/sim.py: the same list, names,
    fields, prefixes, apply modes and coefficients; `params[name]` is the randomizer's parameter (what ADR controls in the reference; 0 = identity)."""
    from robogym_amd.randomization.sim import GenericSimRandomizer, GeomSolimpRandomizer, GeomSolrefRandomizer, GravityRandomizer, JointMarginRandomizer, PidRandomizer

    P = dict(params or {})
    A, names = model.arrays, model.names
    robot_jnt = [j for j, n in enumerate(names["joint"]) if n.startswith("robot0:")]
    robot_dof = [d for d in range(len(A["dof_jntid"])) if int(A["dof_jntid"][d]) in set(robot_jnt)]
    robot_body = [b for b, n in enumerate(names["body"]) if n.startswith("robot0:")]
    g = lambda name, default=(0.0, 0.0): P.get(name, default)
    G = GenericSimRandomizer
    out = [
        GravityRandomizer(param=float(np.atleast_1d(g("gravity", 0.0))[0])),
        JointMarginRandomizer(param=float(np.atleast_1d(g("jnt_margin", 0.0))[0])),
        G("dof_frictionloss_robot", "dof_frictionloss", "uncoupled_mean_variance", param=g("dof_frictionloss_robot"), ids=robot_dof),
        G("dof_damping_robot", "dof_damping", "uncoupled_mean_variance", param=g("dof_damping_robot"), ids=robot_dof),
        G("dof_armature_robot", "dof_armature", "uncoupled_mean_variance", param=g("dof_armature_robot"), ids=robot_dof),
        G("jnt_stiffness_robot", "jnt_stiffness", "variance_mean_additive", param=g("jnt_stiffness_robot"), coef=0.005, ids=robot_jnt, positive_only=True),
        G("body_pos_robot", "body_pos", "variance_additive", param=g("body_pos_robot", (0.0,)), coef=0.02, ids=robot_body),
    ]
    for name in ("pid_kp", "pid_ti", "pid_td", "pid_imax_clamp", "pid_error_deadband"):
        mean, std = (list(np.atleast_1d(g(name))) + [0.0, 0.0])[:2]
        out.append(PidRandomizer(name, mean=float(mean), std=float(std)))
    out += [
        G("actuator_forcerange", "actuator_forcerange", "uncoupled_mean_variance", param=g("actuator_forcerange")),
        GeomSolimpRandomizer(param=g("geom_solimp", (0.0,) * 6)),
        GeomSolrefRandomizer(param=g("geom_solref", (0.0,) * 4)),
        G("geom_margin", "geom_margin", "variance_additive", param=g("geom_margin", (0.0,)), coef=0.0005),
        G("geom_pos", "geom_pos", "variance_additive", param=g("geom_pos", (0.0,)), coef=0.002),
        G("geom_gap", "geom_gap", "max_additive", param=g("geom_gap", (0.0,)), coef=0.01),
        G("geom_friction", "geom_friction", "uncoupled_mean_variance", param=g("geom_friction")),
        G("body_mass", "body_mass", "uncoupled_mean_variance", param=g("body_mass")),
        G("body_inertia", "body_inertia", "variance_additive", param=g("body_inertia", (0.0,))),
    ]
    unknown = sorted(set(P) - {r.name for r in out})
    if unknown:
        raise KeyError("randomizer_params: no simulation randomizer named %s (known: %s)" % (", ".join(unknown), ", ".join(r.name for r in out)))
    return out


def action_bin_array(lower_bound, upper_bound, n_bins, spacing="linear"):
    """BinSpacing.get_bin_array (wrappers/util.py:17-33): the table DiscretizeActionWrapper maps a bin index through.  "linear": n_bins evenly spaced values;
    "exponential" (symmetric range, odd n_bins): -1, -1/2, -1/4, ... 0 ... 1/4, 1/2, 1 times the bound."""
    spacing = str(spacing).lower().split(".")[-1]
    if spacing == "linear":
        return np.linspace(lower_bound, upper_bound, n_bins)
    if spacing != "exponential":
        raise NotImplementedError("action_spacing %r" % spacing)
    assert lower_bound == -upper_bound and n_bins % 2 == 1, "Exponential binning is only supported on symmetric action space with an odd number of bins"
    half = np.array([2.0 ** (-n) for n in range(n_bins // 2)]) * lower_bound
    return np.concatenate([half, [0.0], -half[::-1]])


SUPPORTED_PARAMETERS = {"simulation_params", "robot_control_params", "n_random_initial_steps"}
SUPPORTED_SIMULATION_PARAMS = {"num_objects", "penalty", "used_table_portion"}
SUPPORTED_ROBOT_CONTROL_PARAMS = {"max_position_change", "arm_reset_controller_error", "control_mode", "tcp_solver_mode"}
SUPPORTED_CONSTANTS = {"success_threshold", "successes_needed", "success_reward", "max_timesteps_per_goal_per_obj", "n_action_bins", "action_spacing", "use_goal_distance_reward",
                       "goal_reward_per_object", "normalize_mesh", "randomize"}


def _control_mode_name(mode) -> str:
    """`ControlMode` value or name (robot_interface.py:9-20) -> "tcp+roll+yaw" | "tcp+wrist" | "joint"."""
    name = str(getattr(mode, "value", mode)).lower().split(".")[-1]
    if name in ("tcp+roll+yaw", "tcp_roll_yaw"):
        return "tcp+roll+yaw"
    if name in ("tcp+wrist", "tcp_wrist"):
        return "tcp+wrist"
    if name == "joint":
        return "joint"
    raise ValueError("control_mode %r is not one of the reference's ControlMode values (joint, tcp+roll+yaw, tcp+wrist; robot_interface.py:9-20)" % (mode,))


def _solver_mode_name(mode) -> str:
    """`TcpSolverMode` value or name (robot_interface.py:22-29) -> "mocap_ik" | "mocap"."""
    name = str(getattr(mode, "value", mode)).lower().split(".")[-1]
    if name in ("mocap_ik", "mocap"):
        return name
    raise ValueError("tcp_solver_mode %r is not one of the reference's TcpSolverMode values (mocap, mocap_ik)" % (mode,))


def _check_supported(parameters, sp, rc, constants):
    """A parameter or constant of the reference's env that this env does not implement is an error, not a silently ignored key (the reference's attrs classes reject
    unknown names the same way; the supported subset keeps the reference's names and meaning)."""
    for got, known, where in ((parameters, SUPPORTED_PARAMETERS, "parameters"), (sp, SUPPORTED_SIMULATION_PARAMS, "parameters.simulation_params"),
                              (rc, SUPPORTED_ROBOT_CONTROL_PARAMS, "parameters.robot_control_params"), (constants, SUPPORTED_CONSTANTS, "constants")):
        unknown = sorted(set(got) - known)
        if unknown:
            raise NotImplementedError("%s: %s not implemented by the batched rearrange env (supported: %s)" % (where, ", ".join(unknown), ", ".join(sorted(known))))
    _control_mode_name(rc.get("control_mode", "tcp+roll+yaw"))
    _solver_mode_name(rc.get("tcp_solver_mode", "mocap_ik"))


def make_env(batch_size: int = 4096, device="cuda:0", parameters=None, constants=None, starting_seed: int = 0, apply_wrappers: bool = True, **kw):
    """`BlockRearrangeEnv.build` surface (robot_env.py:1081-1089) for the batched env; `apply_wrappers` (default True, as in the reference) = the rearrange
    wrapper stack of common/base.py:986-996 (MultiDiscrete actions of `constants.n_action_bins` = 11 bins, action smoothing, reward clipping).  `parameters` / `constants` accept the subset this env
    implements (`SUPPORTED_*` below); any other name raises NotImplementedError."""
    parameters, constants = dict(parameters or {}), dict(constants or {})
    sp, rc = dict(parameters.get("simulation_params", {})), dict(parameters.get("robot_control_params", {}))
    _check_supported(parameters, sp, rc, constants)
    args = dict(num_objects=sp.get("num_objects", 5), max_position_change=rc.get("max_position_change", 0.1), arm_reset_controller_error=rc.get("arm_reset_controller_error", True),
                n_random_initial_steps=parameters.get("n_random_initial_steps", 10), starting_seed=starting_seed, wrappers=bool(apply_wrappers),
                n_action_bins=constants.get("n_action_bins", 11), control_mode=rc.get("control_mode", "tcp+roll+yaw"), tcp_solver_mode=rc.get("tcp_solver_mode", "mocap_ik"))      # (+ pipelined_reset=True through **kw: episodes restart inside the step calls)
    if "action_spacing" in constants:
        args["action_spacing"] = constants["action_spacing"]
    for k in ("success_threshold", "successes_needed", "success_reward", "max_timesteps_per_goal_per_obj", "use_goal_distance_reward", "goal_reward_per_object"):
        if k in constants:
            args[k] = constants[k]
    for k in ("penalty", "used_table_portion"):      # (a given penalty dict replaces the default one as a whole, as the reference's attrs field does)
        if k in sp:
            args[k] = sp[k]
    if constants.get("randomize", True) is False:    # RobotEnvConstants.randomize (robot_env.py:155): no simulation randomizers at all
        kw = dict(kw); kw.pop("randomizer_params", None)      # (the rows stay: stabilize_objects' damping change is not a randomizer)
    args.update(kw)
    return BatchedBlockRearrangeEnv(batch_size, device=device, **args)


def make_simple_env(*a, **kw):
    """make_env(apply_wrappers=False) (robot_env.py:1137-1139)"""
    kw["apply_wrappers"] = False
    return make_env(*a, **kw)


class SingleEnvView:
    """B = 1 view of a batched rearrange env with the reference's types (robot_env.py:757-844): observations as numpy arrays without the batch dimension, the reward as a
    LIST of three floats (env, goal, success), a Python bool `done`, an info dict of Python scalars / arrays -- what a caller of `BlockRearrangeEnv` sees.  For porting
    single-env code and tests; not the fast path.  `step` takes the action of the env's mode: floats in [-1, 1] (`make_simple_env`) or bin indices (`make_env`)."""

    def __init__(self, env: BatchedBlockRearrangeEnv):
        assert env.B == 1
        self.env = env
        self.unwrapped = self
        self.mujoco_simulation = self.sim = env.sim
        self.action_shape = env.action_shape[1:]

    @staticmethod
    def _np(t):
        a = t[0].detach().cpu().numpy()
        return a.astype(np.float32) if a.dtype.kind == "f" else a

    def _obs(self, obs):
        return {k: self._np(v) for k, v in obs.items()}

    def reset(self):
        return self._obs(self.env.reset())

    def observe(self):
        return self._obs(self.env.observe())

    def reset_goals(self):
        self.env.reset_goals()
        return self.observe()

    def step(self, action):
        a = np.asarray(action)
        t = torch.as_tensor((a.astype(np.int64) if self.env.wrapped else a.astype(np.float32))[None], device=self.env.device)
        obs, reward, done, info = self.env.step(t.contiguous())
        return self._obs(obs), [float(x) for x in reward[0]], bool(done[0]), {k: (v[0].item() if v[0].dim() == 0 else self._np(v)) for k, v in info.items()}
