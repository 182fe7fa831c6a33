"""dactyl/locked — Shadow hand + locked (solid) cube, batched on MI355X.

Drop-in for the reference's `robogym/envs/dactyl/locked.py` hot path: `make_env()` /
`make_simple_env()` return an env whose `step / reset / observe` keep the reference's keys, shapes
and reward / done / info semantics (robot_env.py:757-844, multi_goal_tracker.py:157-277) with a
leading batch dimension and torch tensors resident in HBM.  The physics, action map, observation
readout and goal distance run in one HIP kernel launch per env.step (include/rgstep.h).

Model assembly mirrors `LockedSimulation` / `CubeSimulationInterface.build`
(locked.py:70-123, cube_env.py:171-218).  A compiled copy of the model ships in
`robogym_amd/models/dactyl_locked.npz` because the robogym asset tree is not redistributed here;
`tools/compile_models.py` regenerates it from an asset checkout.
"""
import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from robogym_amd import _native

from robogym_amd.mujoco.mjcf_compiler import CompiledModel
from robogym_amd.mujoco.mujoco_xml import MujocoXML
from robogym_amd.mujoco.simulation_interface import BatchedSimulationInterface
from robogym_amd.utils import rotation
from robogym_amd.utils.multi_goal_tracker import BatchedMultiGoalTracker

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "models")

FINGERTIP_SITE_NAMES = ["S_fftip", "S_mftip", "S_rftip", "S_lftip", "S_thtip"]
REFERENCE_SITE_NAMES = ["phasespace_ref0", "phasespace_ref1", "phasespace_ref2"]


def build_locked_xml(cube_xml_path="rubik/rubik_locked.xml") -> MujocoXML:
    """The merged MJCF document of dactyl/locked (needs the robogym asset tree)."""
    xml = MujocoXML()
    xml.add_default_compiler_directive()
    xml.append(
        MujocoXML.parse(cube_xml_path)
        .remove_objects_by_name("annotation:outer_bound")
        .add_name_prefix("cube:")
        .set_named_objects_attr("cube:middle", tag="body", pos=[1.0, 0.87, 0.2])
        .set_named_objects_attr("cube:middle", tag="geom", density=421.0)
    )
    xml.append(
        MujocoXML.parse(cube_xml_path)
        .remove_objects_by_name("annotation:outer_bound")
        .add_name_prefix("target:")
        .set_named_objects_attr("target:middle", tag="body", pos=[1.0, 0.87, 0.2])
        .set_objects_attr(tag="geom", group="2", conaffinity="0", contype="0")
    )
    xml.append(MujocoXML.parse("floor/basic_floor.xml").set_named_objects_attr("floor", tag="body", pos=[1, 1, 0]))
    xml.append(
        MujocoXML.parse("robot/shadowhand/main.xml")
        .add_name_prefix("robot0:")
        .set_named_objects_attr("robot0:hand_mount", tag="body", pos=[1.0, 1.25, 0.15], euler=[np.pi / 2, 0, np.pi])
        .remove_objects_by_name("robot0:annotation:outer_bound")
        .remove_objects_by_name("robot0:hand_base")
    )
    return xml


def load_locked_model(recompile: bool = False) -> CompiledModel:
    path = os.path.join(MODEL_DIR, "dactyl_locked.npz")
    if not recompile and os.path.exists(path):
        return CompiledModel.load(path)
    return build_locked_xml().build()


def position_to_control_matrix(model: CompiledModel, hand_prefix="robot0:") -> np.ndarray:
    """[nu][n_hand_joints] map joint positions -> control (hand_interface.py:245-266): a joint
    transmission selects its joint, a fixed-tendon transmission sums the tendon's joints."""
    A = model.arrays
    hand_joints = [j for j, n in enumerate(model.names["joint"]) if n.startswith(hand_prefix)]
    P = np.zeros((len(A["actuator_trntype"]), len(hand_joints)))
    for u in range(P.shape[0]):
        if A["actuator_trntype"][u] == 0:
            P[u, hand_joints.index(int(A["actuator_trnid"][u]))] = 1.0
        else:
            t = int(A["actuator_trnid"][u])
            for w in range(A["tendon_adr"][t], A["tendon_adr"][t] + A["tendon_num"][t]):
                P[u, hand_joints.index(int(A["wrap_objid"][w]))] = 1.0
    return P


class LockedSimulation(BatchedSimulationInterface):
    """Batched `LockedSimulation` (locked.py:70-123): joint groups + the env descriptor of the kernel."""

    def __init__(self, model, batch_size, device="cuda:0", n_substeps=10, relative_action=True, success_threshold=0.4, lib=None):
        super().__init__(model, batch_size, device=device, n_substeps=n_substeps, lib=lib)
        self.register_joint_group("cube_position", prefix="cube:cube_t")
        self.register_joint_group("cube_rotation", prefix="cube:cube_rot")
        self.register_joint_group("target_position", prefix="target:cube_t")
        self.register_joint_group("target_rotation", prefix="target:cube_rot")
        self.register_joint_group("target_all_joints", prefix="target:")
        self.register_joint_group("hand_angle", prefix="robot0:")
        m = model
        hand_q = self.qpos_idxs["hand_angle"]
        tq, tv = self.qpos_idxs["target_all_joints"], self.qvel_idxs["target_all_joints"]
        assert (np.diff(hand_q) == 1).all() and (np.diff(tq) == 1).all() and (np.diff(tv) == 1).all()
        ints = [int(hand_q[0]), len(hand_q), int(self.qpos_idxs["cube_position"][0]), int(self.qpos_idxs["cube_rotation"][0]),
                int(tq[0]), len(tq), int(tv[0]), len(tv), m.name2id("body", "cube:middle")]
        ints += [m.name2id("site", "robot0:" + s) for s in REFERENCE_SITE_NAMES]
        ints += [m.name2id("site", "robot0:" + s) for s in FINGERTIP_SITE_NAMES]
        ints += [1 if relative_action else 0, 0, 0]
        self.pos_to_ctrl = position_to_control_matrix(m)
        self.set_env(ints, self.pos_to_ctrl, success_threshold)
        self.ctrl_lo = torch.tensor(m.actuator_ctrlrange[:, 0], dtype=torch.float32, device=self.device)
        self.ctrl_hi = torch.tensor(m.actuator_ctrlrange[:, 1], dtype=torch.float32, device=self.device)
        self.cube_body_z = float(m.body_pos[m.name2id("body", "cube:middle")][2])

    def denormalize_position_control(self, action: torch.Tensor, relative_action: bool = False) -> torch.Tensor:
        """robot_interface.py:247-278 for the whole batch (host-side use: resets)."""
        lo, hi = self.ctrl_lo, self.ctrl_hi
        if self._params is not None:    # the envs' own ranges (RandomizedJointLimitWrapper rewrites them per episode; the
            cr = self._params["actuator_ctrlrange"]   # reference reads sim.model.actuator_ctrlrange, mujoco_shadow_hand.py:105-113)
            lo, hi = cr[..., 0], cr[..., 1]
        if relative_action:
            p2c = torch.tensor(self.pos_to_ctrl, dtype=torch.float32, device=self.device)
            centre = self.get_qpos("hand_angle") @ p2c.T
        else:
            centre = 0.5 * (hi + lo)
        return torch.minimum(torch.maximum(centre + action * 0.5 * (hi - lo), lo), hi)

    def is_cube_on_palm(self) -> torch.Tensor:
        """cube_utils.on_palm (cube_utils.py:17-23): site cube:center z > 0.04 (after a forward())."""
        self.forward()
        z = self.cube_body_z + self.get_qpos("cube_position")[:, 2]
        return z > 0.04


class LockedParallelGoal:
    """Batched `LockedParallelGoal` (envs/dactyl/goals/locked_parallel.py:12-79)."""

    def __init__(self, sim: LockedSimulation, generator: torch.Generator):
        self.sim = sim
        self.gen = generator
        self.parallel_quats = torch.tensor(rotation.parallel_quats_np(), dtype=torch.float32, device=sim.device)

    def next_goal(self) -> Dict[str, torch.Tensor]:
        B, dev = self.sim.batch_size, self.sim.device
        angle = (torch.rand(B, generator=self.gen, device=dev) * 2 - 1) * np.pi
        z_quat = torch.stack([torch.cos(angle / 2), torch.zeros_like(angle), torch.zeros_like(angle), torch.sin(angle / 2)], dim=-1)
        z_quat = rotation.quat_normalize(z_quat)
        choice = torch.randint(0, 24, (B,), generator=self.gen, device=dev)
        goal_quat = rotation.quat_mul(z_quat, self.parallel_quats[choice])
        qpos_goal = torch.zeros((B, self.sim.nq), dtype=torch.float32, device=dev)
        qpos_goal[:, torch.as_tensor(self.sim.qpos_idxs["cube_rotation"], device=dev)] = goal_quat
        qpos_goal[:, torch.as_tensor(self.sim.qpos_idxs["cube_position"], device=dev)] = torch.tensor([0.0, 0.0, -0.025], device=dev)
        return {"cube_quat": goal_quat, "qpos_goal": qpos_goal}

    @staticmethod
    def goal_distance(goal_quat: torch.Tensor, cube_quat: torch.Tensor) -> torch.Tensor:
        return rotation.quat_magnitude(rotation.quat_difference(goal_quat, cube_quat))


@dataclass
class LockedEnvConstants:
    """The constants of the reference that shape the hot path (cube_env.py:61-124, robot_env.py:104-195, locked.py:49-67)."""

    mujoco_substeps: int = 10
    relative_action: bool = True
    successes_needed: int = 50
    max_timesteps_per_goal: int = 400
    success_reward: float = 5.0
    use_goal_distance_reward: bool = True
    success_threshold: Dict[str, float] = field(default_factory=lambda: {"cube_quat": 0.4})
    max_pose_resets: int = 50
    reset_initial_steps: int = 20
    n_random_initial_steps: int = 10
    cube_position_wiggle_std: float = 0.005


# With pipelined resets the envs inside the reset recipe go through the substep-granular dispatch like everybody else (rollout capacities, scripted ctrl by
# `hold`, per-env forward ticks; an env.step that exceeds the capacities -- the cube landing in the closing hand -- is handed over MID-STEP to the large
# configuration, flags bit 8).  RG_PIPE_ITEMS=0 restores round 2's routing: the recipe's envs in a large-configuration launch on a side stream beside a
# classic rollout launch (1.35 M vs 1.59 M env-steps/s over bench.py --pipelined-reset --steps 60 --warmup 40).
_PIPE_ITEMS = os.environ.get("RG_PIPE_ITEMS", "1") == "1"


class BatchedLockedEnv:
    """B independent dactyl/locked envs stepped in lock-step on one GPU.

    `step` is two launches for the whole batch and nothing else: the physics kernel (rg_batch_step_ex: action map, 10 x
    mj_step, forward ticks, observation rows, goal distances) and the env kernel (rg_env_post_step: reward, success,
    MultiGoalTracker, goal resampling, `done`, and with `pipelined_reset` the reset recipe's phase machine).  Every
    returned tensor is a view of a buffer those kernels wrote; no host synchronisation anywhere."""

    def __init__(self, batch_size: int, device="cuda:0", constants: Optional[LockedEnvConstants] = None, starting_seed: Optional[int] = None,
                 model: Optional[CompiledModel] = None, lib=None, pipelined_reset: bool = False, sort_dispatch: bool = False, parameters=None):
        self.constants = constants or LockedEnvConstants()
        c = self.constants
        self.parameters = parameters
        self.model = model or load_locked_model()
        self.mujoco_simulation = LockedSimulation(self.model, batch_size, device=device, n_substeps=c.mujoco_substeps,
                                                  relative_action=c.relative_action, success_threshold=c.success_threshold["cube_quat"], lib=lib)
        sim = self.mujoco_simulation
        self.batch_size, self.device = sim.batch_size, sim.device
        self.num_actions = sim.nu
        self.seed(starting_seed)
        self.goal_generation = LockedParallelGoal(sim, self._gen)
        B, dev = self.batch_size, self.device
        f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
        b8 = lambda: torch.zeros(B, dtype=torch.bool, device=dev)
        self._obs_buf, self._goal_dist = f32(B, sim.obs_dim), f32(B)
        self._goal_quat = f32(B, 4)
        self._goal_quat[:, 0] = 1
        self._qpos_goal, self._goal_pos = f32(B, sim.nq), f32(B, 3)
        self._prev_dist, self._prev_valid, self._is_successful = f32(B), i32(B), i32(B)
        self.multi_goal_tracker = BatchedMultiGoalTracker(B, dev, c.max_timesteps_per_goal, c.success_reward, c.successes_needed, c.use_goal_distance_reward)
        self.t = i32(B)
        self._needs_reset = True
        # pipelined resets: finished episodes are re-initialised INSIDE the following step
        # launches (reset recipe of cube_env.py:330-355 / locked.py:197-225 as a per-env phase counter), so the
        # other envs never wait for a reset.  Off: `done` envs are the caller's to `reset(mask)` (reference API).
        self.pipelined_reset = bool(pipelined_reset)
        self._phase, self._tries = i32(B), i32(B)   # 0 = live; k > 0: k-1 recipe steps done
        self._preticks = i32(B)
        self._nticks, self._reset_mask, self._live_mask = torch.full((B,), 3, dtype=torch.int32, device=dev), i32(B), torch.ones(B, dtype=torch.int32, device=dev)
        self._reward, self._goal_dist_before, self._info_ssl = f32(B, 3), f32(B), i32(B)
        self._flags = {k: b8() for k in ("done", "goal_reset", "trial_success", "sub_goal_ok", "env_crash", "resetting", "episode_started")}
        self._packed = f32(B, self.packed_dim)
        self._qpos0_rows = torch.tensor(self.model.qpos0, dtype=torch.float32, device=dev).repeat(B, 1)
        self._cube_pos_col = int(sim.qpos_idxs["cube_position"][0])
        self._cube_quat_col = int(sim.qpos_idxs["cube_rotation"][0])
        self._zero_ctrl_rows = (0.5 * (sim.ctrl_lo + sim.ctrl_hi)).repeat(B, 1)   # denormalize_position_control(zero action), absolute
        self.sort_dispatch = bool(sort_dispatch)
        self.stop_on_fall = False    # set by the wrapper stack (StopOnFallWrapper): a dropped cube ends the episode
        self.launch_flags = 0        # diagnostic: extra rg_step_args.flags of the step launches (e.g. 4: without the per-pair collision cache)
        self._order, self._order_age = None, 0
        self._draws = None           # test hook: [B, RG_POST_NDRAW] draws instead of the counter-based generator
        self._goal_override = None   # test hook / scripted goals: [B, 4]
        self._step_count = 0
        self._consts = dict(parallel=self.goal_generation.parallel_quats.contiguous(), qpos0=self._qpos0_rows[0].contiguous(),
                            zero_ctrl=self._zero_ctrl_rows[0].contiguous(), lo=sim.ctrl_lo.contiguous(), hi=sim.ctrl_hi.contiguous())

    # ------------------------------------------------------------------ gym surface
    @property
    def action_space_shape(self) -> Tuple[int]:
        return (self.num_actions,)

    @property
    def action_space(self):
        """`Box(-1, 1, (nu,), float32)` of the reference (robot_env.py:381-385) as a plain description (gym is not a dependency)."""
        return {"low": -1.0, "high": 1.0, "shape": (self.num_actions,), "dtype": "float32"}

    def seed(self, seed=None):
        self._seed = 0 if seed is None else int(seed)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self._seed)
        if hasattr(self, "goal_generation"):
            self.goal_generation.gen = self._gen
        return [self._seed]

    def set_draws(self, draws: Optional[torch.Tensor]):
        """Inject the random draws of the next steps ([B, RG_POST_NDRAW]: u_angle, u_choice, 4 + 3 normals, nu actions) — a
        deterministic test feeds the same numbers to the oracle; None: the in-kernel counter-based generator."""
        self._draws = None if draws is None else torch.as_tensor(draws, dtype=torch.float32, device=self.device).reshape(self.batch_size, _native.RG_POST_NDRAW).contiguous()

    def set_goal_override(self, goals: Optional[torch.Tensor]):
        """Scripted goals: an env that needs a new goal during the next steps receives goals[e] ([B, 4] quaternions)."""
        self._goal_override = None if goals is None else torch.as_tensor(goals, dtype=torch.float32, device=self.device).reshape(self.batch_size, 4).contiguous()

    def _rand_normal(self, *shape):
        return torch.randn(*shape, generator=self._gen, device=self.device)

    def _rand_uniform(self, lo, hi, *shape):
        return lo + (hi - lo) * torch.rand(*shape, generator=self._gen, device=self.device)

    def reset(self, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """RobotEnv.reset (robot_env.py:757-792) for the envs selected by `mask` (default: all)."""
        B, dev = self.batch_size, self.device
        mask = torch.ones(B, dtype=torch.bool, device=dev) if mask is None else mask.to(dev).bool()
        self.t.masked_fill_(mask, 0)
        self._phase.masked_fill_(mask, 0); self._tries.masked_fill_(mask, 0); self._preticks.masked_fill_(mask, 0)
        self._randomize_cube_pose(mask)
        # tracker.reset + reset_goal_generation -> reset_goal (robot_env.py:787-792, 893-909)
        self.multi_goal_tracker.reset(mask)
        self._prev_valid.masked_fill_(mask, 0)
        self._new_goal(mask)
        self._needs_reset = False
        return self.observe()

    def _masked_sim_reset(self, mask):
        """`MjSim.reset` (mj_resetData) of the selected envs: stream-ordered masked row copies, no host sync."""
        sim, F, B, dev = self.mujoco_simulation, _native, self.batch_size, self.device
        sim.copy_rows(F.RG_F_QPOS, self._qpos0_rows, mask)
        for fld, n in ((F.RG_F_QVEL, sim.nv), (F.RG_F_PID, 3 * sim.nu), (F.RG_F_WARMSTART, sim.nv), (F.RG_F_CTRL, sim.nu), (F.RG_F_TIME, 1)):
            sim.copy_rows(fld, torch.zeros((B, n), dtype=torch.float32, device=dev), mask)
        sim.copy_rows(F.RG_F_STATUS, torch.zeros((B, 1), dtype=torch.int32, device=dev), mask)

    def _randomize_cube_pose(self, mask):
        """CubeEnv._reset + LockedEnv._randomize_cube_initial_position (cube_env.py:330-355, locked.py:197-225), synchronous
        form (reference API `reset()`): the recipe steps run in the large kernel configuration."""
        sim, c = self.mujoco_simulation, self.constants
        B = self.batch_size
        need = mask.clone()
        for _ in range(c.max_pose_resets):
            active = need.to(torch.int32).contiguous()
            self._masked_sim_reset(need)
            sim.set_constants(active)   # cube_env.py:346-349: parameters are in the model by now (the wrappers write them before env.reset) -> mj_setConst
            zero = torch.zeros((B, self.num_actions), dtype=torch.float32, device=self.device)
            self._set_ctrl_masked(sim.denormalize_position_control(zero), need)
            for _ in range(c.reset_initial_steps):
                sim.step(active=active, capacity="large")
            sim.add_qpos("cube_position", self._rand_normal(B, 3) * c.cube_position_wiggle_std, need)
            w = self._rand_normal(B, 4)
            sim.set_qpos("cube_rotation", rotation.quat_normalize(w / w.norm(dim=-1, keepdim=True)), need)
            sim.forward(active=active)
            action = self._rand_uniform(-1.0, 1.0, B, self.num_actions)
            self._set_ctrl_masked(sim.denormalize_position_control(action), need)
            for _ in range(c.n_random_initial_steps):
                sim.step(active=active, capacity="large")
            sim.forward(active=active)  # the forward() inside cube_utils.on_palm
            z = sim.cube_body_z + sim.get_qpos("cube_position")[:, 2]
            need = need & ~(z > 0.04)
            if not bool(need.any()):
                break

    def _set_ctrl_masked(self, ctrl, mask):
        self.mujoco_simulation.copy_rows(_native.RG_F_CTRL, ctrl, mask)

    def _new_goal(self, mask):
        """reset_goal (robot_env.py:893-909) outside `step` (the synchronous `reset`): count the goal, sample it, re-observe
        (2 state-less forwards).  Inside `step` the env kernel does the same for the envs whose tracker asks for it."""
        sim = self.mujoco_simulation
        if self._goal_override is not None:
            gq = self._goal_override
            qg = torch.zeros((self.batch_size, sim.nq), dtype=torch.float32, device=self.device)
            qg[:, self._cube_quat_col:self._cube_quat_col + 4] = gq
            qg[:, self._cube_pos_col + 2] = -0.025
            g = {"cube_quat": gq, "qpos_goal": qg}
        else:
            g = self.goal_generation.next_goal()
        m1 = mask[:, None]
        gq = rotation.quat_normalize(g["cube_quat"])   # stored with w >= 0 (what the goal_quat observation reports)
        self._goal_quat.copy_(torch.where(m1, gq, self._goal_quat))
        qg = g["qpos_goal"].clone()
        qg[:, self._cube_quat_col:self._cube_quat_col + 4] = gq
        self._qpos_goal.copy_(torch.where(m1, qg, self._qpos_goal))
        self.multi_goal_tracker.reset_goal_steps(mask)
        sim.env_step(goal_quat=self._goal_quat, obs=self._obs_buf, goal_dist=self._goal_dist, active=mask.to(torch.int32).contiguous(),
                     nsubsteps=0, nforward_ticks=2)
        # _previous_goal_distance = None, then update_goal_info sets it to the current distance
        self._prev_dist.copy_(torch.where(mask, self._goal_dist, self._prev_dist))
        self._prev_valid.masked_fill_(mask, 1)
        self._is_successful.copy_(torch.where(mask, (self._goal_dist < self.constants.success_threshold["cube_quat"]).to(torch.int32), self._is_successful))

    def observe(self) -> Dict[str, torch.Tensor]:
        """Keys, order and shapes of `LockedEnv._default_observation_map` (locked.py:132-146); every value is a view."""
        sim = self.mujoco_simulation
        o, nq, nv = self._obs_buf, sim.nq, sim.nv
        nh = len(sim.qpos_idxs["hand_angle"])
        a = 7
        return {
            "cube_pos": o[:, 0:3],
            "cube_quat": o[:, 3:7],
            "qpos": o[:, a:a + nq],
            "qvel": o[:, a + nq:a + nq + nv],
            "hand_angle": o[:, a + nq + nv:a + nq + nv + nh],
            "fingertip_pos": o[:, a + nq + nv + nh:a + nq + nv + nh + 15],
            "goal_pos": self._goal_pos,
            "goal_quat": self._goal_quat,
            "qpos_goal": self._qpos_goal,
            "is_goal_achieved": self._is_successful[:, None],
        }

    def goal_info(self):
        """RobotEnv.goal_info (robot_env.py:911): (goal-distance reward, is_successful, info) of the last step, batched."""
        return self._reward[:, 1], self._is_successful.bool(), {"goal_dist": {"cube_quat": self._goal_dist_before}, "goal": {"cube_quat": self._goal_quat, "qpos_goal": self._qpos_goal}}

    @property
    def packed_dim(self) -> int:
        return self.mujoco_simulation.obs_dim + 3 + 4 + self.mujoco_simulation.nq + 1 + 3 + 1

    def packed_observation(self, reward: Optional[torch.Tensor] = None, done: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B, 170]: the 166 scalars of the observation dict in key order (locked.py:132-146) followed by the reward
        triple and `done` — the one buffer a replicated learner needs per env.step, and what the multi-GPU all-gather
        moves ("rewards/dones ride in the same buffer").  Written by the env kernel of the last `step`."""
        return self._packed

    _sort_every = int(os.environ.get("RG_SORT_EVERY", "1"))

    def _dispatch_order(self):
        """Longest-expected-first dispatch (rg_step_args.order_dev): the envs sorted by the cycles their previous
        env.step took, so the launch's tail is made of short envs.  Re-sorted every step (every 4th: -1.3 %; RG_SORT_EVERY overrides)."""
        from robogym_amd.mujoco import simulation_interface as _si
        if _si.SUBSTEP_ITEMS and (not self.pipelined_reset or _PIPE_ITEMS) and not self.mujoco_simulation._emul:
            return None      # (the substep-granular dispatch keeps the slots full whatever the order: sorting measured +-0, profiles/r03_ab.txt)
        if not self.sort_dispatch:
            return None
        if self._order is None or self._order_age >= self._sort_every:
            self._order = torch.argsort(self.mujoco_simulation.view(_native.RG_F_COST)[:, 0], descending=True).to(torch.int32)
            self._order_age = 0
        self._order_age += 1
        return self._order

    def _post_args(self):
        sim, c, tr, F = self.mujoco_simulation, self.constants, self.multi_goal_tracker, self._flags
        a = _native.PostArgs()
        P = lambda t: t.data_ptr()
        a.goal_dist, a.obs, a.obs_dim = P(self._goal_dist), P(self._obs_buf), sim.obs_dim
        a.t, a.phase, a.tries = P(self.t), P(self._phase), P(self._tries)
        a.steps, a.steps_since_last_goal, a.successes_so_far = P(tr.steps), P(tr.steps_since_last_goal), P(tr.successes_so_far)
        a.goals_so_far, a.consecutive = P(tr.goals_so_far), P(tr.consecutive_success)
        a.prev_dist, a.prev_valid, a.is_successful = P(self._prev_dist), P(self._prev_valid), P(self._is_successful)
        a.goal_quat, a.qpos_goal, a.preticks, a.reward = P(self._goal_quat), P(self._qpos_goal), P(self._preticks), P(self._reward)
        for k, t in F.items():
            setattr(a, k, P(t))
        a.info_ssl, a.nticks_next, a.reset_mask, a.live_mask = P(self._info_ssl), P(self._nticks), P(self._reset_mask), P(self._live_mask)
        a.goal_dist_before, a.packed = P(self._goal_dist_before), P(self._packed)
        a.draws = None if self._draws is None else P(self._draws)
        a.goal_override = None if self._goal_override is None else P(self._goal_override)
        a.seed, a.step = self._seed & 0xFFFFFFFF, self._step_count & 0xFFFFFFFF
        K = self._consts
        a.parallel_quats, a.qpos0, a.zero_ctrl, a.ctrl_lo, a.ctrl_hi = P(K["parallel"]), P(K["qpos0"]), P(K["zero_ctrl"]), P(K["lo"]), P(K["hi"])
        a.success_threshold, a.success_reward = float(c.success_threshold["cube_quat"]), float(c.success_reward)
        a.wiggle_std, a.cube_body_z = float(c.cube_position_wiggle_std), float(sim.cube_body_z)
        a.max_timesteps_per_goal, a.successes_needed, a.use_goal_distance_reward = int(c.max_timesteps_per_goal), int(c.successes_needed), int(c.use_goal_distance_reward)
        a.pipelined, a.reset_initial_steps, a.n_random_initial_steps, a.max_pose_resets = int(self.pipelined_reset), int(c.reset_initial_steps), int(c.n_random_initial_steps), int(c.max_pose_resets)
        a.cube_pos_col, a.cube_quat_col = self._cube_pos_col, self._cube_quat_col
        a.stop_on_fall = int(self.stop_on_fall)
        return a

    def step(self, action: torch.Tensor):
        """RobotEnv.step (robot_env.py:804-844): returns (obs dict, reward [B,3], done [B], info dict).

        With `pipelined_reset`, envs in the reset recipe ignore `action` (the launch's `hold` mask keeps their scripted
        ctrl), report zero reward, `done` False and `info["resetting"]` True; the step on which the recipe completes
        returns the first observation of the new episode.  The recipe is the reference's, tick for tick: a recipe step is
        `sim.step` (10 mj_step + ONE state-less forward, simulation_interface.py:176-189), the forward after the cube
        perturbation (locked.py:213) and the one inside `on_palm` (cube_utils.py:17-23) are the second tick of recipe steps
        20 and 30 (a tick only touches the PID state, which does not see the cube).
        A crashed simulation (BAD_STATE: NaN / diverged state; the reference raises MujocoException there,
        warning_buffer.py:15-24) reports `done` with `info["env_crash"]`, zero reward and a zeroed observation row."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        sim = self.mujoco_simulation
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.batch_size, self.num_actions).contiguous()
        pipe = self.pipelined_reset
        sim.env_step(action=action, goal_quat=self._goal_quat, obs=self._obs_buf, goal_dist=self._goal_dist, nforward_ticks=3, flags=self.launch_flags,
                     hold=self._reset_mask if pipe else None, nticks=self._nticks if pipe else None, order=self._dispatch_order(),
                     large_mask=self._reset_mask if (pipe and not _PIPE_ITEMS) else None, small_mask=self._live_mask if (pipe and not _PIPE_ITEMS) else None, preticks=self._preticks)
        a = self._post_args()
        stream = None if sim._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _native.check(sim._L, sim._L.rg_env_post_step(sim._bh, ctypes.byref(a), stream), "rg_env_post_step")
        self._step_count += 1
        F, tr = self._flags, self.multi_goal_tracker
        info = {"goal_dist": {"cube_quat": self._goal_dist_before}, "goal_achieved": F["sub_goal_ok"],
                "sub_goal_is_successful": F["sub_goal_ok"], "trial_success": F["trial_success"], "goal_reset": F["goal_reset"],
                "successes_so_far": tr.successes_so_far, "steps_since_last_goal": self._info_ssl, "goals_so_far": tr.goals_so_far,
                "env_crash": F["env_crash"], "resetting": F["resetting"], "episode_started": F["episode_started"],
                "sim_status": sim.view(_native.RG_F_STATUS)[:, 0]}
        return self.observe(), self._reward, F["done"], info

    # ------------------------------------------------------------------ diagnostics
    def sim_status(self) -> torch.Tensor:
        return self.mujoco_simulation.status


#: constants / parameters of the reference's LockedEnv (locked.py:42-67, cube_env.py:61-124, robot_env.py:104-195) this env honours
_PARAMETER_FIELDS = {"n_random_initial_steps", "cube_position_wiggle_std"}
_IGNORED_CONSTANTS = {"randomize": None, "vision_observations": False, "vision_goal": False, "goal_generation": "state", "render_mode": None,
                      "max_steps_goal_unreachable": None, "mujoco_timestep": 0.008}   # accepted when they hold the value the built path implements


def _constants_from(constants, parameters) -> LockedEnvConstants:
    kw = {}
    if isinstance(constants, LockedEnvConstants):
        return constants
    for src, kind in ((constants or {}), "constants"), ((parameters or {}), "parameters"):
        src = dict(src) if isinstance(src, dict) else {k: getattr(src, k) for k in dir(src) if not k.startswith("_")}
        for k, v in src.items():
            if k in LockedEnvConstants.__dataclass_fields__:
                kw[k] = v
            elif k in _IGNORED_CONSTANTS and (_IGNORED_CONSTANTS[k] is None or v == _IGNORED_CONSTANTS[k]):
                continue
            elif k == "simulation_params" and isinstance(v, dict) and not v:
                continue
            else:
                raise NotImplementedError("make_env(%s={%r: %r}) is not supported by the MI355X-native dactyl/locked path "
                                          "(supported: %s)" % (kind, k, v, sorted(LockedEnvConstants.__dataclass_fields__)))
    return LockedEnvConstants(**kw)


def make_env(parameters=None, constants=None, wrapper_params=None, starting_seed=None, apply_wrappers=True, batch_size: int = 1, device="cuda:0", **kwargs):
    """`LockedEnv.build` (locked.py:305, robot_env.py:1081-1139) for a batch of envs: the reference's signature
    `make_env(parameters=None, constants=None, wrapper_params=None, starting_seed=None, apply_wrappers=True)` plus
    `batch_size` / `device`.  `constants` / `parameters` accept the reference's names for everything the built path
    implements and raise for the rest (no silent narrowing).  `apply_wrappers=True` wraps the env in the reference's
    default wrapper stack, vectorised (robogym_amd/wrappers/dactyl_cube.py; dactyl_cube_wrappers.py:8-91): MultiDiscrete
    actions of 11 bins, drop penalty / done on fall, noisy_* / relative_goal / unified goal observations, sin / cos
    angles, clipping, previous action and reward observations, and — `constants["randomize"]`, default True as in the
    reference (robot_env.py:155) — backlash, the thirteen physics / latency randomizations of LockedEnv, observation and action
    noise, occluded / freezing markers.  Without wrappers `randomize` has no effect, as in the reference (cube_env.py:369).
    With `pipelined_reset=True` the wrapped env restarts finished episodes by itself (wrapper `auto_reset`): the randomizations of an
    env are redrawn on the step its episode ends, the reset recipe then runs with them inside the following steps."""
    wc = {}
    if isinstance(constants, dict):   # wrapper-level constants of DactylCubeEnvConstants (cube_env.py:61-124)
        constants = dict(constants)
        for k in ("randomize", "n_action_bins", "relative_goal_wrapper", "drop_reward", "min_episode_length", "fixed_wrist"):
            if k in constants and (apply_wrappers or k != "randomize"):
                wc[k] = constants.pop(k)
    env = BatchedLockedEnv(batch_size, device=device, constants=_constants_from(constants, parameters), starting_seed=starting_seed, parameters=parameters, **kwargs)
    if not apply_wrappers:
        return env
    from robogym_amd.wrappers.dactyl_cube import BatchedDactylCubeWrappers

    wp = dict(wrapper_params or {})
    for k in ("insert_above", "insert_below", "replace", "delete", "wrappers", "adr_wrapper"):
        if wp.get(k):
            raise NotImplementedError("wrapper_params[%r]: editing the wrapper list is not supported (the stack is one vectorised object)" % k)
        wp.pop(k, None)
    env.stop_on_fall = True
    return BatchedDactylCubeWrappers(env, **{"randomize": True, "auto_reset": bool(kwargs.get("pipelined_reset", False)), **wc, **wp})


def make_simple_env(parameters=None, constants=None, starting_seed=None, batch_size: int = 1, device="cuda:0", **kwargs):
    """`make_simple_env` (locked.py:304): no wrappers."""
    return make_env(parameters=parameters, constants=constants, starting_seed=starting_seed, apply_wrappers=False, batch_size=batch_size, device=device, **kwargs)


class SingleEnvView:
    """B = 1 view of the batched env with the reference's types: numpy arrays without the batch dimension, a reward LIST of
    three floats, a Python bool `done`, an info dict of Python scalars — what a reference caller of `RobotEnv` sees
    (robot_env.py:757-844).  For porting single-env code and the reference's own tests; not the fast path."""

    def __init__(self, env: "BatchedLockedEnv"):
        assert env.batch_size == 1
        self.env = env
        self.unwrapped = self
        self.mujoco_simulation = env.mujoco_simulation
        self.sim = env.mujoco_simulation
        self.action_space = env.action_space

    @staticmethod
    def _np(t):
        a = t[0].detach().cpu().numpy()
        return a.astype(np.float32) if a.dtype.kind == "f" else a

    def _obs(self, obs):
        return {k: self._np(v) for k, v in obs.items()}

    def seed(self, seed=None):
        return self.env.seed(seed)

    def reset(self):
        return self._obs(self.env.reset())

    def observe(self):
        return self._obs(self.env.observe())

    def step(self, action):
        obs, reward, done, info = self.env.step(torch.as_tensor(np.asarray(action, dtype=np.float32)[None], device=self.env.device))
        out = {}
        for k, v in info.items():
            out[k] = {kk: float(vv[0]) for kk, vv in v.items()} if isinstance(v, dict) else v[0].item()
        return self._obs(obs), [float(x) for x in reward[0]], bool(done[0]), out

    def goal_info(self):
        r, s_, info = self.env.goal_info()
        return float(r[0]), bool(s_[0]), {"goal_dist": {k: float(v[0]) for k, v in info["goal_dist"].items()}, "goal": {k: self._np(v) for k, v in info["goal"].items()}}
