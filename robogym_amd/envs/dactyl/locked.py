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


class BatchedLockedEnv:
    """B independent dactyl/locked envs stepped in lock-step on one GPU."""

    def __init__(self, batch_size: int, device="cuda:0", constants: Optional[LockedEnvConstants] = None, starting_seed: Optional[int] = None,
                 model: Optional[CompiledModel] = None, lib=None, pipelined_reset: bool = False, sort_dispatch: bool = False):
        self.constants = constants or LockedEnvConstants()
        c = self.constants
        self.model = model or load_locked_model()
        self.mujoco_simulation = LockedSimulation(self.model, batch_size, device=device, n_substeps=c.mujoco_substeps,
                                                  relative_action=c.relative_action, success_threshold=c.success_threshold["cube_quat"], lib=lib)
        sim = self.mujoco_simulation
        self.batch_size, self.device = sim.batch_size, sim.device
        self.num_actions = sim.nu
        self.seed(starting_seed)
        self.goal_generation = LockedParallelGoal(sim, self._gen)
        B, dev = self.batch_size, self.device
        self._obs_buf = torch.zeros((B, sim.obs_dim), dtype=torch.float32, device=dev)
        self._goal_dist = torch.zeros(B, dtype=torch.float32, device=dev)
        self._goal_quat = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self._goal_quat[:, 0] = 1
        self._qpos_goal = torch.zeros((B, sim.nq), dtype=torch.float32, device=dev)
        self._prev_dist = torch.zeros(B, dtype=torch.float32, device=dev)
        self._prev_valid = torch.zeros(B, dtype=torch.bool, device=dev)
        self._is_successful = torch.zeros(B, dtype=torch.bool, device=dev)
        self.multi_goal_tracker = BatchedMultiGoalTracker(B, dev, c.max_timesteps_per_goal, c.success_reward, c.successes_needed, c.use_goal_distance_reward)
        z = lambda: torch.zeros(B, dtype=torch.int32, device=dev)
        self.t = z()
        self._needs_reset = True
        # pipelined resets: finished episodes are re-initialised INSIDE the following step
        # launches (reset recipe of cube_env.py:330-355 / locked.py:197-225 as a per-env phase counter), so the
        # other envs never wait for a reset.  Off: `done` envs are the caller's to `reset(mask)` (reference API).
        self.pipelined_reset = bool(pipelined_reset)
        self._phase, self._tries = z(), z()   # 0 = live; k > 0: k-1 recipe steps done
        self._qpos0_rows = torch.tensor(self.model.qpos0, dtype=torch.float32, device=dev).repeat(B, 1)
        self._cube_pos_col = int(sim.qpos_idxs["cube_position"][0])
        self._cube_quat_col = int(sim.qpos_idxs["cube_rotation"][0])
        self._zero_ctrl_rows = (0.5 * (sim.ctrl_lo + sim.ctrl_hi)).repeat(B, 1)   # denormalize_position_control(zero action), absolute
        self.sort_dispatch = bool(sort_dispatch)

    # ------------------------------------------------------------------ gym surface
    @property
    def action_space_shape(self) -> Tuple[int]:
        return (self.num_actions,)

    def seed(self, seed=None):
        self._seed = 0 if seed is None else int(seed)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self._seed)
        if hasattr(self, "goal_generation"):
            self.goal_generation.gen = self._gen
        return [self._seed]

    def _rand_normal(self, *shape):
        return torch.randn(*shape, generator=self._gen, device=self.device)

    def _rand_uniform(self, lo, hi, *shape):
        return lo + (hi - lo) * torch.rand(*shape, generator=self._gen, device=self.device)

    def reset(self, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """RobotEnv.reset (robot_env.py:757-792) for the envs selected by `mask` (default: all)."""
        sim, c = self.mujoco_simulation, self.constants
        B, dev = self.batch_size, self.device
        mask = torch.ones(B, dtype=torch.bool, device=dev) if mask is None else mask.to(dev).bool()
        self.t = torch.where(mask, torch.zeros_like(self.t), self.t)
        self._randomize_cube_pose(mask)
        # tracker.reset + reset_goal_generation -> reset_goal (robot_env.py:787-792, 893-909)
        self.multi_goal_tracker.reset(mask)
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
        """CubeEnv._reset + LockedEnv._randomize_cube_initial_position (cube_env.py:330-355, locked.py:197-225)."""
        sim, c = self.mujoco_simulation, self.constants
        B = self.batch_size
        need = mask.clone()
        for _ in range(c.max_pose_resets):
            active = need.to(torch.int32).contiguous()
            self._masked_sim_reset(need)
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
        """reset_goal (robot_env.py:893-909): count the goal, sample it, re-observe (2 state-less forwards)."""
        sim = self.mujoco_simulation
        g = self.goal_generation.next_goal()
        m1 = mask[:, None]
        self._goal_quat = torch.where(m1, g["cube_quat"], self._goal_quat).contiguous()
        self._qpos_goal = torch.where(m1, g["qpos_goal"], self._qpos_goal)
        self.multi_goal_tracker.reset_goal_steps(mask)
        sim.env_step(goal_quat=self._goal_quat, obs=self._obs_buf, goal_dist=self._goal_dist, active=mask.to(torch.int32).contiguous(),
                     nsubsteps=0, nforward_ticks=2)
        # _previous_goal_distance = None, then update_goal_info sets it to the current distance
        self._prev_dist = torch.where(mask, self._goal_dist, self._prev_dist)
        self._prev_valid = self._prev_valid | mask
        self._is_successful = torch.where(mask, self._goal_dist < self.constants.success_threshold["cube_quat"], self._is_successful)

    def observe(self) -> Dict[str, torch.Tensor]:
        """Keys, order and shapes of `LockedEnv._default_observation_map` (locked.py:132-146)."""
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
            "goal_pos": torch.zeros((self.batch_size, 3), dtype=torch.float32, device=self.device),
            "goal_quat": rotation.quat_normalize(self._goal_quat),
            "qpos_goal": self._qpos_goal,
            "is_goal_achieved": self._is_successful.to(torch.int32)[:, None],
        }

    def _crashed(self) -> torch.Tensor:
        """Envs whose simulation raised BAD_STATE (NaN / diverged state).  The reference fails loudly there
        (mujoco-py raises MujocoException / MuJoCo auto-resets, warning_buffer.py:15-24); a batch cannot raise for one
        env, so the env reports done with `info["env_crash"]`, zero reward and a zeroed observation row, and is
        re-initialised by the next reset.  Read through the zero-copy status view: stream-ordered, no host sync."""
        return (self.mujoco_simulation.view(_native.RG_F_STATUS)[:, 0] & _native.RG_STATUS_BAD_STATE) != 0

    def _dispatch_order(self):
        """Longest-expected-first dispatch (rg_step_args.order_dev): the envs sorted by the cycles their previous
        env.step took, so the launch's tail is made of short envs."""
        if not self.sort_dispatch:
            return None
        return torch.argsort(self.mujoco_simulation.view(_native.RG_F_COST)[:, 0], descending=True).to(torch.int32)

    @property
    def packed_dim(self) -> int:
        return self.mujoco_simulation.obs_dim + 3 + 4 + self.mujoco_simulation.nq + 1 + 3 + 1

    def packed_observation(self, reward: torch.Tensor, done: torch.Tensor) -> torch.Tensor:
        """[B, 170]: the 166 scalars of the observation dict in key order (locked.py:132-146) followed by the reward
        triple and `done` — the one buffer a replicated learner needs per env.step, and what the multi-GPU all-gather
        moves ("rewards/dones ride in the same buffer")."""
        B, dev = self.batch_size, self.device
        return torch.cat([self._obs_buf, torch.zeros((B, 3), dtype=torch.float32, device=dev), rotation.quat_normalize(self._goal_quat), self._qpos_goal,
                          self._is_successful.to(torch.float32)[:, None], reward, done.to(torch.float32)[:, None]], dim=1)

    def step(self, action: torch.Tensor):
        """RobotEnv.step (robot_env.py:804-844): returns (obs dict, reward [B,3], done [B], info dict)."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        sim, c = self.mujoco_simulation, self.constants
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.batch_size, self.num_actions).contiguous()
        if self.pipelined_reset:
            return self._step_pipelined(action)
        sim.env_step(action=action, goal_quat=self._goal_quat, obs=self._obs_buf, goal_dist=self._goal_dist, nforward_ticks=3, order=self._dispatch_order())
        self.t += 1
        crash = self._crashed()
        self._obs_buf.masked_fill_(crash[:, None], 0.0)
        dist = torch.where(crash, torch.zeros_like(self._goal_dist), self._goal_dist)
        # _get_goal_info (robot_env.py:577-625)
        goal_distance_reward = torch.where(self._prev_valid & ~crash, self._prev_dist - dist, torch.zeros_like(dist))
        self._prev_dist = dist.clone()
        self._prev_valid = torch.ones_like(self._prev_valid)
        is_successful = (dist < c.success_threshold["cube_quat"]) & ~crash
        self._is_successful = is_successful
        goal_dist_before = dist.clone()
        reward, done, new_goal, info = self.multi_goal_tracker.process(is_successful, goal_distance_reward)
        done = done | crash
        info["env_crash"] = crash
        self._new_goal(new_goal)
        info.update({"goal_dist": {"cube_quat": goal_dist_before}, "goal_achieved": is_successful, "goals_so_far": self.multi_goal_tracker.goals_so_far.clone(),
                     "sim_status": sim.view(_native.RG_F_STATUS)[:, 0].clone()})
        return self.observe(), reward, done, info

    def _step_pipelined(self, action: torch.Tensor):
        """`step` with the reset recipe of finished episodes folded into the step launches.  No host
        synchronisation: every decision is a [B] tensor op, every state write a masked row copy on the stream.
        Envs in the recipe ignore `action` (the launch's `hold` mask keeps their scripted ctrl), report zero reward,
        `done` False and `info["resetting"]` True; the step on which the recipe completes returns the first
        observation of the new episode.  The recipe is the reference's, tick for tick: a recipe step is `sim.step`
        (10 mj_step + ONE state-less forward, simulation_interface.py:176-189), the forward after the cube
        perturbation (locked.py:213) and the one inside `on_palm` (cube_utils.py:17-23) are the second tick of recipe
        steps 20 and 30 (a tick only touches the PID state, which does not see the cube)."""
        sim, c = self.mujoco_simulation, self.constants
        B, dev, F = self.batch_size, self.device, _native
        resetting = self._phase > 0
        live = ~resetting
        n1, n2 = c.reset_initial_steps, c.reset_initial_steps + c.n_random_initial_steps
        two = resetting & ((self._phase == n1) | (self._phase == n2))
        nticks = torch.where(live, torch.full_like(self._phase, 3), torch.where(two, torch.full_like(self._phase, 2), torch.ones_like(self._phase))).contiguous()
        hold = resetting.to(torch.int32).contiguous()
        sim.env_step(action=action, goal_quat=self._goal_quat, obs=self._obs_buf, goal_dist=self._goal_dist, nforward_ticks=3,
                     hold=hold, nticks=nticks, order=self._dispatch_order(), large_mask=hold)   # the recipe runs in the large kernel configuration
        self.t += live.to(torch.int32)
        crash = self._crashed()
        self._obs_buf.masked_fill_(crash[:, None], 0.0)
        dist = torch.where(crash, torch.zeros_like(self._goal_dist), self._goal_dist)
        ok_live = live & ~crash
        goal_distance_reward = torch.where(self._prev_valid & ok_live, self._prev_dist - dist, torch.zeros_like(dist))
        self._prev_dist = torch.where(live, dist, self._prev_dist)
        self._prev_valid = self._prev_valid | live
        is_successful = (dist < c.success_threshold["cube_quat"]) & ok_live
        self._is_successful = is_successful
        goal_dist_before = dist.clone()
        reward, done, new_goal, info = self.multi_goal_tracker.process(is_successful, goal_distance_reward, live=live)
        done = done | (crash & live)
        info["env_crash"] = crash
        # ---- recipe progression of the envs that are being reset
        ph = self._phase + resetting.to(torch.int32)
        wiggle = (ph == c.reset_initial_steps + 1) & ~crash
        w = self._rand_normal(B, 4)
        sim.copy_rows(F.RG_F_QPOS, self._obs_buf[:, 0:3] + self._rand_normal(B, 3) * c.cube_position_wiggle_std, wiggle, self._cube_pos_col)
        sim.copy_rows(F.RG_F_QPOS, rotation.quat_normalize(w / w.norm(dim=-1, keepdim=True)), wiggle, self._cube_quat_col)
        sim.copy_rows(F.RG_F_CTRL, sim.denormalize_position_control(self._rand_uniform(-1.0, 1.0, B, self.num_actions)), wiggle)
        finished = (ph == c.reset_initial_steps + c.n_random_initial_steps + 1) & ~crash
        on_palm = (sim.cube_body_z + self._obs_buf[:, 2]) > 0.04
        ok = finished & (on_palm | (self._tries + 1 >= c.max_pose_resets))
        retry = (finished & ~ok) | (crash & resetting)
        start = (done & live) | (crash & live)
        restart = retry | start
        self._tries = torch.where(start, torch.zeros_like(self._tries), self._tries + retry.to(torch.int32))
        self._phase = torch.where(restart, torch.ones_like(ph), torch.where(ok, torch.zeros_like(ph), ph))
        self._masked_sim_reset(restart)
        sim.copy_rows(F.RG_F_CTRL, self._zero_ctrl_rows, restart)
        # ---- envs whose recipe completed start their episode: tracker, clock, goal (robot_env.py:787-792)
        self.multi_goal_tracker.reset(ok)
        self.t = torch.where(ok, torch.zeros_like(self.t), self.t)
        self._prev_valid = self._prev_valid & ~ok
        self._new_goal(new_goal | ok)
        info.update({"goal_dist": {"cube_quat": goal_dist_before}, "goal_achieved": is_successful, "goals_so_far": self.multi_goal_tracker.goals_so_far.clone(),
                     "resetting": self._phase > 0, "episode_started": ok, "sim_status": sim.view(F.RG_F_STATUS)[:, 0].clone()})
        return self.observe(), reward, done, info

    # ------------------------------------------------------------------ diagnostics
    def sim_status(self) -> torch.Tensor:
        return self.mujoco_simulation.status


def make_env(batch_size: int = 1, device="cuda:0", constants=None, parameters=None, starting_seed=None, apply_wrappers=True, **kwargs):
    """`LockedEnv.build` (locked.py:305, robot_env.py:1081-1139) for a batch of envs.

    The wrapper stack of the reference (`apply_wrappers=True`) is not part of the hot path built
    here; the returned env is the unwrapped one in both cases."""
    if isinstance(constants, dict):
        constants = LockedEnvConstants(**constants)
    return BatchedLockedEnv(batch_size, device=device, constants=constants, starting_seed=starting_seed, **kwargs)


def make_simple_env(batch_size: int = 1, device="cuda:0", constants=None, parameters=None, starting_seed=None, **kwargs):
    """`make_simple_env` (locked.py:304): no wrappers."""
    return make_env(batch_size, device, constants, parameters, starting_seed, apply_wrappers=False, **kwargs)
