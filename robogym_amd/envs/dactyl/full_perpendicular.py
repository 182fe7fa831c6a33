"""dactyl/full_perpendicular (BASELINE.json configs[2]) — the Shadow hand with the full Rubik's cube: 26 cubelets on hinge
chains around a free-floating core (nq 170, nv 168, 135 bodies, 117 geoms of which 101 are mesh hulls, condim 6 on the
cubelets).

* MODEL: assembly exactly as `FullPerpendicularSimulation` (/root/reference/robogym/envs/dactyl/full_perpendicular.py:97-117 on
  top of cube_env.py:171-218: the perpendicular cube as "cube:" and, with collisions off, as "target:", spring joints removed,
  floor, hand without its base joint) through the same `MujocoXML` edit calls and the same MJCF compiler (incl. MuJoCo's legacy
  `.msh` mesh format).
* PHYSICS: `rb_step_kernel` (robogym_amd/csrc/rb_kernel.h) through `LargeModelSimulation`.
* ENV: `BatchedFullPerpendicularEnv` — `FullPerpendicularEnv` (full_perpendicular.py:157-420) with its default goal generation
  `face_free` for a batch of envs: `step` is two launches (`rb_batch_step`, `rb_env_post_step`: FaceFreeGoal distances, reward,
  success, MultiGoalTracker, goal generation incl. the target cube's joints, observation row) and no host synchronisation;
  `reset(mask)` is the reference's recipe (cube_env.py:330-355, full_perpendicular.py:286-345) for the selected envs: 20 steps
  under the zero action, cube pose perturbation, scramble, face-angle randomisation, 10 steps under one random action, retried
  while the cube is not on the palm; with `pipelined_reset=True` finished episodes run that recipe by themselves inside the following
  `step` calls (a per-env phase counter in the env kernel, scripted controls through the stepper's `hold` mask), so the other envs never
  wait for a reset.  Goal generation: `face_free` (the default), `full_unconstrained` and `face_curr` (goals/face_free.py, full_unconstrained.py,
  face_curriculum.py).  Unwrapped (`make_simple_env`); the other goal generators of the reference (the four that need pycuber's solver:
  face_cube_solver, release_cube_solver, unconstrained_cube_solver, fixed_fair_scramble) and the wrapper stack are not built for this config.
  The scramble needs the cube-group bookkeeping the reference takes from `pycuber` (not installed here): `scramble_euler`
  applies the same 12 face turns (L, L', R, ... clockwise seen from outside the face) to signed permutation matrices.
"""
import os

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from robogym_amd import _native
from robogym_amd.envs.dactyl.locked import FINGERTIP_SITE_NAMES, MODEL_DIR, REFERENCE_SITE_NAMES
from robogym_amd.mujoco.large_simulation import LargeModelSimulation
from robogym_amd.mujoco.mjcf_compiler import CompiledModel
from robogym_amd.mujoco.mujoco_xml import MujocoXML


def build_full_perpendicular_xml(cube_xml_path: str = "rubik/rubik_perpendicular.xml") -> MujocoXML:
    """The merged MJCF document of dactyl/full_perpendicular (needs the robogym asset tree)."""
    xml = MujocoXML()
    xml.add_default_compiler_directive()
    xml.append(
        MujocoXML.parse(cube_xml_path)
        .add_name_prefix("cube:")
        .set_named_objects_attr("cube:middle", tag="body", pos=[1.0, 0.87, 0.2])
        .remove_objects_by_prefix(prefix="cube:cubelet:spring:", tag="joint")        # "Delete springs for now"
    )
    xml.append(
        MujocoXML.parse(cube_xml_path)
        .add_name_prefix("target:")
        .set_named_objects_attr("target:middle", tag="body", pos=[1.0, 0.87, 0.2])
        .remove_objects_by_prefix(prefix="target:cubelet:spring:", tag="joint")
        .set_objects_attr(tag="geom", group="2", conaffinity="0", contype="0")
    )
    xml.append(MujocoXML.parse("floor/basic_floor.xml").set_named_objects_attr("floor", tag="body", pos=[1, 1, 0]))
    xml.append(
        MujocoXML.parse("robot/shadowhand/main.xml")
        .add_name_prefix("robot0:")
        .set_objects_attr(tag="size", njmax=2000, nconmax=200, nuserdata=100, nuser_actuator=20)     # cube_env.py:239-242, applied at :199
        .set_named_objects_attr("robot0:hand_mount", tag="body", pos=[1.0, 1.25, 0.15], euler=[np.pi / 2, 0, np.pi])
        .remove_objects_by_name("robot0:annotation:outer_bound")
        .remove_objects_by_name("robot0:hand_base")
    )
    return xml


def load_full_perpendicular_model(recompile: bool = False) -> CompiledModel:
    path = os.path.join(MODEL_DIR, "dactyl_full_perpendicular.npz")
    if not recompile and os.path.exists(path):
        return CompiledModel.load(path)
    return build_full_perpendicular_xml().build()


# ------------------------------------------------------------------------------------------------ cube tables (host)
FACE_GEOM_NAMES = ["cube:cubelet:%s_%s" % (s_, a) for a in "xyz" for s_ in ("neg", "pos")]       # FullPerpendicularEnv.FACE_GEOM_NAMES
PYCUBER_ACTIONS = ["L", "L'", "R", "R'", "F", "F'", "B", "B'", "D", "D'", "U", "U'"]           # FullPerpendicularEnv.PYCUBER_ACTIONS
_FACE_OF_LETTER = {"L": (0, 0), "R": (0, 1), "F": (1, 0), "B": (1, 1), "D": (2, 0), "U": (2, 1)}   # cube_manipulator.py:8-15


def cube_tables(model: CompiledModel, prefix: str):
    """Joint tables of one 3x3x3 cube as `rb_env_post_step` / `rb_cube_ops` take them: the qpos column of the first of its 66
    cubelet joints (6 face drivers neg_x .. pos_z, then 20 x (rotx, roty, rotz)) and, per edge / corner cubelet, the offsets of its
    three hinges inside that block and its coordinates in {-1, 0, 1}^3 (CubeManipulator.__init__, cube_manipulator.py:66-146)."""
    names, A = model.names["joint"], model.arrays
    adr = lambda n: int(A["jnt_qposadr"][names.index(prefix + n)])
    drivers = [adr("cubelet:driver:%s_%s" % (s_, a)) for a in "xyz" for s_ in ("neg", "pos")]
    col = drivers[0]
    if drivers != list(range(col, col + 6)):
        raise ValueError("the face drivers of %r are not the first six joints of its cubelet block" % prefix)
    tab = []
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            for z in (-1, 0, 1):
                c = (x, y, z)
                if sum(abs(v) for v in c) < 2:
                    continue
                nm = "_".join("%s_%s" % ("neg" if v < 0 else "pos", a) for a, v in zip("xyz", c) if v)
                off = [adr("cubelet:rot%s:%s" % (a, nm)) - col for a in "xyz"]
                if min(off) < 6 or max(off) >= 66:
                    raise ValueError("cubelet joints of %r are not one contiguous block of 66" % prefix)
                tab.append(off + list(c))
    return col, np.array(tab, dtype=np.int32)


def _euler_of_matrix(m: np.ndarray) -> np.ndarray:
    """rotation.mat2euler (rotation.py:129-148) for batches of matrices whose entries are exactly -1, 0, 1."""
    m = m.astype(np.float64)
    cy = np.sqrt(m[..., 2, 2] ** 2 + m[..., 1, 2] ** 2)
    ok = cy > 0.5
    e = np.empty(m.shape[:-2] + (3,))
    e[..., 2] = np.where(ok, -np.arctan2(m[..., 0, 1], m[..., 0, 0]), -np.arctan2(-m[..., 1, 0], m[..., 1, 1]))
    e[..., 1] = -np.arctan2(-m[..., 0, 2], cy)
    e[..., 0] = np.where(ok, -np.arctan2(m[..., 1, 2], m[..., 2, 2]), 0.0)
    return e


def scramble_euler(tab: np.ndarray, actions: np.ndarray) -> np.ndarray:
    """`_scramble_cube` (full_perpendicular.py:286-294) for a batch: `actions` [B, n] indexes PYCUBER_ACTIONS; returns the
    [B, 60] hinge angles `CubeManipulator.from_pycuber` would write (cube_manipulator.py:189-289; the drivers stay zero),
    in the block's joint order.  A cubelet's orientation is a signed permutation matrix; a clockwise quarter turn of a face, seen
    from outside, is -90 degrees about its outward normal, applied to the cubelets currently on that face."""
    actions = np.asarray(actions)
    B = actions.shape[0]
    coords = tab[:, 3:6].astype(np.int64)                                   # [20, 3]
    mats = np.broadcast_to(np.eye(3, dtype=np.int64), (B, len(tab), 3, 3)).copy()
    turn = np.zeros((12, 3, 3), dtype=np.int64); face = np.zeros((12, 2), dtype=np.int64)
    for k, a in enumerate(PYCUBER_ACTIONS):
        axis, side = _FACE_OF_LETTER[a[0]]
        quarter = -(2 * side - 1) * (-1 if a.endswith("'") else 1)        # +1: +90 degrees about +axis
        i, j = (axis + 1) % 3, (axis + 2) % 3
        turn[k] = np.eye(3); turn[k, i, i] = turn[k, j, j] = 0; turn[k, j, i] = quarter; turn[k, i, j] = -quarter
        face[k] = axis, 2 * side - 1
    for t in range(actions.shape[1]):
        a = actions[:, t]
        cur = np.einsum("bkij,kj->bki", mats, coords)                       # where every cubelet sits now
        on_face = np.take_along_axis(cur, face[a, 0][:, None, None], axis=2)[..., 0] * face[a, 1][:, None] > 0
        turned = np.einsum("bij,bkjl->bkil", turn[a], mats)
        mats = np.where(on_face[..., None, None], turned, mats)
    e = _euler_of_matrix(mats)                                              # [B, 20, 3] (x, y, z)
    out = np.zeros((B, 60))
    for k in range(len(tab)):
        out[:, tab[k, :3] - 6] = e[:, k]
    return out


def _quat_to_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def face_up_quats(model: CompiledModel, candidates: np.ndarray) -> np.ndarray:
    """cube_utils.face_up_quats (cube_utils.py:41-63): per face geom, the axis-aligned orientation of the cube's ball joint that
    lifts it highest.  The reference turns the simulation to each of the 24 orientations and reads geom_xpos; with the cubelet
    joints at zero that height is (R(q) offset)_z for the geom's offset from the ball joint, which is what is evaluated here.
    Four orientations tie per face (the turns about the vertical): the first of `candidates` within 1e-9 is taken (the reference's
    pick among them hangs on MuJoCo's rounding; FaceFreeGoal multiplies by a uniform turn about the vertical anyway)."""
    A, N = model.arrays, model.names
    jb = int(A["jnt_bodyid"][N["joint"].index("cube:cube:rot")])
    anchor = np.array(A["jnt_pos"][N["joint"].index("cube:cube:rot")], dtype=np.float64)
    out = []
    for name in FACE_GEOM_NAMES:
        g = N["geom"].index(name)
        p = np.array(A["geom_pos"][g], dtype=np.float64)
        b = int(A["geom_bodyid"][g])
        while b != jb:                                                      # up the chain with every joint at zero
            p = np.array(A["body_pos"][b], dtype=np.float64) + _quat_to_matrix(np.array(A["body_quat"][b], dtype=np.float64)) @ p
            b = int(A["body_parentid"][b])
            if b == 0:
                raise ValueError("%s does not hang off the cube's ball joint" % name)
        z = np.array([(_quat_to_matrix(q) @ (p - anchor))[2] for q in candidates])
        out.append(candidates[int(np.argmax(z >= z.max() - 1e-9))])
    return np.array(out)


# ------------------------------------------------------------------------------------------------ simulation
class FullPerpendicularSimulation(LargeModelSimulation):
    """Batched `FullPerpendicularSimulation` (full_perpendicular.py:92-155): joint groups, the two cube models' tables on the
    device, `CubeManipulator` operations as a launch."""

    def __init__(self, model: CompiledModel, batch_size: int, **kw):
        super().__init__(model, batch_size, **kw)
        for name, prefix in (("cube_position", "cube:cube:t"), ("cube_rotation", "cube:cube:rot"), ("cube_drivers", "cube:cubelet:driver:"),
                             ("cube_cubelets", "cube:cubelet:"), ("target_position", "target:cube:t"), ("target_rotation", "target:cube:rot"),
                             ("target_drivers", "target:cubelet:driver:"), ("target_cubelets", "target:cubelet:"), ("cube_all_joints", "cube:"),
                             ("target_all_joints", "target:")):
            self.register_joint_group(name, prefix)
        self.cube_col, tab = cube_tables(model, "cube:")
        self.target_col, tab_t = cube_tables(model, "target:")
        if not np.array_equal(tab, tab_t):
            raise ValueError("cube and target differ in their joint layout")
        self.cube_tab_np = tab
        self.cube_tab = torch.tensor(tab, dtype=torch.int32, device=self.device).contiguous()
        N = model.names
        self.face_geoms = [N["geom"].index(n) for n in FACE_GEOM_NAMES]
        self.tip_sites = [N["site"].index("robot0:" + s_) for s_ in FINGERTIP_SITE_NAMES]
        self.ref_sites = [N["site"].index("robot0:" + s_) for s_ in REFERENCE_SITE_NAMES]
        self.center_site = N["site"].index("cube:center")
        self.cube_body_z = float(model.arrays["body_pos"][N["body"].index("cube:middle")][2])      # site cube:center sits at the body's origin; cube_position is relative to it
        self._idx = {}

    def _cols(self, group):
        if group not in self._idx:
            self._idx[group] = torch.as_tensor(self.qpos_idxs[group], dtype=torch.long, device=self.device)
        return self._idx[group]

    def get_qpos(self, group: str) -> torch.Tensor:
        return self.qpos[:, self._cols(group)]

    def set_qpos(self, group: str, value: torch.Tensor, mask: Optional[torch.Tensor] = None):
        cols = self._cols(group)
        value = torch.as_tensor(value, dtype=torch.float32, device=self.device).expand(self.batch_size, len(cols))
        self.qpos[:, cols] = value if mask is None else torch.where(mask[:, None], value, self.qpos[:, cols])

    def get_face_angles(self, target: str) -> torch.Tensor:
        return self.get_qpos("%s_drivers" % target)

    def cube_ops(self, which: str, ops: torch.Tensor, active: Optional[torch.Tensor] = None):
        """CubeManipulator operations on `which` in ("cube", "target"): ops [B, n, 4] = (axis, side, angle, code), see rb_cube_ops."""
        ops = torch.as_tensor(ops, dtype=torch.float32, device=self.device).contiguous()
        assert ops.shape[0] == self.batch_size and ops.shape[2] == 4
        assert active is None or (active.dtype == torch.int32 and active.is_contiguous())
        stream = None if self._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._keep_ops = (ops, active)
        _native.check(self._L, self._L.rb_cube_ops(self._bh, self.cube_col if which == "cube" else self.target_col, ctypes.c_void_p(self.cube_tab.data_ptr()),
                                                     ctypes.c_void_p(ops.data_ptr()), int(ops.shape[1]), None if active is None else ctypes.c_void_p(active.data_ptr()), stream), "rb_cube_ops")

    def forward(self, active=None):
        """SimulationInterface.forward: a state-less forward (kinematics of the stored state into the scratch row + one PID tick)."""
        self.env_step(active=active, nsubsteps=0, nforward_ticks=1)


# ------------------------------------------------------------------------------------------------ env
@dataclass
class FullPerpendicularEnvConstants:
    """FullPerpendicularEnvConstants / Parameters (full_perpendicular.py:44-89, cube_env.py:61-124, robot_env.py:104-195): the fields that
    shape the built path, with the reference's defaults."""

    mujoco_substeps: int = 10
    relative_action: bool = True
    reset_initial_steps: int = 20
    n_random_initial_steps: int = 10
    cube_position_wiggle_std: float = 0.005
    success_threshold: Dict[str, float] = field(default_factory=lambda: {"cube_quat": 0.4, "cube_face_angle": 0.2})
    max_timesteps_per_goal: int = 1600
    successes_needed: int = 50
    success_reward: float = 5.0
    use_goal_distance_reward: bool = True
    goal_generation: str = "face_free"
    goal_directions: List[str] = field(default_factory=lambda: ["cw", "ccw"])
    round_target_face: bool = True
    p_face_flip: float = 0.5
    num_scramble_steps: int = 50
    scramble_face_angles: bool = True
    randomize_face_angles: bool = True
    max_pose_resets: int = 50


#: FullPerpendicularEnv.build_goal_generation (full_perpendicular.py:194-259): the generators the env kernel implements (rb_post_args.goal_mode)
_GOAL_MODES = {"face_free": 0, "full_unconstrained": 1, "face_curr": 2}


class BatchedFullPerpendicularEnv:
    """B independent dactyl/full_perpendicular envs stepped in lock-step on one GPU (see the module docstring)."""

    def __init__(self, batch_size: int, device="cuda:0", constants: Optional[FullPerpendicularEnvConstants] = None, starting_seed: Optional[int] = None,
                 model: Optional[CompiledModel] = None, lib=None, pipelined_reset: bool = False):
        from robogym_amd.utils.multi_goal_tracker import BatchedMultiGoalTracker
        from robogym_amd.utils.rotation import parallel_quats_np

        self.constants = c = constants or FullPerpendicularEnvConstants()
        if c.goal_generation not in _GOAL_MODES:
            raise NotImplementedError("goal_generation=%r: %s are built for dactyl/full_perpendicular" % (c.goal_generation, sorted(_GOAL_MODES)))
        self.model = model or load_full_perpendicular_model()
        kw = dict(lib=lib) if lib is not None else dict(device=device)
        self.mujoco_simulation = sim = FullPerpendicularSimulation(self.model, batch_size, n_substeps=c.mujoco_substeps, relative_action=c.relative_action, **kw)
        self.sim = sim
        self.batch_size, self.device, self.num_actions = sim.batch_size, sim.device, sim.nu
        B, dev = self.batch_size, self.device
        f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
        self.n_hand = len(sim.qpos_idxs["hand_angle"])
        self.obs_dim = 13 + self.n_hand + 15 + 13
        self._obs_buf, self._goal, self._reward, self._goal_dist, self._prev_dist = f32(B, self.obs_dim), f32(B, _native.RB_GOAL_WORDS), f32(B, 3), f32(B, 2), f32(B, 2)
        self._goal[:, 0] = 1
        self._prev_valid, self._is_successful, self._info_ssl, self.t = i32(B), i32(B), i32(B), i32(B)
        self._flags = {k: torch.zeros(B, dtype=torch.bool, device=dev) for k in ("done", "goal_reset", "trial_success", "sub_goal_ok", "env_crash")}
        self.multi_goal_tracker = BatchedMultiGoalTracker(B, dev, c.max_timesteps_per_goal, c.success_reward, c.successes_needed, c.use_goal_distance_reward)
        self.face_up_quats_np = face_up_quats(self.model, parallel_quats_np())
        self._face_up_quats = torch.tensor(self.face_up_quats_np, dtype=torch.float32, device=dev).contiguous()
        self._qpos0_rows = torch.tensor(np.asarray(self.model.arrays["qpos0"]), dtype=torch.float32, device=dev).repeat(B, 1)
        rng_ = np.asarray(self.model.arrays["actuator_ctrlrange"], dtype=np.float32)
        self._ctrl_lo, self._ctrl_hi = torch.tensor(rng_[:, 0], device=dev), torch.tensor(rng_[:, 1], device=dev)
        self.stop_on_fall = False
        # pipelined resets: an env whose episode ended re-initialises itself INSIDE the following step calls (the recipe as a per-env phase
        # counter in rb_post_step_kernel, include/rgstep.h) -- the other envs never wait for a reset.  Off: `done` envs are the caller's to
        # `reset(mask)` (reference API)
        self.pipelined_reset = bool(pipelined_reset)
        self._phase, self._tries, self._hold = i32(B), i32(B), i32(B)
        self._nticks = torch.full((B,), 3, dtype=torch.int32, device=dev)
        self._flags.update({k: torch.zeros(B, dtype=torch.bool, device=dev) for k in ("resetting", "episode_started")})
        self._pipe_draws = None      # test hook: [B, RB_RESET_NDRAW] draws of the in-step recipe instead of the hash generator
        self._draws = None
        self._reset_draws = None
        self._step_count = 0
        self._needs_reset = True
        self._trace = None            # test hook: a list that receives qpos after the recipe's settling steps and after its state writes
        self._physics_events = None   # bench hook: a (start, end) pair of torch.cuda.Event recorded around the physics launch of `step`
        self.seed(starting_seed)
        self.action_space = {"low": -1.0, "high": 1.0, "shape": (self.num_actions,), "dtype": "float32"}

    # ------------------------------------------------------------------ randomness
    def seed(self, seed=None):
        self._seed = 0 if seed is None else int(seed)
        self._gen = torch.Generator(device=self.device); self._gen.manual_seed(self._seed)
        self._np_random = np.random.RandomState(self._seed)
        return [self._seed]

    def set_draws(self, draws):
        """Goal-generation draws of the next steps ([B, RB_POST_NDRAW], include/rgstep.h) instead of the in-kernel generator (tests)."""
        self._draws = None if draws is None else torch.as_tensor(draws, dtype=torch.float32, device=self.device).reshape(self.batch_size, _native.RB_POST_NDRAW).contiguous()

    def set_pipelined_reset_draws(self, draws):
        """The draws of the in-step reset recipe ([B, RB_RESET_NDRAW], include/rgstep.h) instead of the in-kernel generator (tests)."""
        self._pipe_draws = None if draws is None else torch.as_tensor(draws, dtype=torch.float32, device=self.device).reshape(self.batch_size, _native.RB_RESET_NDRAW).contiguous()

    def set_reset_draws(self, draws: Optional[dict]):
        """The draws of the next reset attempt(s) instead of the generators (tests): a dict with the keys of `_draw_reset`."""
        self._reset_draws = draws

    def _draw_reset(self):
        """What one pass of `_randomize_cube_initial_position` draws, for every env, in the reference's order."""
        if self._reset_draws is not None:
            d = self._reset_draws
            return {k: (v if k == "scramble" else torch.as_tensor(v, dtype=torch.float32, device=self.device)) for k, v in d.items()}
        B, c, g, dev = self.batch_size, self.constants, self._gen, self.device
        return {"wiggle": torch.randn((B, 3), generator=g, device=dev), "quat": torch.randn((B, 4), generator=g, device=dev),
                "scramble": self._np_random.randint(len(PYCUBER_ACTIONS), size=(B, c.num_scramble_steps)),
                "face_k": torch.randint(-2, 3, (B, 6), generator=g, device=dev).to(torch.float32),
                "face_angle": (torch.rand((B, 2), generator=g, device=dev) - 0.5) * (np.pi / 2),
                "face_axis": torch.randint(0, 3, (B,), generator=g, device=dev).to(torch.float32),
                "action": torch.rand((B, self.num_actions), generator=g, device=dev) * 2 - 1}

    # ------------------------------------------------------------------ reset
    def _masked_sim_reset(self, mask):
        sim = self.mujoco_simulation
        m1 = mask[:, None]
        sim.qpos.copy_(torch.where(m1, self._qpos0_rows, sim.qpos))
        for t in (sim.qvel, sim.pid, sim.qacc_warmstart, sim.ctrl, sim.view(_native.RG_F_TIME)):
            t.masked_fill_(m1, 0.0)
        sim.view(_native.RG_F_STATUS).masked_fill_(m1, 0)

    def _randomize_cube_initial_position(self, mask):
        """CubeEnv._reset (cube_env.py:330-355) around FullPerpendicularEnv._randomize_cube_initial_position (full_perpendicular.py:301-345)."""
        sim, c, B = self.mujoco_simulation, self.constants, self.batch_size
        need = mask.clone()
        lo, hi = self._ctrl_lo, self._ctrl_hi
        # the recipe's controls are ABSOLUTE: `denormalize_position_control(action)` with its default relative_action=False (robot_interface.py:247-278)
        denorm = lambda a: torch.minimum(torch.maximum(0.5 * (hi + lo) + a * 0.5 * (hi - lo), lo), hi)
        for _ in range(c.max_pose_resets):
            active = need.to(torch.int32).contiguous()
            self._masked_sim_reset(need)              # mujoco_simulation.reset(); the model carries no per-env parameters here: its constants are consistent
            sim.ctrl.copy_(torch.where(need[:, None], denorm(torch.zeros((B, self.num_actions), device=self.device)), sim.ctrl))
            for _ in range(c.reset_initial_steps):
                sim.env_step(active=active, nforward_ticks=1)
            if self._trace is not None:
                self._trace.append(sim.qpos.clone())
            d = self._draw_reset()
            sim.set_qpos("cube_position", sim.get_qpos("cube_position") + d["wiggle"] * c.cube_position_wiggle_std, need)
            q = d["quat"] / d["quat"].norm(dim=-1, keepdim=True)                      # rotation.uniform_quat: normalised, w >= 0
            sim.set_qpos("cube_rotation", torch.where(q[:, :1] < 0, -q, q), need)
            block = torch.zeros((B, 66), dtype=torch.float32, device=self.device)      # from_pycuber zeroes drivers and hinges, then writes the hinges
            block[:, 6:] = torch.as_tensor(scramble_euler(sim.cube_tab_np, d["scramble"]), dtype=torch.float32).to(self.device)
            if c.scramble_face_angles:
                block[:, :6] = d["face_k"] * (np.pi / 2)
            sim.set_qpos("cube_cubelets", block, need)
            if c.randomize_face_angles:
                ops = torch.zeros((B, 2, 4), dtype=torch.float32, device=self.device)
                ops[:, :, 0] = d["face_axis"][:, None]; ops[:, 1, 1] = 1; ops[:, :, 2] = d["face_angle"]
                sim.cube_ops("cube", ops, active)
            if self._trace is not None:
                self._trace.append(sim.qpos.clone())
            sim.forward(active=active)
            sim.ctrl.copy_(torch.where(need[:, None], denorm(d["action"]), sim.ctrl))
            for _ in range(c.n_random_initial_steps):
                sim.env_step(active=active, nforward_ticks=1)
            sim.forward(active=active)                # the forward() inside cube_utils.on_palm
            z = sim.scratch("site_xpos")[:, 3 * sim.center_site + 2]
            need = need & ~(z > 0.04)
            if self._reset_draws is not None or not bool(need.any()):
                break

    def reset(self, mask: Optional[torch.Tensor] = None):
        """RobotEnv.reset (robot_env.py:757-792) for the envs selected by `mask` (default: all)."""
        B, dev = self.batch_size, self.device
        if mask is not None and not bool(mask.any()):      # `reset(mask=done)` on a step where nothing ended: one sync instead of the ~35-launch recipe
            return self.observe()
        mask = torch.ones(B, dtype=torch.bool, device=dev) if mask is None else mask.to(dev).bool()
        self.t.masked_fill_(mask, 0)
        self._phase.masked_fill_(mask, 0); self._tries.masked_fill_(mask, 0); self._hold.masked_fill_(mask, 0); self._nticks.masked_fill_(mask, 3)
        self._randomize_cube_initial_position(mask)
        self.multi_goal_tracker.reset(mask)
        self._prev_valid.masked_fill_(mask, 0)
        self._post(force=mask.to(torch.int32).contiguous())     # reset_goal: first goal, re-observation
        self._needs_reset = False
        return self.observe()

    # ------------------------------------------------------------------ step
    def _post(self, force=None):
        sim, c, tr, F = self.mujoco_simulation, self.constants, self.multi_goal_tracker, self._flags
        a = _native.RbPostArgs()
        P = lambda t: t.data_ptr()
        a.obs, a.obs_dim = P(self._obs_buf), self.obs_dim
        a.t, a.steps, a.steps_since_last_goal, a.successes_so_far = P(self.t), P(tr.steps), P(tr.steps_since_last_goal), P(tr.successes_so_far)
        a.goals_so_far, a.consecutive = P(tr.goals_so_far), P(tr.consecutive_success)
        a.prev_dist, a.prev_valid, a.is_successful, a.goal, a.reward, a.goal_dist = P(self._prev_dist), P(self._prev_valid), P(self._is_successful), P(self._goal), P(self._reward), P(self._goal_dist)
        for k, t in F.items():
            setattr(a, k, P(t))
        a.info_ssl = P(self._info_ssl)
        a.force_new_goal = None if force is None else P(force)
        a.draws = None if self._draws is None else P(self._draws)
        a.seed, a.step = self._seed & 0xFFFFFFFF, self._step_count & 0xFFFFFFFF
        a.cube_tab, a.face_up_quats = P(sim.cube_tab), P(self._face_up_quats)
        a.face_geom = (ctypes.c_int * 6)(*sim.face_geoms); a.tip_site = (ctypes.c_int * 5)(*sim.tip_sites); a.ref_site = (ctypes.c_int * 3)(*sim.ref_sites)
        a.center_site = sim.center_site
        a.cube_pos_col, a.cube_quat_col = int(sim.qpos_idxs["cube_position"][0]), int(sim.qpos_idxs["cube_rotation"][0])
        a.cube_block_col, a.target_block_col, a.hand_col, a.n_hand = sim.cube_col, sim.target_col, int(sim.qpos_idxs["hand_angle"][0]), self.n_hand
        a.quat_threshold, a.face_threshold = float(c.success_threshold["cube_quat"]), float(c.success_threshold["cube_face_angle"])
        a.success_reward, a.p_face_flip, a.round_target_face = float(c.success_reward), float(c.p_face_flip), float(c.round_target_face)
        a.directions = (1 if "cw" in c.goal_directions else 0) | (2 if "ccw" in c.goal_directions else 0)
        a.goal_mode = _GOAL_MODES[c.goal_generation]
        a.max_timesteps_per_goal, a.successes_needed, a.use_goal_distance_reward, a.stop_on_fall = int(c.max_timesteps_per_goal), int(c.successes_needed), int(c.use_goal_distance_reward), int(self.stop_on_fall)
        a.pipelined = int(self.pipelined_reset and force is None)
        if a.pipelined:
            a.phase, a.tries, a.nticks_next, a.hold_next = P(self._phase), P(self._tries), P(self._nticks), P(self._hold)
            a.resetting, a.episode_started = P(F["resetting"]), P(F["episode_started"])
            a.reset_draws = None if self._pipe_draws is None else P(self._pipe_draws)
            a.qpos0, a.ctrl_lo, a.ctrl_hi = P(self._qpos0_rows), P(self._ctrl_lo), P(self._ctrl_hi)
            a.wiggle_std = float(c.cube_position_wiggle_std)
            a.reset_initial_steps, a.n_random_initial_steps, a.max_pose_resets = int(c.reset_initial_steps), int(c.n_random_initial_steps), int(c.max_pose_resets)
            a.num_scramble_steps, a.scramble_face_angles, a.randomize_face_angles = int(c.num_scramble_steps), int(c.scramble_face_angles), int(c.randomize_face_angles)
        stream = None if sim._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._keep_post = (force, self._draws, self._pipe_draws)
        _native.check(sim._L, sim._L.rb_env_post_step(sim._bh, ctypes.byref(a), stream), "rb_env_post_step")
        self._step_count += 1

    def step(self, action: torch.Tensor):
        """RobotEnv.step (robot_env.py:804-844): returns (obs dict, reward [B, 3], done [B], info dict); every tensor is a view of a
        buffer the two launches wrote."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.batch_size, self.num_actions).contiguous()
        ev = self._physics_events
        if ev is not None:
            ev[0].record()
        pipe = self.pipelined_reset
        self.mujoco_simulation.env_step(action=action, nforward_ticks=3, hold=self._hold if pipe else None, nticks=self._nticks if pipe else None)
        if ev is not None:
            ev[1].record()
        self._post()
        F, tr = self._flags, self.multi_goal_tracker
        info = {"goal_dist": {"cube_quat": self._goal_dist[:, 0], "cube_face_angle": self._goal_dist[:, 1]}, "goal_achieved": F["sub_goal_ok"],
                "sub_goal_is_successful": F["sub_goal_ok"], "trial_success": F["trial_success"], "goal_reset": F["goal_reset"], "successes_so_far": tr.successes_so_far,
                "steps_since_last_goal": self._info_ssl, "goals_so_far": tr.goals_so_far, "env_crash": F["env_crash"], "sim_status": self.mujoco_simulation.status,
                "resetting": F["resetting"], "episode_started": F["episode_started"]}
        return self.observe(), self._reward, F["done"], info

    def observe(self) -> Dict[str, torch.Tensor]:
        """Keys and shapes of `FullPerpendicularEnv._default_observation_map` (full_perpendicular.py:177-192); views of the observation row
        and of the state."""
        o, sim, n = self._obs_buf, self.mujoco_simulation, self.n_hand
        t = 13 + n
        return {"cube_pos": o[:, 0:3], "cube_quat": o[:, 3:7], "cube_face_angle": o[:, 7:13], "qpos": sim.qpos, "qvel": sim.qvel, "perp_qpos": sim.qpos, "perp_qvel": sim.qvel,
                "hand_angle": o[:, 13:t], "fingertip_pos": o[:, t:t + 15], "goal_pos": o[:, t + 15:t + 18], "goal_quat": o[:, t + 18:t + 22], "goal_face_angle": o[:, t + 22:t + 28]}

    def relative_goal(self, key: str, current: torch.Tensor) -> torch.Tensor:
        """`FaceFreeGoal.relative_goal` (goals/face_free.py:147-173) of the current goal rows against `current` [B, 4 | 6] (the wrappers pass the observed and the noisy
        state): cube_quat -- a "rotation" goal only asks for the goal's face axis to point up (cube_utils.distance_quat_from_being_up, :168-181), a "flip" goal for
        the orientation itself (quat_difference) --, cube_face_angle -- the angle differences, normalised."""
        from robogym_amd.utils import rotation

        g = self._goal.to(current.dtype)
        if key == "cube_face_angle":
            return rotation.normalize_angles(g[:, 4:10] - current)
        assert key == "cube_quat"
        mode = _GOAL_MODES[self.constants.goal_generation] if hasattr(self, "constants") else 0
        if mode == 1:       # FullUnconstrainedGoal.relative_goal (goals/full_unconstrained.py:91-106): no orientation objective
            return torch.zeros_like(current)
        if mode == 2:       # FaceCurriculumGoal.relative_goal (goals/face_curriculum.py:144-160): always the plain difference
            return rotation.quat_difference(g[:, 0:4], current)
        m = rotation.quat2mat(current)                                                     # [B, 3, 3]
        nr = g[:, 11].long().clamp(0, 2)
        axis = torch.gather(m, 2, nr[:, None, None].expand(-1, 3, 1))[:, :, 0] * g[:, 12:13]
        up = rotation.quat_normalize(rotation.vectors2quat_to_z(axis))
        flip = rotation.quat_difference(g[:, 0:4], current)
        return torch.where(g[:, 10:11] > 0.5, up, flip)

    def goal_info(self):
        g = self._goal
        return {"goal": {"cube_quat": g[:, 0:4], "cube_face_angle": g[:, 4:10], "goal_type": g[:, 10], "axis_nr": g[:, 11], "axis_sign": g[:, 12]},
                "is_successful": self._is_successful}


def make_simple_env(parameters=None, constants=None, starting_seed=None, batch_size: int = 1, device="cuda:0", **kwargs):
    """`make_simple_env` of envs/dactyl/full_perpendicular.py (no wrappers); `pipelined_reset=True`: finished episodes restart by themselves.  `constants` / `parameters`: dicts with the reference's
    names for the fields of FullPerpendicularEnvConstants above; anything else raises."""
    kw = {}
    for src in (constants or {}), (parameters or {}):
        for k, v in dict(src).items():
            if k not in FullPerpendicularEnvConstants.__dataclass_fields__:
                raise NotImplementedError("make_simple_env(%r=%r) is not supported by the MI355X-native dactyl/full_perpendicular path (supported: %s)"
                                          % (k, v, sorted(FullPerpendicularEnvConstants.__dataclass_fields__)))
            kw[k] = v
    return BatchedFullPerpendicularEnv(batch_size, device=device, constants=FullPerpendicularEnvConstants(**kw), starting_seed=starting_seed, **kwargs)


WRAPPER_CONSTANTS = ("randomize", "n_action_bins", "relative_goal_wrapper", "drop_reward", "min_episode_length", "fixed_wrist")


def make_env(parameters=None, constants=None, wrapper_params=None, starting_seed=None, apply_wrappers=True, batch_size: int = 1, device="cuda:0", **kwargs):
    """`FullPerpendicularEnv.build` (full_perpendicular.py:460, robot_env.py:1081-1139) for a batch of envs.  `apply_wrappers=True` wraps the env in the reference's
    default wrapper stack (dactyl_cube_wrappers.py:8-91, vectorised in robogym_amd/wrappers/dactyl_cube.py): MultiDiscrete actions of 11 bins, action smoothing, drop
    penalty / done on fall, noisy_* / achieved / relative goal observations with FaceFreeGoal.relative_goal over pos, quat and face_angle, unified goal vectors, cos / sin
    angles, clipping, previous action and reward observations -- replayed against the reference's own classes (tests/golden/wrappers_full.npz).  The randomization half
    of the stack (`constants["randomize"]`, True by default in the reference) is NOT built for the full cube: RandomizedPerpendicularCubeSizeWrapper, the timestep and
    wind wrappers need per-env mesh scale, timestep and applied forces on rb_step_kernel; asking for it raises instead of silently dropping it."""
    wc = {}
    constants = dict(constants or {})
    for k in WRAPPER_CONSTANTS:
        if k in constants:
            wc[k] = constants.pop(k)
    env = make_simple_env(parameters=parameters, constants=constants, starting_seed=starting_seed, batch_size=batch_size, device=device, **kwargs)
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
