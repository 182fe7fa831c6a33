"""Batched simulation of models beyond the Shadow-hand kernel's compile-time layout (`rb_step_kernel`,
robogym_amd/csrc/rb_kernel.h): BASELINE.json configs[2], dactyl/full_perpendicular.  Mirrors the part of
`SimulationInterface` (/root/reference/robogym/mujoco/simulation_interface.py:92-250) the physics path needs:
state fields as zero-copy `[B, n]` tensor views, `reset`, `step` / `env_step`, joint groups.  No CPU fallback: the gfx950
library and a GPU are required (tests may pass `lib=` = the kernel source on the emulation harness, as for the other kernels)."""
import ctypes
from typing import Dict

import numpy as np
import torch

from robogym_amd import _native
from robogym_amd.mujoco.big_tables import derive_big_tables
from robogym_amd.mujoco.model_blob import pack_model
from robogym_amd.mujoco import simulation_interface

SCRATCH = ["xpos", "xquat", "xipos", "xiquat", "xanchor", "xaxis", "geom_xpos", "geom_xquat", "site_xpos", "rootcom", "cinert", "crb", "cdof", "cdof_dot", "cvel",
           "cacc", "cfrc", "ten_length", "ten_J", "ten_velocity", "Msp", "cand", "contact", "contact_J", "contact_idx", "row", "dofcon_adr", "dofcon", "contact_f", "dbg", "cfrc_ext"]
INFO = ["nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "ntendon", "nM", "npair", "ngroup", "gmax", "maxcon", "maxrow", "scratch_words", "conrec", "rowrec", "conw", "tenw", "lds_bytes", "threads"]


class LargeModelSimulation:
    def __init__(self, model, batch_size: int, device="cuda:0", n_substeps: int = 10, relative_action: bool = True, lib=None, hand: bool = True, env_params: bool = False):
        """`hand`: the model is a Shadow-hand world (joint group `robot0:` = the hand, in-kernel action map through the position -> control matrix);
        False for the rearrange worlds, whose action path is the TCP solver hook (`step_tcp`)."""
        self._emul = lib is not None
        self._L = L = lib if lib is not None else _native.lib()
        self.device = torch.device("cpu") if self._emul else torch.device(device)
        if not self._emul and not torch.cuda.is_available():
            raise _native.NativeError("LargeModelSimulation needs an MI355X (no CPU fallback)")
        if "b_dims" not in model.arrays or "b_tree_desc" not in model.arrays or "b_Mlong" not in model.arrays or "b_tree8" not in model.arrays or "b_geom_aabb" not in model.arrays or "b_body_level" not in model.arrays:
            derive_big_tables(model)
        self.model, self.batch_size, self.n_substeps = model, int(batch_size), int(n_substeps)
        blob = pack_model(model)
        err = ctypes.create_string_buffer(512)
        if not self._emul:
            torch.cuda.set_device(self.device)
        self._mh = L.rb_model_create(blob, len(blob), err, 512)
        if not self._mh:
            raise _native.NativeError("rb_model_create: " + err.value.decode())
        # per-env model parameters: every env gets its own block of the randomisable model fields, read by the kernel instead of the model's arrays
        self._env_params = bool(env_params)
        if self._env_params:
            _native.check(L, L.rb_model_enable_env_params(self._mh), "rb_model_enable_env_params")
        self._params = None
        buf = (ctypes.c_int * 32)()
        n = L.rb_model_info(self._mh, buf, 32)
        assert n == len(INFO)
        self.info = {k: int(buf[i]) for i, k in enumerate(INFO)}
        self._bh = L.rb_batch_create(self._mh, self.batch_size)
        if not self._bh:
            raise _native.NativeError("rb_batch_create: " + L.rg_last_error().decode())
        self.nq, self.nv, self.nu = self.info["nq"], self.info["nv"], self.info["nu"]
        names = model.names["joint"]
        A = model.arrays
        self.qpos_idxs: Dict[str, np.ndarray] = {}
        self.qvel_idxs: Dict[str, np.ndarray] = {}
        self._views = {}
        self._keep = []
        if not hand:
            return
        self.register_joint_group("hand_angle", "robot0:")
        hand_j = [j for j, nm in enumerate(names) if nm.startswith("robot0:")]
        P = np.zeros((self.nu, len(hand_j)), dtype=np.float32)
        for u in range(self.nu):
            if A["actuator_trntype"][u] == 0:
                P[u, hand_j.index(int(A["actuator_trnid"][u]))] = 1
            else:
                t = int(A["actuator_trnid"][u])
                for w in range(A["tendon_adr"][t], A["tendon_adr"][t] + A["tendon_num"][t]):
                    P[u, hand_j.index(int(A["wrap_objid"][w]))] = 1
        self.pos_to_ctrl = P
        hq = self.qpos_idxs["hand_angle"]
        assert (np.diff(hq) == 1).all()
        _native.check(L, L.rb_batch_set_env(self._bh, int(hq[0]), len(hq), 1 if relative_action else 0, P.ctypes.data_as(ctypes.POINTER(ctypes.c_float))), "rb_batch_set_env")
        self._views = {}
        self._keep = []

    def __del__(self):
        try:
            self._L.rb_batch_free(self._bh)
            self._L.rb_model_free(self._mh)
        except Exception:
            pass

    def set_action_map(self, first_qposadr: int, pos_to_ctrl: np.ndarray, relative_action: bool = True, max_position_change: float = 0.0, ctrl_centre_mask: int = 0):
        """The action map of `env_step(action=...)` for a robot that is not the hand (rb_batch_set_env + rb_batch_set_action_limits, include/rgstep.h):
        `pos_to_ctrl` [nu, n] over the n joint positions from `first_qposadr` on."""
        P = np.ascontiguousarray(pos_to_ctrl, dtype=np.float32)
        assert P.ndim == 2 and P.shape[0] == self.nu
        L = self._L
        _native.check(L, L.rb_batch_set_env(self._bh, int(first_qposadr), P.shape[1], 1 if relative_action else 0, P.ctypes.data_as(ctypes.POINTER(ctypes.c_float))), "rb_batch_set_env")
        _native.check(L, L.rb_batch_set_action_limits(self._bh, float(max_position_change), int(ctrl_centre_mask)), "rb_batch_set_action_limits")
        self.pos_to_ctrl = P

    def register_joint_group(self, name, prefix):
        A, names = self.model.arrays, self.model.names["joint"]
        q, v = [], []
        for j, nm in enumerate(names):
            if nm.startswith(prefix):
                nq, nv = {0: (7, 6), 1: (4, 3), 2: (1, 1), 3: (1, 1)}[int(A["jnt_type"][j])]
                q += list(range(A["jnt_qposadr"][j], A["jnt_qposadr"][j] + nq))
                v += list(range(A["jnt_dofadr"][j], A["jnt_dofadr"][j] + nv))
        self.qpos_idxs[name], self.qvel_idxs[name] = np.array(q), np.array(v)

    def view(self, field: int) -> torch.Tensor:
        if field not in self._views:
            n = ctypes.c_int(0)
            p = self._L.rb_batch_field_ptr(self._bh, field, ctypes.byref(n))
            if not p:
                raise _native.NativeError(self._L.rg_last_error().decode())
            dt = torch.int32 if field in (_native.RG_F_STATUS, _native.RB_F_EQ_ACTIVE) else torch.float32
            self._views[field] = simulation_interface.device_tensor(p, (self.batch_size, n.value), dt, self.device)
        return self._views[field]

    qpos = property(lambda self: self.view(_native.RG_F_QPOS))
    qvel = property(lambda self: self.view(_native.RG_F_QVEL))
    ctrl = property(lambda self: self.view(_native.RG_F_CTRL))
    pid = property(lambda self: self.view(_native.RG_F_PID))
    qacc_warmstart = property(lambda self: self.view(_native.RG_F_WARMSTART))
    status = property(lambda self: self.view(_native.RG_F_STATUS)[:, 0])
    stats = property(lambda self: self.view(_native.RG_F_STATS))
    time = property(lambda self: self.view(_native.RG_F_TIME)[:, 0])
    mocap = property(lambda self: self.view(_native.RB_F_MOCAP))            # [B, 7 nmocap]: mocap_pos | mocap_quat per mocap body
    eq_data = property(lambda self: self.view(_native.RB_F_EQ_DATA))        # [B, 7 neq]
    eq_active = property(lambda self: self.view(_native.RB_F_EQ_ACTIVE))    # [B, neq] int32
    sensordata = property(lambda self: self.view(_native.RB_F_SENSORDATA))  # [B, nsensordata], of the last full forward (flags bit 5)

    @property
    def params(self) -> "LargeEnvParams":
        """`sim.model.<field>` of the reference for a batch on this stepper: `[B, ...]` tensor views into the envs' parameter blocks (include/rgstep.h
        rb_model_enable_env_params): gravity, dof_damping / armature / frictionloss / invweight0, jnt_stiffness / margin / range, body_pos / mass / inertia /
        invweight0, actuator_gainprm / forcerange / ctrlrange, geom_pos / margin / gap / friction / solref / solimp, tendon_range / invweight0.  The simulation
        must have been created with `env_params=True`."""
        if not self._env_params:
            raise _native.NativeError("this simulation was created without per-env parameter rows (LargeModelSimulation(..., env_params=True))")
        if self._params is None:
            self._params = LargeEnvParams(self)
        return self._params

    def scratch(self, name: str) -> torch.Tensor:
        """A stage array of the last launch, `[B, words]` (debugging / stage parity tests)."""
        k = SCRATCH.index(name)
        o0 = self._L.rb_scratch_offset(self._mh, k)
        o1 = self._L.rb_scratch_offset(self._mh, k + 1) if k + 1 < len(SCRATCH) else self.info["scratch_words"]
        return self.view(_native.RG_F_DEBUG)[:, o0:o1]

    def reset(self):
        self.sync()
        _native.check(self._L, self._L.rb_batch_reset(self._bh), "rb_batch_reset")

    def env_step(self, action=None, active=None, nsubsteps=None, nforward_ticks=3, flags=0, hold=None, nticks=None):
        """`hold` int32 [B]: envs that keep their stored ctrl row (scripted controls of a reset recipe); `nticks` int32 [B]: per-env count of
        state-less forwards (rb_batch_step_ex)."""
        for t in (action,):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device and t.shape == (self.batch_size, self.nu))
        for t in (active, hold, nticks):
            assert t is None or (t.dtype == torch.int32 and t.is_contiguous() and t.shape == (self.batch_size,))
        flags = int(flags) | (_native.RG_FLAG_MPR_PLANE_DEPTH if simulation_interface.MPR_PLANE_DEPTH else 0)
        stream = None if self._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._keep = [action, active, hold, nticks]
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        _native.check(self._L, self._L.rb_batch_step_ex(self._bh, ptr(action), ptr(active), ptr(hold), ptr(nticks), self.n_substeps if nsubsteps is None else int(nsubsteps),
                                                        int(nforward_ticks), flags, stream), "rb_batch_step_ex")

    def step_tcp(self, main: "LargeModelSimulation", action: torch.Tensor, args: "_native.RbTcpArgs", flags=0, active=None):
        """`JointControlledTcpArm.set_position_control` as one launch of THIS (the TCP solver's) simulation: rb_batch_step_tcp (include/rgstep.h)."""
        assert action is None or (action.dtype == torch.float32 and action.is_contiguous() and action.device == self.device and action.shape == (self.batch_size, 6))
        flags = int(flags) | (_native.RG_FLAG_MPR_PLANE_DEPTH if simulation_interface.MPR_PLANE_DEPTH else 0)
        stream = None if self._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._keep = [action, args, active]
        assert active is None or (active.dtype == torch.int32 and active.is_contiguous() and active.shape == (self.batch_size,))
        _native.check(self._L, self._L.rb_batch_step_tcp(self._bh, main._bh, None if action is None else ctypes.c_void_p(action.data_ptr()), None if active is None else ctypes.c_void_p(active.data_ptr()), ctypes.byref(args), self.n_substeps, flags, stream), "rb_batch_step_tcp")

    def step(self, active=None):
        """SimulationInterface.step: nsubsteps x mj_step, then mj_forward (its PID tick)."""
        self.env_step(nforward_ticks=1, active=active)

    def sync(self):
        if not self._emul:
            torch.cuda.synchronize(self.device)


class LargeEnvParams:
    """Named `[B, ...]` views into the per-env parameter blocks of the scratch rows (rb_prm_layout); same interface as simulation_interface.EnvParams, so the
    randomizers of robogym_amd/randomization/sim.py act on either stepper."""

    def __init__(self, sim: LargeModelSimulation):
        buf = (ctypes.c_int * (2 + 2 * len(_native.RB_PRM_NAMES)))()
        n = sim._L.rb_prm_layout(sim._mh, buf, len(buf))
        assert n == len(buf) and buf[0] == 1, "rb_prm_layout and robogym_amd/_native.py disagree"
        rows = sim.view(_native.RG_F_DEBUG)          # the whole scratch row [B, scratch_words]
        shape_of = dict(gravity=(3,), jnt_range=(-1, 2), body_pos=(-1, 3), body_inertia=(-1, 3), body_invweight0=(-1, 2), actuator_gainprm=(-1, 10), actuator_forcerange=(-1, 2),
                        actuator_ctrlrange=(-1, 2), geom_pos=(-1, 3), geom_friction=(-1, 3), geom_solref=(-1, 2), geom_solimp=(-1, 5), tendon_range=(-1, 2))
        self._views: Dict[str, torch.Tensor] = {}
        B = sim.batch_size
        lo = min(int(buf[2 + 2 * k]) for k in range(len(_native.RB_PRM_NAMES)))
        self.block = rows[:, lo:lo + int(buf[1])]        # every field of an env's block as one [B, words] view (a masked restore of the whole block is one tensor op)
        for k, name in enumerate(_native.RB_PRM_NAMES):
            off, length = int(buf[2 + 2 * k]), int(buf[3 + 2 * k])
            assert lo <= off and off + length <= lo + int(buf[1])
            v = rows[:, off:off + length]
            self._views[name] = v.unflatten(1, shape_of[name]) if (name in shape_of and length) else v

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._views[name]

    def keys(self):
        return self._views.keys()
