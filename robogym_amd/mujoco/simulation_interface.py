"""Batched counterpart of the reference's `SimulationInterface`
(/root/reference/robogym/mujoco/simulation_interface.py:25-250): same method names, with a
leading env-batch dimension on every array and state living in HBM inside the HIP stepper
(C ABI: include/rgstep.h).  One `BatchedSimulationInterface` stands for B independent
(model, MjSim) pairs of the reference (robot_env.py:328-350)."""
import ctypes
import os
from typing import Dict, Optional

import numpy as np
import torch

from robogym_amd import _native
from robogym_amd.mujoco.kernel_tables import derive_kernel_tables
from robogym_amd.mujoco.model_blob import pack_model

#: Contact depth / normal of convex pairs.  False (default): libccd's formula, which is what MuJoCo 2.0 runs (closest point of the final
#: MPR portal triangle).  True: the portal-plane variant (rg_step_args.flags bit 4) — identical whenever the origin projects inside the
#: final triangle, and free of libccd's rounding-level tie breaks on flat contacts; the tight-tolerance parity tests use it on both sides.
#: The flag is round 1's contact generation as a whole: with it box-box pairs go through MPR (one contact) instead of the multi-point routine.
MPR_PLANE_DEPTH = os.environ.get("RG_MPR_PLANE", "0") == "1"
# Substep-granular dispatch of the rollout launches (rg_step_args.flags bit 7, rg_step_items_kernel): persistent workgroups draw
# (env, substep) work items, so the wave slots stay full to the end of a launch.  Bit-identical results; RG_SUBSTEP_ITEMS=0
# (or this switch) keeps one workgroup per env.step.
SUBSTEP_ITEMS = os.environ.get("RG_SUBSTEP_ITEMS", "1") == "1"


def device_tensor(ptr, shape, dtype, device) -> torch.Tensor:
    """Zero-copy tensor over a buffer the native library owns (`__cuda_array_interface__`; host memory under the emulation harness)."""
    is_int = dtype == torch.int32
    if shape[1] == 0:      # e.g. a model without actuators: nothing to alias
        return torch.zeros(shape, dtype=dtype, device=device)
    if torch.device(device).type == "cpu":
        ctype = ctypes.c_int32 if is_int else ctypes.c_float
        return torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape=shape))
    holder = type("_DevArray", (), {})()
    holder.__cuda_array_interface__ = {"shape": shape, "typestr": "<i4" if is_int else "<f4", "data": (int(ptr), False), "version": 2, "strides": None}
    with torch.cuda.device(device):
        t = torch.as_tensor(holder, device=device)
    assert t.data_ptr() == int(ptr) and t.device == torch.device(device)
    return t


class BatchedSimulationInterface:
    def __init__(self, model, batch_size: int, device="cuda:0", n_substeps: int = 10, lib=None):
        self.model = model
        if "k_dims" not in model.arrays:
            derive_kernel_tables(model)
        self._L = lib if lib is not None else _native.lib()
        self._emul = lib is not None
        self.device = torch.device("cpu") if self._emul else torch.device(device)
        if not self._emul:
            if self.device.type != "cuda" or not torch.cuda.is_available():
                raise _native.NativeError("the HIP stepper needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
        self.batch_size = int(batch_size)
        self.n_substeps = int(n_substeps)
        blob = pack_model(model)
        err = ctypes.create_string_buffer(512)
        index = self.device.index or 0
        # model tables and batch state live on `device`, whatever the caller's current device is
        self._mh = self._L.rg_model_create_on(blob, len(blob), index, err, 512)
        if not self._mh:
            raise _native.NativeError("rg_model_create: " + err.value.decode())
        self._bh = self._L.rg_batch_create(self._mh, self.batch_size, index)
        if not self._bh:
            raise _native.NativeError("rg_batch_create: " + self._L.rg_last_error().decode())
        d = model.dims
        self.nq, self.nv, self.nu = int(d[0]), int(d[1]), int(d[2])
        self.npair = int(self._L.rg_model_npair(self._mh))
        self.qpos_idxs: Dict[str, np.ndarray] = {}
        self.qvel_idxs: Dict[str, np.ndarray] = {}
        self._stream = None
        self._views: Dict[int, torch.Tensor] = {}
        self._idx_cache: Dict = {}
        self._keep = []
        self._redo = None
        self._side = None
        self._params = None
        self._xdata = None
        self._data = None
        self.sensors = False      # data.sensordata requested (see BatchedData.sensordata)

    def __del__(self):
        try:
            self._L.rg_batch_free(self._bh)
            self._L.rg_model_free(self._mh)
        except Exception:
            pass

    # ------------------------------------------------------------------ joint groups (simulation_interface.py:92-124)
    def register_joint_group(self, group_name: str, prefix):
        prefixes = [prefix] if isinstance(prefix, str) else list(prefix)
        A = self.model.arrays
        qidx, vidx = [], []
        for j, name in enumerate(self.model.names["joint"]):
            if any(name.startswith(p) for p in prefixes):
                t = int(A["jnt_type"][j])
                nq, nv = {0: (7, 6), 1: (4, 3), 2: (1, 1), 3: (1, 1)}[t]
                qidx += list(range(A["jnt_qposadr"][j], A["jnt_qposadr"][j] + nq))
                vidx += list(range(A["jnt_dofadr"][j], A["jnt_dofadr"][j] + nv))
        self.qpos_idxs[group_name] = np.array(qidx, dtype=np.int64)
        self.qvel_idxs[group_name] = np.array(vidx, dtype=np.int64)

    # ------------------------------------------------------------------ raw field access
    def _ncols(self, field):
        return {_native.RG_F_QPOS: self.nq, _native.RG_F_QVEL: self.nv, _native.RG_F_CTRL: self.nu, _native.RG_F_PID: 3 * self.nu,
                _native.RG_F_WARMSTART: self.nv, _native.RG_F_TIME: 1, _native.RG_F_STATUS: 1, _native.RG_F_STATS: 4,
                _native.RG_F_DEBUG: self._L.rg_debug_size(), _native.RG_F_COST: 1, _native.RG_F_PAIRLB: max(self.npair, 1),
                _native.RG_F_ENVPRM: _native.prm_layout(self._L)["row"]}[field]

    def view(self, field) -> torch.Tensor:
        """Zero-copy [B, n] tensor over the batch's own buffer of `field` (the batched `sim.data.<field>`):
        reads and in-place writes are ordinary stream-ordered torch ops, no host synchronisation, no copy.
        Whoever writes qpos through a view must void the collision cache rows (`touch_qpos`)."""
        t = self._views.get(field)
        if t is None:
            n = ctypes.c_int(0)
            ptr = self._L.rg_batch_field_ptr(self._bh, field, ctypes.byref(n))
            if not ptr:
                raise _native.NativeError("rg_batch_field_ptr: " + self._L.rg_last_error().decode())
            t = device_tensor(ptr, (self.batch_size, int(n.value)), torch.int32 if field == _native.RG_F_STATUS else torch.float32, self.device)
            self._views[field] = t
        return t

    def get_field(self, field) -> torch.Tensor:
        """Snapshot (a copy) of a field."""
        return self.view(field).clone()

    def set_field(self, field, value: torch.Tensor):
        dtype = torch.int32 if field == _native.RG_F_STATUS else torch.float32
        value = torch.as_tensor(value, dtype=dtype, device=self.device).reshape(self.batch_size, self._ncols(field))
        self.view(field).copy_(value)
        if field == _native.RG_F_QPOS:
            self.touch_qpos()

    def touch_qpos(self, mask: Optional[torch.Tensor] = None):
        """qpos was written from outside the stepper: the cached pair distance bounds of those envs are void."""
        lb = self.view(_native.RG_F_PAIRLB)
        if mask is None:
            lb.zero_()
        else:
            lb.mul_((~mask.to(self.device).bool()).to(lb.dtype)[:, None])

    def copy_rows(self, field, value: torch.Tensor, mask: torch.Tensor, col0: int = 0):
        """Asynchronous masked write (rg_batch_copy_rows): rows of the envs with mask != 0, columns
        [col0, col0 + value.shape[1]) <- value.  Stream-ordered with the step launches; no host sync."""
        dtype = torch.int32 if field == _native.RG_F_STATUS else torch.float32
        value = torch.as_tensor(value, dtype=dtype, device=self.device).reshape(self.batch_size, -1).contiguous()
        mask = mask.to(device=self.device, dtype=torch.int32).contiguous()
        _native.check(self._L, self._L.rg_batch_copy_rows(self._bh, field, value.data_ptr(), mask.data_ptr(), int(col0), int(value.shape[1]), self._stream_ptr()), "rg_batch_copy_rows")
        self._keep.append((value, mask))   # the launch is asynchronous: keep the operands alive for a few calls
        del self._keep[:-16]

    def _stream_ptr(self):
        return None if self._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def qpos(self) -> torch.Tensor:
        return self.get_field(_native.RG_F_QPOS)

    @property
    def qvel(self) -> torch.Tensor:
        return self.get_field(_native.RG_F_QVEL)

    def _group_idx(self, table, group):
        idx = table[group]
        key = (table is self.qpos_idxs, group, np.asarray(idx).tobytes())   # keyed by CONTENT: a reassigned group gets a fresh tensor
        if key not in self._idx_cache:
            self._idx_cache[key] = torch.as_tensor(idx, device=self.device)
        return self._idx_cache[key]

    def get_qpos(self, group: str) -> torch.Tensor:
        """Only the group's columns are gathered (no copy of the whole field, no host sync)."""
        return self.view(_native.RG_F_QPOS)[:, self._group_idx(self.qpos_idxs, group)]

    def set_qpos(self, group: str, value, mask: Optional[torch.Tensor] = None):
        idx = self.qpos_idxs[group]
        value = torch.as_tensor(value, dtype=torch.float32, device=self.device).expand(self.batch_size, len(idx))
        contiguous = len(idx) > 0 and (np.diff(idx) == 1).all()
        if mask is not None and contiguous:
            self.copy_rows(_native.RG_F_QPOS, value, mask, int(idx[0]))     # voids the cache rows of the masked envs
            return
        q, ti = self.view(_native.RG_F_QPOS), self._group_idx(self.qpos_idxs, group)
        if mask is None:
            q[:, ti] = value
        else:
            q[:, ti] = torch.where(mask.to(self.device).bool()[:, None], value, q[:, ti])
        self.touch_qpos(mask)

    def add_qpos(self, group: str, value, mask: Optional[torch.Tensor] = None):
        self.set_qpos(group, self.get_qpos(group) + torch.as_tensor(value, dtype=torch.float32, device=self.device), mask)

    def get_qvel(self, group: str) -> torch.Tensor:
        return self.view(_native.RG_F_QVEL)[:, self._group_idx(self.qvel_idxs, group)]

    def set_qvel(self, group: str, value):
        self.view(_native.RG_F_QVEL)[:, self._group_idx(self.qvel_idxs, group)] = torch.as_tensor(value, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ sim.data fields in-tree callers read
    @property
    def data(self) -> "BatchedData":
        """`sim.data.<field>` of the reference for a batch: qpos / qvel / ctrl / time (zero-copy views of the state) and,
        refreshed by every step / forward launch from now on, body_xpos, body_xquat, site_xpos, actuator_force, ncon and the
        contact list (geom1, geom2, dist).  First access switches the readout on (one more row written per env per launch)."""
        if self._data is None:
            lay = (ctypes.c_int * 10)()
            assert self._L.rg_xdata_layout(lay, 10) == 10, "rg_xdata_layout and simulation_interface.py disagree"
            self._xdata = torch.zeros((self.batch_size, int(lay[0])), dtype=torch.float32, device=self.device)
            self._data = BatchedData(self, [int(v) for v in lay])
            self.forward(ticks=0)   # fill it for the current state
        return self._data

    # ------------------------------------------------------------------ per-env model parameters
    @property
    def params(self) -> "EnvParams":
        """`sim.model.<field>` of the reference for a batch: `[B, ...]` tensor views into the per-env parameter rows the
        kernel reads (gravity, timestep, dof_damping, dof_armature, dof_frictionloss, body_mass, body_inertia, jnt_range,
        tendon_range, actuator_gainprm / ctrlrange / forcerange, geom_friction, xfrc_applied, and the mj_setConst outputs
        dof / body / tendon _invweight0).  First access allocates the rows (initialised with the model's own values)."""
        if self._params is None:
            self._params = EnvParams(self)
        return self._params

    def set_constants(self, mask: Optional[torch.Tensor] = None):
        """`SimulationInterface.set_constants` (simulation_interface.py:199-201 -> MjSim.set_constants -> mj_setConst), which the
        reference calls in every `_reset` after the randomizers have written the model (cube_env.py:346-349): recompute
        dof / body / tendon `_invweight0` of the masked envs (None: all) from each env's own mass / inertia / armature /
        site_pos row, on the device (`rg_setconst_kernel`), asynchronously on the current stream.  A batch without
        parameter rows has nothing to refresh (the model's own constants are consistent)."""
        if self._params is None:
            return
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.int32).contiguous()
            assert mask.shape == (self.batch_size,)
        _native.check(self._L, self._L.rg_batch_set_constants(self._bh, self._ptr(mask), self._stream_ptr()), "rg_batch_set_constants")
        self._keep_alive = mask   # (the launch is asynchronous: the mask must outlive it)

    # ------------------------------------------------------------------ state (simulation_interface.py:154-172)
    def get_state(self) -> dict:
        return dict(time=self.get_field(_native.RG_F_TIME), qpos=self.qpos, qvel=self.qvel, pid=self.get_field(_native.RG_F_PID),
                    qacc_warmstart=self.get_field(_native.RG_F_WARMSTART), ctrl=self.get_field(_native.RG_F_CTRL))

    def set_state(self, state: dict):
        names = dict(time=_native.RG_F_TIME, qpos=_native.RG_F_QPOS, qvel=_native.RG_F_QVEL, pid=_native.RG_F_PID,
                     qacc_warmstart=_native.RG_F_WARMSTART, ctrl=_native.RG_F_CTRL)
        for k, v in state.items():
            self.set_field(names[k], v)

    def reset(self):
        """MjSim.reset for every env: qpos0, zero velocities/controls/controller state/time."""
        self.sync()
        _native.check(self._L, self._L.rg_batch_reset(self._bh), "rg_batch_reset")

    @property
    def status(self) -> torch.Tensor:
        return self.view(_native.RG_F_STATUS)[:, 0].clone()

    # ------------------------------------------------------------------ stepping
    def set_env(self, ints, pos_to_ctrl: np.ndarray, success_threshold: float):
        ints = np.asarray(ints, dtype=np.int32)
        p2c = np.ascontiguousarray(pos_to_ctrl, dtype=np.float32)
        _native.check(self._L, self._L.rg_batch_set_env(self._bh, ints.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(ints),
                                                        p2c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), float(success_threshold)), "rg_batch_set_env")
        self.obs_dim = self._L.rg_obs_dim(self._bh)

    def _ptr(self, t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    def env_step(self, action=None, goal_quat=None, obs=None, goal_dist=None, active=None, nsubsteps=None, nforward_ticks=3, flags=0,
                 hold=None, nticks=None, order=None, capacity="auto", large_mask=None, small_mask=None, preticks=None):
        """One reference env.step worth of physics for the whole batch (async on the current stream).
        `active`: optional int32 [B]; envs with 0 are left untouched.  `hold` int32 [B]: envs that keep their
        stored ctrl row; `nticks` int32 [B]: per-env forward-tick counts; `order` int32 [B]: dispatch permutation.
        `capacity`: "rollout" / "large" = one launch of that kernel configuration (rg_step_args.config);
        "auto" = the rollout configuration, then the large one for exactly the envs that exceeded the rollout
        capacities (flagged on the device, no host sync; almost always an empty launch).  `large_mask` (int32 [B], with
        "auto"): envs known to need the large configuration (the reset recipe): they skip the rollout launch and run on a
        side stream concurrently with it (`small_mask`: its complement, if the caller already has it).  `preticks` int32 [B]:
        state-less forwards owed from the previous step's goal reset (rg_step_args.preticks_dev)."""
        for t in (action, goal_quat, obs, goal_dist):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device)
        for t in (active, hold, nticks, order, large_mask, small_mask, preticks):
            assert t is None or (t.dtype == torch.int32 and t.is_contiguous() and t.device == self.device and t.numel() == self.batch_size)
        a = _native.StepArgs()
        a.action_dev, a.goal_quat_dev, a.obs_dev, a.goal_dist_dev = (None if t is None else t.data_ptr() for t in (action, goal_quat, obs, goal_dist))
        a.hold_dev, a.nticks_dev, a.order_dev, a.preticks_dev = (None if t is None else t.data_ptr() for t in (hold, nticks, order, preticks))
        a.xdata_dev = None if self._xdata is None else self._xdata.data_ptr()
        a.nsubsteps = self.n_substeps if nsubsteps is None else int(nsubsteps)
        a.nforward_ticks, a.flags = int(nforward_ticks), int(flags) | (_native.RG_FLAG_MPR_PLANE_DEPTH if MPR_PLANE_DEPTH else 0) | (_native.RG_FLAG_SENSORS if self.sensors else 0)
        self._keep.append((action, goal_quat, obs, goal_dist, active, hold, nticks, order, large_mask))
        del self._keep[:-16]

        def launch(config, active_t, redo_t, stream):
            a.config = config
            a.active_dev = None if active_t is None else active_t.data_ptr()
            a.redo_dev = None if redo_t is None else redo_t.data_ptr()
            a.stream = None if self._emul else stream.cuda_stream
            _native.check(self._L, self._L.rg_batch_step_ex(self._bh, ctypes.byref(a)), "rg_batch_step_ex")

        cur = None if self._emul else torch.cuda.current_stream(self.device)
        if capacity != "auto" or a.nsubsteps == 0:   # (no substeps: no collision stage, nothing can overflow)
            launch(_native.RG_CFG_LARGE if capacity == "large" else _native.RG_CFG_ROLLOUT, active, None, cur)
            return
        if self._redo is None:
            self._redo = torch.zeros(self.batch_size, dtype=torch.int32, device=self.device)
        self._redo.zero_()
        act0 = active
        if large_mask is not None:
            if small_mask is not None and active is None:
                small, big = small_mask, large_mask
            else:
                small = 1 - large_mask if active is None else active * (1 - large_mask)
                big = large_mask if active is None else active * large_mask
            act0 = small.contiguous()
            self._keep.append((act0, big))
            if self._emul:
                launch(_native.RG_CFG_LARGE, big.contiguous(), None, None)
            else:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(cur)
                launch(_native.RG_CFG_LARGE, big.contiguous(), None, self._side)
        base_flags = a.flags
        # (not next to a concurrent large-configuration launch on the side stream -- envs inside the pipelined reset recipe: the
        #  persistent workgroups hold every wave slot until the last work item, so that launch would run AFTER instead of beside
        #  this one; measured with a third of the envs in the recipe: 890 k vs 928 k env-steps/s, profiles/r03_ab.txt)
        #  A rule that follows the share of envs in the recipe (a count sent to pinned host memory behind each step) was tried:
        #  no better than "never beside a recipe launch" in any episode-length mix, profiles/r03_ab.txt.
        if SUBSTEP_ITEMS and large_mask is None:
            a.flags = base_flags | _native.RG_FLAG_SUBSTEP_ITEMS
        launch(_native.RG_CFG_ROLLOUT, act0, self._redo, cur)
        a.flags = base_flags | _native.RG_FLAG_RESUME      # redo[e] - 1 = the substep at which the env was handed over
        launch(_native.RG_CFG_LARGE, self._redo, None, cur)
        a.flags = base_flags
        if large_mask is not None and not self._emul:
            cur.wait_stream(self._side)

    def step(self, with_udd=True, active=None, capacity="auto"):
        """SimulationInterface.step (simulation_interface.py:176-189): nsubsteps x mj_step, then mj_forward."""
        self.env_step(nforward_ticks=1, active=active, capacity=capacity)

    def forward(self, active=None, ticks=1):
        """SimulationInterface.forward (:203-207): state-less, except that the PID callback ticks."""
        self.env_step(nsubsteps=0, nforward_ticks=ticks, active=active)

    def set_ctrl(self, ctrl: torch.Tensor):
        self.set_field(_native.RG_F_CTRL, ctrl)

    def sync(self):
        if not self._emul:
            torch.cuda.synchronize(self.device)


class EnvParams:
    """Named `[B, ...]` views into the per-env model parameter rows (include/rgstep.h RG_F_ENVPRM, rg_prm_layout)."""

    def __init__(self, sim: BatchedSimulationInterface):
        _native.check(sim._L, sim._L.rg_batch_enable_env_params(sim._bh), "rg_batch_enable_env_params")
        lay = _native.prm_layout(sim._L)
        self.rows = sim.view(_native.RG_F_ENVPRM)
        d = sim.model.dims
        nv, nu, nb, nj, ng, ns, nt = int(d[1]), int(d[2]), int(d[3]), int(d[4]), int(d[5]), int(d[6]), int(d[7])
        shapes = dict(gravity=(3,), timestep=(1,), dof_damping=(nv,), dof_armature=(nv,), dof_frictionloss=(nv,), dof_invweight0=(nv,), body_mass=(nb,),
                      body_inertia=(nb, 3), body_invweight0=(nb, 2), jnt_range=(nj, 2), tendon_range=(nt, 2), tendon_invweight0=(nt,),
                      actuator_gainprm=(nu, 10), actuator_ctrlrange=(nu, 2), actuator_forcerange=(nu, 2), geom_friction=(ng, 3), xfrc_applied=(nb, 6), site_pos=(ns, 3), geom_scale=(1,), jnt_margin=(nj,), geom_solref=(ng, 2), geom_solimp=(ng, 5))
        self._views: Dict[str, torch.Tensor] = {}
        B = sim.batch_size
        for name, shape in shapes.items():
            n = int(np.prod(shape))
            self._views[name] = self.rows[:, lay[name]:lay[name] + n].view((B,) + shape)

    def __getitem__(self, name: str) -> torch.Tensor:
        return self._views[name]

    def keys(self):
        return self._views.keys()


class BatchedData:
    """The `mjData` fields the reference's in-tree code reads, batched: every attribute is a `[B, ...]` tensor.
    State fields are views of the stepper's own buffers; the derived ones are views of the readout row each launch writes."""

    def __init__(self, sim: BatchedSimulationInterface, lay):
        self._sim = sim
        row, o_xpos, o_xquat, o_site, o_act, o_ncon, o_con, ncon_slots, o_sensor, sensor_slots = lay
        d = sim.model.dims
        nb, ns, nu = int(d[3]), int(d[6]), int(d[2])
        x, B = sim._xdata, sim.batch_size
        self.body_xpos = x[:, o_xpos:o_xpos + 3 * nb].view(B, nb, 3)
        self.xpos = self.body_xpos
        self.body_xquat = x[:, o_xquat:o_xquat + 4 * nb].view(B, nb, 4)
        self.xquat = self.body_xquat
        self.site_xpos = x[:, o_site:o_site + 3 * ns].view(B, ns, 3)
        self.actuator_force = x[:, o_act:o_act + nu]
        self._ncon = x[:, o_ncon]
        self._contact = x[:, o_con:o_con + 3 * ncon_slots].view(B, ncon_slots, 3)
        nsens = len(sim.model.names.get("sensor", []))
        assert nsens <= sensor_slots
        self._sensordata = x[:, o_sensor:o_sensor + nsens]

    qpos = property(lambda self: self._sim.view(_native.RG_F_QPOS))
    qvel = property(lambda self: self._sim.view(_native.RG_F_QVEL))
    ctrl = property(lambda self: self._sim.view(_native.RG_F_CTRL))
    time = property(lambda self: self._sim.view(_native.RG_F_TIME)[:, 0])
    ncon = property(lambda self: self._ncon.to(torch.int32))

    @property
    def sensordata(self) -> torch.Tensor:
        """data.sensordata (touch sensors), `[B, nsensor]`: the contact normal forces each sensor's site sees, evaluated at the final
        state by the last state-less forward of the step.  That forward then runs in full (collision + solve: about one more
        substep of work), so it is opt-in: first access switches it on (`sim.sensors = True`, rg_step_args.flags bit 5) and the
        values are valid from the next step on."""
        if not self._sim.sensors:
            self._sim.sensors = True
        return self._sensordata

    @property
    def contact(self):
        """(geom1 [B,K] int32, geom2 [B,K] int32, dist [B,K]); entries beyond ncon[e] are stale.  The list is that of the
        last mj_step of the launch (the state-less forward after it does not run the collision stage)."""
        c = self._contact
        return c[:, :, 0].to(torch.int32), c[:, :, 1].to(torch.int32), c[:, :, 2]

    def get_site_xpos(self, name: str) -> torch.Tensor:
        return self.site_xpos[:, self._sim.model.name2id("site", name)]

    def get_body_xpos(self, name: str) -> torch.Tensor:
        return self.body_xpos[:, self._sim.model.name2id("body", name)]

    def get_body_xquat(self, name: str) -> torch.Tensor:
        return self.body_xquat[:, self._sim.model.name2id("body", name)]
