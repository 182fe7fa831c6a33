"""Batched counterpart of the reference's `SimulationInterface`
(/root/reference/robogym/mujoco/simulation_interface.py:25-250): same method names, with a
leading env-batch dimension on every array and state living in HBM inside the HIP stepper
(C ABI: include/rgstep.h).  One `BatchedSimulationInterface` stands for B independent
(model, MjSim) pairs of the reference (robot_env.py:328-350)."""
import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from robogym_amd import _native
from robogym_amd.mujoco.kernel_tables import derive_kernel_tables
from robogym_amd.mujoco.model_blob import pack_model


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
        self._mh = self._L.rg_model_create(blob, len(blob), err, 512)
        if not self._mh:
            raise _native.NativeError("rg_model_create: " + err.value.decode())
        index = self.device.index or 0
        self._bh = self._L.rg_batch_create(self._mh, self.batch_size, index)
        if not self._bh:
            raise _native.NativeError("rg_batch_create: " + self._L.rg_last_error().decode())
        d = model.dims
        self.nq, self.nv, self.nu = int(d[0]), int(d[1]), int(d[2])
        self.qpos_idxs: Dict[str, np.ndarray] = {}
        self.qvel_idxs: Dict[str, np.ndarray] = {}
        self._stream = None

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
                _native.RG_F_DEBUG: self._L.rg_debug_size()}[field]

    def get_field(self, field) -> torch.Tensor:
        dtype = torch.int32 if field == _native.RG_F_STATUS else torch.float32
        out = torch.empty((self.batch_size, self._ncols(field)), dtype=dtype, device=self.device)
        self.sync()
        _native.check(self._L, self._L.rg_batch_copy(self._bh, field, out.data_ptr(), 0, 0 if self._emul else 1), "rg_batch_copy")
        return out

    def set_field(self, field, value: torch.Tensor):
        dtype = torch.int32 if field == _native.RG_F_STATUS else torch.float32
        value = torch.as_tensor(value, dtype=dtype, device=self.device).reshape(self.batch_size, self._ncols(field)).contiguous()
        self.sync()
        _native.check(self._L, self._L.rg_batch_copy(self._bh, field, value.data_ptr(), 1, 0 if self._emul else 1), "rg_batch_copy")

    def copy_rows(self, field, value: torch.Tensor, mask: torch.Tensor, col0: int = 0):
        """Asynchronous masked write (rg_batch_copy_rows): rows of the envs with mask != 0, columns
        [col0, col0 + value.shape[1]) <- value.  Stream-ordered with the step launches; no host sync."""
        dtype = torch.int32 if field == _native.RG_F_STATUS else torch.float32
        value = torch.as_tensor(value, dtype=dtype, device=self.device).reshape(self.batch_size, -1).contiguous()
        mask = mask.to(device=self.device, dtype=torch.int32).contiguous()
        stream = None if self._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _native.check(self._L, self._L.rg_batch_copy_rows(self._bh, field, value.data_ptr(), mask.data_ptr(), int(col0), int(value.shape[1]), stream), "rg_batch_copy_rows")
        self._keepalive = (value, mask)   # the launch is asynchronous: keep the operands alive until the next call

    @property
    def qpos(self) -> torch.Tensor:
        return self.get_field(_native.RG_F_QPOS)

    @property
    def qvel(self) -> torch.Tensor:
        return self.get_field(_native.RG_F_QVEL)

    def get_qpos(self, group: str) -> torch.Tensor:
        return self.qpos[:, torch.as_tensor(self.qpos_idxs[group], device=self.device)]

    def set_qpos(self, group: str, value, mask: Optional[torch.Tensor] = None):
        q = self.qpos
        idx = torch.as_tensor(self.qpos_idxs[group], device=self.device)
        value = torch.as_tensor(value, dtype=torch.float32, device=self.device).expand(self.batch_size, len(idx))
        if mask is None:
            q[:, idx] = value
        else:
            rows = mask.nonzero().flatten()
            q[rows[:, None], idx[None, :]] = value[rows]
        self.set_field(_native.RG_F_QPOS, q)

    def add_qpos(self, group: str, value, mask: Optional[torch.Tensor] = None):
        idx = torch.as_tensor(self.qpos_idxs[group], device=self.device)
        cur = self.qpos[:, idx]
        self.set_qpos(group, cur + torch.as_tensor(value, dtype=torch.float32, device=self.device), mask)

    def get_qvel(self, group: str) -> torch.Tensor:
        return self.qvel[:, torch.as_tensor(self.qvel_idxs[group], device=self.device)]

    def set_qvel(self, group: str, value):
        v = self.qvel
        v[:, torch.as_tensor(self.qvel_idxs[group], device=self.device)] = torch.as_tensor(value, dtype=torch.float32, device=self.device)
        self.set_field(_native.RG_F_QVEL, v)

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
        return self.get_field(_native.RG_F_STATUS)[:, 0]

    # ------------------------------------------------------------------ stepping
    def set_env(self, ints, pos_to_ctrl: np.ndarray, success_threshold: float):
        ints = np.asarray(ints, dtype=np.int32)
        p2c = np.ascontiguousarray(pos_to_ctrl, dtype=np.float32)
        _native.check(self._L, self._L.rg_batch_set_env(self._bh, ints.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(ints),
                                                        p2c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), float(success_threshold)), "rg_batch_set_env")
        self.obs_dim = self._L.rg_obs_dim(self._bh)

    def _ptr(self, t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    def env_step(self, action=None, goal_quat=None, obs=None, goal_dist=None, active=None, nsubsteps=None, nforward_ticks=3, flags=0):
        """One reference env.step worth of physics for the whole batch (async on the current stream).
        `active`: optional int32 [B]; envs with 0 are left untouched."""
        for t in (action, goal_quat, obs, goal_dist):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device)
        assert active is None or (active.dtype == torch.int32 and active.is_contiguous() and active.device == self.device)
        stream = None if self._emul else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _native.check(self._L, self._L.rg_batch_step(self._bh, self._ptr(action), self._ptr(goal_quat), self._ptr(obs), self._ptr(goal_dist), self._ptr(active),
                                                     self.n_substeps if nsubsteps is None else int(nsubsteps), int(nforward_ticks), int(flags), stream), "rg_batch_step")

    def step(self, with_udd=True, active=None):
        """SimulationInterface.step (simulation_interface.py:176-189): nsubsteps x mj_step, then mj_forward."""
        self.env_step(nforward_ticks=1, active=active)

    def forward(self, active=None, ticks=1):
        """SimulationInterface.forward (:203-207): state-less, except that the PID callback ticks."""
        self.env_step(nsubsteps=0, nforward_ticks=ticks, active=active)

    def set_ctrl(self, ctrl: torch.Tensor):
        self.set_field(_native.RG_F_CTRL, ctrl)

    def sync(self):
        if not self._emul:
            torch.cuda.synchronize(self.device)
