"""Data-parallel sharding of the env batch over the GPUs of one node.

Envs never interact, so rank r simply owns envs [r*B, (r+1)*B); model tables are replicated.  The only
exchange the path has is the end-of-step all-gather of the observation rows (RCCL over xGMI via
torch.distributed's "nccl" backend; "gloo" on CPU for tests) for a learner that consumes the full
batch on every rank."""
from typing import Optional

import os

import torch
import torch.distributed as dist


class ShardedObservationGather:
    def __init__(self, local_batch: int, obs_dim: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_batch = local_batch
        self.buffer: Optional[torch.Tensor] = None
        self._staging: Optional[torch.Tensor] = None
        self._work = None
        self._last: Optional[torch.Tensor] = None
        # RG_BENCH_FORCE_DIST=1 (with an initialised process group): run the collective even with one rank, so the RCCL path can be
        # executed on a single-GPU box
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("RG_BENCH_FORCE_DIST") == "1")
        if self.active:
            self.buffer = torch.empty((self.world * local_batch, obs_dim), dtype=torch.float32, device=device)

    def __call__(self, local_obs_rows: torch.Tensor) -> torch.Tensor:
        """[B, obs_dim] on this rank -> [world*B, obs_dim] on every rank (rank-major order)."""
        if not self.active:
            return local_obs_rows
        dist.all_gather_into_tensor(self.buffer, local_obs_rows.contiguous(), group=self.group)
        return self.buffer

    def start(self, local_obs_rows: torch.Tensor):
        """Overlapped variant: snapshot the rows into a staging buffer (so the next env.step may overwrite the
        live buffer) and start the all-gather without making the compute stream wait for it; `finish()` returns
        the gathered tensor of the PREVIOUS `start()`.  At most one gather is in flight."""
        if not self.active:
            self._last = local_obs_rows
            return
        self.finish()
        if self._staging is None:
            self._staging = torch.empty_like(local_obs_rows)
        self._staging.copy_(local_obs_rows)
        self._work = dist.all_gather_into_tensor(self.buffer, self._staging, group=self.group, async_op=True)

    def finish(self) -> Optional[torch.Tensor]:
        if not self.active:
            return self._last
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.buffer

    def global_env_ids(self) -> torch.Tensor:
        return torch.arange(self.rank * self.local_batch, (self.rank + 1) * self.local_batch)
