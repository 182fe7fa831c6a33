"""world_size-2 gloo test of the N>1 path: each rank steps its own shard of envs (kernel source run by the
CPU emulation harness) and the observation rows are all-gathered rank-major."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robogym_amd import _native
    from robogym_amd.distributed import ShardedObservationGather
    from robogym_amd.envs.dactyl.locked import BatchedLockedEnv, LockedEnvConstants

    lib = _native.bind(os.path.join(ROOT, "tests", "emul", "librgstep_emul.so"))
    c = LockedEnvConstants(reset_initial_steps=1, n_random_initial_steps=1, mujoco_substeps=1)
    env = BatchedLockedEnv(2, constants=c, starting_seed=100 + rank, lib=lib)
    env.reset()
    gather = ShardedObservationGather(2, env.mujoco_simulation.obs_dim, "cpu")
    gen = torch.Generator(); gen.manual_seed(rank)
    obs, reward, done, info = env.step(torch.rand((2, 20), generator=gen) * 2 - 1)
    full = gather(env._obs_buf)
    gather.start(env._obs_buf)                       # overlapped variant: same result, one gather in flight
    assert torch.equal(gather.finish(), full)
    assert full.shape == (world * 2, env.mujoco_simulation.obs_dim)
    assert torch.equal(full[2 * rank:2 * rank + 2], env._obs_buf)
    np.save(os.path.join(out_dir, "full_%d.npy" % rank), full.numpy())
    np.save(os.path.join(out_dir, "own_%d.npy" % rank), env._obs_buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_observation_all_gather_gloo(tmp_path, emul_lib):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    full0, full1 = np.load(tmp_path / "full_0.npy"), np.load(tmp_path / "full_1.npy")
    np.testing.assert_array_equal(full0, full1)                     # every rank sees the same global batch
    np.testing.assert_array_equal(full0[0:2], np.load(tmp_path / "own_0.npy"))
    np.testing.assert_array_equal(full0[2:4], np.load(tmp_path / "own_1.npy"))
    assert not np.array_equal(full0[0:2], full0[2:4])               # different seeds -> different shards
