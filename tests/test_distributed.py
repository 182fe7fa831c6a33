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


def test_bench_distributed_path_under_gloo(emul_lib):
    """bench.py's own N > 1 code path (process group, shard per rank, packed observation + reward + done all-gather
    overlapped with the next step, barrier + max-over-ranks timing, one JSON line from rank 0), launched exactly as the
    driver launches it, on CPU: 2 ranks, gloo, the kernel source on the emulation harness (RG_BENCH_EMUL_LIB test hook)."""
    import json
    import subprocess

    port = 29600 + os.getpid() % 2000
    env = dict(os.environ, RG_BENCH_EMUL_LIB=os.path.join(ROOT, "tests", "emul", "librgstep_emul.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 4 and "170 floats" in r["config"]["gathered_row"]


def test_bench_ycb_distributed_path_under_gloo(emul_lib):
    """`bench.py --workload ycb --gpus 2` (BASELINE.json configs[4] is an 8-GPU line: 4096 envs per GPU): the N > 1 code path of the rearrange
    workloads -- a shard of envs per rank, packed observation rows all-gathered behind the next step, barrier + max-over-ranks timing, one JSON
    line from rank 0 -- on CPU with 2 ranks, gloo and the emulation harness (one mj_step per world per env.step, a one-step reset recipe)."""
    import json
    import subprocess

    port = 31600 + os.getpid() % 2000
    env = dict(os.environ, RG_BENCH_EMUL_LIB=os.path.join(ROOT, "tests", "emul", "librgstep_emul.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--workload", "ycb", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0 and r["scaling"] == "weak" and "configs[4]" in r["metric"]
    assert r["config"]["global_batch"] == 2 and r["config"]["collective_backend"] == "gloo" and r["config"]["status_bits"] == 0


def test_bench_spawns_its_own_ranks(emul_lib):
    """`python bench.py --gpus 2` WITHOUT a launcher (--gpus was parsed and ignored): the script
    re-executes itself under torch.distributed.run with 2 ranks, asserts the world size equals --gpus and prints
    n_gpus 2 with the rank count of the process group."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RG_BENCH_EMUL_LIB"] = os.path.join(ROOT, "tests", "emul", "librgstep_emul.so")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["ranks"] == 2 and r["config"]["collective_backend"] == "gloo"
    assert r["config"]["global_batch"] == 4


def test_bench_refuses_a_world_size_that_is_not_gpus(emul_lib):
    import subprocess

    env = dict(os.environ, RG_BENCH_EMUL_LIB=os.path.join(ROOT, "tests", "emul", "librgstep_emul.so"), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 2" in out.stderr
