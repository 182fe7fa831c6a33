"""Residency over the lifetime of one env.step launch: start stamp and duration of every env's wave on the constant-rate counter
(analysis build of the library: hipcc ... -DRG_CLOCK_REALTIME=2 -o ab_realtime.so, RGSTEP_LIB=...), i.e. how many waves are
resident at each moment, how long the ramp-down is, and what the slots' utilisation is.
    RGSTEP_LIB=$PWD/robogym_amd/csrc/ab_realtime.so python tools/residency_curve.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd import _native  # noqa: E402
from robogym_amd.envs.dactyl.locked import make_simple_env  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
env = make_simple_env(batch_size=B, device=dev, starting_seed=20200902, sort_dispatch=True)
env.reset()
sim = env.mujoco_simulation
gen = torch.Generator(device=dev); gen.manual_seed(3)
for _ in range(8):
    env.step(torch.rand((B, 20), generator=gen, device=dev) * 2 - 1)
def sim_span(order, dur, slots=3072):
    import heapq
    h = [0.0] * slots; end = 0.0
    for e in order:
        t = heapq.heappop(h) + dur[e]; end = max(end, t); heapq.heappush(h, t)
    return end


prev = None
for rep in range(4):
    a = torch.rand((B, 20), generator=gen, device=dev) * 2 - 1
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); env.step(a); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    dur = sim.view(_native.RG_F_COST)[:, 0].double().cpu().numpy()
    start = sim.view(_native.RG_F_TIME)[:, 0].double().cpu().numpy()
    start = (start - start.min()) % (1 << 24)
    end = start + dur
    T = end.max()
    ticks_per_ms = 1e5
    print("launch %.3f ms (events); first start .. last end = %.3f ms; waves' durations: mean %.3f ms, min %.3f, max %.3f" % (ms, T / ticks_per_ms, dur.mean() / ticks_per_ms, dur.min() / ticks_per_ms, dur.max() / ticks_per_ms))
    grid = np.linspace(0, T, 41)
    res = [(int(((start <= t) & (end > t)).sum())) for t in grid]
    print("resident waves at 40 equal steps of the launch:", " ".join(str(r) for r in res))
    print("slot utilisation (3072 slots): %.1f %%; peak residency %d; time with >= 95 %% of the peak: %.1f %% of the launch" % (100 * dur.sum() / (3072 * T), max(res), 100 * np.mean(np.array(res) >= 0.95 * max(res))))
    # when did the dispatcher start the last wave, and how long did the ramp-down take
    print("last wave started at %.3f ms; ramp-down (from the last start to the end): %.3f ms" % (start.max() / ticks_per_ms, (T - start.max()) / ticks_per_ms))
    if rep == 0:
        prev = dur
        continue
    rank = np.argsort(np.argsort(start))
    dec = [dur[(rank >= k * B // 10) & (rank < (k + 1) * B // 10)] / ticks_per_ms for k in range(10)]
    print("durations by start decile (ms), mean: " + " ".join("%.2f" % d.mean() for d in dec))
    print("rank correlation of this step's durations with the previous step's (what the dispatch order is sorted by): %.3f" % np.corrcoef(np.argsort(np.argsort(dur)), np.argsort(np.argsort(prev)))[0, 1])
    print("list scheduling of these durations on 3072 slots (ms): sorted by the previous step's %.2f | sorted by their own (perfect prediction) %.2f | env order %.2f | sum / slots %.2f | longest %.2f" % (
        sim_span(np.argsort(-prev), dur) / ticks_per_ms, sim_span(np.argsort(-dur), dur) / ticks_per_ms, sim_span(np.arange(B), dur) / ticks_per_ms, dur.sum() / 3072 / ticks_per_ms, dur.max() / ticks_per_ms))
    prev = dur
