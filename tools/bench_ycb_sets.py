"""rearrange/ycb with K different object sets across the batch (GroupedYcbRearrangeEnv: one compiled model per group, groups on their own streams) next to the
single-set env: env-steps/s at B envs after a shortened reset recipe.      python tools/bench_ycb_sets.py [B] [steps] [multi|streams]
multi (default): the groups' physics phases as ONE launch each (rb_multi_launch); streams: one launch chain per group on its own stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robogym_amd.envs.rearrange.ycb import make_simple_env      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
MULTI = not (len(sys.argv) > 3 and sys.argv[3] == "streams")
print("groups' launches: %s" % ("one launch per phase over all groups (rb_multi_launch)" if MULTI else "one chain per group on its own stream"))
SINGLES = len(sys.argv) > 4 and sys.argv[4] == "singles"      # also every set of the mixed batches on its own: a mixed batch's step is as long as its slowest set's
for sets in ((0,), (0, 1), (0, 1, 2, 4), (0, 1, 2, 3, 4, 5)) + (((1,), (2,), (4,)) if SINGLES else ()):
    if B % len(sets):
        continue
    kw = dict(multi_launch=MULTI) if len(sets) > 1 else {}
    if len(sets) == 1 and sets[0] != 0:
        from robogym_amd.envs.rearrange.xml import load_ycb_model
        kw["main_model"] = load_ycb_model(8, set_index=sets[0])
    env = make_simple_env(batch_size=B, starting_seed=3, object_sets=sets if len(sets) > 1 else None, stabilize_steps=20, n_random_initial_steps=2, settle_steps=20, **kw)
    env.reset()
    dev = env.device
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    act = lambda: torch.rand((B, 6), generator=gen, device=dev) * 2 - 1
    for _ in range(2):
        env.step(act())
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(T):
        env.step(act())
    torch.cuda.synchronize()
    el = time.time() - t0
    status = int(env.status().max()) if hasattr(env, "status") else int(max(env.sim.status.max().item(), env.solver_sim.status.max().item()))
    print("object sets %-20s %d envs (%d per set): %.1f ms per env.step = %.0f env-steps/s, status bits %d" % (sets, B, B // len(sets), 1e3 * el / T, B * T / el, status), flush=True)
    del env
    torch.cuda.empty_cache()
