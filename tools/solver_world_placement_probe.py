"""The TCP solver's world alone under LDS placements of its stage arrays (a library built with -DRB_LDS_ARENA; RB_LDS_PLACE is applied to the SOLVER model only):
launch time of its 40 mj_steps at several batch sizes.  The question it answers: would a configuration with a small LDS image plus an arena for this 8-dof world
(27 bodies, 45 geoms) beat the scratch-row layout at 16 workgroups per CU?
    RGSTEP_LIB=ab_libs/librgstep_arena.so python tools/solver_world_placement_probe.py <placement|none> B [B ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robogym_amd.envs.rearrange.blocks as blk  # noqa: E402

place = sys.argv[1]
orig = blk.LargeModelSimulation
count = [0]


def wrapped(model, *a, **k):
    count[0] += 1
    if count[0] == 2:      # (the env builds the main world first, the solver world second)
        if place != "none":
            os.environ["RB_LDS_PLACE"] = place
        for k2 in ("MAXCON", "MAXROW"):      # SOLVER_MAXCON / SOLVER_MAXROW: capacities of the solver world's scratch row (RB_SCRATCH_MAXCON / RB_SCRATCH_MAXROW for that model only)
            if os.environ.get("SOLVER_" + k2):
                os.environ["RB_SCRATCH_" + k2] = os.environ["SOLVER_" + k2]
    try:
        return orig(model, *a, **k)
    finally:
        for k2 in ("RB_LDS_PLACE", "RB_SCRATCH_MAXCON", "RB_SCRATCH_MAXROW"):
            os.environ.pop(k2, None)


blk.LargeModelSimulation = wrapped
for B in [int(x) for x in sys.argv[2:]]:
    count[0] = 0
    env = blk.BatchedBlockRearrangeEnv(B, stabilize_steps=20, n_random_initial_steps=2, settle_steps=10)
    env.reset()
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(0)
    act = lambda: torch.rand((B, 6), generator=gen, device="cuda:0") * 2 - 1
    for _ in range(3):
        env.step(act())
    a = act()
    ts = []
    for _ in range(5):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); env.solver_sim.step_tcp(env.sim, a, env.tcp); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    st = int(env.solver_sim.status.cpu().numpy().astype("int64").max())
    print("placement %-28s B %5d: solver world launch %.2f ms (min of 5; lds %s B per workgroup, capacity %d contacts / %d rows, max status %d) -> %.0f env-launches/ms" % (
        place, B, min(ts), env.solver_sim.info.get("lds_bytes", "?"), env.solver_sim.info["maxcon"], env.solver_sim.info["maxrow"], st, B / min(ts)), flush=True)
    del env
    torch.cuda.empty_cache()
