"""rearrange/ycb (BASELINE.json configs[4]: 8 YCB objects, one free body per object, one mesh geom per convex part) — the shipped model with
a FIXED object set (robogym_amd/envs/rearrange/xml.py load_ycb_model), on the CPU oracle and on `rb_step_kernel`'s medium configuration
(one wave per env, 56 dofs).  Same protocols as tests/test_rearrange_kernel.py."""
import os

import numpy as np
import pytest
import torch

from robogym_amd.envs.rearrange.xml import load_solver_model, load_ycb_model, object_bounding_boxes
from robogym_amd.mujoco.large_simulation import LargeModelSimulation
from tests.test_rearrange_kernel import sync_from_oracle, tcp_args

N = 8
NV = 56


@pytest.fixture(scope="module")
def models():
    return load_ycb_model(N), load_solver_model()


def _oracle_env(models, n_substeps, settle, seed=0, spread=False, f32=False):
    from oracle import rearrange_oracle as RO

    main, solver = models
    env = RO.OracleRearrangeEnv(main, solver, N, n_substeps=n_substeps, f32=f32)
    bb = object_bounding_boxes(main, N)
    rng = np.random.RandomState(seed)
    table_top = 0.453 + 0.03324
    pos, quat = [], []
    for i in range(N):
        yaw = rng.uniform(0, 2 * np.pi)
        c, s = np.cos(yaw), np.sin(yaw)
        centre = np.array([c * bb[i, 0] - s * bb[i, 1], s * bb[i, 0] + c * bb[i, 1], bb[i, 2]])
        box = np.array([1.25 + 0.15 * (i % 4), 0.52 + 0.3 * (i // 4), table_top + bb[i, 5] + 0.002])     # inside the reference's placement area (simulation/base.py:992-1010)
        if spread:      # (the other shipped sets hold objects up to 40 cm long: a wider grid over the table so that nothing starts interpenetrating)
            box[:2] = [1.02 + 0.27 * (i % 4), 0.40 + 0.52 * (i // 4)]
        pos.append(box - centre); quat.append([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    env.set_object_poses(pos, quat)
    # gripper half open: at qpos0 the two finger pads touch face to face, a degenerate box - box case whose number of clipped points (4 or 5)
    # is decided by rounding
    o, A = env.main.sim, main.arrays
    o.ctrl[env.main.grip_act] = 0.5 * (A["actuator_ctrlrange"][env.main.grip_act, 0] + A["actuator_ctrlrange"][env.main.grip_act, 1])
    env.main.sim.forward()
    for _ in range(settle):
        env.main.step()
    return env


def test_elliptic_dual_solver_agrees_with_newton_on_mesh_contacts(models, oracle_lib):
    """tests/test_rearrange_oracle.py::test_elliptic_dual_solver_agrees_with_newton on the ycb world: eight mesh objects settling on the table (convex parts against the
    table box through MPR, ~19 contacts, ~115 rows, elliptic cones): the oracle's independent dual solver and its Newton solver give the same accelerations."""
    env = _oracle_env(models, 1, settle=40)
    s = env.main.sim
    errs, contacts = [], 0
    for _ in range(3):
        for _ in range(40):
            env.main.step()
        s.forward()
        q, sweeps = s.solve_pgs(max_sweeps=400000, tol=1e-11)
        assert 0 < sweeps < 400000
        errs.append(float(np.abs(q - s.qacc).max() / max(1.0, np.abs(s.qacc).max())))
        contacts += s.ncon
    assert max(errs) < 1e-7 and contacts >= 30, (errs, contacts)


def test_model_and_object_set(models):
    main, _ = models
    A = main.arrays
    assert int(A["dims"][0]) == 7 * N + 8 and int(A["dims"][1]) == NV            # arm 6 + gripper 2 + a free joint per object
    assert main.names["object_mesh"] == ["055_baseball", "058_golf_ball", "070-b_colored_wood_blocks", "072-b_toy_airplane", "072-b_toy_airplane", "011_banana", "025_mug",
                                         "043_phillips_screwdriver"]
    parts = [int((A["geom_bodyid"] == main.name2id("body", "object%d" % i)).sum()) for i in range(N)]
    assert parts == [1, 1, 1, 3, 3, 3, 29, 3]
    for i in range(N):     # make_mesh_object: the body origin is the combined centre of mass, so the body's inertial frame sits at it
        b = main.name2id("body", "object%d" % i)
        assert np.abs(A["body_ipos"][b]).max() < 1e-6 and A["body_mass"][b] > 0 and (A["body_inertia"][b] > 0).all()
    b = main.name2id("body", "object0")     # the baseball: a 3.7 cm sphere at density 1000
    r = object_bounding_boxes(main, N)[0, 3:].mean()
    assert abs(A["body_mass"][b] - 1000 * 4 / 3 * np.pi * r ** 3) < 0.03 * A["body_mass"][b]
    assert np.abs(A["body_inertia"][b] - 0.4 * A["body_mass"][b] * r * r).max() < 0.05 * A["body_inertia"][b].max()


def test_objects_come_to_rest_on_the_table_oracle(models, oracle_lib):
    env = _oracle_env(models, 1, settle=400)
    o, main = env.main.sim, models[0]
    bb = object_bounding_boxes(main, N)
    table_top = 0.453 + 0.03324
    for i in range(N):
        qa = env.obj_q[i]
        z = o.qpos[qa + 2]
        assert abs(z + bb[i, 2] - bb[i, 5] - table_top) < 0.01, (i, z)      # lying on the table as it was placed, not through it
    va = int(main.arrays["jnt_dofadr"][main.names["joint"].index("object0:joint")])
    assert np.abs(o.qvel[va:va + 6 * N]).max() < 0.5 and o.ncon >= N + 4       # (one contact per touching convex part; balls may still roll slowly)


def _stage_dump(models, lib, device, spread=False, min_contacts=12, normal_tol=1e-3):
    env = _oracle_env(models, 1, settle=60, spread=spread)
    o = env.main.sim
    sim = LargeModelSimulation(models[0], 1, device=device, n_substeps=1, lib=lib, hand=False)
    assert sim.info["threads"] == 64 and sim.info["lds_bytes"] <= 15360            # the medium configuration: one wave per env, 10 envs per CU
    sync_from_oracle(sim, o)
    sim.env_step(nsubsteps=1, nforward_ticks=0, flags=1)
    sim.sync()
    o.step()
    dbg = sim.scratch("dbg")[0].cpu().numpy()
    assert int(sim.status[0]) == 0
    ncon_k, nefc_k = int(dbg[0]), int(dbg[1])
    assert ncon_k == o.ncon + o.neq and nefc_k == o.nefc and o.ncon >= min_contacts, (ncon_k, o.ncon, nefc_k, o.nefc)
    con = sim.scratch("contact")[0].cpu().numpy().reshape(-1, 32)[:ncon_k]
    key = lambda t: (t[0], t[1], round(float(t[3][0]), 4), round(float(t[3][1]), 4))
    kc = sorted([(int(c[27]), int(c[28]), float(c[0]), c[1:4].copy(), c[4:7].copy()) for c in con[1:]], key=key)
    oc = sorted([(c["geom1"], c["geom2"], c["dist"], c["pos"], c["frame"][0]) for c in o.contacts()], key=key)
    # Convex parts lying FLAT on the table: the depth and the normal of the MPR contact are well defined, the point is not (any point of the
    # overlapping faces has that depth; libccd returns a vertex-weighted point of the last portal, which rounding moves by millimetres): the
    # keys are (pair, depth), positions are compared as a distribution
    key = lambda t: (t[0], t[1], round(float(t[2]), 5))
    kc, oc = sorted(kc, key=key), sorted(oc, key=key)
    perr = []
    for a, b in zip(kc, oc):
        assert a[0] == b[0] and a[1] == b[1]
        # (the normal of an MPR contact is its last portal's: the portal may tilt by mpr_tolerance / its own size, 1e-6 m over a millimetre -- the fp64 oracle lands
        #  on the table's face normal exactly, the fp32 kernel within that tilt)
        assert abs(a[2] - b[2]) < 5e-6 and np.abs(a[4] - b[4]).max() < normal_tol, (a, b)
        perr.append(np.abs(a[3] - b[3]).max())
    assert np.median(perr) < 2e-5 and max(perr) < 0.03, perr
    assert np.abs(dbg[8:8 + NV] - o.qfrc_bias).max() < 1e-4 * max(1.0, np.abs(o.qfrc_bias).max())
    assert np.abs(dbg[8 + 3 * NV:8 + 4 * NV] - o.qacc_smooth).max() < 1e-4 * np.abs(o.qacc_smooth).max()
    assert np.abs(sim.qpos[0].cpu().numpy() - o.qpos).max() < 5e-6 and np.abs(sim.qvel[0].cpu().numpy() - o.qvel).max() < 2e-3


def _resync(models, lib, device, n_substeps, nsteps, spread=False, classify=False):
    from tests.test_rearrange_kernel import contact_history

    env = _oracle_env(models, n_substeps, settle=30, seed=1, spread=spread)
    om, oc = env.main.sim, env.solver.sim
    sm = LargeModelSimulation(models[0], 1, device=device, n_substeps=n_substeps, lib=lib, hand=False)
    sc = LargeModelSimulation(models[1], 1, device=device, n_substeps=n_substeps, lib=lib, hand=False)
    args = tcp_args(env)
    rng = np.random.RandomState(3)
    errs, same = [], []
    for step in range(nsteps):
        a = rng.uniform(-1, 1, 6)
        sync_from_oracle(sm, om); sync_from_oracle(sc, oc)
        km, kc = sm.stats[0].cpu().numpy().astype(np.float64), sc.stats[0].cpu().numpy().astype(np.float64)
        env.main.ncon_sum = env.main.nefc_sum = env.solver.ncon_sum = env.solver.nefc_sum = 0
        sc.step_tcp(sm, torch.tensor(a[None].astype(np.float32), device=sm.device), args)
        sm.env_step(nforward_ticks=2, flags=32)
        sm.sync()
        env.env_step(a)
        e = lambda x, y: float(np.abs(x.cpu().numpy().astype(np.float64) - y).max())
        errs.append((e(sm.ctrl[0], om.ctrl), e(sm.qpos[0], om.qpos), e(sm.qvel[0], om.qvel)))
        assert int(sm.status[0]) == 0 and int(sc.status[0]) == 0
        same.append(contact_history(sm, env.main, km) and contact_history(sc, env.solver, kc))
    return (np.array(errs), np.array(same)) if classify else np.array(errs)


def _copy_oracle_state(src, dst):
    """state of one oracle world -> its precision twin (the fields sync_from_oracle hands the kernel)"""
    for name in ("qpos", "qvel", "ctrl", "pid", "qacc_warmstart"):
        getattr(dst, name)[:] = getattr(src, name)
    if src.nmocap:
        dst.mocap_pos[:] = src.mocap_pos; dst.mocap_quat[:] = src.mocap_quat
    if src.neq:
        dst.eq_data[:] = src.eq_data; dst.eq_active()[:] = src.eq_active()


def oracle_pair_spread(models, n_substeps, nsteps, spread=False):
    """The `_resync` protocol with the oracle's own FLOAT build in the kernel's place: errs[step] = (ctrl, qpos, qvel) max error of the float twin's main world against
    the double oracle after one env.step from the double oracle's (fp32-rounded) state, same[step] = both went through the same contact / row counts in both
    worlds.  This is what fp32 arithmetic alone does to the protocol; tools/gen_ycb_pair_spread.py records it as tests/golden/ycb_pair_spread.json."""
    env = _oracle_env(models, n_substeps, settle=30, seed=1, spread=spread)
    twin = _oracle_env(models, n_substeps, settle=30, seed=1, spread=spread, f32=True)
    rng = np.random.RandomState(3)
    errs, same = [], []
    for step in range(nsteps):
        a = rng.uniform(-1, 1, 6)
        for w in ("main", "solver"):
            o = getattr(env, w).sim
            for name in ("qpos", "qvel", "ctrl", "pid", "qacc_warmstart"):      # (the rounding sync_from_oracle applies)
                getattr(o, name)[:] = getattr(o, name).astype(np.float32).astype(np.float64)
            _copy_oracle_state(o, getattr(twin, w).sim)
        for e_ in (env, twin):
            e_.main.ncon_sum = e_.main.nefc_sum = e_.solver.ncon_sum = e_.solver.nefc_sum = 0
        env.env_step(a); twin.env_step(a)
        om, tm = env.main.sim, twin.main.sim
        d = lambda n: float(np.abs(np.asarray(getattr(tm, n), dtype=np.float64) - getattr(om, n)).max())
        errs.append((d("ctrl"), d("qpos"), d("qvel")))
        same.append(all(int(getattr(env, w).ncon_sum) == int(getattr(twin, w).ncon_sum) and int(getattr(env, w).nefc_sum) == int(getattr(twin, w).nefc_sum) for w in ("main", "solver")))
    return np.array(errs), np.array(same)


def test_ycb_stage_dump_matches_oracle_emul(models, emul_lib, oracle_lib):
    _stage_dump(models, emul_lib, "cpu")


@pytest.mark.gpu
def test_ycb_stage_dump_matches_oracle_gpu(models, oracle_lib):
    _stage_dump(models, None, "cuda:0")


# ---- the tolerances of the re-synchronised ycb tests are anchored on the ORACLE PAIR: tests/golden/ycb_pair_spread.json holds, per shipped
# object set, what the oracle's own float build does under this very protocol against the double oracle (tools/gen_ycb_pair_spread.py: 12 env.steps, split by contact
# history).  The HIP kernel is held to PAIR_FACTOR x that spread: it is another fp32 evaluation of the same algorithm (different summation orders, fused multiply-adds),
# and its 4-10 sampled steps are compared with the maximum over the pair's 12.  The floors are the resolution of the quantities themselves (ctrl: one fp32 ulp of a
# joint target; qpos / qvel: the pair's medians).
PAIR_FACTOR = 3.0
PAIR_FLOOR = np.array([4e-6, 5e-5, 2e-2])


def _pair_spread(set_index):
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ycb_pair_spread.json")) as f:
        rec = json.load(f)["sets"][str(set_index)]
    return {k: np.asarray(v) for k, v in rec.items() if isinstance(v, list)}


def _assert_within_pair_spread(errs, same, set_index):
    sp = _pair_spread(set_index)
    bound_all = np.maximum(PAIR_FACTOR * sp["max_all"], PAIR_FLOOR)
    bound_same = np.maximum(PAIR_FACTOR * sp["max_same"], PAIR_FLOOR)
    assert (errs < bound_all).all(), (set_index, errs.max(axis=0), bound_all)
    if same.any():
        assert (errs[same] < bound_same).all(), (set_index, errs[same].max(axis=0), bound_same)
    assert (np.median(errs, axis=0) < np.maximum(PAIR_FACTOR * sp["median_all"], PAIR_FLOOR)).all(), (set_index, np.median(errs, axis=0), sp["median_all"])


def test_ycb_pair_spread_fixture_is_what_the_oracle_pair_gives(models, oracle_lib):
    """tests/golden/ycb_pair_spread.json is reproducible from the two oracle builds (set 0, the fixture's own 12 steps)."""
    errs, same = oracle_pair_spread(models, 40, 12)
    sp = _pair_spread(0)
    assert np.allclose(errs.max(axis=0), sp["max_all"], rtol=1e-3, atol=1e-12) and np.allclose(np.median(errs, axis=0), sp["median_all"], rtol=1e-3, atol=1e-12)
    # and the pair itself says what fp32 does to this protocol: positions to ~1e-4, velocities to a few 1e-2 rad/s on parts lying flat (the MPR contact POINT of
    # coplanar faces is defined up to rounding, _stage_dump) -- the scale the kernel's bounds inherit
    assert errs[:, 1].max() < 1e-3 and errs[:, 2].max() < 0.2


@pytest.mark.gpu
def test_ycb_resync_env_steps_gpu(models, oracle_lib):
    """re-synchronised env.steps of the dual simulation, 40 + 40 mj_steps each (protocol of test_rearrange_resync_env_steps_gpu), held to the oracle pair's spread"""
    errs, same = _resync(models, None, "cuda:0", 40, 10, classify=True)
    _assert_within_pair_spread(errs, same, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("set_index", [1, 2, 3, 4, 5])
def test_ycb_other_object_sets_match_oracle_gpu(set_index, oracle_lib):
    """the shipped object sets 1-5 (objects of 20-33 convex parts, 177-305 geoms, up to 10 k static pairs) had no oracle comparison.  Stage
    dump of one mj_step with the objects lying on the table (contact list pair by pair, generalized forces, the new state) and 4 re-synchronised env.steps of
    40 + 40 mj_steps each, bounds of set 0's tests."""
    models = (load_ycb_model(N, set_index=set_index), load_solver_model())
    _stage_dump(models, None, "cuda:0", spread=True, min_contacts=8, normal_tol=2e-3)
    errs, same = _resync(models, None, "cuda:0", 40, 4, spread=True, classify=True)
    print("ycb object set %d: ctrl / qpos / qvel max %s; same contact history in %d of 4 steps" % (set_index, errs.max(axis=0), same.sum()))
    # bounds: PAIR_FACTOR x what the oracle's float build does on THIS set under this protocol (all steps / same-contact-history steps), tests/golden/ycb_pair_spread.json --
    # e.g. set 5 (power drill and cups resting on many coplanar parts): the pair itself shows qvel 0.53 on a step with a contact event and 0.21 on same-history steps
    _assert_within_pair_spread(errs, same, set_index)


def test_ycb_other_object_set_stage_dump_emul(emul_lib, oracle_lib):
    """(one of the other sets on the emulation harness: the same stage dump as the GPU test's)"""
    models = (load_ycb_model(N, set_index=1), load_solver_model())
    _stage_dump(models, emul_lib, "cpu", spread=True, min_contacts=8)


@pytest.mark.gpu
def test_ycb_env_runs_clean_gpu(oracle_lib):
    """The batched env on the shipped model: reset recipe (placement fallback for large objects included) + 6 env.steps at B = 256 -- no status
    bit, finite observation rows of the documented width, every object on the table, rows of different envs differ."""
    from robogym_amd.envs.rearrange.ycb import make_env

    env = make_env(batch_size=256, starting_seed=7, stabilize_steps=30, n_random_initial_steps=2, settle_steps=30)      # (the default: with the wrapper stack)
    obs = env.reset()
    assert env.obs_dim == 36 * N + 23 + 2 * 64 and obs["obj_pos"].shape == (256, N, 3)
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    for _ in range(6):
        obs, reward, done, info = env.step(torch.randint(0, 11, (256, 6), generator=gen, device=env.device))
    env.sync()
    assert int(env.sim.status.max()) == 0 and int(env.solver_sim.status.max()) == 0 and obs["action_ema"].shape == (256, 6) and float(obs["action_ema"].abs().max()) <= 1.0
    assert bool(torch.isfinite(env.packed).all())
    z = obs["obj_pos"][:, :, 2]
    assert float(z.min()) > env.table_height - 0.01 and float((z < env.table_height + 0.2).float().mean()) > 0.99
    assert not torch.equal(obs["obj_pos"][0], obs["obj_pos"][1]) and info["object_names"][0] == "055_baseball"


def _grouped_env_checks(lib, device, B, sets, n_substeps, steps):
    """Different object sets across the batch (GroupedYcbRearrangeEnv): one compiled model per group of envs, one API for the whole batch."""
    from robogym_amd.envs.rearrange.ycb import GroupedYcbRearrangeEnv, make_env

    kw = dict(lib=lib, n_substeps=n_substeps, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0) if lib is not None else dict(stabilize_steps=20, n_random_initial_steps=2, settle_steps=20)
    env = make_env(batch_size=B, device=device, starting_seed=4, object_sets=sets, pipelined_reset=True, **kw)
    assert isinstance(env, GroupedYcbRearrangeEnv) and env.K == len(sets) and env.b == B // len(sets)
    assert env.object_names[0][0] == "055_baseball" and env.object_names[1][0] == "044_flat_screwdriver" and env.object_names[0] != env.object_names[1]
    obs = env.reset()
    assert obs["obj_pos"].shape == (B, N, 3) and obs["qpos"].shape == (B, 64)
    which = env.object_set_of_env()                                                          # (a reset is a new episode: the envs were dealt their sets at random)
    assert sorted(np.bincount(which, minlength=len(sets))) == [env.b] * len(sets)
    half0 = obs["obj_bbox_size"][int(np.nonzero(which == 0)[0][0])].cpu().numpy(); half1 = obs["obj_bbox_size"][int(np.nonzero(which == 1)[0][0])].cpu().numpy()
    assert not np.allclose(np.sort(half0.max(1)), np.sort(half1.max(1)))                    # the two groups really hold different objects
    gen = torch.Generator(device=env.device); gen.manual_seed(2)
    for _ in range(steps):
        obs, reward, done, info = env.step(torch.randint(0, 11, (B, 6), generator=gen, device=env.device))
    env.sync()
    assert reward.shape == (B, 3) and done.shape == (B,) and info["resetting"].shape == (B,) and len(info["object_names"]) == B
    assert int(env.status().max()) == 0 and all(bool(torch.isfinite(g.packed).all()) for g in env.groups)
    z = obs["obj_pos"][:, :, 2]
    assert float(z.min()) > env.groups[0].table_height - 0.01
    return env


def test_ycb_object_sets_across_the_batch_emul(emul_lib):
    _grouped_env_checks(emul_lib, "cpu", B=2, sets=(0, 1), n_substeps=1, steps=2)


def _per_episode_checks(lib, device, B, steps, seed, **recipe):
    """A new object set per episode (the reference rebuilds the simulation at every reset, envs/rearrange/ycb.py:58-84, common/base.py:850-856) = the envs of the batch
    trade physics slots when their episodes end.  Checked against a twin with env i pinned to slot i that is fed the same actions in slot order: every output of the
    trading env is the twin's, read through the env -> slot table, bit for bit, through a reset, episode ends (goal time-out: 8 objects x 1 step per object), the reset
    recipe and the first steps of the next episodes; and the table really changed."""
    from robogym_amd.envs.rearrange.ycb import make_env

    kw = dict(pipelined_reset=True, max_timesteps_per_goal_per_obj=1, starting_seed=seed, object_sets=(0, 1), **recipe)
    if lib is not None:
        kw["lib"] = lib
    env, twin = make_env(batch_size=B, device=device, **kw), make_env(batch_size=B, device=device, resample_object_sets=False, **kw)
    assert twin.object_set_of_env().tolist() == [0] * (B // 2) + [1] * (B // 2)

    def same(a, b, slot):
        for k in a:
            if torch.is_tensor(a[k]):
                assert torch.equal(a[k], b[k][slot]), k

    obs, ref = env.reset(), twin.reset()
    tables = [env._slot.copy()]
    same(obs, ref, torch.as_tensor(env._slot, device=env.device))
    gen = torch.Generator(device=env.device); gen.manual_seed(2)
    ended = started = 0
    for k in range(steps):
        a = torch.randint(0, 11, (B, 6), generator=gen, device=env.device)
        slot = torch.as_tensor(env._slot.copy(), device=env.device)             # the table this step runs on (it changes AFTER the terminal step's outputs are taken)
        inv = torch.empty(B, dtype=torch.long, device=env.device); inv[slot] = torch.arange(B, device=env.device)
        out, tw = env.step(a), twin.step(a[inv].contiguous())
        same(out[0], tw[0], slot); same(out[3], tw[3], slot)
        assert torch.equal(out[1], tw[1][slot]) and torch.equal(out[2], tw[2][slot])
        assert out[3]["object_names"] == [tw[3]["object_names"][int(s)] for s in env._slot_of_step]
        ended += int(out[2].sum()); started += int(out[3]["episode_started"].sum())
        tables.append(env._slot.copy())
    env.sync(); twin.sync()
    assert bool(torch.isfinite(out[0]["obj_pos"]).all()) and (lib is None or int(env.status().max()) == 0)
    assert sorted(env._slot.tolist()) == list(range(B)) and twin.episodes_moved == 0
    assert sorted(np.bincount(env.object_set_of_env(), minlength=2)) == [B // 2, B // 2]
    return env, ended, started, tables


def test_slot_trading_pool_is_everything_inside_the_reset_recipe():
    """GroupedYcbRearrangeEnv._trade_slots on stub groups (no physics): on a step with episode ends, the envs holding slots inside the reset recipe -- just ended or
    ended earlier -- get those same slots back in a new order; live slots and their envs are untouched; a step without ends deals nothing."""
    import types
    from robogym_amd.envs.rearrange.ycb import GroupedYcbRearrangeEnv

    env = object.__new__(GroupedYcbRearrangeEnv)
    env.B, env.b, env.K, env.resample, env.device, env.episodes_moved = 8, 4, 2, True, torch.device("cpu"), 0
    env._rng = np.random.RandomState(0)
    env._slot = np.array([3, 0, 6, 1, 7, 2, 5, 4])                       # env -> slot, some earlier deal
    stage = [np.array([0, 2, 0, 1]), np.array([3, 0, 0, 1])]               # slots 1, 3 (group 0) and 4, 7 (group 1) are inside the recipe
    env.groups = [types.SimpleNamespace(_stage=stage[0], ended_rows=np.array([3]), pipelined=True), types.SimpleNamespace(_stage=stage[1], ended_rows=np.array([3]), pipelined=True)]
    before = env._slot.copy()
    env._trade_slots()
    pool_slots = {1, 3, 4, 7}
    holders = [e for e in range(8) if before[e] in pool_slots]
    assert sorted(env._slot[holders]) == sorted(pool_slots) and sorted(env._slot.tolist()) == list(range(8))
    assert all(env._slot[e] == before[e] for e in range(8) if before[e] not in pool_slots)
    assert not np.array_equal(env._slot, before) and torch.equal(env._slot_dev, torch.as_tensor(env._slot))
    assert torch.equal(env._env_of_slot_dev[env._slot_dev], torch.arange(8))
    for g in env.groups:
        g.ended_rows = np.zeros(0, dtype=np.int64)
    again = env._slot.copy()
    env._trade_slots()
    assert np.array_equal(env._slot, again)


def test_ycb_new_object_set_per_episode_emul(emul_lib):
    env, ended, started, tables = _per_episode_checks(emul_lib, "cpu", B=2, steps=10, seed=8, n_substeps=1, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0)
    assert ended == 2 and started == 2                   # the goal times out on step 8, the next episode starts on step 9
    assert len({tuple(t) for t in tables}) >= 2 and env.episodes_moved == 4, tables      # (seed 8: both draws -- at reset, at the episode end -- swap the two slots)


@pytest.mark.gpu
def test_ycb_new_object_set_per_episode_device_trading_gpu():
    B = 64
    env, ended, started, tables = _per_episode_checks(None, "cuda:0", B=B, steps=22, seed=3, stabilize_steps=4, n_random_initial_steps=1, settle_steps=4, device_reset=True)
    assert env._device_trade and ended >= B and started >= B
    assert len({tuple(t) for t in tables}) >= 2 and env.object_set_of_env() is not None and env.episodes_moved > B // 4      # dealt at reset, dealt again on the device when the goals time out


@pytest.mark.gpu
def test_ycb_object_sets_across_the_batch_gpu():
    _grouped_env_checks(None, "cuda:0", B=256, sets=(0, 1, 2, 4), n_substeps=40, steps=5)
