"""The drop-in boundary: the reference's `make_env` / `make_simple_env` signature, the B = 1 numpy view with the
reference's keys / shapes / dtypes, a port of the reference's `test_observe` (envs/dactyl/tests/test_locked.py:70-97:
observation == simulation state), `goal_info()`, `action_space`, and the `sim.data` fields in-tree callers read
(site_xpos, body xpos / xquat, actuator_force, ncon + contact geoms / dist) against the oracle."""
import numpy as np
import pytest
import torch

from tests.test_env_parity import STATE_FIELDS, _put_rows

pytestmark = pytest.mark.usefixtures("kernel_variant")

OBS_SHAPES = {"cube_pos": (3,), "cube_quat": (4,), "qpos": (38,), "qvel": (36,), "hand_angle": (24,), "fingertip_pos": (15,), "goal_pos": (3,), "goal_quat": (4,),
              "qpos_goal": (38,), "is_goal_achieved": (1,)}


def _check_single_env_view(env):
    from oracle.env_oracle import relative_fingertips
    from robogym_amd.envs.dactyl.locked import FINGERTIP_SITE_NAMES, REFERENCE_SITE_NAMES, SingleEnvView

    view = SingleEnvView(env)
    obs = view.reset()
    # keys, order, shapes and dtypes of LockedEnv._default_observation_map (locked.py:132-146; Box(-inf, inf, shape, float32) per key)
    assert list(obs.keys()) == list(OBS_SHAPES.keys())
    for k, shape in OBS_SHAPES.items():
        assert obs[k].shape == shape, k
        assert obs[k].dtype == (np.int32 if k == "is_goal_achieved" else np.float32), k
    assert view.action_space["shape"] == (20,) and view.action_space["low"] == -1.0 and view.action_space["high"] == 1.0
    # ---- port of the reference's test_observe: the observation matches the simulation state
    simulation = view.mujoco_simulation
    obs = view.observe()
    qpos = simulation.qpos[0].cpu().numpy(); qpos[simulation.qpos_idxs["target_all_joints"]] = 0.0
    qvel = simulation.qvel[0].cpu().numpy(); qvel[simulation.qvel_idxs["target_all_joints"]] = 0.0
    cube_quat = simulation.get_qpos("cube_rotation")[0].cpu().numpy()
    sx = simulation.data.site_xpos[0].cpu().numpy().astype(np.float64)
    tips = [simulation.model.name2id("site", "robot0:" + n) for n in FINGERTIP_SITE_NAMES]
    refs = [simulation.model.name2id("site", "robot0:" + n) for n in REFERENCE_SITE_NAMES]
    true_obs = {
        "cube_pos": simulation.get_qpos("cube_position")[0].cpu().numpy(),
        "cube_quat": cube_quat * (-1.0 if cube_quat[0] < 0 else 1.0),
        "hand_angle": simulation.get_qpos("hand_angle")[0].cpu().numpy(),
        "fingertip_pos": relative_fingertips(sx[tips], sx[refs]).ravel(),   # MuJoCoObservation.fingertip_positions() (mujoco_shadow_hand.py:18-46)
        "qpos": qpos, "qvel": qvel,
    }
    for key, true_val in true_obs.items():
        assert np.allclose(obs[key], true_val, atol=2e-6), "Value for obs %s %s doesn't match true value %s." % (key, obs[key], true_val)
    # ---- step: reference return types
    o2, reward, done, info = view.step(np.zeros(20))
    assert isinstance(reward, list) and len(reward) == 3 and all(isinstance(r, float) for r in reward) and isinstance(done, bool)
    assert set(["goal_dist", "goal_achieved", "successes_so_far", "steps_since_last_goal", "goals_so_far", "trial_success", "env_crash"]) <= set(info)
    assert isinstance(info["goal_dist"]["cube_quat"], float) and info["goals_so_far"] == 1 and info["steps_since_last_goal"] == 1
    gd_reward, is_successful, ginfo = view.goal_info()
    assert gd_reward == reward[1] and isinstance(is_successful, bool) and abs(ginfo["goal_dist"]["cube_quat"] - info["goal_dist"]["cube_quat"]) < 1e-7
    np.testing.assert_allclose(ginfo["goal"]["cube_quat"], o2["goal_quat"])


def test_single_env_view_and_observe_emul(locked_model, emul_lib):
    from robogym_amd.envs.dactyl.locked import make_simple_env

    env = make_simple_env(constants={"reset_initial_steps": 2, "n_random_initial_steps": 1, "mujoco_substeps": 3}, starting_seed=0, batch_size=1, model=locked_model, lib=emul_lib)
    _check_single_env_view(env)


@pytest.mark.gpu
def test_single_env_view_and_observe_gpu(locked_model):
    from robogym_amd.envs.dactyl.locked import make_simple_env

    _check_single_env_view(make_simple_env(starting_seed=0, batch_size=1, model=locked_model))


def test_make_env_signature_and_constants(locked_model, emul_lib):
    from robogym_amd.envs.dactyl.locked import make_env, make_simple_env

    env = make_simple_env(parameters={"n_random_initial_steps": 3}, constants={"max_timesteps_per_goal": 77, "successes_needed": 5, "randomize": False},
                          starting_seed=4, batch_size=2, model=locked_model, lib=emul_lib)
    assert env.constants.n_random_initial_steps == 3 and env.constants.max_timesteps_per_goal == 77 and env.constants.successes_needed == 5
    with pytest.raises(NotImplementedError):
        make_simple_env(constants={"vision_observations": True}, batch_size=1, model=locked_model, lib=emul_lib)
    wrapped = make_env(batch_size=2, model=locked_model, lib=emul_lib)          # the reference's defaults: apply_wrappers=True, randomize=True
    assert wrapped.action_space["nvec"] == [11] * 20 and wrapped.unwrapped.stop_on_fall and wrapped.randomize
    assert not make_env(constants={"randomize": False}, batch_size=1, model=locked_model, lib=emul_lib).randomize
    assert make_env(constants={"fixed_wrist": True}, batch_size=1, model=locked_model, lib=emul_lib).fixed_wrist
    with pytest.raises(NotImplementedError):
        make_env(wrapper_params={"delete": ["StopOnFallWrapper"]}, batch_size=1, model=locked_model, lib=emul_lib)


def _check_data_fields(sim, ora):
    d = sim.data
    ora.sim.reset(); ora.settle(50)
    st = ora.get_state_f32(); ora.set_state_f32(st)
    _put_rows(sim, np.arange(sim.batch_size), {k: np.repeat(st[k][None], sim.batch_size, 0) for k in STATE_FIELDS})
    rng = np.random.RandomState(3)
    a = rng.uniform(-1, 1, 20).astype(np.float32)
    sim.env_step(action=torch.as_tensor(np.repeat(a[None], sim.batch_size, 0), device=sim.device), nforward_ticks=1)
    ctrl = ora.denormalize(np.clip(a.astype(np.float64), -1, 1), True)
    ora.sim.ctrl[:] = ctrl
    # the oracle's contact list of the LAST mj_step (the state-less forward afterwards recomputes it at the final state)
    for _ in range(ora.n_substeps - 1):
        ora.sim.step()
    ora.sim.fwd_position()
    want_con = sorted((c["geom1"], c["geom2"], round(c["dist"], 5)) for c in ora.sim.contacts())
    ora.sim.step(); ora.sim.forward()
    nb, ns = 31, 36
    np.testing.assert_allclose(d.body_xpos[0].cpu().numpy(), ora.sim.xpos.reshape(nb, 3), atol=2e-6)
    np.testing.assert_allclose(d.body_xquat[0].cpu().numpy(), ora.sim.xquat.reshape(nb, 4), atol=2e-6)
    np.testing.assert_allclose(d.site_xpos[0].cpu().numpy(), ora.sim.site_xpos.reshape(ns, 3), atol=2e-6)
    np.testing.assert_allclose(d.actuator_force[0].cpu().numpy(), ora.sim.actuator_force, atol=2e-3)
    g1, g2, dist = d.contact
    n = int(d.ncon[0])
    got = sorted((int(g1[0, i]), int(g2[0, i]), round(float(dist[0, i]), 5)) for i in range(n))
    assert n == len(want_con) and [g[:2] for g in got] == [w[:2] for w in want_con]
    np.testing.assert_allclose([g[2] for g in got], [w[2] for w in want_con], atol=2e-5)
    assert n >= 1
    np.testing.assert_allclose(d.get_site_xpos("robot0:S_fftip")[0].cpu().numpy(), ora.sim.site_xpos.reshape(ns, 3)[sim.model.name2id("site", "robot0:S_fftip")], atol=2e-6)
    assert torch.equal(d.qpos, sim.view(0)) and d.time.shape == (sim.batch_size,)


def test_data_fields_match_oracle_emul(locked_model, emul_lib, oracle_lib):
    from oracle.env_oracle import OracleLockedEnvPhysics
    from robogym_amd.envs.dactyl.locked import LockedSimulation

    _check_data_fields(LockedSimulation(locked_model, 1, lib=emul_lib, n_substeps=3), OracleLockedEnvPhysics(locked_model, n_substeps=3))


@pytest.mark.gpu
def test_data_fields_match_oracle_gpu(locked_model, oracle_lib):
    from oracle.env_oracle import OracleLockedEnvPhysics
    from robogym_amd.envs.dactyl.locked import LockedSimulation

    _check_data_fields(LockedSimulation(locked_model, 3, device="cuda:0"), OracleLockedEnvPhysics(locked_model))


def test_product_library_loads_and_exports_the_header():
    """The gfx950 build of the C ABI (robogym_amd/csrc/librgstep.so) loads without a GPU and exports every function that
    include/rgstep.h declares (no compute calls here); the binding's list and struct size agree with it; and the LDS
    footprint of the rollout configuration still fits 10 allocation granules of 1280 B (12 envs per CU, which is also what
    the 168-VGPR budget allows: what the measured throughput rests on), the large configuration 16 granules (8 per CU)."""
    import ctypes
    import os
    import re

    from robogym_amd import _native

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "robogym_amd", "csrc", "librgstep.so")
    assert os.path.exists(path), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(path)
    declared = sorted(set(re.findall(r"\b(r[gba]_[a-z_0-9]+)\s*\(", open(os.path.join(root, "include", "rgstep.h")).read())))
    # (rb_*: the large-model path, the full-cube env kernel and the rearrange TCP hook; ra_*: the rearrange env kernel)
    assert len(declared) >= 55 and sum(n.startswith("rb_") for n in declared) == 23 and sum(n.startswith("ra_") for n in declared) == 4
    assert L.ra_post_args_size() == ctypes.sizeof(_native.RaPostArgs) and L.rb_post_args_size() == ctypes.sizeof(_native.RbPostArgs)
    for name in declared:
        assert hasattr(L, name), name
    assert set(_native.EXPORTS) == set(declared)
    assert L.rg_post_args_size() == ctypes.sizeof(_native.PostArgs)
    assert L.rg_lds_bytes_cfg(0) <= 10 * 1280 and 160 * 1024 // (10 * 1280) == 12
    assert L.rg_lds_bytes_cfg(1) <= 16 * 1280 and 160 * 1024 // (16 * 1280) == 8


def _check_touch_sensors(sim, ora, model, nsteps, atol=(0.02, 0.02), pid_atol=1e-2):
    """Fingers closing on the cube: data.sensordata (mj_sensorAcc, touch) of the step's last forward against the oracle's, each
    step from the oracle's state."""
    d = sim.data
    d.sensordata                                   # switches the sensor pass on
    ora.sim.reset(); ora.settle(40)
    a = np.zeros(20, dtype=np.float32)
    for i, n in enumerate(model.names["actuator"]):
        if n.endswith(("FJ2", "FJ1", "THJ1", "THJ0", "THJ4")):
            a[i] = 1.0
    B, seen = sim.batch_size, 0.0
    for t in range(nsteps):
        st = ora.get_state_f32(); ora.set_state_f32(st)
        _put_rows(sim, np.arange(B), {k: np.repeat(st[k][None], B, 0) for k in STATE_FIELDS})
        sim.env_step(action=torch.as_tensor(np.repeat(a[None], B, 0), device=sim.device), nforward_ticks=3)
        ora.env_step(a)
        got, want = d.sensordata[0].cpu().numpy().astype(np.float64), ora.sim.sensordata.copy()
        np.testing.assert_allclose(got, want, atol=atol[0] + atol[1] * np.abs(want).max(), err_msg="step %d" % t)
        seen = max(seen, float(want.max()))
        np.testing.assert_allclose(sim.view(3)[0].cpu().numpy(), ora.sim.pid, atol=pid_atol)     # the full forward is still exactly ONE controller tick (filtered derivatives under contact: 1e-2)
    return seen


def test_touch_sensors_match_oracle_emul(locked_model, emul_lib, oracle_lib):
    from oracle.env_oracle import OracleLockedEnvPhysics
    from robogym_amd.envs.dactyl.locked import LockedSimulation

    _check_touch_sensors(LockedSimulation(locked_model, 1, lib=emul_lib, n_substeps=10), OracleLockedEnvPhysics(locked_model), locked_model, 2)


@pytest.mark.gpu
def test_touch_sensors_match_oracle_gpu(locked_model, oracle_lib, kernel_variant):
    """Stated tolerance on a sensor reading (newtons): plane 0.02 + 2 % of the largest reading; default 0.2 + 5 % (a fingertip
    contact whose libccd normal falls on the other side of a tie break moves a few per cent of the grip force between sensors; PID state 1e-2 / 3e-1: the filtered derivative is d(joint error)/dt, 1e-3 rad over one 8 ms substep is 0.12)."""
    from oracle.env_oracle import OracleLockedEnvPhysics
    from robogym_amd.envs.dactyl.locked import LockedSimulation

    seen = _check_touch_sensors(LockedSimulation(locked_model, 2, device="cuda:0"), OracleLockedEnvPhysics(locked_model), locked_model, 20,
                                atol=kernel_variant.tol((0.02, 0.02), (0.2, 0.05)), pid_atol=kernel_variant.tol(1e-2, 3e-1))
    assert seen > 1.0       # the fingertips did press on the cube (several newtons)


def test_model_blob_is_a_documented_interface(locked_model, emul_lib):
    """SURVEY 8(b) lists `rg_compile_mjcf`; compilation stays host-side here and the RGMODEL1 blob is the interface (include/rgstep.h): its
    directory is enumerable through the ABI, the library reports which arrays each model kind reads, every one of them is in the blob the
    Python compiler packs, and a blob that lacks one is refused with an error naming it."""
    import ctypes

    from robogym_amd.envs.rearrange.xml import load_solver_model
    from robogym_amd.mujoco.big_tables import derive_big_tables
    from robogym_amd.mujoco.model_blob import pack_model

    L = emul_lib
    solver = load_solver_model(); derive_big_tables(solver)
    for model, create, free, keys_fn in ((locked_model, L.rg_model_create, L.rg_model_free, L.rg_model_blob_keys), (solver, L.rb_model_create, L.rb_model_free, L.rb_model_blob_keys)):
        blob = pack_model(model)
        n = L.rg_blob_entry(blob, len(blob), -1, None, None, None)
        assert n == len(model.arrays)
        seen = {}
        for i in range(n):
            name, dt, cnt = ctypes.create_string_buffer(41), ctypes.c_int(), ctypes.c_uint()
            assert L.rg_blob_entry(blob, len(blob), i, name, ctypes.byref(dt), ctypes.byref(cnt)) == n
            seen[name.value.decode()] = (dt.value, cnt.value)
        assert set(seen) == set(model.arrays)
        for k, (dt, cnt) in seen.items():
            a = np.asarray(model.arrays[k])
            assert cnt == a.size and dt == (0 if a.dtype == np.float64 else 2 if a.dtype == np.float32 else 1 if np.issubdtype(a.dtype, np.integer) else 0), k
        err = ctypes.create_string_buffer(512)
        create.restype = ctypes.c_void_p
        h = create(blob, len(blob), err, 512)
        assert h, err.value
        buf = ctypes.create_string_buffer(1 << 14)
        need = keys_fn(ctypes.c_void_p(h), buf, len(buf))
        keys = buf.value.decode().split(",")
        assert 0 < need <= len(buf) and len(keys) > 40 and set(keys) <= set(model.arrays) and "dims" in keys and "qpos0" in keys
        free(ctypes.c_void_p(h))
        victim = keys[len(keys) // 2]
        m2 = model.copy_with(); del m2.arrays[victim]
        blob2 = pack_model(m2)
        assert not create(blob2, len(blob2), err, 512) and victim in err.value.decode()
    assert L.rg_blob_entry(b"NOTABLOB" + bytes(8), 16, -1, None, None, None) < 0
    trunc = pack_model(solver)[:4096]      # the directory points past the end: refused, not read
    assert L.rg_blob_entry(trunc, len(trunc), L.rg_blob_entry(trunc, len(trunc), -1, None, None, None) - 1, None, None, None) < 0
