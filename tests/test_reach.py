"""dactyl/reach (BASELINE.json configs[0]): the second model through the same compiler -> oracle -> kernel stack.
CPU: model dimensions of SURVEY 8 and one-mj_step parity on the emulation harness; GPU: the same through the
C ABI on the MI355X plus batch properties."""
import numpy as np
import pytest
import torch

from oracle.env_oracle import OracleReachPhysics
from robogym_amd.envs.dactyl.reach import ReachSimulation, load_reach_model

pytestmark = pytest.mark.usefixtures("kernel_variant")


@pytest.fixture(scope="module")
def reach_model():
    from robogym_amd.mujoco.kernel_tables import derive_kernel_tables

    m = load_reach_model()
    derive_kernel_tables(m)
    return m


def _sync(sim, ora):
    st = ora.get_state_f32()
    ora.set_state_f32(st)
    sim.set_state({k: torch.tensor(v[None], device=sim.device).expand(sim.batch_size, -1).contiguous() for k, v in st.items()})


def _resync_env_step_errors(sim, ora, actions):
    out = []
    for a in actions:
        _sync(sim, ora)
        sim.env_step(action=torch.tensor(np.repeat(a[None].astype(np.float32), sim.batch_size, 0), device=sim.device), nforward_ticks=3)
        ora.env_step(a)
        q = sim.qpos.cpu().numpy()[0].astype(np.float64); v = sim.qvel.cpu().numpy()[0].astype(np.float64)
        out.append((np.abs(q - ora.sim.qpos).max(), np.abs(v - ora.sim.qvel).max()))
    return np.array(out)


def test_reach_model_dimensions(reach_model):
    """SURVEY 8 model table, dactyl/reach row: nq 24, nv 24, nu 20, 29 bodies, 63 geoms, 39 sites, 12 tendons."""
    d = reach_model.dims
    assert [int(d[i]) for i in (0, 1, 2, 3, 5, 6, 7)] == [24, 24, 20, 29, 63, 39, 12]
    assert not any(n.startswith("robot0:hand_base") for n in reach_model.names["joint"])   # the hand is immovable


def test_reach_oracle_zero_control_settles(reach_model, oracle_lib):
    """ReachSimulation.build (reach.py:131-141): 20 steps under the zero control move the fingers towards the
    range centres and leave a finite, slowly moving hand; no finger-finger contact explosion."""
    ora = OracleReachPhysics(reach_model)
    ora.zero_control_settle(20)
    assert np.isfinite(ora.sim.qpos).all() and np.abs(ora.sim.qvel).max() < 5.0
    tips = ora.fingertip_pos()
    assert tips.shape == (5, 3) and (np.linalg.norm(tips - tips.mean(0), axis=1) < 0.15).all()


def test_reach_resync_errors_emul(reach_model, emul_lib, oracle_lib):
    """fp32 kernel source (emulation harness) vs fp64 oracle on the reach model, re-synchronised env.steps
    under random relative actions (self-collisions of the fingers included).  qpos 5e-6, qvel 5e-3."""
    sim = ReachSimulation(reach_model, 1, lib=emul_lib)
    ora = OracleReachPhysics(reach_model)
    ora.zero_control_settle(20)
    rng = np.random.RandomState(8)
    errs = _resync_env_step_errors(sim, ora, rng.uniform(-1, 1, (4, 20)))
    assert errs[:, 0].max() < 5e-6 and errs[:, 1].max() < 5e-3, errs
    assert int(sim.status.max()) == 0


@pytest.mark.gpu
def test_reach_resync_errors_gpu(reach_model, oracle_lib):
    """The same comparison through the C ABI on the MI355X over 60 env.steps; plus the absolute fingertip
    observation (site positions of the five tips) against the oracle after the last step."""
    sim = ReachSimulation(reach_model, 2, device="cuda:0")
    ora = OracleReachPhysics(reach_model)
    ora.zero_control_settle(20)
    rng = np.random.RandomState(9)
    errs = _resync_env_step_errors(sim, ora, rng.uniform(-1, 1, (60, 20)))
    assert np.median(errs[:, 0]) < 1e-6 and errs[:, 0].max() < 1e-4, (np.median(errs[:, 0]), errs[:, 0].max())
    assert np.median(errs[:, 1]) < 2e-4 and errs[:, 1].max() < 5e-2, (np.median(errs[:, 1]), errs[:, 1].max())
    _sync(sim, ora)
    ora.sim.forward()
    sim.env_step(nsubsteps=1, nforward_ticks=0, flags=1)      # stage dump: kinematics of the synchronised configuration
    dump = sim.get_field(8).cpu().numpy()[0]
    site0 = 32 * 3 + 32 * 4
    tips = np.array([dump[site0 + 3 * s: site0 + 3 * s + 3] for s in sim.tip_sites])
    assert np.abs(tips - ora.fingertip_pos()).max() < 5e-6
    assert int(sim.status.max().item()) == 0


@pytest.mark.gpu
def test_reach_batch_rollout_gpu(reach_model):
    """2048 hands, 100 env.steps of random actions: finite, inside the joint ranges (+ margin), no status bits;
    identical envs given identical actions stay bit-identical."""
    B = 2048
    sim = ReachSimulation(reach_model, B, device="cuda:0")
    sim.settle(20)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(4)
    for _ in range(100):
        a = torch.rand((1, 20), generator=gen, device="cuda:0").expand(B, -1).contiguous() * 2 - 1
        sim.env_step(action=a, nforward_ticks=3)
    q = sim.qpos
    assert torch.isfinite(q).all() and (q == q[0]).all()
    rng_ = torch.tensor(reach_model.arrays["jnt_range"], dtype=torch.float32, device="cuda:0")
    assert ((q >= rng_[:, 0] - 0.05) & (q <= rng_[:, 1] + 0.05)).all()
    assert int(sim.status.max().item()) == 0


# ---------------------------------------------------------------------------------------------------------------------------
# the env proper: FingertipPosGoal's second simulation, RobotEnv bookkeeping, the B = 1 1000-step run
def _reach_env_run(env, ora, nsteps, seed, resync, flag_band=0.0):
    """Both envs get the same goal draws and actions; with `resync` the kernel's hand state is re-written from the oracle's
    before every step (the goal simulations are never re-synchronised: they see two steps per goal only)."""
    from tests.test_env_parity import STATE_FIELDS, _put_rows

    rng = np.random.RandomState(seed)
    B = env.batch_size
    normals = rng.randn(4000, 24)
    k = [0]

    def next_normal():
        k[0] += 1
        return normals[k[0] - 1]

    ora.next_normal_fn = next_normal
    n0 = next_normal()
    env.set_draws(torch.as_tensor(np.repeat(n0[None], B, 0)))
    obs = env.reset()
    ora.reset(n0)
    np.testing.assert_allclose(obs["goal_fingertip_pos"][0].cpu().numpy(), ora.goal_tips, atol=2e-4)
    stats = dict(goal_err=[], tip_err=[], rew_err=[], goals=0, timeouts=0, mismatch=0)
    for t in range(nsteps):
        a = rng.uniform(-1, 1, 20)
        if t < nsteps // 2 and rng.rand() < 0.3:   # first half: now and then drive straight at the goal (successes); second half: random actions only (timeouts)
            a = np.clip((ora.goal_joint_pos - ora.sim.qpos[ora.hand_q]) @ ora.P.T / np.maximum(ora.P.sum(1), 1) * 4.0, -1, 1)
        if resync:
            st = ora.get_state_f32(); ora.set_state_f32(st)
            _put_rows(env.sim, np.arange(B), {f: np.repeat(st[f][None], B, 0) for f in STATE_FIELDS})
            env._prev_dist[:] = float(ora.prev_dist)
            env._goal[:] = torch.as_tensor(ora.goal_tips, dtype=torch.float32)
        goals_before = ora.tracker.goals_so_far
        env.set_draws(torch.as_tensor(np.repeat(normals[k[0]][None], B, 0)))      # used only if this step ends in a goal reset
        obs, reward, done, info = env.step(torch.as_tensor(np.repeat(a[None], B, 0), dtype=torch.float32))
        r, d, inf = ora.step(a)
        new_goal = ora.tracker.goals_so_far != goals_before
        stats["goals"] += int(new_goal); stats["timeouts"] += int(d and not inf["trial_success"])
        stats["tip_err"].append(np.abs(obs["fingertip_pos"][0].cpu().numpy() - ora.tips()).max())
        stats["rew_err"].append(abs(float(reward[0, 1]) - r[1]))
        if resync:
            assert bool(done[0]) == bool(d) and float(reward[0, 2]) == r[2], t
            assert int(info["successes_so_far"][0]) == inf["successes_so_far"] and int(info["goals_so_far"][0]) == inf["goals_so_far"], t
            # (the flag of the observation is `distance to the CURRENT goal < threshold` after a possible goal reset; within
            #  `flag_band` of the threshold the two precisions may decide differently -- it feeds no counter)
            if abs(ora.goal_distance() - ora.SUCCESS_THRESHOLD) >= flag_band:
                assert int(obs["is_goal_achieved"][0, 0]) == int(inf["is_goal_achieved"]), t
        if new_goal:
            stats["goal_err"].append(np.abs(obs["goal_fingertip_pos"][0].cpu().numpy() - ora.goal_tips).max())
        if d:
            n = next_normal()
            env.set_draws(torch.as_tensor(np.repeat(n[None], B, 0)))
            env.reset(); ora.reset(n)
    for key in obs:
        assert (obs[key] == obs[key][0]).all()        # identical rows stay identical
    return stats


def test_reach_env_emul(reach_model, emul_lib, oracle_lib):
    from oracle.env_oracle import OracleReachEnv
    from robogym_amd.envs.dactyl.reach import BatchedReachEnv, goal_simulation_model

    env = BatchedReachEnv(1, model=reach_model, lib=emul_lib, mujoco_substeps=3, max_timesteps_per_goal=4)
    ora = OracleReachEnv(reach_model, goal_simulation_model(reach_model), n_substeps=3, max_timesteps_per_goal=4)
    obs = env.observe()
    assert [k for k in obs] == ["qpos", "qvel", "fingertip_pos", "goal_fingertip_pos", "is_goal_achieved"]
    assert obs["qpos"].shape == (1, 24) and obs["fingertip_pos"].shape == (1, 15) and obs["is_goal_achieved"].dtype == torch.int32
    st = _reach_env_run(env, ora, 6, 0, resync=True)
    assert max(st["tip_err"]) < 2e-5 and max(st["rew_err"]) < 5e-5 and st["timeouts"] >= 1


@pytest.mark.gpu
def test_reach_env_1000_steps_gpu(reach_model, oracle_lib, kernel_variant):
    """BASELINE configs[0]: dactyl/reach, 1000 env.steps (B = 2 identical rows so that row handling is in the picture), every
    step re-synchronised from the oracle env: fingertip observation, goal-distance reward, success flag, tracker counters,
    timeouts (150 steps) and every goal the second simulation produces."""
    from oracle.env_oracle import OracleReachEnv
    from robogym_amd.envs.dactyl.reach import BatchedReachEnv, goal_simulation_model

    env = BatchedReachEnv(2, device="cuda:0", model=reach_model)
    ora = OracleReachEnv(reach_model, goal_simulation_model(reach_model))
    st = _reach_env_run(env, ora, 1000, 1, resync=True, flag_band=kernel_variant.tol(0.0, 5e-4))
    print("reach env, 1000 re-synchronised steps: fingertip obs max err %.2e | goal reward max err %.2e | %d goals reached, %d timeouts | goal fingertip (second sim) max err %.2e"
          % (max(st["tip_err"]), max(st["rew_err"]), st["goals"], st["timeouts"], max(st["goal_err"]) if st["goal_err"] else float("nan")))
    t = kernel_variant.tol   # (default: finger-finger mesh contacts through libccd's depth; tails as in profiles/r03_precision.txt)
    assert max(st["tip_err"]) < t(2e-4, 1e-3) and np.median(st["tip_err"]) < 2e-6 and max(st["rew_err"]) < t(5e-4, 1e-3)
    assert st["goals"] >= 3 and st["timeouts"] >= 2 and max(st["goal_err"]) < t(5e-4, 1e-3)
    assert int(env.sim.status.max()) == 0 and int(env.goal_simulation.status.max()) == 0


# ---------------------------------------------------------------------------------------------------------------------------
# FREE-RUNNING parity on a contact-light protocol (north_star: "qpos drift <= 1e-4 vs reference over 1000 steps"; SURVEY section 7 hard part 3;
# VERDICT r03 "missing" 7 / item 6 i): no re-synchronisation — both sides start from the same bytes and run 1000 env.steps on their own.
def _smooth_actions(nsteps, nu, amp, seed):
    """a slowly varying action stream: per actuator a sum of two sinusoids (periods 60 - 400 env.steps), amplitude `amp`"""
    rng = np.random.RandomState(seed)
    t = np.arange(nsteps)[:, None]
    p1, p2 = rng.uniform(60, 200, nu), rng.uniform(200, 400, nu)
    ph1, ph2 = rng.uniform(0, 2 * np.pi, nu), rng.uniform(0, 2 * np.pi, nu)
    return amp * (0.6 * np.sin(2 * np.pi * t / p1 + ph1) + 0.4 * np.sin(2 * np.pi * t / p2 + ph2))


@pytest.mark.gpu
def test_reach_free_running_1000_steps_gpu(reach_model, oracle_lib):
    """dactyl/reach (configs[0], the hand alone), 1000 env.steps = 10 000 mj_steps, NO re-synchronisation: both sides start from the same bytes and
    run on their own under a smooth ABSOLUTE action stream around the range centres (the fingers breathe by a fifth of their ranges; no finger
    rams another, no joint reaches its limit: the contact-light protocol SURVEY section 7 hard-part 3 asks for).  The position-controlled hand is an attractor:
    the run stays at 1e-7 ... 1e-6 and returns there, but a finger-finger contact that opens a substep apart on the two sides leaves a transient of ~1e-3
    whose tail decays over hundreds of steps -- and WHETHER a stream has such an event is decided by rounding.  Round 6: the bound is therefore anchored on
    what fp32 alone does to this protocol -- the oracle's own source built in float, run beside the kernel on the same three action streams (the float build
    itself leaves 1e-4 on 7 / 10 / 12 % of the steps in the portal-plane configuration and on 12 / 29 / 38 % in the default one; rounds 4-5 asserted
    "<= 15 %" from ONE stream on which the kernel happened to show 10 %).  Asserted: the kernel's fraction of steps beyond 1e-4, its largest excursion and its
    run median are within small factors of the float build's, it ends where it started (<= 3e-4), and no status bit.
    (Under relative actions the targets random-walk into finger-finger collisions, and the run leaves 1e-4 at the first impact that the
    two precisions resolve a substep apart: measured 4e-3 -- that is what the re-synchronised protocol is for.)"""
    from oracle import rg_oracle
    from robogym_amd.mujoco.model_blob import pack_model

    stats_k, stats_f = [], []
    for seed in (11, 12, 13):
        sim = ReachSimulation(reach_model, 2, device="cuda:0", relative_action=False)
        ora = OracleReachPhysics(reach_model, relative_action=False)
        ora.zero_control_settle(20)
        _sync(sim, ora)
        twin = OracleReachPhysics(reach_model, relative_action=False)
        twin.sim = rg_oracle.OracleSim(pack_model(reach_model), f32=True)
        for name in ("qpos", "qvel", "ctrl", "pid", "qacc_warmstart"):
            getattr(twin.sim, name)[:] = getattr(ora.sim, name)
        acts = _smooth_actions(1000, 20, 0.2, seed)
        err, e32 = np.zeros(1000), np.zeros(1000)
        for k in range(1000):
            sim.env_step(action=torch.tensor(np.repeat(acts[k][None].astype(np.float32), 2, 0), device=sim.device), nforward_ticks=3)
            a = acts[k].astype(np.float32).astype(np.float64)
            ora.env_step(a); twin.env_step(a)
            err[k] = np.abs(sim.qpos[0].cpu().numpy().astype(np.float64) - ora.sim.qpos).max()
            e32[k] = np.abs(twin.sim.qpos.astype(np.float64) - ora.sim.qpos).max()
        assert int(sim.status.max()) == 0
        stats_k.append((err[-1], np.median(err), np.mean(err > 1e-4), err.max())); stats_f.append((e32[-1], np.median(e32), np.mean(e32 > 1e-4), e32.max()))
        print("reach free-running seed %d: kernel end %.1e median %.1e beyond %.3f max %.1e | float oracle end %.1e median %.1e beyond %.3f max %.1e" % ((seed,) + stats_k[-1] + stats_f[-1]))
    K, F = np.array(stats_k), np.array(stats_f)
    assert K[:, 2].mean() <= 1.5 * F[:, 2].mean() + 0.05, (K[:, 2], F[:, 2])          # steps beyond 1e-4
    assert K[:, 3].max() <= 3.0 * F[:, 3].max(), (K[:, 3], F[:, 3])                   # the largest excursion
    assert np.median(K[:, 1]) <= 3.0 * np.median(F[:, 1]) + 1e-6, (K[:, 1], F[:, 1])  # run medians
    assert K[:, 0].max() <= 3e-4                                                      # and back at the attractor at the end
