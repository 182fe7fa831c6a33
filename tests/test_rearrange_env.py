"""`BatchedBlockRearrangeEnv.step` (three launches: TCP solver world, main world, env kernel) against `OracleRearrangeEnv.env_step`
started from the SAME state: observation row (24 keys of envs/rearrange/common/base.py:376-421), rewards, done flags, goal distances, the
tracker's counters and the gripper hand-over to the solver world.  CPU: kernel source on the emulation harness; `-m gpu`: the MI355X."""
import numpy as np
import pytest
import torch

from robogym_amd import _native
from robogym_amd.envs.rearrange.blocks import BatchedBlockRearrangeEnv


def _oracle_from_kernel(env, row, n_substeps):
    """An OracleRearrangeEnv holding env `row`'s state (both worlds), goal and previous success count."""
    from oracle import rearrange_oracle as RO

    model = env.model
    if getattr(env, "per_env_parameters", False):
        # the env's own model: its parameter block as the kernel reads it (the reference's randomizers have written it at reset -- GeomSolimpRandomizer clips dmin / dmax
        # into [0.5, 0.99] even at parameter 0, randomization/sim.py:183-268 -- and so has stabilize_objects' damping change)
        P = env.sim.params
        model = env.model.copy_with(**{("opt_gravity" if k == "gravity" else k): P[k][row].cpu().numpy().astype(np.float64) for k in P.keys() if P[k].shape[1] > 0})
    o = RO.OracleRearrangeEnv(model, None if env.solver_sim is None else env.solver_model, env.N, n_substeps=n_substeps, max_position_change=env.max_position_change,
                              wrist_only=env.wrist_only, ideal_arm=getattr(env, "ideal_arm", False))
    for sim, os_ in [(env.sim, o.main.sim)] + ([] if env.solver_sim is None else [(env.solver_sim, o.solver.sim)]):
        for name, f in (("qpos", sim.qpos), ("qvel", sim.qvel), ("ctrl", sim.ctrl), ("pid", sim.pid), ("qacc_warmstart", sim.qacc_warmstart)):
            getattr(os_, name)[:] = f[row].cpu().numpy().astype(np.float64)
        os_._L.ro_set_time(os_.d, float(sim.time[row]))
        if os_.nmocap:
            mc = sim.mocap[row].cpu().numpy().astype(np.float64); os_.mocap_pos[:] = mc[:3]; os_.mocap_quat[:] = mc[3:]
        if os_.neq:
            os_.eq_data[:] = sim.eq_data[row].cpu().numpy().astype(np.float64)
    o.set_goal(env.goal[row, :, :3].cpu().numpy().astype(np.float64), env.goal_rot[row].cpu().numpy().astype(np.float64))
    return o


# per-key tolerance of ONE re-synchronised env.step (40 + 40 mj_steps in fp32 against the double-precision oracle) whose mj_steps held the same contact and row
# counts on both sides (tests/test_rearrange_kernel.py contact_history): median over (step, env) <= tol, every such step <= SAME_HISTORY_TAIL x tol
STEP_TOL = dict(obj_pos=5e-5, obj_rel_pos=5e-5, obj_vel_pos=3e-3, obj_rot=2e-4, obj_vel_rot=3e-2, robot_joint_pos=5e-6, gripper_pos=1e-5, gripper_velp=2e-3,
                gripper_controls=1e-6, gripper_qpos=5e-5, gripper_vel=2e-3, qpos=2e-5, goal_obj_pos=1e-6, goal_obj_rot=1e-6, rel_goal_obj_pos=2e-5, rel_goal_obj_rot=2e-4,
                obj_gripper_contact=0, tcp_force=3e-2, tcp_torque=3e-3)
SAME_HISTORY_TAIL = 10.0
# an env.step with a contact EVENT resolved a substep apart by the two precisions (differing contact history): bounded loosely
EVENT_TAIL = 300.0


def _check_steps(lib, device, B, n_substeps, nsteps, tol_scale=1.0, make=None, tol=None, min_same_fraction=0.0, down_bias=True, pos_sum_factor=None):
    from tests.test_rearrange_kernel import contact_history

    if make is None:
        env = BatchedBlockRearrangeEnv(B, device=device, lib=lib, n_substeps=n_substeps, stabilize_steps=1 if lib is not None else 20, n_random_initial_steps=0 if lib is not None else 1, settle_steps=0 if lib is not None else 10, starting_seed=3)
    else:
        env = make()
    N = env.N
    obs = env.reset()
    assert set(obs) == {k for k, _ in __import__("robogym_amd.envs.rearrange.blocks", fromlist=["OBS_KEYS"]).OBS_KEYS} and env.obs_dim == 36 * N + 23 + 2 * env.nq
    assert obs["obj_pos"].shape == (B, N, 3) and obs["qpos"].shape == (B, env.nq) and obs["obj_colors"].shape == (B, N, 4)
    # after reset: objects on the table inside the placement area, targets elsewhere, nothing flagged
    z = obs["obj_pos"][..., 2].cpu().numpy()
    if make is None:
        assert np.all(np.abs(z - (env.table_height + 0.0254)) < 2e-3)
    assert z.min() > env.table_height - 0.01 and int(env.sim.status.max()) == 0
    tol = dict(STEP_TOL if tol is None else tol)
    rng = np.random.RandomState(5)
    worst, same, mag = {}, [], {}
    for step in range(nsteps):
        a = rng.uniform(-1, 1, (B, env.action_dim)).astype(np.float32)
        if not env.joint_control and down_bias:
            a[:, 2] = -np.abs(a[:, 2])            # downwards: towards the blocks
        oracles = [_oracle_from_kernel(env, r, n_substeps) for r in range(B)]
        prev_valid = env.prev_valid.cpu().numpy().copy(); prev_ns = env.prev_nsucc.cpu().numpy().copy()
        t0 = env.t.cpu().numpy().copy()
        km = env.sim.stats.cpu().numpy().astype(np.float64)
        kc = None if env.solver_sim is None else env.solver_sim.stats.cpu().numpy().astype(np.float64)
        obs, rew, done, info = env.step(torch.tensor(a, device=env.device))
        env.sync()
        for r in range(B):
            o = oracles[r]
            if prev_valid[r]:
                o.prev_dist = None   # (the oracle recomputes its own previous count below)
                o.main.sim.fwd_position()      # (body frames of the copied state: the count is read from xpos / xquat; no controller tick)
            before = o.num_success(o.goal_distance()) if prev_valid[r] else None
            oobs, orew, ogoal_rew, odone, oinfo = o.env_step(a[r].astype(np.float64))
            if before is not None:
                ogoal_rew = o.num_success(o.goal_distance()) - before
                assert abs(before - prev_ns[r]) < 1e-6
            same.append(contact_history(env.sim, o.main, km[r], row=r) and (env.solver_sim is None or contact_history(env.solver_sim, o.solver, kc[r], row=r)))
            for k, tl in tol.items():
                got = obs[k][r].cpu().numpy().astype(np.float64).reshape(np.asarray(oobs[k]).shape)
                err = float(np.abs(got - oobs[k]).max())
                worst.setdefault(k, []).append(err)
                mag.setdefault(k, []).append(float(np.abs(np.asarray(oobs[k], dtype=np.float64)).max()))
            if abs(float(np.linalg.norm(oobs["tcp_force"])) - 150.0) > 7.5:      # (the flag is a threshold on the force's norm, SAFETY_STOP_FORCE_THRESHOLD: compared away from it)
                assert bool(obs["safety_stop"][r, 0]) == bool(oobs["safety_stop"][0])
            assert abs(float(rew[r, 0]) - orew) < 1e-6 and abs(float(rew[r, 1]) - ogoal_rew) < 1e-6 and bool(done[r]) == bool(odone)
            d = o.goal_distance()
            # (sums over the N objects: N times an object's own tolerance on a same-history step -- 1e-4 / 2e-3 for the five blocks, as before --, the event tail otherwise)
            gd = (max(1e-4, N * tol["obj_pos"] * ((0.4 if N == 5 else 1.0) if pos_sum_factor is None else pos_sum_factor)), max(2e-3, N * tol["obj_rot"])) if same[-1] else (EVENT_TAIL * tol["obj_pos"], EVENT_TAIL * tol["obj_rot"])
            assert abs(float(env.goal_dist[r, 0]) - d["obj_pos"].sum()) < gd[0] and abs(float(env.goal_dist[r, 1]) - d["obj_rot"].sum()) < gd[1], (gd, bool(same[-1]))
            assert int(env.t[r]) == t0[r] + 1
            if not env.joint_control:      # the mocap target the action produced (TCP pose + denormalised action; tcp+wrist: orientation aligned with the vertical)
                ksim, osim = (env.sim, o.main.sim) if env.solver_sim is None else (env.solver_sim, o.solver.sim)      # (tcp_solver_mode mocap: the env's own world)
                mc = ksim.mocap[r].cpu().numpy().astype(np.float64)
                assert np.abs(mc[:3] - osim.mocap_pos.reshape(-1)[:3]).max() < 2e-6 and np.abs(mc[3:] - osim.mocap_quat.reshape(-1)[:4]).max() < 2e-6
            # gripper hand-over to the solver world
            assert env.solver_sim is None or float(env.solver_sim.qpos[r, env.solver_grip_q]) == float(env.sim.qpos[r, env.grip_q]) and float(env.solver_sim.ctrl[r, env.solver_grip_act]) == float(env.sim.ctrl[r, env.grip_act])
        assert int(env.sim.status.max()) == 0 and (env.solver_sim is None or int(env.solver_sim.status.max()) == 0)
    # Re-synchronised env.steps with the gripper pushing objects, classified by contact history: the (step, env) pairs whose 80 mj_steps held
    # the same contact / row counts on both sides carry the stated fp32 tolerance -- median <= tol, each <= SAME_HISTORY_TAIL x tol, no `tol_scale` --; a pair
    # with a differing history is an env.step with a contact event resolved a substep apart and is bounded loosely.
    same = np.array(same)
    print("env.step vs oracle: %d of %d (step, env) pairs with the same contact history; worst same-history error / tolerance per key: %s" % (
        same.sum(), len(same), {k: round(float(np.array(worst[k])[same].max() / max(tl, 1e-12)), 2) for k, tl in tol.items() if same.any()}))
    assert same.mean() >= min_same_fraction, same
    # (Round 6: ONE pair per run may sit beyond the same-history tail, inside the event bound.  Equal SUMS of contact / row counts over the 80 mj_steps are equal
    #  histories up to cancelling differences, and an arm pressed into the table for a dozen steps reaches states where one mj_step's solve is decided at rounding
    #  level: profiles/r06_rb_outliers.txt -- six action streams x 48 pairs on the round-5 build AND on round 6's register Newton step: one such pair in 288 on
    #  either build, on different streams (2.7e-4 / 7.0e-4 in the arm joints), every other pair at 1e-7 ... 2e-6.  Which stream has it is a property of the rounding path.)
    beyond = np.zeros(len(same), dtype=bool)
    for k, tl in tol.items():
        e = np.array(worst[k])
        if same.any():
            assert np.median(e[same]) <= tl, (k, np.median(e[same]), int(same.sum()))
            beyond |= same & (e > max(SAME_HISTORY_TAIL * tl, 1e-6))
        if tl > 0:      # (a 0 / 1 contact flag can differ on a step whose contact history differs: that is what the classification says)
            # (... and so can the wrist's force / torque reading, by the whole wrench of the contact that exists on one side only: bounded by the reading's own size)
            ev = max(EVENT_TAIL * tl * tol_scale, 1e-6) if k not in ("tcp_force", "tcp_torque") else max(EVENT_TAIL * tl * tol_scale, max(mag[k]))
            assert e.max() <= ev, (k, np.median(e), e.max(), ev)
    assert beyond.sum() <= 1, (int(beyond.sum()), {k: float(np.array(worst[k])[beyond].max()) for k in tol})
    return env


def _goal_and_tracker_checks(env):
    """Teleport the blocks onto their goals: every object within both thresholds -> goal reward = +N on the step it happens, success reward,
    `goal_reset` raised, a new goal drawn by `reset_goals`, the tracker's counters as MultiGoalTracker.process leaves them; then a block pushed
    off the table ends the episode with the penalty."""
    B = env.B
    z = torch.zeros(B, env.action_dim, device=env.device)
    ssl0 = int(env.ssl[1])
    assert int(env.prev_valid.min()) == 1      # (the observation that ended the reset / the last step established the success count the next reward is measured from)
    for i, qa in enumerate(env.obj_q):                          # env 0: all blocks at their goals
        env.sim.qpos[0, qa:qa + 7] = env.goal[0, i]
    env.step(z)
    assert float(env.reward[0, 1]) == 5.0 and float(env.reward[0, 2]) == 5.0 and float(env.reward[1, 1]) == 0.0
    assert bool(env.goal_reset[0]) and not bool(env.goal_reset[1]) and int(env.successes[0]) == 1 and int(env.ssl[0]) == 0 and int(env.ssl[1]) == ssl0 + 1
    assert float(env.observe()["is_goal_achieved"][0, 0]) == 1.0
    old = env.goal[0].clone()
    rew_before = env.reward.clone()
    env.reset_goals()
    assert not torch.equal(old[:, :3], env.goal[0, :, :3]) and torch.allclose(env.goal[0, :, 3:], old[:, 3:], atol=1e-6) and int(env.prev_valid[0]) == 1
    o = env.observe()      # the observation carries the NEW goal (re-observed for that env only), the step's reward / counters are untouched
    assert torch.allclose(o["goal_obj_pos"][0], env.goal[0, :, :3]) and float(o["is_goal_achieved"][0, 0]) == 0.0 and torch.equal(env.reward, rew_before) and int(env.successes[0]) == 1
    assert torch.allclose(o["rel_goal_obj_pos"][0], env.goal[0, :, :3] - o["obj_pos"][0], atol=1e-6)
    env.sim.qpos[1, env.obj_q[2]] = 3.0
    env.step(z)
    assert bool(env.done[1]) and bool(env.objects_off_table[1]) and float(env.reward[1, 0]) == -1.0 and not bool(env.done[0])


def test_rearrange_env_step_matches_oracle_emul(emul_lib, oracle_lib):
    """(the emulation harness runs about one mj_step of these worlds per 4 s: one short env.step here, the full 40 + 40 on the GPU)"""
    env = _check_steps(emul_lib, "cpu", B=2, n_substeps=1, nsteps=1)
    _goal_and_tracker_checks(env)


@pytest.mark.gpu
def test_rearrange_env_step_matches_oracle_gpu(oracle_lib):
    """the full 40 + 40 mj_steps per env.step, four envs, 12 steps with the arm pressing down"""
    env = _check_steps(None, "cuda:0", B=4, n_substeps=40, nsteps=12, min_same_fraction=0.4)
    _goal_and_tracker_checks(env)


def _joint_env(lib, device, B, n_substeps, **kw):
    return lambda: BatchedBlockRearrangeEnv(B, device=device, lib=lib, n_substeps=n_substeps, control_mode="joint", **kw)


def test_rearrange_joint_control_env_step_matches_oracle_emul(emul_lib, oracle_lib):
    """control_mode "joint" (robot_interface.py:9-20; JointControlledArm + MujocoRobotiqGripper, no TCP solver world): the action map at the head of the main world's
    launch, env kernel without a solver world -- one short env.step here, the full 40 mj_steps on the GPU"""
    env = _check_steps(emul_lib, "cpu", B=2, n_substeps=1, nsteps=2, make=_joint_env(emul_lib, "cpu", 2, 1, stabilize_steps=1, n_random_initial_steps=1, settle_steps=1))
    assert env.solver_sim is None and env.action_shape == (2, 7)
    _goal_and_tracker_checks(env)


@pytest.mark.gpu
def test_rearrange_joint_control_env_step_matches_oracle_gpu(oracle_lib):
    """the arm's joints commanded directly, +- max_position_change = 0.1 rad per step, gripper opening / closing: 12 steps of four envs after the full reset recipe"""
    env = _check_steps(None, "cuda:0", B=4, n_substeps=40, nsteps=12, make=_joint_env(None, "cuda:0", 4, 40, stabilize_steps=20, n_random_initial_steps=2, settle_steps=10),
                       min_same_fraction=0.4)
    _goal_and_tracker_checks(env)


def test_rearrange_wrist_control_env_step_matches_oracle_emul(emul_lib, oracle_lib):
    """control_mode "tcp+wrist" (robot_interface.py:9-20; FreeWristTcpArm in the solver world): 4 + 1 action numbers, no roll, the commanded orientation aligned with
    the vertical (MocapSolver.align_axis; the oracle's version is pinned by tests/golden/rearrange_tcp_wrist.npz) -- short env.steps here, the full 40 + 40 on the GPU"""
    mk = lambda: BatchedBlockRearrangeEnv(2, device="cpu", lib=emul_lib, n_substeps=1, control_mode="tcp+wrist", stabilize_steps=1, n_random_initial_steps=1, settle_steps=1)
    env = _check_steps(emul_lib, "cpu", B=2, n_substeps=1, nsteps=2, make=mk)
    assert env.wrist_only and env.action_shape == (2, 5) and env.tcp.wrist_only == 1
    _goal_and_tracker_checks(env)


@pytest.mark.gpu
def test_rearrange_wrist_control_env_step_matches_oracle_gpu(oracle_lib):
    mk = lambda: BatchedBlockRearrangeEnv(4, device="cuda:0", n_substeps=40, control_mode="tcp+wrist", stabilize_steps=20, n_random_initial_steps=2, settle_steps=10)
    env = _check_steps(None, "cuda:0", B=4, n_substeps=40, nsteps=12, make=mk, min_same_fraction=0.4)
    _goal_and_tracker_checks(env)
    # through make_env with the wrapper stack: MultiDiscrete [B, 5], action_ema of 5 numbers
    from robogym_amd.envs.rearrange.blocks import make_env
    w = make_env(batch_size=4, parameters={"robot_control_params": {"control_mode": "tcp+wrist"}}, stabilize_steps=5, n_random_initial_steps=1, settle_steps=5)
    w.reset()
    obs = w.step(torch.full((4, 5), 9, dtype=torch.int64, device=w.device))[0]
    w.sync()
    assert obs["action_ema"].shape == (4, 5) and np.allclose(obs["action_ema"].cpu().numpy(), 0.8, atol=1e-6) and int(w.sim.status.max()) == 0


@pytest.mark.parametrize("mode", ["tcp+roll+yaw", "tcp+wrist"])
def test_rearrange_mocap_solver_mode_env_step_matches_oracle_emul(emul_lib, oracle_lib, mode):
    """tcp_solver_mode "mocap" (robot_interface.py:22-29; the branch of build_composite_robot, robot/composite/ur_gripper_arm.py:126-128, the reference's
    test_robot_polymorphism.py parametrises next to mocap_ik): MujocoIdealURGripperCompositeRobot -- the main world's arm hangs on the mocap weld, no joint actuators, ONE
    world and one physics launch per env.step (rb_tcp_args.self_world) -- against OracleRearrangeEnv(ideal_arm=True)."""
    mk = lambda: BatchedBlockRearrangeEnv(2, device="cpu", lib=emul_lib, n_substeps=1, control_mode=mode, tcp_solver_mode="mocap", stabilize_steps=1, n_random_initial_steps=1,
                                          settle_steps=1)
    env = _check_steps(emul_lib, "cpu", B=2, n_substeps=1, nsteps=2, make=mk)
    assert env.ideal_arm and env.solver_sim is None and env.sim.nu == 1 and env.action_shape == (2, 5 if mode == "tcp+wrist" else 6) and env.tcp.self_world == 1
    _goal_and_tracker_checks(env)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tcp+roll+yaw", "tcp+wrist"])
def test_rearrange_mocap_solver_mode_env_step_matches_oracle_gpu(oracle_lib, mode):
    # (the weld drives the arm with whatever force it takes: no pressing into the table here, and 3 cm per step instead of 10)
    mk = lambda: BatchedBlockRearrangeEnv(4, device="cuda:0", n_substeps=40, control_mode=mode, tcp_solver_mode="mocap", stabilize_steps=20, n_random_initial_steps=2, settle_steps=10,
                                          max_position_change=0.03)
    # (objects in contact with a weld-driven gripper: the sum of the five position distances is held to five times one object's tolerance, not to two)
    # (... and the state vector, which holds the pushed objects' poses, to the tolerance of the object positions themselves)
    env = _check_steps(None, "cuda:0", B=4, n_substeps=40, nsteps=10, make=mk, min_same_fraction=0.4, down_bias=False, pos_sum_factor=1.0, tol=dict(STEP_TOL, qpos=5e-5))
    _goal_and_tracker_checks(env)


# ycb: convex parts lying FLAT on the table make the MPR contact POINT ill-defined at the millimetre level (tests/test_rearrange_ycb.py _stage_dump), so velocities and
# wrench readings of one env.step carry more rounding than the blocks' box - box contacts do; positions do not
YCB_STEP_TOL = dict(STEP_TOL, obj_pos=2e-4, obj_rel_pos=2e-4, obj_vel_pos=3e-2, obj_rot=2e-3, obj_vel_rot=0.3, qpos=2e-4, rel_goal_obj_pos=2e-4, rel_goal_obj_rot=2e-3,
                    gripper_velp=5e-3, gripper_vel=5e-3, tcp_force=0.1, tcp_torque=1e-2)


@pytest.mark.gpu
def test_ycb_env_step_observation_row_matches_oracle_gpu(oracle_lib):
    """the N = 8 observation row (24 keys, 439 scalars), rewards, done flags, goal distances and tracker counters of `env.step` on the ycb
    world against `OracleRearrangeEnv.env_step` started from the same state -- the protocol of the blocks test above on the mesh objects."""
    from robogym_amd.envs.rearrange.ycb import BatchedYcbRearrangeEnv

    _check_steps(None, "cuda:0", B=3, n_substeps=40, nsteps=6, tol=YCB_STEP_TOL,
                 make=lambda: BatchedYcbRearrangeEnv(3, n_substeps=40, stabilize_steps=30, n_random_initial_steps=1, settle_steps=10, starting_seed=3))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["joint", "tcp+wrist"])
def test_ycb_env_step_other_control_modes_match_oracle_gpu(oracle_lib, mode):
    """the same protocol on the ycb world in the two other control modes (7 joint numbers, no solver world; xyz + wrist with the vertical alignment)"""
    from robogym_amd.envs.rearrange.ycb import BatchedYcbRearrangeEnv

    _check_steps(None, "cuda:0", B=3, n_substeps=40, nsteps=5, tol=YCB_STEP_TOL,
                 make=lambda: BatchedYcbRearrangeEnv(3, n_substeps=40, stabilize_steps=30, n_random_initial_steps=1, settle_steps=10, starting_seed=3, control_mode=mode))


def test_ycb_env_step_observation_row_matches_oracle_emul(emul_lib, oracle_lib):
    from robogym_amd.envs.rearrange.ycb import BatchedYcbRearrangeEnv

    _check_steps(emul_lib, "cpu", B=1, n_substeps=1, nsteps=1, tol=YCB_STEP_TOL,
                 make=lambda: BatchedYcbRearrangeEnv(1, device="cpu", lib=emul_lib, n_substeps=1, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0, starting_seed=3))


@pytest.mark.gpu
def test_rearrange_env_batch_4096_runs_clean_gpu():
    """BASELINE.json configs[3] at its batch size: reset + 8 random-action steps, no status bits, finite rows, blocks stay on the table."""
    env = BatchedBlockRearrangeEnv(4096, stabilize_steps=20, n_random_initial_steps=2, settle_steps=10)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(8):
        obs, rew, done, info = env.step((torch.rand(4096, 6, generator=g) * 2 - 1).to(env.device))
    env.sync()
    assert int(env.sim.status.max()) == 0 and int(env.solver_sim.status.max()) == 0 and bool(torch.isfinite(env.packed).all())
    assert float(done.float().mean()) < 0.02


def test_make_env_robot_polymorphism_emul(emul_lib):
    """envs/rearrange/tests/test_robot_polymorphism.py on the batched envs: the default robot (tcp+roll+yaw over mocap_ik, max_position_change 0.1, controller error reset,
    6 action numbers), every (control_mode, tcp_solver_mode) pair of its parametrisation -> action width 5 / 6, one world for mocap and two for mocap_ik -- and joint
    control with max_position_change = 2.4 -> 7; the ycb `make_env` hands the modes on as well."""
    from robogym_amd.envs.rearrange import blocks as Bk, ycb as Yc

    kw = dict(batch_size=1, device="cpu", lib=emul_lib, n_substeps=1, apply_wrappers=False)
    env = Bk.make_env(**kw)
    assert env.control_mode == "tcp+roll+yaw" and env.tcp_solver_mode == "mocap_ik" and env.max_position_change == 0.1 and env.tcp.reset_controller_error == 1
    assert env.action_shape == (1, 6) and env.solver_sim is not None
    for mode, dims, solver, one_world in (("tcp+wrist", 5, "mocap", True), ("tcp+wrist", 5, "mocap_ik", False), ("tcp+roll+yaw", 6, "mocap", True), ("tcp+roll+yaw", 6, "mocap_ik", False)):
        env = Bk.make_env(parameters=dict(robot_control_params=dict(control_mode=mode, tcp_solver_mode=solver, max_position_change=0.1)), **kw)
        assert env.action_shape == (1, dims) and (env.solver_sim is None) == one_world and env.ideal_arm == one_world and env.max_position_change == 0.1, (mode, solver)
        assert env.tcp.wrist_only == (1 if mode == "tcp+wrist" else 0) and env.sim.nu == (1 if one_world else 7)
    env = Bk.make_env(parameters=dict(robot_control_params=dict(control_mode="joint", max_position_change=2.4)), **kw)
    assert env.joint_control and env.action_shape == (1, 7) and env.max_position_change == 2.4 and env.solver_sim is None
    env = Yc.make_env(parameters=dict(robot_control_params=dict(control_mode="joint")), **kw)
    assert env.joint_control and env.action_shape == (1, 7) and env.N == 8


def test_grouped_ycb_multi_launch_is_bit_identical_emul(emul_lib):
    """GroupedYcbRearrangeEnv: the groups' physics phases as ONE launch each (rb_multi_begin / rb_multi_launch -> rb_step_multi_kernel: workgroup k steps env k % b of
    batch k / b, each batch with its own model) against one launch chain per group -- the same bytes in every observation row, reward and state vector."""
    from robogym_amd.envs.rearrange.ycb import GroupedYcbRearrangeEnv

    outs = []
    for multi in (True, False):
        env = GroupedYcbRearrangeEnv(2, device="cpu", lib=emul_lib, object_sets=(0, 1), starting_seed=4, n_substeps=1, stabilize_steps=1, n_random_initial_steps=1, settle_steps=1,
                                     resample_object_sets=False, multi_launch=multi)
        assert env.multi_launch == multi
        env.reset()
        g = torch.Generator().manual_seed(1)
        for _ in range(2):
            obs, rew, done, info = env.step(torch.rand((2, 6), generator=g) * 2 - 1)
        outs.append((torch.cat([gr.packed for gr in env.groups]).clone(), rew.clone(), torch.cat([gr.sim.qpos for gr in env.groups]).clone(),
                     torch.cat([gr.solver_sim.qpos for gr in env.groups]).clone(), torch.cat([gr.sim.ctrl for gr in env.groups]).clone()))
        assert int(max(gr.sim.status.max() for gr in env.groups)) == 0
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


def test_single_env_view_has_the_reference_types_emul(emul_lib):
    """`SingleEnvView(make_simple_env(batch_size=1))`: numpy observations without the batch dimension, reward list of three floats, bool done, scalar info values"""
    from robogym_amd.envs.rearrange.blocks import SingleEnvView, make_simple_env

    env = SingleEnvView(make_simple_env(batch_size=1, device="cpu", lib=emul_lib, n_substeps=1, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0))
    obs = env.reset()
    assert isinstance(obs["obj_pos"], np.ndarray) and obs["obj_pos"].shape == (5, 3) and obs["obj_pos"].dtype == np.float32 and obs["qpos"].shape == (env.env.nq,)
    obs, reward, done, info = env.step(np.zeros(6))
    assert isinstance(reward, list) and len(reward) == 3 and all(isinstance(x, float) for x in reward) and isinstance(done, bool) and not done
    assert isinstance(info["goal_dist_obj_pos"], float) and isinstance(info["successes_so_far"], int) and isinstance(info["goal_reset"], bool)
    assert np.allclose(obs["rel_goal_obj_pos"], obs["goal_obj_pos"] - obs["obj_pos"], atol=1e-6)


# ------------------------------------------------------------------------------------------------ the rearrange wrapper stack (RearrangeEnv.apply_wrappers)
def _wrapper_stack_replay(lib, device, n_substeps):
    """`make_env()`'s default stack -- DiscretizeActionWrapper(11 bins) -> ClipRewardWrapper -> SmoothActionWrapper(0.3) -- runs inside the launches
    (rb_tcp_args.action_index / bins / ema_*, ra_post_args.reward_clip).  Replay of tests/golden/rearrange_wrappers.npz, recorded from the reference's own
    classes (tools/gen_golden_rearrange_wrappers.py): the action that reaches the env and the `action_ema` observation step by step, a reset in between."""
    import os

    from robogym_amd.envs.rearrange.blocks import make_env

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rearrange_wrappers.npz"))
    # (emulation harness: ONE mj_step per env.step; smooth_alpha is chosen so that the step-adjusted alpha is the reference's 0.3 ^ (0.04 / 0.08) all the same)
    kw = dict(lib=lib, n_substeps=n_substeps, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0, smooth_alpha=0.3 ** (40 / n_substeps)) if lib is not None else dict(stabilize_steps=5, n_random_initial_steps=1, settle_steps=2)
    env = make_env(batch_size=2, device=device, starting_seed=3, penalty=dict(table_collision=0.0, objects_off_table=250.0, wrist_collision=0.0), **kw)
    assert env.wrapped and env.n_action_bins == 11 and abs(env.tcp_wrapped.ema_alpha - 0.3 ** 0.5) < 1e-6        # alpha ^ (0.001 x 40 / 0.08)
    obs = env.reset()
    assert np.array_equal(obs["action_ema"].cpu().numpy(), np.zeros((2, 6))) and np.array_equal(g["action_ema_at_reset"], np.zeros(6))
    T = len(g["idx"]) if lib is None else 8
    with pytest.raises(AssertionError):
        env.step(torch.zeros((2, 6), dtype=torch.float32, device=env.device))                                      # MultiDiscrete actions only
    for t in range(T):
        if t == int(g["reset_at"][0]):
            env.reset()
        idx = torch.tensor(np.stack([g["idx"][t], g["idx"][(t + 3) % len(g["idx"])]]), device=env.device)       # row 1: another stream (rows are independent)
        obs, rew, done, info = env.step(idx)
        env.sync()
        got = obs["action_ema"][0].cpu().numpy().astype(np.float64)
        assert np.abs(got - g["action_to_env"][t]).max() < 2e-6 and np.abs(got - g["action_ema"][t]).max() < 2e-6, (t, got, g["action_to_env"][t])
        assert float(rew.abs().max()) <= 100.0
    # ClipRewardWrapper: an object pushed off the table costs 250 in this configuration and arrives as -100
    env.sim.qpos[0, env.obj_q[0]:env.obj_q[0] + 3] = torch.tensor([3.0, 3.0, 0.5], device=env.device)
    obs, rew, done, info = env.step(torch.full((2, 6), 5, dtype=torch.int64, device=env.device))
    env.sync()
    assert bool(info["objects_off_table"][0]) and float(rew[0, 0]) == -100.0 and bool(done[0]) and not bool(info["objects_off_table"][1])
    return env


def test_rearrange_wrapper_stack_matches_reference_classes_emul(emul_lib):
    _wrapper_stack_replay(emul_lib, "cpu", n_substeps=1)


@pytest.mark.gpu
def test_rearrange_wrapper_stack_matches_reference_classes_gpu():
    _wrapper_stack_replay(None, "cuda:0", n_substeps=40)


def test_exponential_action_bins_reach_the_launch_emul(emul_lib):
    """`constants.action_spacing = "exponential"` (BinSpacing.EXPONENTIAL, wrappers/util.py:17-33): the env's device table is the reference's exponential array and
    the launch maps bin indices through it -- after the first step of an episode the smoothing wrapper's output is the mapped action itself."""
    from robogym_amd.envs.rearrange.blocks import make_env

    env = make_env(batch_size=1, device="cpu", lib=emul_lib, constants={"action_spacing": "exponential"}, n_substeps=1, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0)
    table = [-1.0, -0.5, -0.25, -0.125, -0.0625, 0.0, 0.0625, 0.125, 0.25, 0.5, 1.0]
    assert env.bins.shape == (6, 11) and all(env.bins[d].tolist() == table for d in range(6))
    env.reset()
    idx = torch.tensor([[3, 9, 5, 0, 10, 6]])
    obs = env.step(idx)[0]
    assert np.allclose(obs["action_ema"][0].numpy(), [table[i] for i in idx[0].tolist()], atol=1e-7)


# ------------------------------------------------------------------------------------------------ pipelined resets
def _pipelined_reset_sequence(lib, device, n_substeps, B, **extra):
    """Episodes that end (goal time-out after 5 steps here) restart INSIDE the following step calls: recipe stages of 2 + 1 + 2 steps (stabilise, one random
    action, settle), during which the env reports `resetting`, zero reward and done = False, then `episode_started` with fresh counters, objects placed on the
    table, a new goal and an observation row that already carries it."""
    kw = dict(lib=lib) if lib is not None else {}
    env = BatchedBlockRearrangeEnv(B, device=device, n_substeps=n_substeps, stabilize_steps=2, n_random_initial_steps=1, settle_steps=2, max_timesteps_per_goal_per_obj=1,
                                   pipelined_reset=True, starting_seed=11, **kw, **extra)
    env.reset()
    goal0 = env.goal.clone()
    act = lambda: torch.zeros((B, env.action_dim), dtype=torch.float32, device=env.device)
    seen = []
    for k in range(1, 12):
        obs, rew, done, info = env.step(act())
        env.sync()
        seen.append((bool(done.all()), bool(info["resetting"].all()), bool(info["episode_started"].all()), float(rew.abs().max())))
        assert int(env.sim.status.max()) == 0 and (env.solver_sim is None or int(env.solver_sim.status.max()) == 0) and bool(torch.isfinite(env.packed).all())
        if k == 5:
            assert bool(done.all()) and bool(info["resetting"].all())                      # the time-out step: terminal observation, recipe scheduled
            z = obs["obj_pos"][..., 2]                                                     # (the observation is still the old episode's)
        if 6 <= k <= 9:
            assert not bool(done.any()) and bool(info["resetting"].all()) and float(rew.abs().max()) == 0.0 and int(env.steps.max()) == 5
        if k == 10:
            assert bool(info["episode_started"].all()) and not bool(info["resetting"].any()) and not bool(done.any())
            assert int(env.steps.max()) == 0 and int(env.t.max()) == 0 and int(env.ssl.max()) == 0
            assert not torch.equal(env.goal, goal0)                                          # a new goal ...
            assert torch.allclose(obs["goal_obj_pos"], env.goal[:, :, :3]) and float(rew.abs().max()) == 0.0   # ... which the returned observation already carries
            zz = obs["obj_pos"][..., 2].cpu().numpy()
            assert np.all(np.abs(zz - (env.table_height + 0.0254)) < 5e-3)               # blocks on the table where the new episode placed them
        if k == 11:
            assert int(env.steps.max()) == 1 and not bool(info["episode_started"].any())   # the new episode is live
    assert [s[0] for s in seen] == [False] * 4 + [True] + [False] * 6
    return env


def test_rearrange_pipelined_reset_sequence_emul(emul_lib):
    _pipelined_reset_sequence(emul_lib, "cpu", n_substeps=1, B=2)


def _device_recipe_checks(env):
    """After `_pipelined_reset_sequence(device_reset=True)`: nothing of the recipe lives on the host, no placement ran out of trials, the per-env parameter rows are back
    at the model's values (stabilize_objects' damping restored, zero-parameter randomizers)."""
    assert env.device_reset and int(env.placement_failed.max()) == 0 and int(env.stage.max()) == 0
    if env.per_env_parameters:
        d = env.sim.params["dof_damping"]
        assert torch.equal(d[:, env.obj_dofs], env._param_defaults["dof_damping"][env.obj_dofs][None, :].expand(env.B, -1))


def test_rearrange_device_reset_sequence_emul(emul_lib):
    """the same protocol with the recipe's stage machine, begin-of-episode state and placement / goal sampling in ra_recipe_kernel (no flag readback)"""
    _device_recipe_checks(_pipelined_reset_sequence(emul_lib, "cpu", n_substeps=1, B=2, device_reset=True))
    _device_recipe_checks(_pipelined_reset_sequence(emul_lib, "cpu", n_substeps=1, B=2, device_reset=True, control_mode="joint"))
    _device_recipe_checks(_pipelined_reset_sequence(emul_lib, "cpu", n_substeps=1, B=2, device_reset=True, control_mode="tcp+wrist"))


def test_mocap_solver_mode_with_the_wrapper_stack_emul(emul_lib):
    """`make_env` with tcp_solver_mode mocap and the default wrapper stack: MultiDiscrete actions through the hook on the env's own world (bin lookup and smoothing inside
    the launch), the smoothed action as observed, the mocap target moving by action x max_position_change from the TCP's pose."""
    from robogym_amd.envs.rearrange.blocks import make_env

    env = make_env(batch_size=2, device="cpu", lib=emul_lib, parameters={"robot_control_params": {"tcp_solver_mode": "mocap", "max_position_change": 0.05}}, n_substeps=1,
                   stabilize_steps=1, n_random_initial_steps=0, settle_steps=0, smooth_alpha=0.3 ** 40)
    assert env.ideal_arm and env.wrapped and env.solver_sim is None
    obs = env.reset()
    assert float(obs["action_ema"].abs().max()) == 0.0
    tcp0 = obs["gripper_pos"].clone()
    idx = torch.tensor([[10, 5, 5, 5, 5, 5], [5, 0, 5, 5, 5, 5]])
    obs = env.step(idx)[0]
    a = obs["action_ema"].numpy()
    assert np.allclose(a[0], [1, 0, 0, 0, 0, 0], atol=1e-6) and np.allclose(a[1], [0, -1, 0, 0, 0, 0], atol=1e-6)      # (first step of the filter: the action itself)
    mc = env.sim.mocap.numpy()
    assert np.allclose(mc[0, :3] - tcp0[0].numpy(), [0.05, 0, 0], atol=2e-5) and np.allclose(mc[1, :3] - tcp0[1].numpy(), [0, -0.05, 0], atol=2e-5)
    assert int(env.sim.status.max()) == 0


def _device_placement_statistics(lib, device, B, rounds, ycb=False):
    """ra_recipe_kernel's `place_objects_in_grid` (common/utils.py:719-829) on its own: every env is told its episode ended, `rounds` times.  Each placement: the
    blocks' yawed bounding boxes inside the placement area and pairwise disjoint (distinct grid cells), resting on the table, the start pose everywhere else, a fresh
    goal drawn the same way when the episode starts; over all draws the cell of object 0 is uniform over the grid (chi-square against the uniform law)."""
    kw = dict(lib=lib) if lib is not None else {}
    if ycb:      # 8 mesh objects: the grid has fewer cells than objects -> the kernel's rejection sampling (place_objects_with_no_constraint)
        from robogym_amd.envs.rearrange.ycb import BatchedYcbRearrangeEnv as cls
    else:
        cls = BatchedBlockRearrangeEnv
    env = cls(B, device=device, n_substeps=1, stabilize_steps=0, n_random_initial_steps=0, settle_steps=0, pipelined_reset=True, device_reset=True, starting_seed=5, **kw)
    (off_x, off_y, _), (width, height, _) = env.placement_area()
    lo = np.array([off_x, off_y]) - env.table_size[:2] + env.table_pos[:2]
    half0 = env.obj_half[:, :2]
    cells = []
    qpos0 = torch.tensor(env.model.arrays["qpos0"].astype(np.float32), device=env.device)
    for _ in range(rounds):
        env.stage.zero_(); env.done.fill_(True); env.goal_reset.fill_(False)
        env._advance_recipes_device()
        env.sync()
        assert bool(env.ended.all()) and bool(env.resetting.all()) and int(env.stage.min()) == 1 and int(env.placement_failed.max()) == 0
        q = env.sim.qpos.cpu().numpy().astype(np.float64)
        yaw = env.yaw.cpu().numpy().astype(np.float64)
        assert yaw.min() >= 0 and yaw.max() <= 2 * np.pi + 1e-5
        half = env._aabb_half(yaw)[..., :2]
        cen = np.stack([np.cos(yaw) * env.obj_center[:, 0] - np.sin(yaw) * env.obj_center[:, 1], np.sin(yaw) * env.obj_center[:, 0] + np.cos(yaw) * env.obj_center[:, 1]], -1)
        xy = np.stack([q[:, qa:qa + 2] for qa in env.obj_q], 1) + cen                    # centre of the yawed bounding box (blocks: the body origin itself)
        quat = np.stack([q[:, qa + 3:qa + 7] for qa in env.obj_q], 1)
        assert np.abs(quat[..., 0] - np.cos(yaw / 2)).max() < 1e-6 and np.abs(quat[..., 3] - np.sin(yaw / 2)).max() < 1e-6 and np.abs(quat[..., 1:3]).max() == 0
        assert np.all(xy - half >= lo - 1e-5) and np.all(xy + half <= lo + [width, height] + 1e-5)
        z = np.stack([q[:, qa + 2] for qa in env.obj_q], 1)
        assert np.abs(z + env.obj_center[:, 2] - env.obj_half[:, 2] - env.table_height).max() < 1e-5       # the bounding box rests on the table top
        for i in range(env.N):
            for j in range(i + 1, env.N):
                apart = np.abs(xy[:, i] - xy[:, j]) >= half[:, i] + half[:, j] - 1e-6
                assert apart.any(-1).all()
        assert np.abs(q[:, env.arm_q] - np.asarray(__import__("robogym_amd.envs.rearrange.blocks", fromlist=["x"]).TABLETOP_EXPERIMENT_INITIAL_POS)).max() < 1e-6
        assert float(env.sim.qvel.abs().max()) == 0 and float(env.sim.time.abs().max()) == 0
        so = env.static_obs.cpu().numpy()
        assert np.abs(so[..., :2] - half).max() < 1e-6 and so[..., 3:6].min() >= 0 and so[..., 3:6].max() < 1 and np.all(so[..., 6] == 1)
        ncol, nrow = (width // (2 * half[..., 0].max(1))).astype(int), (height // (2 * half[..., 1].max(1))).astype(int)
        if ycb:
            assert (ncol * nrow < env.N).mean() > 0.5        # (most yaw draws leave fewer cells than objects)
            ncol, nrow = np.maximum(ncol, 1), np.maximum(nrow, 1)
        col = np.floor((xy[:, 0, 0] - half[:, 0, 0] - lo[0]) / (width / ncol) + 0.5).astype(int); row = np.floor((xy[:, 0, 1] - half[:, 0, 1] - lo[1]) / (height / nrow) + 0.5).astype(int)
        assert ycb or (col.min() >= 0 and (col < ncol).all() and row.min() >= 0 and (row < nrow).all())
        cells.append(np.stack([col / ncol, row / nrow], -1))                             # (grids differ with the yaw draw: compare the cell's relative position)
        # the episode starts on the next call (all three stage lengths are zero): first goal = another placement with the same yaw
        env.done.fill_(False)
        env._advance_recipes_device()
        env.sync()
        assert bool(env.episode_started.all()) and int(env.stage.max()) == 0 and int(env.prev_valid.max()) == 1     # (re-observed: the first step's reward has its reference)
        g = env.goal.cpu().numpy().astype(np.float64)
        assert np.all(g[..., :2] + cen - half >= lo - 1e-5) and np.all(g[..., :2] + cen + half <= lo + [width, height] + 1e-5) and np.abs(g[..., :2] + cen - xy).max() > 1e-3
        ez = np.mod(yaw + np.pi, 2 * np.pi) - np.pi
        assert np.abs(env.goal_rot[..., 2].cpu().numpy() - ez).max() < 1e-5 and np.abs(np.abs(g[..., 3]) - np.abs(np.cos(ez / 2))).max() < 1e-5
        qg = env.qpos_goal.cpu().numpy()
        assert np.abs(qg[:, env.obj_q[1]:env.obj_q[1] + 7] - g[:, 1]).max() < 1e-6 and np.abs(qg[:, env.arm_q] - q[:, env.arm_q]).max() < 1e-6
    c = np.concatenate(cells)
    # uniformity of object 0's cell: quadrants of the placement area equally likely (4 bins, chi-square with 3 dof: 16.3 = p 0.001)
    quad = (c[:, 0] >= 0.5).astype(int) * 2 + (c[:, 1] >= 0.5).astype(int)
    counts = np.bincount(quad, minlength=4)
    return env, counts


def test_device_placement_is_valid_emul(emul_lib):
    env, counts = _device_placement_statistics(emul_lib, "cpu", B=64, rounds=2)
    assert counts.sum() == 128 and counts.min() > 0
    _device_placement_statistics(emul_lib, "cpu", B=32, rounds=1, ycb=True)


@pytest.mark.gpu
def test_device_placement_is_valid_and_uniform_gpu():
    env, counts = _device_placement_statistics(None, "cuda:0", B=4096, rounds=4)
    n = counts.sum()
    # (grids with an odd number of columns / rows put the middle column on the upper side of the split: compare with the expectation of the observed grids instead
    # of 1/4 each -- loosely: every quadrant between 15 % and 35 %)
    assert n == 4 * 4096 and counts.min() > 0.15 * n and counts.max() < 0.35 * n, counts
    _device_placement_statistics(None, "cuda:0", B=4096, rounds=2, ycb=True)


@pytest.mark.gpu
def test_rearrange_pipelined_reset_sequence_gpu():
    _pipelined_reset_sequence(None, "cuda:0", n_substeps=40, B=64)


@pytest.mark.gpu
def test_rearrange_device_reset_sequence_gpu():
    _device_recipe_checks(_pipelined_reset_sequence(None, "cuda:0", n_substeps=40, B=64, device_reset=True))
    _device_recipe_checks(_pipelined_reset_sequence(None, "cuda:0", n_substeps=40, B=64, device_reset=True, control_mode="joint"))


def _joint_control_wrapped_pipelined(lib, device, n_substeps, B):
    """control_mode "joint" through `make_env` with the wrapper stack and pipelined resets: MultiDiscrete [B, 7] actions, the smoothing filter's output (the reference's
    IncrementalExpAvg, wrappers/util.py:142-160, recomputed here in float64) is the action that reaches the robot -- checked on the control row the main world's launch
    wrote: ctrl[:6] = clip(q + a * min(range / 2, 0.1)), the gripper around its previous control --, and an episode that times out restarts inside the step calls with
    the controls held while the objects stabilise."""
    from robogym_amd.envs.rearrange.blocks import make_env

    kw = dict(lib=lib) if lib is not None else {}
    env = make_env(batch_size=B, device=device, starting_seed=4, parameters={"robot_control_params": {"control_mode": "joint", "max_position_change": 0.1}},
                   constants={"max_timesteps_per_goal_per_obj": 1}, n_substeps=n_substeps, stabilize_steps=2, n_random_initial_steps=1, settle_steps=2, pipelined_reset=True,
                   smooth_alpha=0.3 ** (40 / n_substeps), **kw)
    assert env.joint_control and env.wrapped and env.solver_sim is None and env.bins.shape == (7, 11)
    obs = env.reset()
    assert obs["action_ema"].shape == (B, 7) and float(obs["action_ema"].abs().max()) == 0.0
    A = env.model.arrays
    lo, hi = A["actuator_ctrlrange"][:, 0].astype(np.float64), A["actuator_ctrlrange"][:, 1].astype(np.float64)
    rng = np.random.RandomState(2)
    al, value, t = 0.3 ** 0.5, np.zeros((B, 7)), 0
    for k in range(1, 12):
        idx = rng.randint(0, 11, (B, 7))
        q = env.sim.qpos[:, env.arm_q[0]:env.arm_q[0] + 6].cpu().numpy().astype(np.float64)
        g0 = env.sim.ctrl[:, env.grip_act].cpu().numpy().astype(np.float64)
        c0 = env.sim.ctrl.cpu().numpy().copy()
        live = not bool(env.resetting.any())
        obs, rew, done, info = env.step(torch.tensor(idx, device=env.device))
        env.sync()
        assert int(env.sim.status.max()) == 0 and bool(torch.isfinite(env.packed).all())
        ctrl = env.sim.ctrl.cpu().numpy().astype(np.float64)
        if live:
            value = value * al + (1 - al) * (idx / 5.0 - 1.0); t += 1
            a = value / (1 - al ** t)
            assert np.abs(obs["action_ema"].cpu().numpy() - a).max() < 2e-6
        if live and not bool(done.any()):      # (an episode that ended on this step already carries the next episode's start state)
            assert np.abs(ctrl[:, :6] - np.clip(q + a[:, :6] * np.minimum(0.5 * (hi - lo)[:6], 0.1), lo[:6], hi[:6])).max() < 2e-6
            assert np.abs(ctrl[:, 6] - np.clip(g0 + a[:, 6] * 0.5 * (hi - lo)[6], lo[6], hi[6])).max() < 2e-6
        assert bool(done.all()) == (k == 5)     # the goal's time-out: 5 objects x 1 step; the recipe's 2 + 1 + 2 steps follow
        if k in (6, 7):     # stabilise: the stored controls stay
            assert bool(info["resetting"].all()) and np.array_equal(ctrl, c0)
        if k == 8:          # the recipe's one random action moved the arm's targets
            assert bool(info["resetting"].all()) and not np.array_equal(ctrl[:, :6], c0[:, :6])
        if k == 10:
            assert bool(info["episode_started"].all()) and float(obs["action_ema"].abs().max()) == 0.0
            value, t = np.zeros((B, 7)), 0
    assert int(env.steps.max()) == 1
    return env


def test_rearrange_joint_control_wrapped_pipelined_emul(emul_lib):
    _joint_control_wrapped_pipelined(emul_lib, "cpu", n_substeps=1, B=2)


@pytest.mark.gpu
def test_rearrange_joint_control_wrapped_pipelined_gpu():
    _joint_control_wrapped_pipelined(None, "cuda:0", n_substeps=40, B=16)


# ------------------------------------------------------------------------------------------------ the reference's impulse-response pin on the batched env itself
def _impulse_response_on_the_kernel(lib, device, stabilize_steps, cases=((True, 0.165, 0.036, 5, 1e-3), (False, 0.05, 0.0363, 12, 1.5e-3))):
    """envs/rearrange/tests/test_rearrange_sim.py:135-230 stepped on the PRODUCT (BatchedBlockRearrangeEnv, no oracle in the loop): three envs per case, env i gets the
    impulse on TCP axis i after two zero steps, then 40 zero steps, the actions passed through the smoothing wrapper's filter as the reference's
    `make_env(...).env` does (tests/test_rearrange_oracle.py::_impulse_trajectory is the same protocol on the oracle).  Asserted: 90 % of the steady-state
    displacement within `rise` steps, on every axis, and the displacement itself within the reference's 1e-3 of the expected value for the default (controller-error
    reset on: 0.03616 / 0.03644 / 0.03656 for 0.036 on the emulation harness, the oracle gives 0.03615 / 0.03644 / 0.03651).  Without the reset the harness gives
    0.0370 / 0.0372 / 0.0370 for 0.0363 (oracle 0.0365 / 0.0367 / 0.0365): inside 1e-3 with little to spare -- the hook's stated deviation (DESIGN.md section 4 (i):
    it reads the TCP pose from fresh kinematics, the reference one substep stale in this mode) is worth 5e-4 here -- so that case is asserted at 1.5e-3."""
    kw = dict(lib=lib) if lib is not None else {}
    alpha = 0.3 ** (0.001 * 40 / 0.08)
    out = []
    for rce, mpc, expected, rise, tol in cases:
        env = BatchedBlockRearrangeEnv(3, device=device, max_position_change=mpc, arm_reset_controller_error=rce, n_random_initial_steps=0, stabilize_steps=stabilize_steps,
                                       settle_steps=0, starting_seed=0, **kw)
        env.reset()
        P, ema = [], np.zeros((3, 6))
        for k in range(43):
            imp = np.zeros((3, 6))
            if k == 2:
                imp[np.arange(3), np.arange(3)] = 1.0
            ema = ema * alpha + (1 - alpha) * imp
            obs = env.step(torch.tensor((ema / (1 - alpha ** (k + 1))).astype(np.float32), device=env.device))[0]
            P.append(obs["gripper_pos"].cpu().numpy().astype(np.float64).copy())
        env.sync()
        assert int(env.sim.status.max()) == 0 and int(env.solver_sim.status.max()) == 0
        P = np.array(P) - P[0]
        total = P[-1][np.arange(3), np.arange(3)]
        out.append(total)
        assert np.abs(total - expected).max() < tol, (rce, mpc, total)
        assert (np.abs(P[2 + rise][np.arange(3), np.arange(3)]) > 0.9 * total).all(), (rce, mpc, P[2 + rise], total)
    return out


# ------------------------------------------------------------------------------------------------ the goal layer against the reference's own code
def _goal_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "rearrange_goal.npz"))


def test_oracle_goal_distance_matches_reference_code():
    """oracle/rearrange_oracle.py's relative_goal / goal_distance against tests/golden/rearrange_goal.npz = `ObjectStateGoal.relative_goal / goal_distance`
    (goals/object_state.py:492-599), whose source tools/gen_golden_rearrange_goal.py executes as it stands: 64 random object / goal states."""
    from oracle import rearrange_oracle as RO

    g = _goal_golden()
    for t in range(len(g["cur_pos"])):
        rel_pos = g["goal_pos"][t] - g["cur_pos"][t]
        rel_rot = RO.normalize_angles(RO.subtract_euler(g["goal_rot"][t], g["cur_rot"][t]))
        assert np.abs(rel_pos - g["rel_pos"][t]).max() < 1e-12 and np.abs(rel_rot - g["rel_rot"][t]).max() < 1e-9
        d_rot = RO.quat_magnitude(RO.quat_normalize(RO.euler2quat(rel_rot)))
        assert np.abs(np.linalg.norm(rel_pos, axis=-1) - g["dist_pos"][t]).max() < 1e-12 and np.abs(d_rot - g["dist_rot"][t]).max() < 1e-9
    # _calculate_num_success / _calculate_goal_distance_reward (common/base.py:824-848): the count of objects inside both thresholds, and its change
    env = RO.OracleRearrangeEnv.__new__(RO.OracleRearrangeEnv)
    env.success_threshold, env.goal_reward_per_object = {"obj_pos": 0.04, "obj_rot": 0.2}, 1.0
    ns = np.array([env.num_success({"obj_pos": g["thr_dist_pos"][t], "obj_rot": g["thr_dist_rot"][t]}) for t in range(len(g["num_success"]))])
    assert np.array_equal(ns, g["num_success"]) and np.array_equal(ns[1:] - ns[:-1], g["goal_reward"]) and 0 < ns.min() + 1 and ns.max() >= 3


def _env_kernel_goal_layer(lib, device, B):
    """The env kernel's goal entries (rel_goal_obj_pos, rel_goal_obj_rot, the summed distances in info, is_goal_achieved) for object poses and goals taken
    from the golden: the objects are put at (cur_pos, euler2quat(cur_rot)), one forward, then the observation row."""
    from oracle import rearrange_oracle as RO

    g = _goal_golden()
    kw = dict(lib=lib, n_substeps=1) if lib is not None else {}
    env = BatchedBlockRearrangeEnv(B, device=device, stabilize_steps=1, n_random_initial_steps=0, settle_steps=0, **kw)
    env.reset()
    T = B * 2 if lib is not None else len(g["cur_pos"]) // B * B
    worst = np.zeros(4)
    for t0 in range(0, T, B):
        sl = slice(t0, t0 + B)
        pos = g["cur_pos"][sl] + np.array([1.45, 0.77, 0.9])                      # (somewhere above the table: nothing touches)
        quat = RO.euler2quat(g["cur_rot"][sl])
        for i, qa in enumerate(env.obj_q):
            env.sim.qpos[:, qa:qa + 7] = torch.tensor(np.concatenate([pos[:, i], quat[:, i]], -1).astype(np.float32), device=env.device)
        gq = RO.euler2quat(g["goal_rot"][sl])
        env.goal[:] = torch.tensor(np.concatenate([g["goal_pos"][sl] + np.array([1.45, 0.77, 0.9]), gq], -1).astype(np.float32), device=env.device)
        env.goal_rot[:] = torch.tensor(g["goal_rot"][sl].astype(np.float32), device=env.device)
        env.sim.env_step(nsubsteps=0, nforward_ticks=1, flags=32)
        env._observe_only()
        env.sync()
        obs = env.observe()
        rp, rr = obs["rel_goal_obj_pos"].cpu().numpy(), obs["rel_goal_obj_rot"].cpu().numpy()
        e_rr = np.abs(RO.normalize_angles(rr - g["rel_rot"][sl]))                  # (angles compare modulo 2 pi)
        # Euler triples are not unique at the gimbal lock (|pitch| ~ pi / 2): the rotation they stand for is what has to agree there
        q_k, q_g = RO.euler2quat(rr.astype(np.float64)), RO.euler2quat(g["rel_rot"][sl])
        e_q = np.minimum(np.abs(q_k - q_g).max(-1), np.abs(q_k + q_g).max(-1))
        lock = np.abs(np.abs(g["rel_rot"][sl][..., 1]) - np.pi / 2) < 0.05
        worst = np.maximum(worst, [np.abs(rp - g["rel_pos"][sl]).max(), e_rr[~lock].max() if (~lock).any() else 0.0, e_q.max(),
                                   np.abs(env.goal_dist[:, 0].cpu().numpy() - g["dist_pos"][sl].sum(-1)).max() + np.abs(env.goal_dist[:, 1].cpu().numpy() - g["dist_rot"][sl].sum(-1)).max()])
    print("env kernel goal layer vs the reference's code: rel pos %.1e, rel rot (Euler, away from gimbal lock) %.1e, rel rot as a quaternion %.1e, summed distances %.1e" % tuple(worst))
    assert worst[0] < 2e-6 and worst[1] < 2e-5 and worst[2] < 2e-6 and worst[3] < 5e-5


def test_env_kernel_goal_layer_matches_reference_code_emul(emul_lib):
    _env_kernel_goal_layer(emul_lib, "cpu", B=4)


@pytest.mark.gpu
def test_env_kernel_goal_layer_matches_reference_code_gpu():
    _env_kernel_goal_layer(None, "cuda:0", B=16)
