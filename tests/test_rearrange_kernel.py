"""The rearrange worlds on the large-model stepper (`rb_step_kernel`) against the CPU oracle: BASELINE.json configs[3] (rearrange/blocks,
num_objects = 5: UR16e + 2f-85 gripper + table, elliptic cones, equality rows, cascaded-PI actuators, F/T sensors) and the TCP solver's own
world (mocap weld), through the C ABI (`rb_batch_step_ex`, `rb_batch_step_tcp`).  The same checks run on the CPU against the kernel SOURCE
(fiber-emulation harness) and, under `-m gpu`, on the MI355X.

Protocol (as for the dactyl models): re-synchronised errors — before every compared step both sides start from the oracle's
state rounded to fp32 — because contact-rich trajectories diverge chaotically in any precision."""
import numpy as np
import pytest
import torch

from robogym_amd import _native
from robogym_amd.envs.rearrange.xml import load_blocks_model, load_solver_model
from robogym_amd.mujoco.large_simulation import LargeModelSimulation

NV = 38


@pytest.fixture(scope="module")
def models():
    return load_blocks_model(5), load_solver_model()


def _oracle_env(models, n_substeps, settle=3, seed=0):
    from oracle import rearrange_oracle as RO

    main, solver = models
    env = RO.OracleRearrangeEnv(main, solver, 5, n_substeps=n_substeps)
    rng = np.random.RandomState(seed)
    ztop = 0.453 + 0.03324 + 0.0254
    pos = [[1.2 + 0.11 * i + 0.01 * rng.rand(), 0.55 + 0.1 * i, ztop] for i in range(5)]
    yaw = rng.uniform(0, 2 * np.pi, 5)
    env.set_object_poses(pos, [[np.cos(a / 2), 0, 0, np.sin(a / 2)] for a in yaw])
    # gripper half open: at qpos0 the two finger pads touch face to face, a degenerate box - box case whose number of clipped points (4 or 5) is
    # decided by rounding (fp32 kernel vs fp64 oracle)
    cr = main.arrays["actuator_ctrlrange"][env.main.grip_act]
    env.main.sim.ctrl[env.main.grip_act] = 0.5 * (cr[0] + cr[1])
    env.main.sim.forward()
    for _ in range(settle):
        env.main.step()
    return env


def sync_from_oracle(sim, o, row=0):
    """kernel rows <- oracle state rounded to fp32; oracle <- the same rounded values"""
    for name, f in (("qpos", sim.qpos), ("qvel", sim.qvel), ("ctrl", sim.ctrl), ("pid", sim.pid), ("qacc_warmstart", sim.qacc_warmstart)):
        v = getattr(o, name).astype(np.float32)
        f[row] = torch.tensor(v, device=sim.device)
        getattr(o, name)[:] = v.astype(np.float64)
    sim.view(_native.RG_F_TIME)[row, 0] = float(o.time)
    if o.nmocap:
        mc = np.concatenate([o.mocap_pos, o.mocap_quat]).astype(np.float32)
        sim.mocap[row] = torch.tensor(mc, device=sim.device)
        o.mocap_pos[:] = mc[:3]; o.mocap_quat[:] = mc[3:]
    if o.neq:
        ed = o.eq_data.astype(np.float32)
        sim.eq_data[row] = torch.tensor(ed, device=sim.device)
        o.eq_data[:] = ed
        sim.eq_active[row] = torch.tensor(o.eq_active().astype(np.int32), device=sim.device)


def tcp_args(env, mpc=0.1, rce=True):
    from oracle import rearrange_oracle as RO

    Am, solver = env.main.model.arrays, env.solver.model
    a = _native.RbTcpArgs()
    for k in range(6):
        a.arm_qposadr[k] = int(env.solver.arm_q[k]); a.main_arm_qposadr[k] = int(env.main.arm_q[k])
    a.main_gripper_actuator = env.main.grip_act; a.tcp_body = env.solver.tcp_body; a.wrist_joint = solver.names["joint"].index("robot0:J6")
    a.reset_controller_error = 1 if rce else 0
    a.max_position_change = mpc; a.speed_roll = float(RO.SPEED_ROLL); a.speed_pitch = float(RO.SPEED_PITCH); a.joint_drift_threshold = float(RO.JOINT_DRIFT_THRESHOLD)
    a.gripper_ctrl_lo = float(Am["actuator_ctrlrange"][6, 0]); a.gripper_ctrl_hi = float(Am["actuator_ctrlrange"][6, 1])
    return a


def _stage_dump(models, lib, device):
    """One mj_step of the main world with ~25 contacts: contact list, row count, generalized forces, qacc_smooth, qacc, the new state."""
    env = _oracle_env(models, 1, settle=40)
    o = env.main.sim
    sim = LargeModelSimulation(models[0], 1, device=device, n_substeps=1, lib=lib, hand=False)
    sync_from_oracle(sim, o)
    sim.env_step(nsubsteps=1, nforward_ticks=0, flags=1)
    sim.sync()
    o.step()
    dbg = sim.scratch("dbg")[0].cpu().numpy()
    assert int(sim.status[0]) == 0
    ncon_k, nefc_k = int(dbg[0]), int(dbg[1])
    assert ncon_k == o.ncon + o.neq and nefc_k == o.nefc and o.ncon >= 20          # (the kernel's list carries the equality as its first record)
    # contacts: same geom pairs, distances, positions, normals (order may differ: pair list vs body loop)
    con = sim.scratch("contact")[0].cpu().numpy().reshape(-1, 32)[:ncon_k]
    assert int(con[0, 31]) == 2 and int(con[0, 26]) == 1                              # the finger coupling (joint equality, one row)
    kc = sorted([(int(c[27]), int(c[28]), float(c[0]), c[1:4].copy(), c[4:7].copy()) for c in con[1:]], key=lambda t: (t[0], t[1], round(t[3][0], 4), round(t[3][1], 4)))
    oc = sorted([(c["geom1"], c["geom2"], c["dist"], c["pos"], c["frame"][0]) for c in o.contacts()], key=lambda t: (t[0], t[1], round(t[3][0], 4), round(t[3][1], 4)))
    for a, b in zip(kc, oc):
        assert a[0] == b[0] and a[1] == b[1]
        assert abs(a[2] - b[2]) < 2e-6 and np.abs(a[3] - b[3]).max() < 5e-6 and np.abs(a[4] - b[4]).max() < 2e-4
    assert np.abs(dbg[8:8 + NV] - o.qfrc_bias).max() < 1e-4 * max(1.0, np.abs(o.qfrc_bias).max())
    assert np.abs(dbg[8 + 2 * NV:8 + 3 * NV] - o.qfrc_actuator).max() < 1e-4 * max(1.0, np.abs(o.qfrc_actuator).max())
    assert np.abs(dbg[8 + 3 * NV:8 + 4 * NV] - o.qacc_smooth).max() < 1e-4 * np.abs(o.qacc_smooth).max()
    assert np.abs(dbg[8 + 4 * NV:8 + 5 * NV] - o.qacc).max() < 2e-3 * np.abs(o.qacc).max()
    # ... and against the oracle's INDEPENDENT solver on the same rows (dual problem, cone projection per elliptic contact: oracle/rg_oracle.c ro_solve_pgs), which
    # shares no code with either Newton implementation
    q_dual, sweeps = o.solve_pgs(max_sweeps=400000, tol=1e-11)
    assert sweeps > 0 and np.abs(q_dual - o.qacc).max() < 1e-7 * np.abs(o.qacc).max()
    assert np.abs(dbg[8 + 4 * NV:8 + 5 * NV] - q_dual).max() < 2e-3 * np.abs(q_dual).max()
    assert np.abs(sim.qpos[0].cpu().numpy() - o.qpos).max() < 2e-6 and np.abs(sim.qvel[0].cpu().numpy() - o.qvel).max() < 2e-5
    assert np.abs(sim.pid[0].cpu().numpy() - o.pid).max() < 1e-5


def _obb_prune_is_exact(models, lib, device):
    """The broadphase's oriented-box test (after the bounding spheres) only removes pairs that cannot hold a contact: with the test switched
    off (flags bit 11) the contact list and the step are bit-identical, in both worlds."""
    env = _oracle_env(models, 1, settle=40)
    for model, o in ((models[0], env.main.sim), (models[1], env.solver.sim)):
        out = []
        for flags in (1, 1 | 2048):
            sim = LargeModelSimulation(model, 1, device=device, n_substeps=1, lib=lib, hand=False)
            sync_from_oracle(sim, o)
            sim.env_step(nsubsteps=1, nforward_ticks=0, flags=flags)
            sim.sync()
            dbg = sim.scratch("dbg")[0].cpu().numpy()
            ncon = int(dbg[0])
            out.append((ncon, sim.scratch("contact")[0].cpu().numpy().reshape(-1, 32)[:ncon].copy(), sim.qpos[0].cpu().numpy().copy(), sim.qvel[0].cpu().numpy().copy()))
            assert int(sim.status[0]) == 0
        a, b = out
        assert a[0] == b[0] and a[0] >= 1 and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


def contact_history(sim, oarm, before, row=0):
    """Did kernel and oracle hold the same NUMBER of contacts and of constraint rows in every mj_step since `before` (= `sim.stats[0]` then, with the oracle's
    `ncon_sum / nefc_sum` zeroed at the same moment)?  The kernel's statistics row and OracleArmSim sum both counts per mj_step; equal sums over an env.step are
    (up to cancelling differences) equal counts in each of its mj_steps: the classification VERDICT r04 weak 1 asks the rearrange parity bounds to be split by."""
    d = sim.stats[row].cpu().numpy().astype(np.float64) - before
    return int(round(d[0])) == int(oarm.ncon_sum) and int(round(d[1])) == int(oarm.nefc_sum)


def _resync_env_steps(models, lib, device, n_substeps, nsteps, seed=1, classify=False):
    """Re-synchronised env.steps of the dual simulation: `rb_batch_step_tcp` (sync, forward, mocap target, solver mj_steps, main ctrl) then the
    main world's mj_steps + two state-less forwards, the last in full (sensors) — against OracleRearrangeEnv.env_step.  `classify`: also returns, per step,
    whether both worlds went through the same contact / row counts on the two sides (`contact_history`)."""
    env = _oracle_env(models, n_substeps, settle=10, seed=seed)
    om, oc = env.main.sim, env.solver.sim
    sm = LargeModelSimulation(models[0], 1, device=device, n_substeps=n_substeps, lib=lib, hand=False)
    sc = LargeModelSimulation(models[1], 1, device=device, n_substeps=n_substeps, lib=lib, hand=False)
    args = tcp_args(env)
    rng = np.random.RandomState(seed)
    errs, same = [], []
    for step in range(nsteps):
        a = rng.uniform(-1, 1, 6)
        if step % 3 == 2:
            a[2] = -1.0          # push down towards the blocks / table from time to time
        sync_from_oracle(sm, om); sync_from_oracle(sc, oc)
        km, kc = sm.stats[0].cpu().numpy().astype(np.float64), sc.stats[0].cpu().numpy().astype(np.float64)
        env.main.ncon_sum = env.main.nefc_sum = env.solver.ncon_sum = env.solver.nefc_sum = 0
        sc.step_tcp(sm, torch.tensor(a[None].astype(np.float32), device=sm.device), args)
        sm.env_step(nforward_ticks=2, flags=32)
        # on_observations_updated (the env kernel's last act; joint_controlled_tcp_arm.py:114-129): the solver world's gripper follows the main world's
        sc.qpos[0, env.solver.grip_q] = sm.qpos[0, env.main.grip_q]; sc.ctrl[0, env.solver.grip_act] = sm.ctrl[0, env.main.grip_act]
        sm.sync()
        env.env_step(a)
        e = lambda x, y: float(np.abs(x.cpu().numpy().astype(np.float64) - y).max())
        errs.append((e(sc.qpos[0], oc.qpos), e(sc.mocap[0], np.concatenate([oc.mocap_pos, oc.mocap_quat])), e(sm.ctrl[0], om.ctrl), e(sm.qpos[0], om.qpos),
                     e(sm.qvel[0], om.qvel), e(sm.pid[0], om.pid), e(sm.sensordata[0], om.sensordata) / max(1.0, np.abs(om.sensordata).max())))
        assert int(sm.status[0]) == 0 and int(sc.status[0]) == 0
        same.append(contact_history(sm, env.main, km) and contact_history(sc, env.solver, kc))
    return (np.array(errs), np.array(same)) if classify else np.array(errs)


RESYNC_COLUMNS = ["solver qpos", "mocap", "main ctrl", "main qpos", "main qvel", "main pid", "sensordata (rel)"]
# per-step bounds of an ORDINARY re-synchronised env.step (80 mj_steps in fp32 against the double-precision oracle) ...
RESYNC_STEP_BOUND = {"solver qpos": 1e-4, "mocap": 2e-6, "main ctrl": 2e-5, "main qpos": 2e-4, "main qvel": 2e-2, "main pid": 1e-4, "sensordata (rel)": 1e-2}
# ... and of an env.step with a contact EVENT that the two precisions resolve a substep apart (a block's flat box-box contact point appearing, the solver's gripper
# slide leaving its limit row): the emulation harness and the MI355X round differently, so WHICH step of a trajectory is such a step differs between them --
# the harness shows one in 25 on this protocol (main qpos 9e-4, qvel 5e-2), rounds 3-4's GPU runs happened to show none
RESYNC_EVENT_BOUND = {"solver qpos": 2e-3, "mocap": 2e-5, "main ctrl": 2e-3, "main qpos": 5e-3, "main qvel": 0.5, "main pid": 5e-3, "sensordata (rel)": 0.2}


def _assert_resync(errs):
    names = RESYNC_COLUMNS
    med = dict(zip(names, np.median(errs, axis=0))); mx = dict(zip(names, errs.max(axis=0)))
    # medians: fp32 rounding.  (main ctrl = the solver's six arm angles.  The solver's whole qpos includes its gripper slides, which sit ON their upper limit
    #  (q = 0, range [-0.04473, 0]): whether the limit row is active is a rounding-level decision, worth ~1e-5 m on a joint with armature 100)
    assert med["solver qpos"] < 5e-6 and med["mocap"] < 1e-6 and med["main ctrl"] < 2e-6 and med["main qpos"] < 2e-6 and med["main qvel"] < 1e-4, (med, mx)
    assert med["main pid"] < 2e-5 and med["sensordata (rel)"] < 1e-3, (med, mx)
    # every step inside the ordinary bound, except at most one step in ten, which stays inside the event bound
    ordinary = np.array([[row[i] < RESYNC_STEP_BOUND[n] for i, n in enumerate(names)] for row in errs]).all(axis=1)
    assert (~ordinary).sum() <= max(1, len(errs) // 10), (errs[~ordinary], med, mx)
    for i, n in enumerate(names):
        assert errs[:, i].max() < RESYNC_EVENT_BOUND[n], (n, med, mx)


# ------------------------------------------------------------------------------------------------ CPU: the kernel source on the emulation harness
def test_rearrange_stage_dump_matches_oracle_emul(models, emul_lib, oracle_lib):
    _stage_dump(models, emul_lib, "cpu")


def test_oriented_box_prune_is_exact_emul(models, emul_lib, oracle_lib):
    _obb_prune_is_exact(models, emul_lib, "cpu")


def test_rearrange_resync_env_steps_emul(models, emul_lib, oracle_lib):
    _assert_resync(_resync_env_steps(models, emul_lib, "cpu", n_substeps=3, nsteps=4))


def test_register_and_matrix_pipe_solves_match_the_lds_cholesky_emul(models, emul_lib, oracle_lib, tmp_path):
    """Round 6 replaced rb_step_kernel's blocked LDS Cholesky + substitutions (rb_chol / rb_chol_solve, still in the source behind -DRB_LDS_CHOL) by a factor + solve in
    registers on the one-wave configurations (rb_reg_solve_n) and by a matrix-pipe factorisation with single-wave substitutions on the large one (rb_chol_mfma,
    rb_chol_solve_wave).  Same linear systems, different floating-point paths: one mj_step of the rearrange main world (38-dof group, ~25 contacts) and of the full cube
    (96-dof group, ~28 contacts) from the same state on a harness build of either kind."""
    import os
    import subprocess

    from robogym_amd import _native
    from robogym_amd.envs.dactyl.full_perpendicular import load_full_perpendicular_model

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "librgstep_emul_ldschol.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-std=c++17", "-U_FORTIFY_SOURCE", "-DRG_EMUL", "-DRB_LDS_CHOL", "-I" + os.path.join(root, "tests", "emul"), "-I" + os.path.join(root, "robogym_amd", "csrc"),
                           "-w", "-shared", "-o", so, os.path.join(root, "tests", "emul", "hip_emul.cpp"), "-x", "c++", os.path.join(root, "robogym_amd", "csrc", "rg_api.hip")])
    lds_lib = _native.bind(so)
    # rearrange main world: the oracle provides a contact-rich state
    env = _oracle_env(models, 1, settle=40)
    o = env.main.sim
    out = []
    for lib in (emul_lib, lds_lib):
        sim = LargeModelSimulation(models[0], 1, device="cpu", n_substeps=1, lib=lib, hand=False)
        sync_from_oracle(sim, o)
        sim.env_step(nsubsteps=1, nforward_ticks=0, flags=1)
        dbg = sim.scratch("dbg")[0].cpu().numpy()
        assert int(sim.status[0]) == 0 and int(dbg[0]) >= 20
        out.append((dbg[8 + 4 * NV:8 + 5 * NV].copy(), sim.qpos[0].cpu().numpy().copy(), sim.qvel[0].cpu().numpy().copy()))
    assert np.abs(out[0][0] - out[1][0]).max() < 2e-4 * np.abs(out[1][0]).max()          # qacc of the constrained solve
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-6 and np.abs(out[0][2] - out[1][2]).max() < 2e-5
    # the full cube on the palm: the 96-dof group through the matrix-pipe factorisation
    from tests.test_large_model import OracleFullCube, _put
    full = load_full_perpendicular_model()
    sims = [LargeModelSimulation(full, 1, lib=lib, n_substeps=1) for lib in (emul_lib, lds_lib)]
    ora = OracleFullCube(full, sims[0].pos_to_ctrl, sims[0].qpos_idxs["hand_angle"])
    ora.hold_pose()
    for _ in range(60):
        ora.sim.step()
    st = ora.state_f32()
    res = []
    for sim in sims:
        _put(sim, st)
        sim.env_step(nsubsteps=1, nforward_ticks=0, flags=0)
        assert int(sim.status[0]) == 0
        res.append((sim.qpos[0].cpu().numpy().copy(), sim.qvel[0].cpu().numpy().copy()))
    assert np.abs(res[0][0] - res[1][0]).max() < 2e-6 and np.abs(res[0][1] - res[1][1]).max() < 5e-4, (np.abs(res[0][0] - res[1][0]).max(), np.abs(res[0][1] - res[1][1]).max())


# ------------------------------------------------------------------------------------------------ MI355X
@pytest.mark.gpu
def test_rearrange_stage_dump_matches_oracle_gpu(models, oracle_lib):
    _stage_dump(models, None, "cuda:0")


@pytest.mark.gpu
def test_oriented_box_prune_is_exact_gpu(models, oracle_lib):
    _obb_prune_is_exact(models, None, "cuda:0")


@pytest.mark.gpu
def test_rearrange_resync_env_steps_gpu(models, oracle_lib):
    """the full 40 + 40 mj_steps per env.step"""
    _assert_resync(_resync_env_steps(models, None, "cuda:0", n_substeps=40, nsteps=25))


# an env.step whose 80 mj_steps held the same contact and row counts on both sides: fp32 rounding of one trajectory, no event -- bounds per step, no exceptions
# (measured on the MI355X, 150 steps of this protocol, profiles/r05_parity_rearrange.txt: 131 same-history steps with solver qpos <= 1.8e-6, main qpos <= 3.1e-6,
#  qvel <= 1.9e-4, pid <= 2.2e-6, sensordata <= 4.2e-4; the bounds leave a factor of ~10)
RESYNC_SAME_HISTORY_BOUND = {"solver qpos": 2e-5, "mocap": 2e-6, "main ctrl": 2e-5, "main qpos": 3e-5, "main qvel": 2e-3, "main pid": 3e-5, "sensordata (rel)": 5e-3}


def _assert_resync_classified(errs, same, min_same_fraction=0.5):
    """The errors of the re-synchronised env.steps split by `contact_history`: steps that went through the same contact / row counts on both sides are held to
    RESYNC_SAME_HISTORY_BOUND one by one (no "one in ten" allowance); steps whose histories differ -- a contact appearing a substep apart -- to the event bound."""
    names = RESYNC_COLUMNS
    assert same.mean() >= min_same_fraction, same
    for i, n in enumerate(names):
        assert errs[same, i].max() < RESYNC_SAME_HISTORY_BOUND[n], (n, errs[same, i].max(), int(same.sum()))
        assert errs[:, i].max() < RESYNC_EVENT_BOUND[n], (n, errs[:, i].max())


def test_rearrange_resync_classified_emul(models, emul_lib, oracle_lib):
    errs, same = _resync_env_steps(models, emul_lib, "cpu", n_substeps=3, nsteps=3, classify=True)
    _assert_resync_classified(errs, same, min_same_fraction=0.0)


@pytest.mark.gpu
def test_rearrange_resync_errors_by_contact_history_gpu(models, oracle_lib):
    """40 re-synchronised env.steps (40 + 40 mj_steps each) classified by whether kernel and oracle went through the same contact and
    row counts; the agreeing steps carry the fp32 tolerance, the disagreeing ones the event bound."""
    errs, same = _resync_env_steps(models, None, "cuda:0", n_substeps=40, nsteps=40, seed=7, classify=True)
    print("rearrange/blocks, 40 re-synchronised env.steps: %d with the same contact history (main qpos max %.1e, qvel max %.1e), %d with a differing one (qpos max %.1e, qvel max %.1e)" % (
        same.sum(), errs[same, 3].max(), errs[same, 4].max(), (~same).sum(), errs[~same, 3].max() if (~same).any() else 0, errs[~same, 4].max() if (~same).any() else 0))
    _assert_resync_classified(errs, same)


@pytest.mark.gpu
def test_rearrange_batch_is_deterministic_and_rows_are_independent_gpu(models, oracle_lib):
    """B = 512 identical rows give bit-identical results, run to run and row to row."""
    env = _oracle_env(models, 40, settle=10)
    sims = []
    for rep in range(2):
        sm = LargeModelSimulation(models[0], 512, n_substeps=40, hand=False)
        for r in (0,):
            sync_from_oracle(sm, env.main.sim, row=r)
        for f in (sm.qpos, sm.qvel, sm.ctrl, sm.pid, sm.qacc_warmstart, sm.eq_data):
            f[1:] = f[0:1]
        sm.eq_active[1:] = sm.eq_active[0:1]
        sm.env_step(nforward_ticks=2, flags=32)
        sm.sync()
        sims.append(sm)
    a, b = sims
    assert torch.equal(a.qpos, b.qpos) and torch.equal(a.qvel, b.qvel) and torch.equal(a.sensordata, b.sensordata)
    assert torch.equal(a.qpos[1:], a.qpos[:1].expand(511, -1)) and int(a.status.max()) == 0
