"""The rearrange features of the CPU oracle (elliptic cones + impratio, weld-to-mocap and joint-polycoef equality rows, negative
solref, mocap bodies, mujoco-py's cascaded-PI actuator, jointpos / force / torque sensors) and the rearrange env oracle
(oracle/rearrange_oracle.py), pinned by

* closed forms of the documented constraint model (no MuJoCo available: SURVEY.md §8c),
* goldens generated from the reference's own rotation utilities (tools/gen_golden_rearrange.py),
* the reference's own property tests re-expressed on the oracle:
  /root/reference/robogym/envs/rearrange/tests/test_rearrange_sim.py:10-55 (sizes), :96-132 (gripper sync, final -0.04473 +- 1e-4),
  :135-230 (mocap-IK impulse response), tests/test_rearrange_robots.py:45-78,194-243 (action scaling tables).
"""
import xml.etree.ElementTree as et

import numpy as np
import pytest

from oracle import rearrange_oracle as RO
from oracle.rg_oracle import OracleSim
from robogym_amd.envs.rearrange.xml import load_blocks_model, load_solver_model
from robogym_amd.mujoco.mjcf_compiler import compile_mjcf
from robogym_amd.mujoco.model_blob import pack_model


def _sim(xml):
    m = compile_mjcf(et.XML(xml))
    return m, OracleSim(pack_model(m))


@pytest.fixture(scope="module")
def models():
    return load_blocks_model(5), load_solver_model()


def _env(models, mpc=0.1, rce=True, stabilize=100):
    main, solver = models
    env = RO.OracleRearrangeEnv(main, solver, 5, max_position_change=mpc, arm_reset_controller_error=rce)
    ztop = 0.453 + 0.03324 + 0.0254
    env.set_object_poses([[1.25 + 0.1 * i, 0.9 + 0.08 * i, ztop] for i in range(5)], [[1, 0, 0, 0]] * 5)
    env.main.sim.forward()
    for _ in range(stabilize):     # stabilize_objects (common/utils.py:76-92): 100 simulation steps at reset, the arm holding its start pose
        env.main.step()
    return env


# ------------------------------------------------------------------------------------------------ model
def test_arm_actuation_range_table(models):
    """robot/test/test_robot_interface.py:25-48, the reference's own numbers: the arm's actuation range (half the control range of its six actuators; what a JOINT
    arm reports for absolute actions and, without max_position_change, for relative ones) is 6.1959, 6.1959, 2.8, 6.1959, 6.1959, 2.312561."""
    cr = models[0].arrays["actuator_ctrlrange"][:6]
    assert np.allclose(0.5 * (cr[:, 1] - cr[:, 0]), [6.1959, 6.1959, 2.8, 6.1959, 6.1959, 2.312561], atol=1e-6)


def test_model_sizes_match_the_reference(models):
    """test_rearrange_sim.py:10-55 and SURVEY §8's dimension table (43 / 38 / 7 / 37 / 55 / 3; solver 8 / 8 / 1 / 27 / 45 / 3)."""
    main, solver = models
    assert list(main.dims[:7]) == [43, 38, 7, 37, 13, 55, 3]
    assert list(solver.dims[:7]) == [8, 8, 1, 27, 8, 45, 3]
    assert list(main.size_int[:4]) == [2000, 500, 2000, 16]
    assert list(solver.size_int[:3]) == [200, 200, 200]
    assert main.opt_timestep[0] == 0.001 and solver.opt_timestep[0] == 0.001
    assert main.opt_int[1] == 1 and main.opt_impratio[0] == 10.0            # cone elliptic, impratio 10 (ur16e/base.xml:3)
    # main world: joint actuated -> no weld, the finger coupling only; solver world: mocap weld + coupling (base.xml:52-54, gripper_actuators.xml:3)
    assert list(main.eq_type) == [2] and list(solver.eq_type) == [1, 2]
    assert main.names["actuator"][:6] == ["ur_actuator_%d" % k for k in range(1, 7)] and list(main.actuator_user[:6]) == [1] * 6
    assert np.allclose(main.actuator_gainprm[0], [12, 0, 0, 0, 0, 70, .05, 1.0, .97, 2.094])
    assert solver.nmocap[0] == 1 and solver.body_mocapid[solver.name2id("body", "robot0:mocap")] == 0
    # robot/test/test_robot_interface.py:26-48: ctrl ranges of the arm actuators
    assert np.allclose(main.actuator_ctrlrange[:6, 1], [6.1959, 6.1959, 2.8, 6.1959, 6.1959, 5.49778714])
    assert np.allclose(main.actuator_ctrlrange[6], [-0.04473, 0])


def test_rotation_helpers_match_the_reference_goldens():
    g = np.load("tests/golden/rearrange_rotation.npz")
    assert np.abs(RO.euler2quat(g["euler"]) - g["euler2quat"]).max() < 1e-14
    assert np.abs(RO.quat2mat(g["quat"]) - g["quat2mat"]).max() < 1e-14
    assert np.abs(RO.mat2euler(g["quat2mat"]) - g["mat2euler"]).max() < 1e-14
    assert np.abs(RO.mat2quat(g["quat2mat"]) - g["mat2quat"]).max() < 1e-12
    assert np.abs(RO.subtract_euler(g["euler"], g["euler2"]) - g["subtract_euler"]).max() < 1e-13
    assert np.abs(RO.normalize_angles(3 * g["euler"]) - g["normalize_angles"]).max() < 1e-14
    assert np.abs(RO.quat_magnitude(RO.quat_normalize(RO.euler2quat(g["rel_rot"]))) - g["rot_distance"]).max() < 1e-13


# ------------------------------------------------------------------------------------------------ closed forms
WELD = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="0.001" gravity="0 0 0"/>
  <worldbody>
    <body mocap="true" name="mocap" pos="0 0 0"/>
    <body name="puck" pos="0 0 0">
      <joint type="free"/>
      <geom type="sphere" size="0.05" mass="2.0" contype="0" conaffinity="0"/>
    </body>
  </worldbody>
  <equality>
    <weld body1="mocap" body2="puck" solimp="0.9 0.95 0.001" solref="0.02 1"/>
  </equality>
</mujoco>
"""


def test_weld_to_mocap_follows_the_documented_critically_damped_law():
    """A free body welded to a mocap body that jumps by delta: with a_1 = d a_ref (exact here: one body, A = diagApprox, so the
    regulariser R = (1 - d) / d A gives weight d on a_ref) the residual obeys r'' = -d (b r' + k r), b = 2 / (dmax tc),
    k = d / (dmax^2 tc^2): critically damped, omega = d / (dmax tc).  Checked against that ODE integrated with the same
    semi-implicit Euler rule, and against the analytic envelope."""
    m, s = _sim(WELD)
    assert s.neq == 1 and s.nmocap == 1
    delta = np.array([0.05, -0.02, 0.03])
    s.mocap_pos[:] = delta
    h, d, dmax, tc = 0.001, 0.95, 0.95, 0.02
    b, k = 2 / (dmax * tc), d / (dmax ** 2 * tc ** 2)
    x, v = np.zeros(3), np.zeros(3)
    for i in range(60):
        s.step()
        a = d * (-b * v - k * (x - delta)); v = v + h * a; x = x + h * v
        assert s.ne == 6 and s.nefc == 6
        assert np.abs(s.qpos[:3] - x).max() < 2e-6, i
    w = d / (dmax * tc)
    assert np.abs(s.qpos[:3] / delta - (1 - (1 + w * 0.06) * np.exp(-w * 0.06))).max() < 0.01
    # orientation rows: a mocap rotation about z is followed too (half-angle residual, same law)
    ang = 0.2
    s.mocap_quat[:] = [np.cos(ang / 2), 0, 0, np.sin(ang / 2)]
    for i in range(400):
        s.step()
    q = s.qpos[3:7]
    assert abs(2 * np.arctan2(q[3], q[0]) - ang) < 1e-4 and np.abs(s.qpos[:3] - delta).max() < 1e-5


COUPLED = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="0.001" gravity="0 0 -9.81"/>
  <worldbody>
    <body name="a" pos="0 0 0"><joint name="ja" type="slide" axis="1 0 0" damping="5"/><geom type="box" size=".02 .02 .02" mass="0.5" contype="0" conaffinity="0"/></body>
    <body name="b" pos="0 0.2 0"><joint name="jb" type="slide" axis="1 0 0" damping="5"/><geom type="box" size=".02 .02 .02" mass="0.5" contype="0" conaffinity="0"/></body>
  </worldbody>
  <equality><joint joint1="jb" joint2="ja" polycoef="0.01 2 0 0 0" solref="-50000 -100"/></equality>
  <actuator><general joint="ja" gainprm="1"/></actuator>
</mujoco>
"""


def test_joint_equality_with_negative_solref_couples_two_slides():
    """gripper_actuators.xml:3 shape: q_b = c0 + c1 q_a held by a direct (stiffness, damping) solref.  At rest under a constant
    force the residual is f_constraint / (stiffness-scaled) small, and the Jacobian carries -poly'."""
    m, s = _sim(COUPLED)
    s.ctrl[0] = 1.5
    for _ in range(3000):
        s.step()
    qa, qb = s.qpos[0], s.qpos[1]
    assert abs(s.qvel[0]) < 0.31 and s.ne == 1
    assert abs(qb - (0.01 + 2 * qa)) < 2e-4          # stiffness 50000 / dmax^2 against forces of order 1 N
    J = s.efc_J[: m.dims[1]]
    assert np.allclose(J, [-2.0, 1.0])


SLOPE_ELLIPTIC = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="0.001" cone="elliptic" impratio="10" gravity="{gx} 0 {gz}"/>
  <worldbody>
    <geom name="floor" type="plane" size="5 5 1" condim="3" friction="{mu} 0.005 0.0001"/>
    <body name="brick" pos="0 0 0.05">
      <joint type="free"/>
      <geom name="brick" type="box" size="0.1 0.1 0.05" condim="{condim}" friction="{mu} 0.005 0.0001" density="800"/>
    </body>
  </worldbody>
</mujoco>
"""


@pytest.mark.parametrize("mu,condim", [(0.3, 3), (0.8, 4), (0.5, 6)])
def test_elliptic_cone_coulomb_friction_on_a_slope(mu, condim):
    """A brick on a plane under tilted gravity: below the friction angle it stays (creep of the soft constraint only), above it
    it accelerates with g (sin t - mu cos t).  impratio = 10 makes the friction rows 10 x stiffer than the normal one."""
    g = 9.81
    for theta, slides in ((np.arctan(mu) * 0.7, False), (np.arctan(mu) + 0.15, True)):
        m, s = _sim(SLOPE_ELLIPTIC.format(gx=g * np.sin(theta), gz=-g * np.cos(theta), mu=mu, condim=condim))
        assert m.opt_int[1] == 1
        for _ in range(300):
            s.step()
        v0, t0 = s.qvel[0], s.time
        for _ in range(300):
            s.step()
        acc = (s.qvel[0] - v0) / (s.time - t0)
        if slides:   # (the sliding brick rocks on its leading edge: contacts come and go, the mean acceleration is Coulomb's)
            assert abs(acc - g * (np.sin(theta) - mu * np.cos(theta))) < 0.06 * g * np.sin(theta)
        else:
            assert s.ncon == 4 and s.nefc == 4 * condim
            assert abs(acc) < 5e-3 and abs(s.qvel[0]) < 2e-2


def test_elliptic_newton_solution_is_a_stationary_point(models):
    """KKT of the primal problem at the solver's answer: M (qacc - qacc_smooth) = J' f with f the cone forces the oracle reports,
    and every contact force inside its friction cone."""
    env = _env(models, stabilize=5)
    s = env.main.sim
    rng = np.random.RandomState(3)
    for k in range(30):
        env.env_step(rng.uniform(-1, 1, 6))
    s.forward()
    nv = 38
    M = s.qM.reshape(nv, nv)
    J = s.efc_J[: s.nefc * nv].reshape(s.nefc, nv)
    f = s.efc_force[: s.nefc]
    res = M @ (s.qacc - s.qacc_smooth) - J.T @ f
    assert np.abs(res).max() < 1e-6 * max(1.0, np.abs(J.T @ f).max())
    assert s.ncon >= 20
    for c in s.contacts():
        a, dim = c["efc_address"], c["dim"]
        fn = f[a]
        assert fn >= -1e-9
        ft = f[a + 1:a + dim] / c["friction"][: dim - 1]
        assert np.linalg.norm(ft) <= fn * (1 + 1e-6) + 1e-9          # elliptic cone: |f_t / mu| <= f_n


CASCADE = """
<mujoco>
  <compiler angle="radian"/>
  <size nuserdata="20" nuser_actuator="1"/>
  <option timestep="0.001" gravity="0 0 0"/>
  <worldbody>
    <body name="link"><joint name="j" type="hinge" axis="0 0 1" damping="0.01" armature="0.01"/>
      <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.02" mass="1.0" contype="0" conaffinity="0"/></body>
  </worldbody>
  <actuator>
    <general gaintype="user" biastype="user" name="a" joint="j" ctrlrange="-6 6" forcerange="-56 56" gainprm="12 0 0 0 0 10 .10 1.5 .97 3.142" user="1"/>
  </actuator>
</mujoco>
"""


def test_cascaded_pi_controller_tracks_and_respects_its_velocity_cap():
    """mjpid.pyx's cascade on a single hinge: the commanded position is reached, the joint speed never exceeds max_vel (+ the PI
    loop's overshoot), the smoothed set-point follows ctrl with the EMA's time constant, and the state is 3 numbers per actuator."""
    m, s = _sim(CASCADE)
    s.ctrl[0] = 0.0
    s.step()                                   # time 0: warm start of the smoothed set-point
    s.ctrl[0] = 2.0
    vmax = 0
    ema = []
    for i in range(3000):
        s.step()
        vmax = max(vmax, abs(s.qvel[0]))
        ema.append(s.pid[2])
    assert abs(ema[0] - 2.0 * 0.03) < 1e-12 and abs(ema[99] - 2.0 * (1 - 0.97 ** 100)) < 1e-9
    assert vmax < 3.142 * 1.1 and vmax > 3.0   # position error 2 rad x kp 12 saturates the velocity set-point
    assert abs(s.qpos[0] - 2.0) < 2e-3 and abs(s.qvel[0]) < 2e-2


def test_force_torque_sensor_reads_the_static_load(models):
    """toolhead_force / toolhead_torque (ur16e/base.xml:47-49) at rest: the interaction force at the `robot0:grip` site equals the weight
    of everything distal to it (the gripper subtree), rotated into the site frame."""
    env = _env(models, stabilize=60)
    s, m = env.main.sim, env.main.model
    A = m.arrays
    body = int(A["site_bodyid"][m.names["site"].index("robot0:grip")])
    mass = A["body_subtreemass"][body]
    f = s.sensordata[env.force_adr:env.force_adr + 3]
    assert abs(np.linalg.norm(f) - mass * 9.81) < 0.02 * mass * 9.81
    R = s.site_xmat[9 * m.names["site"].index("robot0:grip"):][:9].reshape(3, 3)
    assert np.abs(R @ f - np.array([0, 0, mass * 9.81])).max() < 0.03 * mass * 9.81      # the parent holds the child UP
    assert np.allclose(s.sensordata[:6], s.qpos[:6])                                         # jointpos sensors


# ------------------------------------------------------------------------------------------------ reference pins
@pytest.mark.parametrize("mpc,control,pos,angle,grip", [
    (1.0, np.ones(6), [1.0] * 3, [1.0, 1.0], 0.0), (0.05, np.ones(6), [0.05] * 3, [0.05, 0.05], 0.0),
    (1.0, -np.ones(6), [-1.0] * 3, [-1.0, -1.0], -0.022365), (0.05, -np.ones(6), [-0.05] * 3, [-0.05, -0.05], -0.022365)])
def test_tcp_arm_denormalization_table(models, mpc, control, pos, angle, grip):
    """envs/rearrange/tests/test_rearrange_robots.py:45-78 (xyz x max_position_change, angles x (200, 600) deg x mpc, gripper -1 -> -0.022365)"""
    main, solver = models
    env = RO.OracleRearrangeEnv(main, solver, 5, max_position_change=mpc)
    arm, g = env.denormalize(control)
    assert np.allclose(arm, np.concatenate([pos, np.array(angle) * np.deg2rad([200, 600])]))
    assert np.isclose(g, grip)


def test_tcp_action_scaling_table(models):
    """test_rearrange_robots.py:194-243: TCP_ROLL_YAW at mpc 0.05 -> (0.05 x 3, 10 deg, 30 deg); at 0.27 -> (0.27 x 3, 54 deg, 162 deg)"""
    for mpc, exp in ((0.05, [0.05, 0.05, 0.05, np.deg2rad(10), np.deg2rad(30)]), (0.27, [0.27, 0.27, 0.27, np.deg2rad(54), np.deg2rad(162)])):
        env = RO.OracleRearrangeEnv(models[0], models[1], 5, max_position_change=mpc)
        assert np.allclose(env.denormalize(np.ones(6))[0], exp)


def test_dual_sim_gripper_sync(models):
    """test_rearrange_sim.py:96-132: closed at the start; 25 steps of "open": both simulations agree (0.012 for the first five steps,
    0.001 after), and end at the joint limit -0.04473 +- 1e-4.  Both clocks advance alike (:56-84)."""
    env = _env(models)
    m, c = env.main, env.solver
    assert abs(m.sim.qpos[m.grip_q]) < 1e-4 and abs(c.sim.qpos[c.grip_q]) < 1e-4
    a = np.zeros(6); a[-1] = -1
    t0 = m.sim.time - c.sim.time
    for i in range(25):
        env.env_step(a)
        assert abs(m.sim.qpos[m.grip_q] - c.sim.qpos[c.grip_q]) < (0.012 if i < 5 else 0.001)
        assert abs((m.sim.time - c.sim.time) - t0) < 1e-5
    assert abs(m.sim.qpos[m.grip_q] + 0.04473) < 1e-4 and abs(c.sim.qpos[c.grip_q] + 0.04473) < 1e-4


def _impulse_trajectory(models, mpc, rce, dim):
    """The reference test's trajectory (test_rearrange_sim.py:176-207): two zero steps, one full action on TCP axis `dim` (None: no impulse), 40 zero steps, stepped
    through the SmoothActionWrapper as `make_env(...).env` is (alpha 0.3 per 0.08 s, bias-corrected: wrappers/util.py:142-160,203-211 -- the robot sees 0.54, 0.27,
    0.14, ... sum 1.11).  Returns the TCP displacements relative to the first step and the arm's joint displacement over the run."""
    alpha = 0.3 ** (0.001 * 40 / 0.08)
    env = _env(models, mpc, rce)
    z = np.zeros(6); imp = z.copy()
    if dim is not None:
        imp[dim] = 1
    q0 = env.main.sim.qpos[env.main.arm_q].copy()
    P, ema = [], np.zeros(6)
    for k in range(43):
        ema = ema * alpha + (1 - alpha) * (imp if k == 2 else z)
        env.env_step(ema / (1 - alpha ** (k + 1)))
        P.append(env.main.body_xpos(env.main.tcp_body))
    return np.array(P) - P[0], env.main.sim.qpos[env.main.arm_q] - q0


@pytest.mark.parametrize("rce,mpc,expected,rise", [(True, 0.165, 0.036, 5), (False, 0.05, 0.0363, 12), (True, 0.1, 0.022, 5), (False, 0.03, 0.022, 12)])
def test_mocap_ik_impulse_response(models, rce, mpc, expected, rise):
    """test_rearrange_sim.py:135-230, as the reference states it: an impulse action on one TCP axis after two zero steps, then 40 zero steps; (i) 90 % of the steady-state
    displacement within `rise` steps of the impulse, (ii) the steady-state displacement itself within 1e-3 of the expected value, on each of x, y, z, for all four
    (controller-error reset, max_position_change) cases.  Measured: 0.0362 / 0.0364 / 0.0365 for 0.036, 0.0365 / 0.0367 / 0.0365 for 0.0363, 0.02197 / 0.02209 / 0.02206
    and 0.0219 / 0.0220 / 0.0218 for 0.022.
    History: rounds 3-4 fed the raw impulse (no smoothing wrapper) and read the missing factor 1.11 as a 10 % deviation; with the wrapper in the loop the two cases with
    controller-error reset still missed by 1.4-1.6e-3 on x and y until the cascaded-PI controller got its bias feed-forward (next test)."""
    for dim in range(3):
        P, _ = _impulse_trajectory(models, mpc, rce, dim)
        total = P[-1, dim]
        assert abs(total - expected) < 1e-3, (dim, total)
        assert abs(P[2 + rise, dim]) > 0.9 * total


def test_cascaded_pi_bias_feed_forward_is_what_the_reference_pins_ask_for(models, oracle_lib):
    """mjpid.pyx is not in the tree, so whether the cascaded-PI controller adds the actuated dof's bias force (gravity + Coriolis compensation) to the velocity loop's
    output has to be inferred.  The wrist joints' velocity loops are P-only (gainprm `kp_v 20, ti_v 0` for J5, `1, 0` for J6: joint_actuations.xml:9-10) and the
    wrist carries an off-axis camera (0.42 kg, 6 cm).  WITHOUT the feed-forward J5 creeps by -2.2e-4 rad per env.step under that weight -- with the controller-error
    reset every step's joint target is the position the arm has crept to -- the TCP drifts +2.1e-3 / -1.5e-3 over the 43 steps of the impulse-response test with
    ZERO actions, and the reference's pin fails on x and y (0.0374 / 0.0346 for 0.036 +- 1e-3); WITH it the drift is 2e-5 and all four cases hold (test above).  Both
    runs are kept here: the oracle's default is the feed-forward, `ro_set_cascade_bias_ff(0)` is the comparison."""
    try:
        oracle_lib.lib().ro_set_cascade_bias_ff(0)
        drift0, dq0 = _impulse_trajectory(models, 0.165, True, None)
        x0 = _impulse_trajectory(models, 0.165, True, 0)[0][-1, 0]
    finally:
        oracle_lib.lib().ro_set_cascade_bias_ff(1)
    drift1, dq1 = _impulse_trajectory(models, 0.165, True, None)
    assert 1e-3 < np.abs(drift0[-1, :2]).max() < 3e-3 and np.argmax(np.abs(dq0)) == 4 and abs(dq0[4]) > 5 * np.abs(np.delete(dq0, 4)).max(), (drift0[-1], dq0)
    assert abs(x0 - 0.036) > 1e-3                                          # the reference's own assertion fails without the term ...
    assert np.abs(drift1[-1]).max() < 2e-4 and np.abs(dq1).max() < 1e-3, (drift1[-1], dq1)      # ... and the arm holds still with it


def test_bias_force_at_rest_is_the_gradient_of_the_potential_energy(models):
    """What the cascaded-PI controller feeds forward: at zero velocity `qfrc_bias` (mj_rne: gravity + Coriolis) is the gradient of the potential energy
    V(q) = - sum_b m_b g . xipos_b with respect to the joint angles -- checked by central differences over forward kinematics alone, for the six arm joints and the
    gripper's, in five random arm poses (1e-6 of the largest gravity torque, ~70 N m at the shoulder)."""
    env = _env(models, 0.1, True, stabilize=0)
    m, s = env.main, env.main.sim
    A = m.model.arrays
    mass, g = A["body_mass"], np.asarray(A["opt_gravity"], dtype=float).reshape(3)
    rng = np.random.RandomState(2)
    qadr = list(m.arm_q) + [m.grip_q]
    dofs = [int(A["jnt_dofadr"][list(A["jnt_qposadr"]).index(q)]) for q in qadr]

    def potential():
        s.forward()
        return -float((mass[:, None] * s.field("xipos").reshape(-1, 3) @ g).sum())

    for _ in range(5):
        s.qvel[:] = 0
        s.qpos[m.arm_q] = RO.TABLETOP_EXPERIMENT_INITIAL_POS + rng.uniform(-0.4, 0.4, 6)
        s.forward()
        bias = s.qfrc_bias.copy()
        grad = []
        for q in qadr:
            q0, eps = s.qpos[q], 1e-5
            s.qpos[q] = q0 + eps; vp = potential()
            s.qpos[q] = q0 - eps; vm = potential()
            s.qpos[q] = q0
            grad.append((vp - vm) / (2 * eps))
        s.forward()
        assert np.abs(bias[dofs]).max() > 20 and np.abs(bias[dofs] - grad).max() < 1e-6 * np.abs(bias[dofs]).max() + 1e-7, (bias[dofs], grad)


def test_elliptic_dual_solver_agrees_with_newton(models):
    """The independent cross-check of tests/test_oracle.py::test_pgs_dual_solver_agrees_with_newton for the rearrange worlds, i.e. for ELLIPTIC cones with
    impratio 10, weld / joint equalities with negative solref, friction loss and limits: `ro_solve_pgs` minimises the DUAL problem -- forces, each elliptic
    contact's rows confined to the friction cone f_0 >= 0, sum (f_j / friction_j)^2 <= f_0^2 by a second-order-cone projection -- and shares nothing with the
    Newton solver but the constraint rows (no three-zone cone cost, no Hessian, no line search).  States: the gripper closing on a block and pushing it into the table
    (main world: ~20 contacts, ~125 rows; solver world: the gripper on the table plane).  The two accelerations agree to 1e-10 in the median, 1e-6 at worst (the
    Newton solver's own tolerance shows in one state, 6e-8)."""
    env = _env(models, 0.1, True, stabilize=20)
    rng = np.random.RandomState(0)
    tcp = env.main.body_xpos(env.main.tcp_body)
    ztop = 0.453 + 0.03324 + 0.0254
    env.set_object_poses([[tcp[0], tcp[1], ztop]] + [[1.25 + 0.1 * i, 0.9 + 0.08 * i, ztop] for i in range(1, 5)], [[1, 0, 0, 0]] * 5)
    env.main.sim.forward()
    errs, contacts = [], 0
    for k in range(22):
        a = rng.uniform(-1, 1, 6); a[2] = -0.6; a[5] = 1.0 if k > 8 else -1.0
        env.env_step(a)
        if k % 3 != 0 and k < 18:
            continue
        for s in (env.main.sim, env.solver.sim):
            s.forward()
            q, sweeps = s.solve_pgs(max_sweeps=400000, tol=1e-11)
            assert 0 < sweeps < 400000
            errs.append(float(np.abs(q - s.qacc).max() / max(1.0, np.abs(s.qacc).max())))
            contacts += s.ncon
    assert np.median(errs) < 1e-10 and max(errs) < 1e-6 and contacts > 100, (errs, contacts)


@pytest.mark.parametrize("z_action", [1, -1])
def test_table_collision_penalty(models, z_action):
    """envs/rearrange/tests/test_rearrange_envs.py:323-399 (blocks, MOCAP_IK, max_position_change = the solver mode's default 0.1, no random initial steps): 20 steps of
    a full action along world z through the smoothing wrapper (the reference steps `make_env(...).env`).  Towards the table the step's reward is exactly minus the
    table_collision penalty (the finger pads reach `table_collision_plane`, 1 mm above the table top, in both worlds -- the solver world holds the same plane, so the
    arm is stopped there gently: no safety stop); away from it the reward is not negative.  Before the first step no penalty applies."""
    env = _env(models, 0.1, True)
    env.penalty = dict(table_collision=0.2, objects_off_table=1.0)
    assert not env.gripper_table_contact()
    alpha, ema = 0.3 ** (0.001 * 40 / 0.08), np.zeros(6)
    a = np.zeros(6); a[2] = z_action
    for k in range(20):
        ema = ema * alpha + (1 - alpha) * a
        obs, reward, _, done, _ = env.env_step(ema / (1 - alpha ** (k + 1)))
    if z_action < 0:
        assert reward == -0.2 and not done and not obs["safety_stop"][0]
        assert abs(obs["gripper_pos"][2] - env.table_height) < 1e-3
    else:
        assert reward >= 0.0 and obs["gripper_pos"][2] > env.table_height + 0.3


def test_gripper_table_proximity(models):
    """test_rearrange_envs.py:140-176 (TCP_ROLL_YAW): raw full actions towards the table bring the grip site within REACH_THRESHOLD = 0.02 of the table top in at most
    STEPS_THRESHOLD = 10 env.steps, and never below it."""
    env = _env(models, 0.1, True)
    a = np.zeros(6); a[2] = -1.0
    z, z_min, t = env.main.body_xpos(env.main.tcp_body)[2], np.inf, 0
    while z > env.table_height + 0.02 and t < 10:
        z = env.env_step(a)[0]["gripper_pos"][2]
        z_min, t = min(z_min, z), t + 1
    assert z <= env.table_height + 0.02 and z_min >= env.table_height and t <= 5


def test_randomized_initial_robot_position_comes_to_rest(models):
    """test_rearrange_envs.py:296-320 with common/base.py:484-496 (`_randomize_robot_initial_position`: one random action for n_random_initial_steps = 10 steps, then 100
    steps of the zero action): resets end at different TCP positions with the gripper "not in motion", |gripper_velp| <= 3e-3 per axis in the reference's test.  The
    reference's random actions come from gym's action_space.sample() and cannot be replayed; six uniform random actions here.  Four poses come to rest at
    < 1e-4 m/s.  In the other two the random action has folded the elbow to 2.75 rad, next to J3's control range of 2.8, where its actuator sits at or near its
    force limit (150 N m) and the arm is still unfolding at 4-5e-3 m/s after the 100 steps: asserted as such (speed <= 1e-2 and J3's actuator force above 100 N m),
    not as at rest.  (Before the cascaded-PI controller had its bias feed-forward, three further poses crept at 3-4e-3 m/s.)"""
    rng, ends, speed, elbow = np.random.RandomState(1), [], [], []
    for _ in range(6):
        env = _env(models, 0.1, True)
        a = rng.uniform(-1, 1, 6)
        for _ in range(10):
            env.env_step(a)
        for _ in range(100):
            obs = env.env_step(np.zeros(6))[0]
        assert not obs["safety_stop"][0]
        speed.append(np.abs(obs["gripper_velp"]).max()); ends.append(obs["gripper_pos"])
        elbow.append(abs(env.main.sim.field("actuator_force")[2]))
    speed, elbow = np.array(speed), np.array(elbow)
    at_rest = speed <= 3e-3                                             # the reference's tolerance
    assert at_rest.sum() >= 4 and speed[at_rest].max() < 2e-4, speed
    assert (speed[~at_rest] <= 1e-2).all() and (elbow[~at_rest] > 100).all() and (elbow[at_rest] < 60).all(), (speed, elbow)
    assert min(np.linalg.norm(ends[i] - ends[j]) for i in range(6) for j in range(i)) > 1e-2


def _wrist_only(models, value, steps=100):
    """test_rearrange_robots.py:306-378's protocol on the default env (TCP_ROLL_YAW, MOCAP_IK, max_position_change 0.1, no random initial steps): `steps` steps of the
    wrapped env with every action bin at 5 (zero) except the wrist's (index -2) at 7 or 3, i.e. +-0.4 through the smoothing wrapper.  Returns the arm's joint
    displacement over the run."""
    env = _env(models, 0.1, True)
    m = env.main
    q0 = m.sim.qpos[m.arm_q].copy()
    alpha, ema = 0.3 ** (0.001 * 40 / 0.08), np.zeros(6)
    a = np.zeros(6); a[-2] = value
    for k in range(steps):
        ema = ema * alpha + (1 - alpha) * a
        env.env_step(ema / (1 - alpha ** (k + 1)))
    return m.sim.qpos[m.arm_q] - q0


def test_wrist_isolation(models, oracle_lib):
    """test_rearrange_robots.py:306-378, as stated: 100 steps of a pure wrist action, clockwise and counter-clockwise, leave J1 .. J5 within 0.7 deg of where they
    started while J6 turns (to its range's end less JOINT_DRIFT_THRESHOLD, where `constrain_quat_ctrl` holds it).  This is the reference's second statement that
    needs the cascaded-PI controller's bias feed-forward: without it J5 alone creeps by more than a degree in those 100 steps (checked below)."""
    tol = np.deg2rad(0.7)
    for value in (0.4, -0.4):
        dq = _wrist_only(models, value)
        assert np.abs(dq[:5]).max() < tol and abs(dq[5]) > 0.5, np.rad2deg(dq)
    try:
        oracle_lib.lib().ro_set_cascade_bias_ff(0)
        dq = _wrist_only(models, 0.4)
    finally:
        oracle_lib.lib().ro_set_cascade_bias_ff(1)
    assert np.abs(dq[:5]).max() > tol and np.argmax(np.abs(dq[:5])) == 4, np.rad2deg(dq)


def test_free_wrist_reach(models):
    """test_rearrange_robots.py:142-190 (TCP_ROLL_YAW, MOCAP_IK, max_position_change 0.1, controller-error reset) with robot/utils/reach_helper.py:731-762: a
    proportional controller on the wrist dimension of the DENORMALISED control -- Kp x clip(error, +-30 deg) per step, the arm auto-stepping -- brings J6 to each of 12
    angles between its control range's ends less 5 deg within 200 steps (tolerance 1e-2 rad, Kp = 2); the angles 5 deg OUTSIDE the range are not reached (Kp = 1,
    100 steps): `constrain_quat_ctrl` keeps the wrist JOINT_DRIFT_THRESHOLD = 1 deg inside.  (The reference resets with 10 random initial steps; here every reach
    starts from the tabletop start pose.)"""
    def reach(target, Kp=1, max_steps=100, speed_limit=np.deg2rad(30), atol=1e-2):
        env = _env(models, 0.1, True, stabilize=20)
        m = env.main
        j6 = lambda: m.sim.qpos[m.arm_q[5]]
        step, error = 0, target - j6()
        while step < max_steps and not abs(error) < atol:
            ctrl = np.zeros(5); ctrl[-1] = Kp * np.clip(error, -speed_limit, speed_limit)
            env.set_control(ctrl, m.sim.ctrl[m.grip_act])
            m.step(); env.observe_sync()                                  # autostep + the observation that syncs the solver world's gripper
            error, step = target - j6(), step + 1
        return abs(error) < atol, step, j6()

    A = models[0].arrays
    lo, hi = A["actuator_ctrlrange"][5]
    assert np.allclose([lo, hi], [0.872665, 5.49778714])                 # joint_actuations.xml:10
    buf = np.deg2rad(5)
    steps = []
    for target in np.linspace(lo + buf, hi - buf, 12):
        ok, n, got = reach(target, Kp=2, max_steps=200)
        assert ok, (np.rad2deg(target), np.rad2deg(got))
        steps.append(n)
    assert max(steps) <= 60, steps                                        # (measured: at most a few tens of steps; the reference allows 200)
    for target in (lo - buf, hi + buf):
        ok, n, got = reach(target)
        assert not ok and abs(got - np.clip(target, lo + np.deg2rad(1), hi - np.deg2rad(1))) < np.deg2rad(1.5), (np.rad2deg(target), np.rad2deg(got))


def test_discretize_action_tables():
    """wrappers/tests/test_action_wrappers.py:7-31, the reference's own numbers: 11 linear bins = linspace(-1, 1, 11), 11 exponential bins =
    -1, -0.5, -0.25, -0.125, -0.0625, 0, ... 1; and the env's device table is that array for each of the six action dimensions."""
    from robogym_amd.envs.rearrange.blocks import action_bin_array

    assert np.array_equal(action_bin_array(-1.0, 1.0, 11, "linear"), np.linspace(-1, 1, 11))
    assert np.array_equal(action_bin_array(-1.0, 1.0, 11, "exponential"), [-1.0, -0.5, -0.25, -0.125, -0.0625, 0.0, 0.0625, 0.125, 0.25, 0.5, 1.0])
    with pytest.raises(AssertionError):
        action_bin_array(-1.0, 1.0, 10, "exponential")
    with pytest.raises(NotImplementedError):
        action_bin_array(-1.0, 1.0, 11, "log")


def test_make_env_rejects_what_it_does_not_implement():
    """`make_env(parameters=..., constants=...)` of the batched rearrange envs: a name of the reference's parameter / constant classes that is not implemented raises
    (the reference's attrs classes raise on unknown names; silently ignoring e.g. `success_pause_range_s` would change the task without a word), and so does a
    solver mode other than mocap_ik; a control mode that is none of the reference's three ControlMode values is a ValueError (all three are built)."""
    from robogym_amd.envs.rearrange import blocks as Bk, ycb as Yc

    for mk in (Bk.make_env, Yc.make_env):
        for bad in (dict(constants={"success_pause_range_s": (1.0, 1.0)}), dict(parameters={"simulation_params": {"object_groups": []}}),
                    dict(parameters={"object_scale_high": 0.5})):
            with pytest.raises(NotImplementedError):
                mk(batch_size=1, device="cpu", **bad)
        with pytest.raises(ValueError):
            mk(batch_size=1, device="cpu", parameters={"robot_control_params": {"tcp_solver_mode": "ik"}})
    # tcp_solver_mode mocap: built for the blocks world with the synchronous reset; the ycb sets and pipelined resets say so
    with pytest.raises(NotImplementedError):
        Yc.make_env(batch_size=1, device="cpu", parameters={"robot_control_params": {"tcp_solver_mode": "mocap"}})
    with pytest.raises(NotImplementedError):
        Bk.make_env(batch_size=1, device="cpu", parameters={"robot_control_params": {"tcp_solver_mode": "mocap"}}, pipelined_reset=True)
        with pytest.raises(ValueError):
            mk(batch_size=1, device="cpu", parameters={"robot_control_params": {"control_mode": "tcp+pitch"}})
    assert [Bk._control_mode_name(x) for x in ("joint", "tcp+roll+yaw", "tcp+wrist", "ControlMode.TCP_WRIST")] == ["joint", "tcp+roll+yaw", "tcp+wrist", "tcp+wrist"]


def test_crowded_table_placement_keeps_objects_apart():
    """`BatchedBlockRearrangeEnv._grid_placement` on the ycb object set, where the grid has fewer cells than objects (8 large meshes on the 0.61 x 0.58 m area):
    the fallback (place_objects_with_no_constraint, common/utils.py:829-880, vectorised over envs, largest object first) leaves no two bounding boxes overlapping
    and every box inside the placement area; the blocks set still takes the grid branch."""
    from robogym_amd.envs.rearrange import blocks as Bk
    from robogym_amd.envs.rearrange.xml import load_blocks_model, load_ycb_model, object_bounding_boxes

    for model, N, crowded in ((load_ycb_model(8), 8, True), (load_blocks_model(5), 5, False)):
        class Stub:
            pass
        e = Stub(); e.N = N
        bb = object_bounding_boxes(model, N); e.obj_center, e.obj_half = bb[:, :3], bb[:, 3:]
        A, gn = model.arrays, model.names["geom"]
        tb, tg = model.name2id("body", "table"), gn.index("table")
        e.table_pos, e.table_size, e.used_table_portion, e._rng = A["body_pos"][tb].copy(), A["geom_size"][tg].copy(), 1.0, np.random.RandomState(3)
        e._aabb_half = lambda yaw, e=e: Bk.BatchedBlockRearrangeEnv._aabb_half(e, yaw)
        e.placement_area = lambda e=e: Bk.BatchedBlockRearrangeEnv.placement_area(e)
        B = 256
        yaw = e._rng.uniform(0, 2 * np.pi, (B, N))
        half = e._aabb_half(yaw)
        tsx, tsy = 2 * e.table_size[0], 2 * e.table_size[1]
        width, height = 0.5 * tsx, 0.38 * tsy
        ncells = (width // (2 * half[:, :, 0].max(1))).astype(int) * (height // (2 * half[:, :, 1].max(1))).astype(int)
        assert bool((ncells < N).any()) == crowded
        out = Bk.BatchedBlockRearrangeEnv._grid_placement(e, yaw, np.arange(B))
        c, s_ = np.cos(yaw), np.sin(yaw)
        ctr = out[:, :, :2] + np.stack([c * e.obj_center[:, 0] - s_ * e.obj_center[:, 1], s_ * e.obj_center[:, 0] + c * e.obj_center[:, 1]], -1)
        for i in range(N):
            for j in range(i):
                assert not ((np.abs(ctr[:, i] - ctr[:, j]) < half[:, i, :2] + half[:, j, :2] - 1e-9).all(-1)).any()
        lo = e.table_pos[:2] - e.table_size[:2] + [0.5 * tsx - width / 2, 0.44 * tsy - height / 2]
        assert (ctr - half[:, :, :2] >= lo - 1e-9).all() and (ctr + half[:, :, :2] <= lo + [width, height] + 1e-9).all()
        assert np.allclose(out[:, :, 2] + e.obj_center[:, 2] - half[:, :, 2], e.table_pos[2] + e.table_size[2])      # every box stands on the table top


@pytest.mark.parametrize("portion,offset,size", [(1.0, (0.3038, 0.38275, 0.06648), (0.6075, 0.58178, 0.26)), (0.8, (0.3645, 0.44093, 0.06648), (0.486, 0.46542, 0.26)),
                                                 (0.6, (0.4253, 0.49911, 0.06648), (0.3645, 0.3491, 0.26)), (0.4, (0.486, 0.55728, 0.06648), (0.243, 0.23271, 0.26))])
def test_block_placement_area_table(portion, offset, size):
    """envs/rearrange/tests/test_placement.py:7-47, the reference's own numbers: the placement area of the blocks env (its default of ONE object) for four values
    of `used_table_portion`, to the reference's 1e-4; and the clip to a tenth of the table per object that `get_table_setting` applies (five blocks: never below 0.5)."""
    from robogym_amd.envs.rearrange import blocks as Bk
    from robogym_amd.envs.rearrange.xml import load_blocks_model

    model = load_blocks_model(5)

    class Stub:
        pass
    e = Stub()
    e.table_size, e.used_table_portion, e.N = model.arrays["geom_size"][model.names["geom"].index("table")].copy(), portion, 1
    off, sz = Bk.BatchedBlockRearrangeEnv.placement_area(e)
    assert np.allclose(off, offset, atol=1e-4) and np.allclose(sz, size, atol=1e-4)
    e.N = 5
    off5, sz5 = Bk.BatchedBlockRearrangeEnv.placement_area(e)
    assert np.isclose(sz5[0], 0.6075 * max(portion, 0.5), atol=1e-4)


def test_tcp_action_path_matches_reference_code():
    """Normalised action -> denormalised control -> wrist-constrained angles -> the quaternion difference handed to the mocap solver: the oracle's
    `denormalize` scaling and `tcp_quat_control` against tests/golden/rearrange_tcp.npz, i.e. the reference's own `FreeDOFTcpArm.denormalize_position_control /
    constrain_quat_ctrl` and `MocapSolver.get_tcp_quat` source run by tools/gen_golden_rearrange_tcp.py (96 samples, the wrist constraint binding in 18)."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rearrange_tcp.npz"))
    lo, hi = g["wrist_range"]
    for t in range(len(g["action"])):
        a, mpc = g["action"][t], float(g["mpc"][t])
        den = np.concatenate([a[:3] * mpc, a[3:5] * np.array([RO.SPEED_ROLL, RO.SPEED_PITCH]) * mpc])          # OracleRearrangeEnv.denormalize, the arm's five numbers
        assert np.abs(den - g["denorm"][t]).max() < 1e-15
        pos, dq = RO.tcp_quat_control(den, g["q"][t][5], lo, hi, g["gripper_quat"][t])
        assert np.abs(pos - g["denorm"][t][:3]).max() == 0 and np.abs(dq - g["dquat"][t]).max() < 1e-14
    assert abs(RO.JOINT_DRIFT_THRESHOLD - np.deg2rad(1)) < 1e-15
    # control_mode tcp+wrist: FreeWristTcpArm's class attributes through the same source, with MocapSolver.align_axis (tests/golden/rearrange_tcp_wrist.npz)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rearrange_tcp_wrist.npz"))
    for t in range(len(g["action"])):
        a, mpc = g["action"][t], float(g["mpc"][t])
        den = np.concatenate([a[:3] * mpc, a[3:4] * RO.SPEED_PITCH * mpc])
        assert np.abs(den - g["denorm"][t]).max() < 1e-15
        pos, dq = RO.tcp_quat_control(den, g["q"][t][5], lo, hi, g["gripper_quat"][t])
        assert np.abs(pos - g["denorm"][t][:3]).max() == 0 and np.abs(dq - g["dquat"][t]).max() < 1e-13, (t, dq, g["dquat"][t])


def test_solver_quaternion_response_holds_the_references_own_test():
    """robot/control/tcp/test/test_solver.py (test_zero_control, test_wrist_rotations, test_quat_second_dof_rotation) re-expressed on the oracle: at the arm's start
    pose, for rotations of +-30 ... 180 degrees in every controlled dimension of both TCP control modes, the quaternion difference handed to the mocap solver must give
    the rotation the test derives independently -- current orientation times the control, then, for tcp+wrist, the shortest rotation that puts the frame's
    most-vertical axis back onto the vertical (`calculate_quat_adjustment_due_to_axis_alignment`, :46-71) -- to the test's own 1e-5 on the Euler angles of the
    difference."""
    from robogym_amd.envs.rearrange.xml import load_solver_model

    ora = RO.OracleArmSim(load_solver_model(), 40)
    ora.sim.qpos[ora.arm_q] = RO.TABLETOP_EXPERIMENT_INITIAL_POS
    ora.sim.forward()
    cur = np.array(ora.body_xquat(ora.tcp_body), dtype=float)
    cur = RO.euler2quat(RO.quat2euler(cur))                    # (the test goes through tcp_rot(): Euler angles of the TCP)

    def rot_vec(q, v):
        return RO.quat_mul(q, RO.quat_mul(np.array([0.0, *v]), RO.quat_conjugate(q)))[1:]

    def adjustment(from_quat, axis):
        t = np.zeros(3); t[axis] = 1.0
        c = rot_vec(from_quat, t)
        d = float(np.dot(c, t))
        t = t * np.sign(d)
        ang, ax = np.arccos(abs(d)), np.cross(c, t)
        if np.linalg.norm(ax) < 1e-12:
            return np.array([1.0, 0, 0, 0])
        ax = ax / np.linalg.norm(ax)
        return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])

    n = 0
    for dims, align in (((2,), 2), ((0, 2), None)):            # FreeWristTcpArm: DOF_DIMS = [PITCH (value 2)], ALIGN_AXIS = PITCH; FreeRollYawTcpArm: [ROLL, PITCH]
        for which in range(len(dims)):
            for deg in (0, 30, 60, 90, 160, 180):
                for sign in (1.0, -1.0):
                    ctrl = np.zeros(len(dims)); ctrl[which] = sign * np.deg2rad(deg)
                    eul = np.zeros(3)
                    for k, d in enumerate(dims):
                        eul[d] = ctrl[k]
                    target = RO.quat_mul(cur, RO.euler2quat(eul))
                    if align is not None:
                        target = RO.quat_mul(adjustment(target, align), target)
                    _, dq = RO.tcp_quat_control(np.concatenate([np.zeros(3), ctrl]), 0.0, -1e9, 1e9, cur)      # (wide joint range: get_tcp_quat alone, as the test calls it)
                    diff = RO.quat2euler(RO.quat_normalize(RO.quat_mul(cur + (target - cur), RO.quat_conjugate(cur + dq))))      # rotation.quat_difference
                    assert np.allclose(diff, 0.0, atol=1e-5), (dims, which, deg, sign, diff)
                    n += 1
    assert n == 36


def test_observation_keys_and_order_are_the_reference_methods():
    """The packed observation row of the batched env (robogym_amd/envs/rearrange/blocks.py OBS_KEYS, written by ra_post_step_kernel) and the oracle env's
    observation have the keys of `RearrangeEnv._observe_simple` in its order: tests/golden/rearrange_obs_keys.json is what executing that method's own source on a
    recording stub returns (tools/gen_golden_rearrange_obs_keys.py)."""
    import json
    import os

    from robogym_amd.envs.rearrange.blocks import OBS_KEYS

    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rearrange_obs_keys.json")))
    assert [k for k, _ in OBS_KEYS] == [k for k, _ in ref] and len(ref) == 24
    src = dict(ref)
    assert src["qpos"] == "sim.qpos" and src["gripper_pos"].endswith("tcp_xyz()") and src["rel_goal_obj_rot"] == "goal[rel_goal_obj_rot]"
    widths = dict(OBS_KEYS)      # per-object keys are per-object in the row too
    assert all(str(widths[k]).startswith("N") for k in ("obj_pos", "obj_rel_pos", "obj_vel_pos", "obj_rot", "obj_vel_rot", "goal_obj_pos", "goal_obj_rot", "rel_goal_obj_pos",
                                                         "rel_goal_obj_rot", "obj_gripper_contact", "obj_bbox_size", "obj_colors"))
