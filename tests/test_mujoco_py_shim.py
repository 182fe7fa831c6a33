"""The `mujoco_py`-shaped B = 1 seam (robogym_amd/mujoco_py_shim.py): names / shapes / address helpers of
PyMjModel, in-place `sim.data` arrays, step / forward / reset / get_state / set_state, writable vs read-only model fields,
`load_model_from_xml`, all against the oracle; on the GPU a port of the reference's test_mujoco_move_hand written against
this surface."""
import numpy as np
import pytest

BOX_ON_PLANE = """
<mujoco>
  <compiler angle="radian" coordinate="local"/>
  <option timestep="0.004"/>
  <worldbody>
    <geom name="floor" type="plane" size="1 1 0.1" pos="0 0 0"/>
    <body name="slider" pos="0 0 0.3">
      <joint name="lift" type="slide" axis="0 0 1" damping="0.5"/>
      <joint name="swing" type="hinge" axis="0 1 0" damping="0.1"/>
      <geom name="arm" type="capsule" fromto="0 0 0 0.2 0 0" size="0.02" density="800"/>
      <site name="tip" pos="0.2 0 0"/>
      <body name="puck" pos="0.2 0 -0.05">
        <joint name="free" type="free"/>
        <geom name="puck" type="box" size="0.03 0.03 0.03" density="500"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def _oracle(compiled):
    from oracle.rg_oracle import OracleSim
    from robogym_amd.mujoco.model_blob import pack_model

    return OracleSim(pack_model(compiled))


def test_model_surface_and_step_against_oracle(locked_model, emul_lib, oracle_lib):
    from robogym_amd import mujoco_py_shim as mujoco_py

    oracle_lib.set_kernel_variant(False)
    model = mujoco_py.PyMjModel(locked_model)
    sim = mujoco_py.MjSim(model, nsubsteps=2, lib=emul_lib)
    assert (model.nq, model.nv, model.nu, model.nbody) == (38, 36, 20, 31)
    assert model.body_name2id("cube:middle") == locked_model.names["body"].index("cube:middle") and model.site_id2name(0) == "cube:center"
    assert model.get_joint_qpos_addr("cube:cube_rot") == (3, 7) and isinstance(model.get_joint_qpos_addr("robot0:WRJ1"), int)
    assert model.get_joint_qvel_addr("cube:cube_rot") == (3, 6) and model.actuator_gainprm.shape == (20, 10)
    with pytest.raises(ValueError):
        model.geom_name2id("no such geom")
    with pytest.raises(ValueError):           # numpy: assignment destination is read-only
        model.geom_size[0] = 1.0
    ora = _oracle(locked_model)
    qpos_ref = sim.data.qpos                    # the arrays are updated in place, like mujoco_py's views of mjData
    np.testing.assert_allclose(sim.data.qpos, ora.qpos, atol=1e-7)
    ctrl = np.clip(0.5 * (locked_model.actuator_ctrlrange[:, 0] + locked_model.actuator_ctrlrange[:, 1]) + 0.1, locked_model.actuator_ctrlrange[:, 0], locked_model.actuator_ctrlrange[:, 1])
    sim.data.ctrl[:] = ctrl; ora.ctrl[:] = ctrl
    for _ in range(3):
        sim.step()
        for _ in range(2):
            ora.step()
    assert qpos_ref is sim.data.qpos and abs(sim.data.time - 6 * 0.008) < 1e-6
    np.testing.assert_allclose(sim.data.qpos, ora.qpos, atol=2e-5)
    np.testing.assert_allclose(sim.data.qvel, ora.qvel, atol=5e-3)
    ora.fwd_position()
    np.testing.assert_allclose(sim.data.get_site_xpos("robot0:S_fftip"), ora.site_xpos.reshape(-1, 3)[model.site_name2id("robot0:S_fftip")], atol=2e-5)
    for _ in range(14):                         # the cube lands on the palm
        sim.step()
    assert sim.data.ncon == len(sim.data.contact) and sim.data.ncon >= 1 and sim.data.contact[0].dist < 0.01
    # state round trip, then a writable model field: gravity off -> the cube no longer presses on the palm
    st = sim.get_state()
    sim.model.opt.gravity[:] = 0.0
    sim.data.xfrc_applied[model.body_name2id("cube:middle"), 2] = 0.5      # and a lifting force on the cube
    for _ in range(8):
        sim.step()
    z_up = sim.data.get_joint_qpos("cube:cube_tz")
    sim.model.opt.gravity[:] = [0, 0, -9.81]; sim.data.xfrc_applied[:] = 0
    sim.set_state(st)
    for _ in range(8):
        sim.step()
    assert z_up > sim.data.get_joint_qpos("cube:cube_tz") + 1e-3
    sim.forward()
    assert sim.data.sensordata.shape == (5,) and (sim.data.sensordata >= 0).all()
    sim.reset()
    np.testing.assert_allclose(sim.data.qpos, locked_model.qpos0, atol=1e-7)
    assert sim.data.time == 0.0 and not sim.data.qvel.any()


def test_load_model_from_xml_matches_oracle(emul_lib, oracle_lib):
    """MJCF text -> model -> MjSim: a slider + hinge arm carrying a free box that drops onto the floor plane."""
    from robogym_amd import mujoco_py_shim as mujoco_py

    oracle_lib.set_kernel_variant(False)
    model = mujoco_py.load_model_from_xml(BOX_ON_PLANE)
    assert model.joint_names == ("lift", "swing", "free") and model.nq == 9 and model.nv == 8 and abs(model.opt.timestep - 0.004) < 1e-12
    sim = mujoco_py.MjSim(model, nsubsteps=5, lib=emul_lib)
    ora = _oracle(model._compiled)
    for k in range(26):
        sim.step()
        for _ in range(5):
            ora.step()
        if k == 3:
            np.testing.assert_allclose(sim.data.qpos, ora.qpos, atol=5e-5)     # 20 free-running fp32 substeps
    assert sim.data.ncon >= 1                     # the box has landed
    np.testing.assert_allclose(sim.data.qpos[:2], ora.qpos[:2], atol=2e-3)
    np.testing.assert_allclose(sim.data.get_site_xpos("tip"), ora.site_xpos.reshape(-1, 3)[0], atol=2e-3)
    assert abs(sim.data.get_body_xpos("puck")[2] - 0.03) < 5e-3 or sim.data.get_body_xpos("puck")[2] > 0.02


@pytest.mark.gpu
def test_move_hand_through_the_shim_gpu():
    """Port of the reference's test_mujoco_move_hand (robot/shadow_hand/test/test_mujoco_hand.py:44-75) against the mujoco_py
    surface: one actuator at a time is sent to a random target inside its control range (the rest of its group spread out of
    the way is omitted: targets stay near the zero pose), 100 sim.step() of 10 substeps, every actuator within 7.5 degrees."""
    from robogym_amd import mujoco_py_shim as mujoco_py
    from robogym_amd.envs.dactyl.locked import position_to_control_matrix
    from robogym_amd.envs.dactyl.reach import load_reach_model

    compiled = load_reach_model()
    model = mujoco_py.PyMjModel(compiled)
    sim = mujoco_py.MjSim(model, nsubsteps=10)
    mujoco_py.cymj.set_pid_control(sim.model, sim.data)
    P = position_to_control_matrix(compiled)
    lo, hi = model.actuator_ctrlrange[:, 0], model.actuator_ctrlrange[:, 1]
    hand_q = [model.get_joint_qpos_addr(n) for n in model.joint_names if n.startswith("robot0:")]
    rng = np.random.RandomState(0)
    zero = np.clip(np.zeros(model.nu), lo, hi)
    for name in ("robot0:A_WRJ1", "robot0:A_FFJ2", "robot0:A_MFJ1", "robot0:A_RFJ2", "robot0:A_LFJ4", "robot0:A_THJ4", "robot0:A_THJ1"):
        u = model.actuator_name2id(name)
        control = zero.copy()
        control[u] = lo[u] + rng.uniform(0.0, 1.0) * (hi[u] - lo[u])
        sim.data.ctrl[:] = control
        for _ in range(100):
            sim.step()
        observed = P @ sim.data.qpos[hand_q]
        assert np.rad2deg(np.abs(observed - control)).max() < 7.5, (name, np.rad2deg(np.abs(observed - control)))


# ---------------------------------------------------------------------------------------------------------------------------
# the KERNEL against closed forms (no oracle in the loop): Coulomb's law on a slope, the period of a compound pendulum
SLOPE = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="0.002" gravity="{gx} 0 {gz}"/>
  <worldbody>
    <geom name="floor" type="plane" size="2 2 1" condim="3" friction="{mu} 0.005 0.0001"/>
    <body name="box" pos="0 0 0.0195">
      <joint name="free" type="free"/>
      <geom name="box" type="box" size="0.05 0.05 0.02" density="600" condim="3" friction="{mu} 0.005 0.0001"/>
    </body>
  </worldbody>
</mujoco>
"""
PENDULUM = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="0.0005" gravity="0 0 -9.81"/>
  <worldbody>
    <body name="rod" pos="0 0 1">
      <joint name="hinge" type="hinge" axis="0 1 0"/>
      <geom name="rod" type="box" size="0.01 0.02 0.15" pos="0 0 -0.15" density="1000"/>
    </body>
  </worldbody>
</mujoco>
"""


def _slope(mujoco_py, kw, mu, tan_theta, T):
    g, th = 9.81, np.arctan(tan_theta)
    sim = mujoco_py.MjSim(mujoco_py.load_model_from_xml(SLOPE.format(gx=g * np.sin(th), gz=-g * np.cos(th), mu=mu)), nsubsteps=int(round(T / 0.002)), **kw)
    sim.step()
    return float(sim.data.qvel[0]), g * (np.sin(th) - mu * np.cos(th)) * T, g * np.sin(th) * T


def _pendulum_period(mujoco_py, kw):
    sim = mujoco_py.MjSim(mujoco_py.load_model_from_xml(PENDULUM), nsubsteps=1, **kw)
    sim.data.qpos[0] = 0.02
    crossings, prev, t = [], 0.02, 0.0
    while len(crossings) < 2 and t < 3.0:
        sim.step(); t += 0.0005
        q = float(sim.data.qpos[0])
        if prev > 0 >= q:
            crossings.append(t - 0.0005 * (0 - q) / (prev - q))
        prev = q
    a, b, c, l = 0.01, 0.02, 0.15, 0.15
    mass = 1000 * 8 * a * b * c
    return crossings[1] - crossings[0], 2 * np.pi * np.sqrt((mass * ((2 * a) ** 2 + (2 * c) ** 2) / 12 + mass * l * l) / (mass * 9.81 * l))


def test_kernel_obeys_coulomb_friction_emul(emul_lib):
    from robogym_amd import mujoco_py_shim as mujoco_py

    vx, sliding, free = _slope(mujoco_py, dict(lib=emul_lib), 0.3, 1.4 * 0.3, 0.2)
    assert abs(vx - sliding) < 0.06 * sliding
    vx, _, free = _slope(mujoco_py, dict(lib=emul_lib), 0.3, 0.7 * 0.3, 0.2)
    assert abs(vx) < 0.01 * free


@pytest.mark.gpu
def test_kernel_obeys_closed_forms_gpu():
    """The HIP kernel itself (fp32, MI355X) against physics closed forms, no oracle involved: a = g (sin theta - mu cos theta)
    above the friction angle and rest below it, for two friction coefficients; T = 2 pi sqrt(I / (m g l)) of a compound pendulum
    to 0.2 %."""
    from robogym_amd import mujoco_py_shim as mujoco_py

    for mu in (0.3, 0.8):
        vx, sliding, free = _slope(mujoco_py, {}, mu, 1.4 * mu, 0.6)
        assert abs(vx - sliding) < 0.06 * sliding, (mu, vx, sliding)
        vx, _, free = _slope(mujoco_py, {}, mu, 0.7 * mu, 0.6)
        assert abs(vx) < 0.01 * free, (mu, vx)
    got, want = _pendulum_period(mujoco_py, {})
    assert abs(got - want) < 2e-3 * want, (got, want)


@pytest.mark.gpu
def test_kernel_resting_depth_matches_the_documented_soft_constraint_law_gpu():
    """The HIP kernel against the closed form of MuJoCo's documented soft-constraint law (tests/test_oracle.py:
    rest_depth_closed_form): a frictionless sphere at rest on a plane, three (solref, solimp) settings."""
    from robogym_amd import mujoco_py_shim as mujoco_py
    from tests.test_oracle import REST, REST_CASES, rest_depth_closed_form

    for case in REST_CASES:
        sim = mujoco_py.MjSim(mujoco_py.load_model_from_xml(REST.format(tc=case[0], dr=case[1], d0=case[2], d1=case[3], w=case[4])), nsubsteps=4000)
        sim.step()
        assert abs(sim.data.qvel[2]) < 1e-5
        np.testing.assert_allclose(sim.data.qpos[2] - 0.05, rest_depth_closed_form(*case), rtol=2e-3)


def _damped_wheel(mujoco_py, kw, n):
    from tests.test_oracle import DAMPED, damped_wheel_closed_form

    sim = mujoco_py.MjSim(mujoco_py.load_model_from_xml(DAMPED), nsubsteps=n, **kw)
    sim.data.qvel[0] = 3.0
    sim.step()
    return float(sim.data.qvel[0]), damped_wheel_closed_form(n)


def test_kernel_implicit_damping_emul(emul_lib):
    from robogym_amd import mujoco_py_shim as mujoco_py

    got, want = _damped_wheel(mujoco_py, dict(lib=emul_lib), 50)
    assert abs(got - want) < 2e-5 * want


@pytest.mark.gpu
def test_kernel_implicit_damping_gpu():
    """w' = w I / (I + h b) per step (mj_Euler's implicit joint damping), 250 steps, on the MI355X."""
    from robogym_amd import mujoco_py_shim as mujoco_py

    got, want = _damped_wheel(mujoco_py, {}, 250)
    assert abs(got - want) < 1e-4 * want, (got, want)
