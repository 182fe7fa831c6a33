"""Pin the oracle and the host logic against the committed golden fixtures (tests/golden/*.npz), which
tools/gen_golden.py produced by running the real reference modules (the pure-numpy slices of the hot
path that import without MuJoCo)."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_rotation_helpers_match_reference():
    from oracle import env_oracle
    from robogym_amd.utils import rotation

    g = np.load(os.path.join(G, "rotation.npz"))
    q, p = torch.tensor(g["q"]), torch.tensor(g["p"])
    np.testing.assert_allclose(rotation.quat_mul(q, p).numpy(), g["quat_mul"], atol=1e-12)
    np.testing.assert_allclose(rotation.quat_difference(q, p).numpy(), g["quat_difference"], atol=1e-12)
    np.testing.assert_allclose(rotation.quat_magnitude(rotation.quat_difference(q, p)).numpy(), g["quat_magnitude"], atol=1e-9)
    np.testing.assert_allclose(rotation.quat_normalize(q).numpy(), g["quat_normalize"], atol=0)
    ours, ref = rotation.parallel_quats_np(), g["parallel_quats"]
    match = np.abs(ours @ ref.T) > 1 - 1e-12      # same rotation <=> |<q, p>| = 1
    assert ours.shape == ref.shape == (24, 4) and (match.sum(0) == 1).all() and (match.sum(1) == 1).all()   # the same SET of 24 rotations
    assert (ours[:, 0] >= 0).all() and np.allclose(np.linalg.norm(ours, axis=1), 1.0, atol=1e-15)
    for a, b, d, mg in zip(g["q"], g["p"], g["quat_difference"], g["quat_magnitude"]):
        np.testing.assert_allclose(env_oracle.quat_difference(a, b), d, atol=1e-12)
        assert abs(env_oracle.quat_magnitude(env_oracle.quat_difference(a, b)) - mg) < 1e-9


def test_hand_tables_and_action_map_match_reference(locked_model):
    """ctrl ranges: XML == table to 1e-8 (reference test_hand_interface.py:56-63); joint / actuator order
    (test_locked.py:17-52); the action -> ctrl map of robot_interface.py:247-278."""
    from oracle.env_oracle import OracleLockedEnvPhysics
    from robogym_amd.envs.dactyl.locked import position_to_control_matrix

    g = np.load(os.path.join(G, "hand_control.npz"))
    m = locked_model
    assert [n[len("robot0:"):] for n in m.names["joint"][8:]] == list(g["joints"])
    assert m.names["joint"][:8] == ["cube:cube_tx", "cube:cube_ty", "cube:cube_tz", "cube:cube_rot", "target:cube_tx", "target:cube_ty", "target:cube_tz", "target:cube_rot"]
    assert [n[len("robot0:"):] for n in m.names["actuator"]] == list(g["actuators"])
    np.testing.assert_allclose(m.actuator_ctrlrange[:, 0], g["ctrl_lo"], atol=1e-8)
    np.testing.assert_allclose(m.actuator_ctrlrange[:, 1], g["ctrl_hi"], atol=1e-8)
    np.testing.assert_array_equal(position_to_control_matrix(m), g["position_to_control"])
    ora = OracleLockedEnvPhysics(m)
    for a, qh, cr, ca in zip(g["actions"], g["hand_qpos"], g["ctrl_relative"], g["ctrl_absolute"]):
        ora.sim.qpos[ora.hand_q] = qh
        np.testing.assert_allclose(ora.denormalize(np.clip(a, -1, 1), True), cr, atol=1e-12)
        np.testing.assert_allclose(ora.denormalize(np.clip(a, -1, 1), False), ca, atol=1e-12)


def test_fingertip_observation_matches_reference_fk(locked_model):
    """The reference pins its numpy hand FK to MuJoCo at 1e-6 (test_mujoco_hand.py:19-41); the same FK
    (run on the real asset files) pins our model compiler + oracle kinematics + fingertip frame here."""
    from oracle.env_oracle import OracleLockedEnvPhysics

    g = np.load(os.path.join(G, "hand_fk.npz"))
    ora = OracleLockedEnvPhysics(locked_model)
    worst = 0.0
    for ang, tips in zip(g["joint_angles"], g["relative_fingertips"]):
        ora.sim.qpos[ora.hand_q] = ang
        ora.sim.fwd_position()
        row = ora.obs_row()
        worst = max(worst, np.abs(row[-15:] - tips.ravel()).max())
    assert worst < 1e-6, worst


def test_pendulum_forward_kinematics_matches_reference_fk(oracle_lib):
    """A second model through MJCF compiler -> oracle kinematics, against the reference's own numpy forward kinematics on its double-pendulum test model
    (tests/golden/pendulum_fk.npz from tools/gen_golden_pendulum_fk.py; the reference pins that FK to MuJoCo's site_xpos at 1e-6 on this very model,
    mujoco/test/test_mujoco_utils.py:148-179): site positions and the hinge2 joint anchor for 64 joint configurations, 1e-9.  (The XML's `integrator="RK4"` is
    taken out before compiling -- this stepper is Euler only, as every robogym world is -- which kinematics do not see.)"""
    from oracle.rg_oracle import OracleSim
    from robogym_amd.mujoco.model_blob import pack_model
    from robogym_amd.mujoco.mujoco_xml import MujocoXML

    x = MujocoXML.parse("test/inverted_pendulum/inverted_double_pendulum.xml").add_name_prefix("ivp:").add_default_compiler_directive()
    for opt in x.root_element.iter("option"):
        opt.attrib.pop("integrator", None)
    m = x.build()
    assert m.names["joint"] == ["ivp:hinge", "ivp:hinge2"] and int(m.arrays["dims"][0]) == 2
    s = OracleSim(pack_model(m))
    g = np.load(os.path.join(G, "pendulum_fk.npz"))
    sites = [m.names["site"].index(n) for n in g["site_names"]]
    j2 = m.names["joint"].index("ivp:hinge2")
    worst = 0.0
    for q, pos in zip(g["joint_angles"], g["positions"]):
        s.qpos[:] = q
        s.fwd_position()
        sx = s.field("site_xpos").reshape(-1, 3)
        worst = max(worst, np.abs(sx[sites] - pos[:2]).max(), np.abs(s.field("xanchor").reshape(-1, 3)[j2] - pos[-1]).max())
    assert worst < 1e-9, worst


def test_cube_mass_and_model_dims(locked_model):
    m = locked_model
    assert tuple(int(x) for x in m.dims[:3]) == (38, 36, 20)            # test_locked.py joint list: nq 38, nv 36, nu 20
    assert abs(m.body_subtreemass[m.name2id("body", "cube:middle")] - 0.078) < 1e-3   # test_locked.py:59-63


def test_batched_tracker_matches_reference_sequence():
    from robogym_amd.utils.multi_goal_tracker import BatchedMultiGoalTracker

    g = np.load(os.path.join(G, "tracker.npz"))
    tr = BatchedMultiGoalTracker(1, "cpu")
    one = torch.ones(1, dtype=torch.bool)
    tr.reset(one); tr.reset_goal_steps(one)
    for t in range(len(g["done"])):
        reward, done, new_goal, info = tr.process(torch.tensor([bool(g["is_successful"][t])]), torch.tensor([float(g["goal_distance_reward"][t])], dtype=torch.float32))
        tr.reset_goal_steps(new_goal)
        np.testing.assert_allclose(reward[0].numpy(), g["reward"][t], atol=1e-6)
        assert bool(done[0]) == bool(g["done"][t]), t
        assert int(info["successes_so_far"][0]) == int(g["successes_so_far"][t])
        assert int(info["steps_since_last_goal"][0]) == int(g["steps_since_last_goal"][t]), t
        assert bool(new_goal[0]) == bool(g["goal_reset"][t])
        if done[0]:
            tr.reset(one); tr.reset_goal_steps(one)


def test_oracle_tracker_matches_reference_sequence():
    """The scalar tracker restatement used as the env-level oracle (oracle/env_oracle.py) against the same
    reference-generated 1500-step sequence (a 400-step timeout, 50 successes -> trial success)."""
    from oracle.env_oracle import OracleMultiGoalTracker

    g = np.load(os.path.join(G, "tracker.npz"))
    tr = OracleMultiGoalTracker()
    tr.reset(); tr.reset_goal_steps()
    for t in range(len(g["done"])):
        reward, done, info = tr.process(bool(g["is_successful"][t]), float(g["goal_distance_reward"][t]), tr.reset_goal_steps)
        np.testing.assert_allclose(reward, g["reward"][t], atol=1e-12)
        assert done == bool(g["done"][t]), t
        assert info["successes_so_far"] == int(g["successes_so_far"][t])
        assert info["steps_since_last_goal"] == int(g["steps_since_last_goal"][t]), t
        assert info["goal_reset"] == bool(g["goal_reset"][t])
        if done:
            tr.reset(); tr.reset_goal_steps()
