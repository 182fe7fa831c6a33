"""tests/golden/pendulum_fk.npz: the reference's numpy forward kinematics (robogym/mujoco/forward_kinematics.py, which the reference's own test pins to MuJoCo's
site_xpos at 1e-6: mujoco/test/test_mujoco_utils.py:148-179) on its double-pendulum test model, for 64 joint configurations.  Imports /root/reference with
`mujoco_py` stubbed (the kinematics are pure numpy over the parsed XML); the fixture travels with the repository.

    python tools/gen_golden_pendulum_fk.py
"""
import os
import sys
import types

import numpy as np

np.float = float  # the reference predates numpy 1.24
stub = types.ModuleType("mujoco_py")
stub.MjSim = object; stub.MjSimState = object; stub.cymj = types.SimpleNamespace(); stub.const = types.SimpleNamespace(); stub.__path__ = []
gen = types.ModuleType("mujoco_py.generated"); gen.__path__ = []
const = types.ModuleType("mujoco_py.generated.const")
gen.const = const; stub.generated = gen
sys.modules["mujoco_py"] = stub; sys.modules["mujoco_py.generated"] = gen; sys.modules["mujoco_py.generated.const"] = const
sys.path.insert(0, "/root/reference")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from robogym.mujoco.forward_kinematics import ForwardKinematics
    from robogym.mujoco.mujoco_xml import MujocoXML

    mxml = MujocoXML.parse("test/inverted_pendulum/inverted_double_pendulum.xml").add_name_prefix("ivp:")
    joint_names = ["ivp:hinge", "ivp:hinge2"]
    site_names = ["ivp:hinge2_site", "ivp:tip"]
    kin = ForwardKinematics.prepare(mxml, "ivp:cart", np.zeros(3), np.zeros(3), site_names, joint_names)
    rng = np.random.RandomState(20200901)
    q = rng.uniform(-np.pi, np.pi, (64, 2))
    q[0] = 0
    out = np.array([kin.compute(a, return_joint_pos=True) for a in q])      # [64, 2 sites + 2 joint anchors, 3]
    np.savez(os.path.join(OUT, "pendulum_fk.npz"), joint_angles=q, positions=out, site_names=np.array(site_names), joint_names=np.array(joint_names))
    print("pendulum_fk.npz:", out.shape, "tip at zero angles", out[0, 1])


if __name__ == "__main__":
    main()
