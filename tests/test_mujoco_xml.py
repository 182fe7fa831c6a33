"""The reference's tests of its MJCF assembly helper (`MujocoXML`: /root/reference/robogym/mujoco/test/test_mujoco_utils.py:40-146,182-197), restated on
`robogym_amd.mujoco.mujoco_xml.MujocoXML` + the MJCF compiler at the level of the compiled model (names, addresses, body positions): composition of prefixed
copies, attribute setting with mixed magnitudes, removal by tag, the serialised form."""
import numpy as np

from robogym_amd.mujoco.mujoco_xml import MujocoXML

def _doc(*bodies):
    return "<mujoco><worldbody>%s</worldbody></mujoco>" % "".join(bodies)


# the reference tests' two little worlds: a free ball; a hinge arm with a sliding forearm one unit along x
XML_BALL = _doc('<body name="ball"><freejoint name="ball_joint"/><geom name="sphere" type="sphere" size="0.1 0.1 0.1" pos="0.00 0.00 0.00"/></body>')
XML_ARM = _doc('<body name="arm"><joint name="hinge_joint" type="hinge" axis="0 0 1"/><geom name="sphere" type="sphere" size="0.1 0.1 0.1" pos="0.00 0.00 0.00"/>'
               '<body name="forearm" pos="1 0 0"><joint name="slide_joint" type="slide" axis="1 0 0"/><geom name="box" type="box" size="0.1 0.1 0.1" pos="0.00 0.00 0.00"/></body></body>')


def _group(model, prefix):
    """SimulationInterface.register_joint_group (simulation_interface.py:100-136): the qpos / qvel addresses of the joints whose name starts with `prefix`"""
    A, names = model.arrays, model.names["joint"]
    width = {0: (7, 6), 1: (4, 3), 2: (1, 1), 3: (1, 1)}      # free, ball, slide, hinge
    qa, va = [], []
    for j, n in enumerate(names):
        if n.startswith(prefix):
            nq, nv = width[int(A["jnt_type"][j])]
            qa += list(range(int(A["jnt_qposadr"][j]), int(A["jnt_qposadr"][j]) + nq)); va += list(range(int(A["jnt_dofadr"][j]), int(A["jnt_dofadr"][j]) + nv))
    return qa, va


def test_simple_mujoco_setup():
    """:40-80: two prefixed copies of a free ball, each moved by `set_named_objects_attr`, appended to an empty document"""
    one = MujocoXML.from_string(XML_BALL).add_name_prefix("ball_one:").set_named_objects_attr("ball_one:ball", pos=[1, 0, 0])
    two = MujocoXML.from_string(XML_BALL).add_name_prefix("ball_two:").set_named_objects_attr("ball_two:ball", pos=[-1, 0, 0])
    m = MujocoXML().add_default_compiler_directive().append(one).append(two).build()
    assert m.names["joint"] == ["ball_one:ball_joint", "ball_two:ball_joint"] and m.names["body"][1:] == ["ball_one:ball", "ball_two:ball"]
    for prefix, x in (("ball_one:ball_joint", 1.0), ("ball_two:ball_joint", -1.0)):
        qa, va = _group(m, prefix)
        assert len(qa) == 7 and len(va) == 6
        assert np.allclose(m.arrays["qpos0"][qa], [x, 0, 0, 1, 0, 0, 0])           # a free joint's qpos0 is the body's pose
    assert int(m.arrays["dims"][0]) == 14 and int(m.arrays["dims"][1]) == 12


def test_more_complex_mujoco_setup():
    """:83-123: two prefixed two-joint arms; joint groups by prefix select the right addresses"""
    xml = MujocoXML().add_default_compiler_directive()
    for p in ("arm_one:", "arm_two:"):
        xml.append(MujocoXML.from_string(XML_ARM).add_name_prefix(p))
    m = xml.build()
    assert m.names["joint"] == ["arm_one:hinge_joint", "arm_one:slide_joint", "arm_two:hinge_joint", "arm_two:slide_joint"]
    for p in ("arm_one:", "arm_two:"):
        assert [len(x) for x in _group(m, p)] == [2, 2] and [len(x) for x in _group(m, p + "hinge_joint")] == [1, 1]
    assert _group(m, "arm_one:hinge_joint")[0] == [0] and _group(m, "arm_two:hinge_joint")[0] == [2] and _group(m, "arm_two:")[0] == [2, 3]


def test_set_attributes_mixed_precision():
    """:126-145: values of very different magnitude survive the attribute formatting (relative error, as the reference measures it)"""
    m = MujocoXML().add_default_compiler_directive().append(MujocoXML.from_string(XML_BALL).set_named_objects_attr("ball", pos=[1, 1e-8, 1e-12])).build()
    pos = m.arrays["body_pos"][m.name2id("body", "ball")]
    assert np.linalg.norm(pos / np.array([1, 1e-8, 1e-12]) - 1) < 1e-6


def test_remove_elem():
    """:182-197: removal by tag, and the serialised document"""
    ball = MujocoXML.from_string(XML_BALL).remove_objects_by_tag("freejoint")
    ref = _doc('<body name="ball"><geom name="sphere" pos="0.00 0.00 0.00" size="0.1 0.1 0.1" type="sphere"/></body>')
    import xml.etree.ElementTree as et

    # (the reference compares strings, which holds on the Python it pins -- ElementTree sorted attributes alphabetically up to 3.7; the canonical forms are compared here)
    assert et.canonicalize(ref, strip_text=True) == et.canonicalize(ball.xml_string(), strip_text=True)
    assert [e.tag for e in ball.root_element.iter()] == ["mujoco", "worldbody", "body", "geom"]
    m = MujocoXML().add_default_compiler_directive().append(ball).build()
    assert m.names["joint"] == [] and int(m.arrays["dims"][0]) == 0
