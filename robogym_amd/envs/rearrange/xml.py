"""MJCF assembly of the rearrange worlds (UR16e + 2f-85 gripper + table + N objects) through the same `MujocoXML`
edit calls the reference makes.

Reference call sites restated here:
* `ArmSimulationInterface.make_world_xml / make_robot_xml / build`
  (/root/reference/robogym/robot/ur16e/mujoco/simulation/base.py:60-115): base.xml, timestep and `<size>` overrides,
  joint-actuated arm = mocap weld removed + `jointspec/calibrations/<dir>/{ur16e_ik_class,joint_actuations}.xml`,
  mocap-actuated arm = `jointspec/ur16e_mocap_class.xml`; gripper actuators appended last.
* `RearrangeSimulationInterface.make_xml / make_world_xml`
  (/root/reference/robogym/envs/rearrange/simulation/base.py:258-315): sizes njmax 2000 / nconmax 500, the
  (object, target) XML pairs appended in object order with the object group's material arguments.
* `make_block / make_target / make_blocks_and_targets` (/root/reference/robogym/envs/rearrange/common/utils.py:195-291).
* `build_solver_sim` (/root/reference/robogym/robot/composite/ur_gripper_arm.py:143-160): the arm-only world with the
  mocap weld that turns TCP commands into joint targets (sizes 200 / 200 / 200).
* materials: /root/reference/robogym/envs/rearrange/materials/default.jsonnet (the reference's default
  `material_names = ["default"]`, envs/rearrange/common/base.py:210).
"""
import numpy as np

from robogym_amd.mujoco.mujoco_xml import MujocoXML

BASE_XML = "robot/ur16e/base.xml"

#: envs/rearrange/materials/default.jsonnet
DEFAULT_MATERIAL = {"geom": {"condim": "6", "margin": 0.00005}, "joint": {"damping": "0.01", "armature": "0.001"}}


def make_world_xml(mujoco_timestep: float, contact_params: dict) -> MujocoXML:
    xml = MujocoXML.parse(BASE_XML).set_objects_attr(tag="option", timestep=mujoco_timestep)
    if contact_params:
        xml.set_objects_attr(tag="size", **contact_params)
    return xml.add_default_compiler_directive()


def make_robot_xml(xml: MujocoXML, joint_actuated: bool, arm_joint_calibration_path: str = "cascaded_pi") -> MujocoXML:
    if joint_actuated:
        xml.remove_objects_by_name("mocap_weld")
        sub = "robot/ur16e/jointspec/calibrations/%s" % arm_joint_calibration_path
        xml.append(MujocoXML.parse(sub + "/ur16e_ik_class.xml"))
        xml.append(MujocoXML.parse(sub + "/joint_actuations.xml"))
    else:
        xml.append(MujocoXML.parse("robot/ur16e/jointspec/ur16e_mocap_class.xml"))
    xml.append(MujocoXML.parse("robot/ur16e/gripper_actuators.xml"))
    return xml


def make_block(name: str, object_size) -> MujocoXML:
    src = """
    <mujoco>
      <worldbody>
        <body name="%s" pos="0.0 0.0 0.0">
          <geom type="box" rgba="0.0 0.0 0.0 0.0" material="block_mat"/>
          <joint name="%s:joint" type="free"/>
        </body>
      </worldbody>
    </mujoco>
    """ % (name, name)
    return MujocoXML.from_string(src).set_objects_attr(tag="geom", size=np.asarray(object_size, dtype=float))


def make_target(xml: MujocoXML) -> MujocoXML:
    import copy

    t = MujocoXML(copy.deepcopy(xml.root_element))
    return (t.remove_objects_by_tag("joint").add_name_prefix("target:", exclude_attribs=["material", "mesh", "class"])
            .set_objects_attr(tag="geom", contype=0, conaffinity=0))


def set_objects_attrs(xml: MujocoXML, tag_args: dict) -> MujocoXML:
    for tag, args in tag_args.items():
        xml.set_objects_attr(tag=tag, **args)
    return xml


def build_blocks_xml(num_objects: int = 5, object_size: float = 0.0254, mujoco_timestep: float = 0.001,
                     joint_actuated: bool = True, material: dict = DEFAULT_MATERIAL) -> MujocoXML:
    """BlockRearrangeSim.build (simulation/blocks.py:27-33 + simulation/base.py:236-300), default parameters."""
    xml = make_world_xml(mujoco_timestep, dict(njmax=2000, nconmax=500, nuserdata=2000, nuser_actuator=16))
    size = np.tile(float(object_size), 3)
    for i in range(num_objects):
        obj = make_block("object%d" % i, size.copy())
        tgt = make_target(obj)
        set_objects_attrs(obj, material)
        xml.append(obj)
        xml.append(tgt)
    return make_robot_xml(xml, joint_actuated)


def build_solver_xml(mujoco_timestep: float = 0.001) -> MujocoXML:
    """The controller arm's own simulation: ArmSimulationInterface.build with tcp_solver_mode = mocap."""
    xml = make_world_xml(mujoco_timestep, dict(njmax=200, nconmax=200, nuserdata=200))
    return make_robot_xml(xml, joint_actuated=False)


# ----------------------------------------------------------------------------------------- compiled models (what ships)
import os  # noqa: E402

from robogym_amd.mujoco.mjcf_compiler import CompiledModel  # noqa: E402

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "models")


def load_blocks_model(num_objects: int = 5, recompile: bool = False) -> CompiledModel:
    """The main world of rearrange/blocks with `num_objects` blocks (BASELINE.json configs[3]: num_objects = 5)."""
    path = os.path.join(MODEL_DIR, "rearrange_blocks%d.npz" % num_objects)
    if not recompile and os.path.exists(path):
        return CompiledModel.load(path)
    return build_blocks_xml(num_objects).build()


def load_solver_model(recompile: bool = False) -> CompiledModel:
    """The TCP solver's own world (arm + gripper, mocap weld)."""
    path = os.path.join(MODEL_DIR, "ur16e_solver.npz")
    if not recompile and os.path.exists(path):
        return CompiledModel.load(path)
    return build_solver_xml().build()
