"""dactyl/full_perpendicular (BASELINE.json configs[2]) — the Shadow hand with the full Rubik's cube: 26 cubelets on hinge
chains around a free-floating core (nq 170, nv 168, 135 bodies, 117 geoms of which 101 are mesh hulls, condim 6 on the
cubelets).

What exists for this config is the MODEL and the ORACLE side: assembly exactly as `FullPerpendicularSimulation`
(/root/reference/robogym/envs/dactyl/full_perpendicular.py:97-117 on top of cube_env.py:171-218: the perpendicular cube as
"cube:" and, with collisions off, as "target:", spring joints removed, floor, hand without its base joint) through the same
`MujocoXML` edit calls and the same MJCF compiler (incl. MuJoCo's legacy `.msh` mesh format), and the CPU oracle stepping
it (tests/test_full_perpendicular.py: the reference's cube-mass pin, the cube resting on the palm with its cubelets held
together).  The HIP kernel does NOT run this model: its per-env LDS layout assumes nv <= 36 / 30 constrained dofs with a
dense Hessian and pyramids of condim <= 4; 168 dofs and condim 6 need the tree-sparse-only solver layout of DESIGN.md §9.
`rg_model_create` refuses the model loudly (dimension limits), it is never silently narrowed.
"""
import os

import numpy as np

from robogym_amd.envs.dactyl.locked import MODEL_DIR
from robogym_amd.mujoco.mjcf_compiler import CompiledModel
from robogym_amd.mujoco.mujoco_xml import MujocoXML


def build_full_perpendicular_xml(cube_xml_path: str = "rubik/rubik_perpendicular.xml") -> MujocoXML:
    """The merged MJCF document of dactyl/full_perpendicular (needs the robogym asset tree)."""
    xml = MujocoXML()
    xml.add_default_compiler_directive()
    xml.append(
        MujocoXML.parse(cube_xml_path)
        .add_name_prefix("cube:")
        .set_named_objects_attr("cube:middle", tag="body", pos=[1.0, 0.87, 0.2])
        .remove_objects_by_prefix(prefix="cube:cubelet:spring:", tag="joint")        # "Delete springs for now"
    )
    xml.append(
        MujocoXML.parse(cube_xml_path)
        .add_name_prefix("target:")
        .set_named_objects_attr("target:middle", tag="body", pos=[1.0, 0.87, 0.2])
        .remove_objects_by_prefix(prefix="target:cubelet:spring:", tag="joint")
        .set_objects_attr(tag="geom", group="2", conaffinity="0", contype="0")
    )
    xml.append(MujocoXML.parse("floor/basic_floor.xml").set_named_objects_attr("floor", tag="body", pos=[1, 1, 0]))
    xml.append(
        MujocoXML.parse("robot/shadowhand/main.xml")
        .add_name_prefix("robot0:")
        .set_objects_attr(tag="size", njmax=2000, nconmax=200, nuserdata=100, nuser_actuator=20)     # cube_env.py:239-242, applied at :199
        .set_named_objects_attr("robot0:hand_mount", tag="body", pos=[1.0, 1.25, 0.15], euler=[np.pi / 2, 0, np.pi])
        .remove_objects_by_name("robot0:annotation:outer_bound")
        .remove_objects_by_name("robot0:hand_base")
    )
    return xml


def load_full_perpendicular_model(recompile: bool = False) -> CompiledModel:
    path = os.path.join(MODEL_DIR, "dactyl_full_perpendicular.npz")
    if not recompile and os.path.exists(path):
        return CompiledModel.load(path)
    return build_full_perpendicular_xml().build()
