"""dactyl/reach — the Shadow hand alone, reaching fingertip targets (BASELINE.json configs[0], the reference's
own CPU-runnable case).

What is built for this config is the physics path: model assembly exactly as `ReachSimulation.build`
(/root/reference/robogym/envs/dactyl/reach.py:79-143: floor, five target sites, the hand with its base joint
removed, 20 settling steps under the zero control), the batched simulation with the hand's action map, and the
observation quantities of `ReachEnv._default_observation_map` (:163-174) that the physics produces (hand joint
positions / velocities, absolute fingertip positions).  It serves as the second model through the same
compiler -> oracle -> kernel stack and as a parity case (tests/test_reach.py).  The reach goal generator
(a second physics simulation that samples collision-free hand poses, `FingertipPosGoal`) and the env
bookkeeping around it are not part of this round.
"""
import os

import numpy as np
import torch

from robogym_amd.envs.dactyl.locked import FINGERTIP_SITE_NAMES, MODEL_DIR, position_to_control_matrix
from robogym_amd.mujoco.mjcf_compiler import CompiledModel
from robogym_amd.mujoco.mujoco_xml import MujocoXML
from robogym_amd.mujoco.simulation_interface import BatchedSimulationInterface


def build_reach_xml() -> MujocoXML:
    """The merged MJCF document of dactyl/reach (needs the robogym asset tree)."""
    xml = MujocoXML()
    xml.add_default_compiler_directive()
    xml.append(MujocoXML.parse("floor/basic_floor.xml").set_named_objects_attr("floor", tag="body", pos=[1, 1, 0]))
    target = MujocoXML.parse("shadowhand_reach/target.xml")
    colors = [[1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 1.0], [0.0, 0.0, 1.0, 1.0], [1.0, 1.0, 0.0, 1.0], [1.0, 0.0, 1.0, 1.0]]
    for site, color in zip(FINGERTIP_SITE_NAMES, colors):
        target.set_named_objects_attr("target_%s" % site, pos=[0.5, 0.5, 0.0], type="sphere", rgba=color, size=0.005)
    xml.append(target)
    xml.append(
        MujocoXML.parse("robot/shadowhand/main.xml")
        .add_name_prefix("robot0:")
        .set_named_objects_attr("robot0:hand_mount", tag="body", pos=[1.0, 1.25, 0.15], euler=[np.pi / 2, 0, np.pi])
        .remove_objects_by_name("robot0:annotation:outer_bound")
        .remove_objects_by_name("robot0:hand_base")
    )
    return xml


def load_reach_model(recompile: bool = False) -> CompiledModel:
    path = os.path.join(MODEL_DIR, "dactyl_reach.npz")
    if not recompile and os.path.exists(path):
        return CompiledModel.load(path)
    return build_reach_xml().build()


class ReachSimulation(BatchedSimulationInterface):
    """Batched `ReachSimulation` (reach.py:59-143)."""

    def __init__(self, model: CompiledModel, batch_size: int, device="cuda:0", n_substeps: int = 10, relative_action: bool = True, lib=None):
        super().__init__(model, batch_size, device=device, n_substeps=n_substeps, lib=lib)
        m = model
        hand_joints = [n for n in m.names["joint"] if n.startswith("robot0:")]
        self.register_joint_group("hand_angle", hand_joints)
        hand_q = self.qpos_idxs["hand_angle"]
        assert (np.diff(hand_q) == 1).all()
        self.pos_to_ctrl = position_to_control_matrix(m)
        self.ctrl_lo = torch.tensor(m.arrays["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=self.device)
        self.ctrl_hi = torch.tensor(m.arrays["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=self.device)
        # the kernel's env descriptor: only the action map is meaningful here (no cube, no goal distance)
        ints = [int(hand_q[0]), len(hand_q), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0] + [m.name2id("site", "robot0:" + s) for s in FINGERTIP_SITE_NAMES]
        ints += [1 if relative_action else 0, 0, 0]
        self.set_env(ints, self.pos_to_ctrl, 0.0)
        self.tip_sites = [m.name2id("site", "robot0:" + s) for s in FINGERTIP_SITE_NAMES]

    def settle(self, nsteps: int = 20):
        """`ReachSimulation.build` tail (reach.py:131-141): zero control (range centres), 20 sim steps."""
        self.set_ctrl((0.5 * (self.ctrl_lo + self.ctrl_hi)).expand(self.batch_size, -1).contiguous())
        for _ in range(nsteps):
            self.step()
