"""Generate tests/golden/*.npz by importing the parts of the real reference that run in this
container (pure numpy once `mujoco_py` is stubbed): rotation utilities, the Shadow-hand
control tables and action map, the numpy hand forward kinematics, LockedParallelGoal's distance and
MultiGoalTracker.  Needs /root/reference; the resulting fixtures travel with the repository.

    python tools/gen_golden.py
"""
import os
import sys
import types

import numpy as np

np.float = float  # the reference predates numpy 1.24
stub = types.ModuleType("mujoco_py")
stub.MjSim = object
stub.MjSimState = object
stub.cymj = types.SimpleNamespace()
stub.const = types.SimpleNamespace()
stub.__path__ = []
gen = types.ModuleType("mujoco_py.generated"); gen.__path__ = []
const = types.ModuleType("mujoco_py.generated.const")
gen.const = const; stub.generated = gen
sys.modules["mujoco_py"] = stub
sys.modules["mujoco_py.generated"] = gen
sys.modules["mujoco_py.generated.const"] = const
sys.path.insert(0, "/root/reference")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(20200901)
    from robogym.utils import rotation

    # ---- rotation.py
    q = rng.randn(64, 4); q /= np.linalg.norm(q, axis=1, keepdims=True)
    p = rng.randn(64, 4); p /= np.linalg.norm(p, axis=1, keepdims=True)
    diff = np.array([rotation.quat_difference(a, b) for a, b in zip(q, p)])
    mag = np.array([rotation.quat_magnitude(d) for d in diff])
    from robogym.envs.dactyl.common import cube_utils
    np.savez(os.path.join(OUT, "rotation.npz"), q=q, p=p, quat_mul=rotation.quat_mul(q, p), quat_difference=diff, quat_magnitude=mag,
             quat_normalize=rotation.quat_normalize(q.copy()), parallel_quats=np.array(cube_utils.PARALLEL_QUATS))

    # ---- hand tables + action map (robot_interface.py:220-278, hand_interface.py)
    from robogym.robot.shadow_hand import hand_interface as hi

    lo = np.array([hi.ACTUATOR_CTRLRANGE[a][0] for a in hi.ACTUATORS])
    up = np.array([hi.ACTUATOR_CTRLRANGE[a][1] for a in hi.ACTUATORS])

    class FakeHand(hi.Hand):
        def __init__(self):
            self.qpos = np.zeros(24)
        def actuator_ctrl_range_upper_bound(self): return up
        def actuator_ctrl_range_lower_bound(self): return lo
        def get_current_position(self): return self.qpos
        def get_name(self): return "fake"
        def observe(self): raise NotImplementedError
        def set_position_control(self, c): pass
        def set_effort_control(self, c): pass
        def set_joint_control_mode(self, j, m): pass
        def zero_effort_control(self): return np.zeros(20)
    for name in list(getattr(FakeHand, "__abstractmethods__", [])):
        setattr(FakeHand, name, lambda self, *a, **k: None)
    FakeHand.__abstractmethods__ = frozenset()
    hand = FakeHand()
    actions = rng.uniform(-1.3, 1.3, (32, 20))
    qh = rng.uniform(-0.5, 1.5, (32, 24))
    rel, absl = [], []
    for a, qq in zip(actions, qh):
        hand.qpos = qq
        a = np.clip(a, -1, 1)
        rel.append(hand.denormalize_position_control(a, relative_action=True))
        absl.append(hand.denormalize_position_control(a, relative_action=False))
    np.savez(os.path.join(OUT, "hand_control.npz"), actuators=np.array(hi.ACTUATORS), joints=np.array(hi.JOINTS), ctrl_lo=lo, ctrl_hi=up,
             position_to_control=hi.POSITION_TO_CONTROL_MATRIX, actions=actions, hand_qpos=qh, ctrl_relative=np.array(rel), ctrl_absolute=np.array(absl))

    # ---- numpy hand FK (hand_forward_kinematics.py) -> relative fingertip observation
    from robogym.robot.shadow_hand.hand_forward_kinematics import compute_forward_kinematics_fingertips
    jl = np.array([hi.JOINT_LIMITS[j] for j in hi.JOINTS])
    angles = jl[:, 0] + rng.rand(48, 24) * (jl[:, 1] - jl[:, 0])
    angles[0] = 0
    tips = np.array([compute_forward_kinematics_fingertips(a) for a in angles])
    np.savez(os.path.join(OUT, "hand_fk.npz"), joint_angles=angles, relative_fingertips=tips)

    # ---- MultiGoalTracker (multi_goal_tracker.py:157-277), dactyl settings
    for missing in ("_jsonnet", "gym", "gym.spaces"):
        if missing not in sys.modules:
            mod = types.ModuleType(missing); mod.__path__ = []
            mod.Space = object; mod.Box = object; mod.Dict = object; mod.Tuple = object; mod.Discrete = object; mod.MultiDiscrete = object; mod.Env = object; mod.Wrapper = object; mod.spaces = mod
            sys.modules[missing] = mod
    from robogym.utils.multi_goal_tracker import MultiGoalTracker

    sim = types.SimpleNamespace(mj_sim=types.SimpleNamespace(nsubsteps=10, model=types.SimpleNamespace(opt=types.SimpleNamespace(timestep=0.008))))
    events = {"n": 0}
    def reset_goal():
        events["n"] += 1
        tracker.reset_goal_steps()
        return {"new": events["n"]}
    tracker = MultiGoalTracker(mujoco_simulation=sim, reset_goal_generation_fn=reset_goal, reset_goal_fn=reset_goal, max_timesteps_per_goal=400,
                               success_reward=5.0, successes_needed=50, success_pause_range_s=(0.0, 0.0), max_steps_goal_unreachable=10,
                               check_goal_reachable=False, use_goal_distance_reward=True, goal_types={"flip"}, random_state=np.random.RandomState(1))
    tracker.reset(); tracker.reset_goal_steps()
    T = 1500
    succ = rng.rand(T) < 0.05
    succ[600:1010] = False  # provoke a 400-step timeout
    gdr = rng.randn(T) * 0.1
    rew, done, info_s, info_steps, goal_reset = [], [], [], [], []
    for t in range(T):
        info = {}
        obs, r, d, info = tracker.process({}, 0.0, False, info, float(gdr[t]), bool(succ[t]), {"goal": {"goal_type": "flip"}})
        rew.append(r); done.append(d); info_s.append(info["successes_so_far"]); info_steps.append(info["steps_since_last_goal"]); goal_reset.append(bool(info.get("goal_reset", False)))
        if d:
            tracker.reset(); tracker.reset_goal_steps()
    np.savez(os.path.join(OUT, "tracker.npz"), is_successful=succ, goal_distance_reward=gdr, reward=np.array(rew), done=np.array(done),
             successes_so_far=np.array(info_s), steps_since_last_goal=np.array(info_steps), goal_reset=np.array(goal_reset))
    print("golden fixtures written to", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
