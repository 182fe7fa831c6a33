"""Golden vectors for the rearrange goal layer (tests/golden/rearrange_goal.npz): the REAL `ObjectStateGoal.relative_goal / goal_distance`
(/root/reference/robogym/envs/rearrange/goals/object_state.py:492-599, rot_dist_type "full") evaluated on random object / goal states, with a stub in
place of the MuJoCo simulation (the two methods read five attributes of it and nothing else).  Needs /root/reference; the fixture travels.

    python tools/gen_golden_rearrange_goal.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
np.float = float      # (the reference's rotation module predates numpy 1.24)

# The module's import chain (gym.envs.robotics, _jsonnet, trimesh, collision, ...) is not installed here; the two methods and the rotation-distance function
# they call are pure numpy + robogym.utils.rotation.  Their SOURCE is taken from the reference file as it stands and executed on its own.
import ast  # noqa: E402

sys.path.insert(0, "/root/reference")
from robogym.utils import rotation  # noqa: E402

REF = "/root/reference/robogym/envs/rearrange/goals/object_state.py"
tree = ast.parse(open(REF).read())
want_funcs, want_methods = {"full_euler_angle_difference"}, {"relative_goal", "goal_distance"}
body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want_funcs]
cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ObjectStateGoal"][0]
methods = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want_methods]
assert len(body) == 1 and len(methods) == 2
shell = ast.ClassDef(name="ObjectStateGoal", bases=[], keywords=[], body=methods, decorator_list=[])
module = ast.Module(body=body + [shell], type_ignores=[])
ast.fix_missing_locations(module)
OS = types.SimpleNamespace()
ns = {"np": np, "rotation": rotation, "dict": dict}
exec(compile(module, REF, "exec"), ns)
OS.ObjectStateGoal, OS.full_euler_angle_difference = ns["ObjectStateGoal"], ns["full_euler_angle_difference"]
OS.GoalArgs = lambda: types.SimpleNamespace(rot_dist_type="full")


def main():
    rng = np.random.RandomState(11)
    N = 5
    sim = types.SimpleNamespace(num_objects=N, num_groups=N, max_num_objects=N, goal_pos_offset=0.0, goal_rot_weight=1.0,
                                object_groups=[types.SimpleNamespace(object_ids=[i]) for i in range(N)])
    g = OS.ObjectStateGoal.__new__(OS.ObjectStateGoal)
    g.mujoco_simulation = sim
    g.args = OS.GoalArgs()
    g.rot_dist_func = OS.full_euler_angle_difference
    T = 64
    cur_pos, goal_pos = rng.uniform(-0.3, 0.3, (T, N, 3)), rng.uniform(-0.3, 0.3, (T, N, 3))
    cur_rot, goal_rot = rng.uniform(-np.pi, np.pi, (T, N, 3)), rng.uniform(-np.pi, np.pi, (T, N, 3))
    cur_rot[:8, :, :2] = 0; goal_rot[:8, :, :2] = 0                       # pure yaw cases (objects flat on the table)
    goal_rot[8:12] = cur_rot[8:12] + rng.uniform(-1e-3, 1e-3, (4, N, 3))  # near the goal
    rel_pos, rel_rot, d_pos, d_rot = [], [], [], []
    for t in range(T):
        out = g.goal_distance({"obj_pos": goal_pos[t], "obj_rot": goal_rot[t]}, {"obj_pos": cur_pos[t], "obj_rot": cur_rot[t]})
        rel_pos.append(out["relative_goal"]["obj_pos"]); rel_rot.append(out["relative_goal"]["obj_rot"]); d_pos.append(out["obj_pos"]); d_rot.append(out["obj_rot"])
    # RearrangeEnv._calculate_num_success / _calculate_goal_distance_reward (envs/rearrange/common/base.py:824-848), same treatment: their source on a stub env
    BASE = "/root/reference/robogym/envs/rearrange/common/base.py"
    tree2 = ast.parse(open(BASE).read())
    env_cls = [n for n in tree2.body if isinstance(n, ast.ClassDef) and n.name == "RearrangeEnv"][0]
    meths = [n for n in env_cls.body if isinstance(n, ast.FunctionDef) and n.name in ("_calculate_num_success", "_calculate_goal_distance_reward")]
    assert len(meths) == 2
    for n in meths:
        n.returns = None
    m2 = ast.Module(body=[ast.ClassDef(name="RearrangeEnv", bases=[], keywords=[], body=meths, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(m2)
    ns2 = {"np": np}
    exec(compile(m2, BASE, "exec"), ns2)
    env = types.SimpleNamespace(constants=types.SimpleNamespace(success_threshold={"obj_pos": 0.04, "obj_rot": 0.2}, goal_reward_per_object=1.0))
    env._calculate_num_success = types.MethodType(ns2["RearrangeEnv"]._calculate_num_success, env)
    dp, dr = np.array(d_pos), np.array(d_rot)
    near = rng.rand(T, N) < 0.5                                            # some objects inside the thresholds
    dp = np.where(near, dp * 0.05, dp); dr = np.where(near, dr * 0.05, dr)
    nsucc = np.array([env._calculate_num_success({"obj_pos": dp[t], "obj_rot": dr[t]}) for t in range(T)])
    greward = np.array([ns2["RearrangeEnv"]._calculate_goal_distance_reward(env, {"obj_pos": dp[t - 1], "obj_rot": dr[t - 1]}, {"obj_pos": dp[t], "obj_rot": dr[t]}) for t in range(1, T)])
    out = os.path.join(HERE, "..", "tests", "golden", "rearrange_goal.npz")
    np.savez_compressed(out, cur_pos=cur_pos, cur_rot=cur_rot, goal_pos=goal_pos, goal_rot=goal_rot, rel_pos=np.array(rel_pos), rel_rot=np.array(rel_rot),
                        dist_pos=np.array(d_pos), dist_rot=np.array(d_rot), thr_dist_pos=dp, thr_dist_rot=dr, num_success=nsucc, goal_reward=greward)
    print("wrote", out, "dist_rot[0]", np.array(d_rot)[0])


if __name__ == "__main__":
    main()
