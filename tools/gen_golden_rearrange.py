"""Generate tests/golden/rearrange_*.npz from the parts of the REAL reference that run in this container with `mujoco_py` / `gym`
stubbed: `robogym.utils.rotation` (the Euler / quaternion / matrix conversions the rearrange observation and goal
code goes through), and the rearrange goal-distance arithmetic (`full_euler_angle_difference`, the distance formulas of
`ObjectStateGoal.goal_distance`, envs/rearrange/goals/object_state.py:67-68,584-599).  Needs /root/reference.

    python tools/gen_golden_rearrange.py
"""
import os
import sys
import types

import numpy as np

np.float = float
sys.path.insert(0, "/root/reference")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from robogym.utils import rotation as R

    rng = np.random.RandomState(20200904)
    e = rng.uniform(-np.pi, np.pi, (96, 3))
    e[:4, 1] = [np.pi / 2, -np.pi / 2, np.pi / 2 - 1e-9, 0.0]     # gimbal branch of mat2euler
    q = rng.randn(96, 4); q /= np.linalg.norm(q, axis=1, keepdims=True)
    e2 = rng.uniform(-np.pi, np.pi, (96, 3))
    mats = R.quat2mat(q)
    rel = R.normalize_angles(R.subtract_euler(e, e2))
    np.savez(os.path.join(OUT, "rearrange_rotation.npz"), euler=e, euler2=e2, quat=q, euler2quat=R.euler2quat(e), quat2mat=mats,
             mat2euler=R.mat2euler(mats), mat2quat=R.mat2quat(mats), quat2euler=R.quat2euler(q), subtract_euler=R.subtract_euler(e, e2),
             normalize_angles=R.normalize_angles(3.0 * e), rel_rot=rel,
             rot_distance=R.quat_magnitude(R.quat_normalize(R.euler2quat(rel))))
    print("written", os.path.join(OUT, "rearrange_rotation.npz"))


if __name__ == "__main__":
    main()
