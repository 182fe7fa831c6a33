"""Golden vectors for the rearrange wrapper stack (tests/golden/rearrange_wrappers.npz): the REAL reference classes stacked as
`RearrangeEnv.apply_wrappers` stacks them (/root/reference/robogym/envs/rearrange/common/base.py:986-996: SmoothActionWrapper(alpha = 0.3) ->
ClipRewardWrapper -> DiscretizeActionWrapper(11 bins, linear)) around a scripted inner env that records the action it receives and emits
pre-drawn rewards.  `gym` (absent here) is the stub of tools/gen_golden_wrappers.py.  Needs /root/reference; the fixture travels.

    python tools/gen_golden_rearrange_wrappers.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "gen_golden_wrappers.py")).read()
exec(compile(src[:src.index("from robogym.envs.dactyl.common.dactyl_cube_wrappers")], "gym_stub", "exec"))      # the gym / mujoco_py stubs + sys.path

from robogym.wrappers.util import BinSpacing, ClipRewardWrapper, DiscretizeActionWrapper, SmoothActionWrapper  # noqa: E402


class Inner:
    def __init__(self, rewards):
        self.action_space = Box(-1.0, 1.0, shape=(6,))      # noqa: F821  (robot action space of the TCP arm + gripper: 6 numbers in [-1, 1])
        self.observation_space = Dict({})                     # noqa: F821
        self.sim = types.SimpleNamespace(model=types.SimpleNamespace(opt=types.SimpleNamespace(timestep=0.001)), nsubsteps=40)
        self.rewards, self.k, self.seen = rewards, 0, []
        self.unwrapped = self

    def reset(self):
        self.k = 0
        return {}

    def step(self, a):
        self.seen.append(np.asarray(a, dtype=np.float64).copy())
        r = self.rewards[self.k]; self.k += 1
        return {}, r, False, {}


def main():
    rng = np.random.RandomState(7)
    T = 40
    idx = rng.randint(0, 11, size=(T, 6))
    rewards = rng.uniform(-3, 3, size=(T, 3)); rewards[5] = [150.0, -120.0, 3.0]; rewards[17] = [-250.0, 99.0, 100.5]
    inner = Inner(rewards)
    env = DiscretizeActionWrapper(ClipRewardWrapper(SmoothActionWrapper(inner, alpha=0.3)), n_action_bins=11, bin_spacing=BinSpacing.LINEAR)
    obs0 = env.reset()
    ema, rew = [], []
    for t in range(T):
        if t == 25:                       # a second episode: the filter restarts
            env.reset()
        o, r, d, i = env.step(idx[t])
        ema.append(np.asarray(o["action_ema"], dtype=np.float64)); rew.append(np.asarray(r, dtype=np.float64))
    out = os.path.join(HERE, "..", "tests", "golden", "rearrange_wrappers.npz")
    np.savez_compressed(out, idx=idx, rewards=rewards, action_to_env=np.array(inner.seen), action_ema=np.array(ema), clipped_reward=np.array(rew),
                        action_ema_at_reset=np.asarray(obs0["action_ema"], dtype=np.float64), reset_at=np.array([25]))
    print("wrote", out, "first smoothed actions", np.array(inner.seen)[:2])


if __name__ == "__main__":
    main()
