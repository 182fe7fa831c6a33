"""Golden vectors for the vectorised default wrapper stack (tests/golden/wrappers.npz): the REAL reference wrapper classes
(robogym/wrappers/{util,cube,dactyl,randomizations}.py), stacked by the reference's own construct_default_wrappers +
apply_named_wrappers (randomize=False: the no-noise, no-delay configuration of LockedEnv), around a scripted inner env that
emits pre-drawn observations / rewards / dones.  `gym` (absent here) is replaced by a minimal stub of exactly the classes the
wrappers subclass.  Needs /root/reference; the fixture travels with the repository.

    python tools/gen_golden_wrappers.py
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np

np.float = float


def mod(name, **attrs):
    m = types.ModuleType(name); m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


mod("mujoco_py", MjSim=object, MjSimState=object, cymj=types.SimpleNamespace(), const=types.SimpleNamespace())
mod("mujoco_py.generated"); mod("mujoco_py.generated.const")


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.full(self.shape, low, dtype=np.float32) if np.isscalar(low) else np.asarray(low)
        self.high = np.full(self.shape, high, dtype=np.float32) if np.isscalar(high) else np.asarray(high)
        self.dtype = dtype
        self.np_random = np.random.RandomState(0)


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = OrderedDict(spaces)


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec); self.shape = self.nvec.shape

    def seed(self, s):
        pass


class Wrapper:
    def __init__(self, env):
        self.env = env
        self.observation_space = getattr(env, "observation_space", None)
        self.action_space = getattr(env, "action_space", None)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, a):
        return self.env.step(a)

    def reset(self, *a, **k):
        return self.env.reset(*a, **k)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)


class ObservationWrapper(Wrapper):
    def reset(self, *a, **k):
        return self.observation(self.env.reset(*a, **k))

    def step(self, a):
        o, r, d, i = self.env.step(a)
        return self.observation(o), r, d, i


class ActionWrapper(Wrapper):
    def step(self, a):
        return self.env.step(self.action(a))


class RewardWrapper(Wrapper):
    def step(self, a):
        o, r, d, i = self.env.step(a)
        return o, self.reward(r), d, i


spaces = mod("gym.spaces", Box=Box, Dict=Dict, MultiDiscrete=MultiDiscrete, Space=Space, Tuple=object, Discrete=object)
mod("gym", Wrapper=Wrapper, ObservationWrapper=ObservationWrapper, ActionWrapper=ActionWrapper, RewardWrapper=RewardWrapper, Env=object, spaces=spaces, Space=Space)
mod("gym.wrappers")
sys.path.insert(0, "/root/reference")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "tests", "golden")

from robogym.envs.dactyl.common.dactyl_cube_wrappers import apply_wrappers  # noqa: E402
from robogym.utils import rotation  # noqa: E402


class _RelativeGoal:
    """LockedParallelGoal.relative_goal (envs/dactyl/goals/locked_parallel.py:54-66) on the reference's own rotation helpers
    (the class itself cannot be imported here: its module pulls robot_env.py, which needs mujoco_py.cymj and Python < 3.9)."""

    def relative_goal(self, goal_state, current_state):
        return {"cube_pos": np.zeros(3), "cube_quat": rotation.quat_difference(goal_state["cube_quat"], current_state["cube_quat"])}

OBS_SHAPES = OrderedDict([("cube_pos", 3), ("cube_quat", 4), ("qpos", 38), ("qvel", 36), ("hand_angle", 24), ("fingertip_pos", 15), ("goal_pos", 3),
                          ("goal_quat", 4), ("qpos_goal", 38), ("is_goal_achieved", 1)])


class ScriptedEnv:
    """Stands in for the unwrapped LockedEnv: emits the scripted episode, records the actions it receives."""

    def __init__(self, script):
        self.script, self.t = script, 0
        self.unwrapped = self
        self.observation_space = Dict({k: Box(-np.inf, np.inf, (n,), np.float32) for k, n in OBS_SHAPES.items()})
        self.action_space = Box(-1.0, 1.0, (20,), np.float32)
        self._random_state = np.random.RandomState(5)
        self.reward_names = ["env", "goal", "success"]
        self.goal_generation = _RelativeGoal()
        self.received = []
        model = types.SimpleNamespace(site_name2id=lambda n: 0, opt=types.SimpleNamespace(timestep=0.008))
        self.sim = types.SimpleNamespace(model=model, nsubsteps=10, data=types.SimpleNamespace(site_xpos=np.zeros((1, 3))))

    def _emit(self):
        o = OrderedDict((k, self.script["obs_" + k][self.t].copy()) for k in OBS_SHAPES)
        self._goal = {"cube_quat": o["goal_quat"].copy(), "cube_pos": o["goal_pos"].copy()}
        self.sim.data.site_xpos[0] = [1.0, 0.87, 0.2 + o["cube_pos"][2]]      # site cube:center (body cube:middle at z = 0.2)
        return o

    def reset(self):
        self.t = 0
        return self._emit()

    def step(self, action):
        self.received.append(np.asarray(action, dtype=np.float64).copy())
        self.t += 1
        info = {"successes_so_far": int(self.script["successes_so_far"][self.t])}
        return self._emit(), list(self.script["reward"][self.t]), bool(self.script["done"][self.t]), info


def main():
    rng = np.random.RandomState(20200902)
    T = 40
    script = {}
    for k, n in OBS_SHAPES.items():
        script["obs_" + k] = rng.randn(T + 1, n) * (150.0 if k == "qvel" else 1.0)   # some values beyond the +-100 clip
    for k in ("cube_quat", "goal_quat"):
        q = script["obs_" + k]; q /= np.linalg.norm(q, axis=1, keepdims=True)
    script["obs_cube_pos"] *= 0.05
    script["obs_cube_pos"][25:, 2] = -0.19            # the cube falls at step 25 (centre z = 0.01 < 0.04)
    script["obs_is_goal_achieved"] = (rng.rand(T + 1, 1) < 0.3).astype(np.float64)
    script["reward"] = np.stack([np.zeros(T + 1), rng.randn(T + 1) * 0.2, (rng.rand(T + 1) < 0.1) * 5.0], axis=1)
    script["reward"][7, 1] = 250.0                     # beyond the reward clip
    script["done"] = np.zeros(T + 1, bool); script["done"][33] = True
    script["successes_so_far"] = np.cumsum(script["reward"][:, 2] > 0)
    actions = rng.randint(0, 11, size=(T, 20))

    inner = ScriptedEnv(script)
    default_wrappers = {"default_no_noise_levels": {"fingertip_pos": {}, "hand_angle": {}, "cube_pos": {}, "cube_quat": {}},
                        "default_no_observation_delay_levels": {"interpolators": {}, "groups": {}}}
    env = apply_wrappers(inner, randomize=False, n_action_bins=None, fixed_wrist=False, relative_goal_wrapper=True, drop_reward=-20.0,
                         default_wrappers=default_wrappers, min_episode_length=-1)
    out = {("script_" + k): v for k, v in script.items()}
    out["actions"] = actions
    obs = env.reset()
    keys = list(obs.keys())
    rec = {k: [np.asarray(obs[k], dtype=np.float64).ravel()] for k in keys}
    rewards, dones, infos = [], [], {"fell_down": [], "drops_so_far": [], "first_drop": []}
    for t in range(T):
        obs, rew, done, info = env.step(actions[t])
        assert list(obs.keys()) == keys
        for k in keys:
            rec[k].append(np.asarray(obs[k], dtype=np.float64).ravel())
        rewards.append(np.asarray(rew, dtype=np.float64)); dones.append(done)
        for k in infos:
            infos[k].append(int(info[k]))
    for k in keys:
        out["wobs_" + k] = np.stack(rec[k])
    out["obs_keys"] = np.array(keys)
    out["wreward"] = np.stack(rewards); out["wdone"] = np.array(dones)
    for k, v in infos.items():
        out["winfo_" + k] = np.array(v)
    out["received_actions"] = np.stack(inner.received)
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **out)
    print("wrapped observation keys:", keys)
    print("reward shape", out["wreward"].shape, "fell at", int(np.argmax(out["winfo_fell_down"])), "dones", np.nonzero(out["wdone"])[0][:5])


if __name__ == "__main__":
    main()
