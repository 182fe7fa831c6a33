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



# ------------------------------------------------------------------------------------------------------------------------
# randomize=True: the full default stack of LockedEnv (locked.py:228-282, cube_env.py:379-388) — Backlash, the thirteen
# pre-noise randomizations, observation noise, the phasespace freezing / occlusion wrappers, action noise — around a scripted
# env whose `sim` stub carries the compiled dactyl/locked model's arrays.  Every draw the wrappers take from the env's
# RandomState is logged, so the vectorised stack can be replayed on the same draws (tests/test_wrappers.py).
class RecordingRandomState:
    def __init__(self, seed):
        self._rs, self.log = np.random.RandomState(seed), []

    def _rec(self, name, out):
        self.log.append((name, np.asarray(out, dtype=np.float64).ravel().copy()))
        return out

    def uniform(self, low=0.0, high=1.0, size=None):
        return self._rec("uniform", self._rs.uniform(low, high, size))

    def randn(self, *shape):
        return self._rec("randn", self._rs.randn(*shape))

    def random_sample(self, size=None):
        return self._rec("random_sample", self._rs.random_sample(size))

    def exponential(self, scale=1.0, size=None):
        return self._rec("exponential", self._rs.exponential(scale, size))

    def randint(self, low, high=None, size=None):
        return self._rec("randint", self._rs.randint(low, high, size))

    def choice(self, a):
        return self._rec("choice", self._rs.choice(a))


class ModelStub:
    """The fields of mujoco_py's PyMjModel the wrappers touch, on the compiled model's numbers."""

    def __init__(self, cm):
        A, N = cm.arrays, cm.names
        self.body_inertia, self.body_mass, self.body_pos = A["body_inertia"].copy(), A["body_mass"].copy(), A["body_pos"].copy()
        self.geom_friction, self.geom_size = A["geom_friction"].copy(), A["geom_size"].copy()
        self.site_pos = A["site_pos"].copy()
        self.dof_damping, self.dof_jntid = A["dof_damping"].copy(), A["dof_jntid"].copy()
        self.jnt_range, self.tendon_range = A["jnt_range"].copy(), A["tendon_range"].copy()
        self.actuator_gainprm, self.actuator_ctrlrange = A["actuator_gainprm"].copy(), A["actuator_ctrlrange"].copy()
        self.nv = len(self.dof_damping)
        self.opt = types.SimpleNamespace(gravity=A["opt_gravity"].copy(), timestep=float(A["opt_timestep"][0]))
        self.body_names, self.geom_names, self.site_names = list(N["body"]), list(N["geom"]), list(N["site"])
        self.joint_names, self.actuator_names, self.tendon_names = list(N["joint"]), list(N["actuator"]), list(N["tendon"])
        self._qadr = {n: int(A["jnt_qposadr"][j]) for j, n in enumerate(self.joint_names)}
        for kind in ("body", "geom", "site", "joint", "actuator", "tendon"):
            names = getattr(self, kind + "_names")
            setattr(self, kind + "_name2id", (lambda names: (lambda n: names.index(n)))(names))

    def get_joint_qpos_addr(self, name):
        return self._qadr[name]


class RandomizedScriptedEnv(ScriptedEnv):
    def __init__(self, script, cm, pos_to_ctrl):
        super().__init__(script)
        self._random_state = RecordingRandomState(11)
        self.cm, self.P = cm, pos_to_ctrl
        model = ModelStub(cm)
        data = types.SimpleNamespace(site_xpos=np.zeros((len(model.site_names), 3)), xfrc_applied=np.zeros((len(model.body_names), 6)), ctrl=np.zeros(20),
                                     qpos=np.zeros(38), ncon=0, contact=[])
        data.get_joint_qpos = lambda name: data.qpos[model.get_joint_qpos_addr(name)]
        self.sim = types.SimpleNamespace(model=model, nsubsteps=10, data=data)
        self.constants = types.SimpleNamespace(relative_action=True)
        self.hand_q = np.array([model.get_joint_qpos_addr(n) for n in model.joint_names if n.startswith("robot0:")])
        self.snap = []

    def _emit(self):
        o = OrderedDict((k, self.script["obs_" + k][self.t].copy()) for k in OBS_SHAPES)
        self._goal = {"cube_quat": o["goal_quat"].copy(), "cube_pos": o["goal_pos"].copy()}
        d = self.sim.data
        d.site_xpos[self.sim.model.site_name2id("cube:center")] = [1.0, 0.87, 0.2 + o["cube_pos"][2]]
        d.qpos[:] = o["qpos"]
        cons = self.script["contacts"][self.t]
        d.ncon = len(cons)
        d.contact = [types.SimpleNamespace(geom1=int(c[0]), geom2=int(c[1]), dist=float(c[2])) for c in cons]
        return o

    def _set_action(self, action):
        """RobotEnv._set_action (robot_env.py:497-504) -> Robot.denormalize_position_control (robot_interface.py:247-278) with
        the live ctrl range -> set_position_control: restated (the class needs mujoco_py)."""
        m, d = self.sim.model, self.sim.data
        lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
        centre = self.P @ d.qpos[self.hand_q] if self.constants.relative_action else 0.5 * (hi + lo)
        d.ctrl[:] = np.clip(centre + np.clip(action, -1.0, 1.0) * (hi - lo) / 2.0, lo, hi)

    def step(self, action):
        m, d = self.sim.model, self.sim.data
        self.snap.append(dict(timestep=m.opt.timestep, xfrc=d.xfrc_applied.copy()))
        return super().step(action)


def main_randomized():
    sys.path.insert(0, ROOT)
    from robogym_amd.envs.dactyl.locked import load_locked_model

    cm = load_locked_model()
    g0 = np.load(os.path.join(OUT, "hand_tables.npz")) if os.path.exists(os.path.join(OUT, "hand_tables.npz")) else None
    rng = np.random.RandomState(20200903)
    T = 40
    script = {}
    for k, n in OBS_SHAPES.items():
        script["obs_" + k] = rng.randn(T + 1, n) * (150.0 if k == "qvel" else 1.0)
    for k in ("cube_quat", "goal_quat"):
        q = script["obs_" + k]; q /= np.linalg.norm(q, axis=1, keepdims=True)
    script["obs_cube_pos"] *= 0.05
    script["obs_qpos"] *= 0.2
    script["obs_cube_pos"][36:, 2] = -0.19
    script["obs_is_goal_achieved"] = (rng.rand(T + 1, 1) < 0.3).astype(np.float64)
    script["reward"] = np.stack([np.zeros(T + 1), rng.randn(T + 1) * 0.2, (rng.rand(T + 1) < 0.1) * 5.0], axis=1)
    script["done"] = np.zeros(T + 1, bool)
    script["successes_so_far"] = np.cumsum(script["reward"][:, 2] > 0)
    occl = [cm.names["geom"].index(n) for n in ("robot0:ffocclusion", "robot0:mfocclusion", "robot0:rfocclusion", "robot0:lfocclusion", "robot0:thocclusion")]
    cube = cm.names["geom"].index("cube:middle")
    contacts = []
    for t in range(T + 1):
        c = [(cube, 5 + int(rng.randint(40)), float(rng.randn() * 5e-4)) for _ in range(int(rng.randint(0, 4)))]
        for g in occl:
            if rng.rand() < 0.25:
                c.append((cube, g, float(rng.randn() * 3e-4 - 1e-4)))
        contacts.append(c)
    script["contacts"] = contacts
    actions = rng.randint(0, 11, size=(T, 20))
    resets_at = [0, 25]          # a second episode: the randomizations restart from the ORIGINAL values, not from the previous draw

    A = cm.arrays
    hand_joints = [j for j, n in enumerate(cm.names["joint"]) if n.startswith("robot0:")]
    Pm = np.zeros((20, len(hand_joints)))
    for u in range(20):
        if A["actuator_trntype"][u] == 0:
            Pm[u, hand_joints.index(int(A["actuator_trnid"][u]))] = 1
        else:
            t_ = int(A["actuator_trnid"][u])
            for w in range(A["tendon_adr"][t_], A["tendon_adr"][t_] + A["tendon_num"][t_]):
                Pm[u, hand_joints.index(int(A["wrap_objid"][w]))] = 1
    inner = RandomizedScriptedEnv(script, cm, Pm)
    locked_defaults = {
        "default_observation_noise_levels": {"fingertip_pos": {"uncorrelated": 0.002, "additive": 0.001}, "hand_angle": {"additive": 0.1, "uncorrelated": 0.1},
                                             "cube_pos": {"additive": 0.005, "uncorrelated": 0.001}, "cube_quat": {"additive": 0.1, "uncorrelated": 0.09}},
        "default_observation_delay_levels": {"interpolators": {"cube_quat": "QuatInterpolator"}, "groups": {}},
        "pre_obsnoise_randomizations": [["RandomizedActionLatency"], ["RandomizedCubeSizeWrapper"], ["RandomizedBodyInertiaWrapper"], ["RandomizedTimestepWrapper"],
                                        ["RandomizedRobotFrictionWrapper"], ["RandomizedCubeFrictionWrapper"], ["RandomizedGravityWrapper"], ["RandomizedWindWrapper"],
                                        ["RandomizedPhasespaceFingersWrapper"], ["RandomizedRobotDampingWrapper"], ["RandomizedRobotKpWrapper"],
                                        ["RandomizedJointLimitWrapper"], ["RandomizedTendonRangeWrapper"]],      # locked.py:264-279
        "post_obsnoise_randomizations": [["FingersOccludedPhasespaceMarkers"], ["FingersFreezingPhasespaceMarkers"], ["CubeFreezingPhasespaceBody"],
                                         ["ActionNoiseWrapper"]],                                                # cube_env.py:382-387
    }
    env = apply_wrappers(inner, randomize=True, n_action_bins=None, fixed_wrist=False, relative_goal_wrapper=True, drop_reward=-20.0,
                         default_wrappers=locked_defaults, min_episode_length=-1)
    out = {("script_" + k): v for k, v in script.items() if k != "contacts"}
    cflat = [(t, c[0], c[1], c[2]) for t, cs in enumerate(contacts) for c in cs]
    out["script_contacts"] = np.array(cflat, dtype=np.float64).reshape(-1, 4)
    out["actions"] = actions; out["resets_at"] = np.array(resets_at)
    rec, keys, rewards, dones = None, None, [], []
    infos = {"fell_down": [], "drops_so_far": [], "first_drop": []}
    model_after_reset = []
    draw_mark = []

    def snap_model():
        m = inner.sim.model
        return dict(body_inertia=m.body_inertia.copy(), geom_friction=m.geom_friction.copy(), gravity=m.opt.gravity.copy(), dof_damping=m.dof_damping.copy(),
                    actuator_kp=m.actuator_gainprm[:, 0].copy(), jnt_range=m.jnt_range.copy(), actuator_ctrlrange=m.actuator_ctrlrange.copy(),
                    tendon_range=m.tendon_range.copy(), site_pos=m.site_pos.copy(), cube_size=m.geom_size[cube].copy())

    def record(obs):
        nonlocal rec, keys
        if rec is None:
            keys = list(obs.keys()); rec = {k: [] for k in keys}
        assert list(obs.keys()) == keys
        for k in keys:
            rec[k].append(np.asarray(obs[k], dtype=np.float64).ravel())

    t_inner = 0
    for t in range(T):
        if t in resets_at:
            inner.t = t       # the scripted episode continues at the same script row after a reset
            _orig_reset = inner.reset
            inner.reset = lambda: inner._emit()
            record(env.reset())
            inner.reset = _orig_reset
            model_after_reset.append(snap_model())
            draw_mark.append(len(inner._random_state.log))
        obs, rew, done, info = env.step(actions[t])
        record(obs)
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
    out["step_timestep"] = np.array([s_["timestep"] for s_ in inner.snap])        # model.opt.timestep each inner env.step ran with
    out["step_xfrc"] = np.stack([s_["xfrc"][cm.names["body"].index("cube:middle"), :3] for s_ in inner.snap])
    for r, snap in enumerate(model_after_reset):
        for k, v in snap.items():
            out["model%d_%s" % (r, k)] = v
    log = inner._random_state.log
    out["draw_names"] = np.array([n for n, _ in log])
    out["draw_offsets"] = np.cumsum([0] + [len(v) for _, v in log])
    out["draw_values"] = np.concatenate([v for _, v in log])
    out["pos_to_ctrl"] = Pm
    np.savez_compressed(os.path.join(OUT, "wrappers_randomized.npz"), **out)
    print("randomize=True: %d observation keys, %d draws (%d values)" % (len(keys), len(log), len(out["draw_values"])))
    print(keys)



def main_fixed_wrist():
    """constants.fixed_wrist=True (FixedWristWrapper innermost, randomize=False): the action that reaches the env holds the wrist
    flexion actuator at `(0 - qpos[WRJ0]) / half control range` whatever the policy asks for."""
    sys.path.insert(0, ROOT)
    from robogym_amd.envs.dactyl.locked import load_locked_model

    cm = load_locked_model()
    rng = np.random.RandomState(20200904)
    T = 12
    script = {}
    for k, n in OBS_SHAPES.items():
        script["obs_" + k] = rng.randn(T + 1, n)
    for k in ("cube_quat", "goal_quat"):
        q = script["obs_" + k]; q /= np.linalg.norm(q, axis=1, keepdims=True)
    script["obs_cube_pos"] *= 0.02
    script["obs_qpos"] *= 0.2
    script["obs_is_goal_achieved"] = np.zeros((T + 1, 1))
    script["reward"] = np.zeros((T + 1, 3)); script["done"] = np.zeros(T + 1, bool); script["successes_so_far"] = np.zeros(T + 1, int)
    script["contacts"] = [[] for _ in range(T + 1)]
    actions = rng.randint(0, 11, size=(T, 20))
    inner = RandomizedScriptedEnv(script, cm, np.zeros((20, 24)))
    env = apply_wrappers(inner, randomize=False, n_action_bins=None, fixed_wrist=True, relative_goal_wrapper=True, drop_reward=-20.0,
                         default_wrappers={"default_no_noise_levels": {"fingertip_pos": {}, "hand_angle": {}, "cube_pos": {}, "cube_quat": {}},
                                           "default_no_observation_delay_levels": {"interpolators": {}, "groups": {}}}, min_episode_length=-1)
    env.reset()
    for t in range(T):
        env.step(actions[t])
    out = {("script_" + k): v for k, v in script.items() if k != "contacts"}
    out["actions"] = actions; out["received_actions"] = np.stack(inner.received)
    np.savez_compressed(os.path.join(OUT, "wrappers_fixed_wrist.npz"), **out)
    u = cm.names["actuator"].index("robot0:A_WRJ0")
    print("fixed wrist: received wrist actions", np.round(out["received_actions"][:4, u], 4))


# ------------------------------------------------------------------------------------------------------------------------
# The same stack around the FULL cube's observation set (FullPerpendicularEnv._default_observation_map, full_perpendicular.py:177-192; its default_no_noise_levels
# :390-396 add cube_face_angle) with FaceFreeGoal.relative_goal (goals/face_free.py:147-173) -- the method's own source, lifted out of its class with ast (the module
# imports robot_env.py, which needs mujoco_py.cymj) and run on the reference's cube_utils / rotation: tests/golden/wrappers_full.npz.
# (qpos / qvel and their perp_ duplicates only pass through the clip: 12 / 11 numbers instead of the model's 173 / 168 keep the fixture small)
FULL_OBS_SHAPES = OrderedDict([("cube_pos", 3), ("cube_quat", 4), ("cube_face_angle", 6), ("qpos", 12), ("qvel", 11), ("perp_qpos", 12), ("perp_qvel", 11), ("hand_angle", 24),
                               ("fingertip_pos", 15), ("goal_pos", 3), ("goal_quat", 4), ("goal_face_angle", 6)])


def _face_free_relative_goal():
    import ast

    path = "/root/reference/robogym/envs/dactyl/goals/face_free.py"
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "FaceFreeGoal"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "relative_goal"][0]
    m_ = ast.Module(body=[fn], type_ignores=[]); ast.fix_missing_locations(m_)
    mod("robogym.mujoco.helpers", joint_qpos_ids_from_prefix=None)      # (cube_utils imports one helper it does not need here)
    from robogym.envs.dactyl.common import cube_utils

    ns = {"np": np, "rotation": rotation, "cube_utils": cube_utils}
    exec(compile(m_, path, "exec"), ns)
    return ns["relative_goal"]


class FullScriptedEnv(ScriptedEnv):
    def __init__(self, script):
        super().__init__(script)
        self.observation_space = Dict({k: Box(-np.inf, np.inf, (n,), np.float32) for k, n in FULL_OBS_SHAPES.items()})
        rel = _face_free_relative_goal()
        self.goal_generation = types.SimpleNamespace(relative_goal=lambda goal_state, current_state: rel(None, goal_state, current_state))

    def _emit(self):
        o = OrderedDict((k, self.script["obs_" + k][self.t].copy()) for k in FULL_OBS_SHAPES)
        self._goal = {"cube_quat": o["goal_quat"].copy(), "cube_pos": o["goal_pos"].copy(), "cube_face_angle": o["goal_face_angle"].copy(),
                      "goal_type": "rotation" if self.script["goal_rotation"][self.t] else "flip", "axis_nr": int(self.script["axis_nr"][self.t]),
                      "axis_sign": int(self.script["axis_sign"][self.t])}
        self.sim.data.site_xpos[0] = [1.0, 0.87, 0.2 + o["cube_pos"][2]]
        return o


def main_full():
    rng = np.random.RandomState(20200907)
    T = 40
    script = {}
    for k, n in FULL_OBS_SHAPES.items():
        script["obs_" + k] = rng.randn(T + 1, n) * (150.0 if k in ("qvel", "perp_qvel") else 1.0)
    for k in ("cube_quat", "goal_quat"):
        q = script["obs_" + k]; q /= np.linalg.norm(q, axis=1, keepdims=True)
    script["obs_cube_face_angle"] *= 2.0; script["obs_goal_face_angle"] = np.round(script["obs_goal_face_angle"] * 2.0) * (np.pi / 2)      # (differences beyond +-pi: normalize_angles at work)
    script["obs_cube_pos"] *= 0.05
    script["obs_cube_pos"][25:, 2] = -0.19
    script["goal_rotation"] = rng.rand(T + 1) < 0.5
    script["axis_nr"] = rng.randint(0, 3, T + 1); script["axis_sign"] = rng.choice([-1, 1], T + 1)
    script["reward"] = np.stack([np.zeros(T + 1), rng.randn(T + 1) * 0.2, (rng.rand(T + 1) < 0.1) * 5.0], axis=1)
    script["done"] = np.zeros(T + 1, bool); script["done"][33] = True
    script["successes_so_far"] = np.cumsum(script["reward"][:, 2] > 0)
    actions = rng.randint(0, 11, size=(T, 20))
    inner = FullScriptedEnv(script)
    default_wrappers = {"default_no_noise_levels": {"fingertip_pos": {}, "hand_angle": {}, "cube_pos": {}, "cube_quat": {}, "cube_face_angle": {}},
                        "default_no_observation_delay_levels": {"interpolators": {}, "groups": {}}}
    env = apply_wrappers(inner, randomize=False, n_action_bins=None, fixed_wrist=False, relative_goal_wrapper=True, drop_reward=-20.0,
                         default_wrappers=default_wrappers, min_episode_length=-1)
    out = {("script_" + k): np.asarray(v) for k, v in script.items()}
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
    np.savez_compressed(os.path.join(OUT, "wrappers_full.npz"), **out)
    print("full cube: wrapped observation keys:", keys)


if __name__ == "__main__":
    if "--full-only" in sys.argv:
        main_full()
        sys.exit(0)
    main()
    main_randomized()
    main_fixed_wrist()
    main_full()
