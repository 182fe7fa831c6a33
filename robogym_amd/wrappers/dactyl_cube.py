"""The reference's default wrapper stack of the dactyl cube envs, vectorised over the env batch.

`construct_default_wrappers` (envs/dactyl/common/dactyl_cube_wrappers.py:8-91) builds, innermost first:
    ClipActionWrapper -> StopOnFallWrapper -> [BacklashWrapper, physics randomizations]* -> ObservationDelayWrapper ->
    RandomizeObservationWrapper -> SmoothActionWrapper -> RelativeGoalWrapper -> [post-noise randomizations]* ->
    AngleObservationWrapper -> UnifiedGoalObservationWrapper -> ClipObservationWrapper -> ClipRewardWrapper ->
    PreviousActionObservationWrapper -> RewardObservationWrapper -> DiscretizeActionWrapper            (* randomize=True only)
(wrappers/util.py:36-343, wrappers/cube.py:106-182, wrappers/dactyl.py:190-221, wrappers/randomizations.py:314-393).
Here the whole stack is ONE object whose `step` / `reset` apply the same transformations in the same order as `[B, ...]`
tensor ops around the batched env; every wrapper of the list is either implemented (with the reference's formula) or
named in `NOT_BUILT` and refused when asked for.

Implemented: ClipAction, StopOnFall (drop reward, min_episode_length, fell_down / drops_so_far / first_drop), ObservationDelay
with no delay groups (the default of every dactyl env: locked.py:244-262 comments the groups out), RandomizeObservation
(additive per-episode bias, multiplicative bias, uncorrelated per-step noise, quaternion noise with the 1.96 correction),
SmoothAction (bias-corrected EMA, alpha adjusted to the step length), RelativeGoal (LockedParallelGoal.relative_goal),
AngleObservation, UnifiedGoalObservation, ClipObservation, ClipReward, PreviousActionObservation, RewardObservation,
DiscretizeAction (linear bins), and with `randomize=True` the physics randomizations that are writes into `sim.model` /
`sim.data` — per-env parameter rows here (robogym_amd/randomization/sim.py, include/rgstep.h RG_F_ENVPRM):
RandomizedBodyInertia, RandomizedRobotFriction, RandomizedCubeFriction, RandomizedGravity, RandomizedRobotDamping,
RandomizedRobotKp, RandomizedWind.
"""
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from robogym_amd.utils import rotation

NOT_BUILT = ["BacklashWrapper", "RandomizedActionLatency", "RandomizedCubeSizeWrapper", "RandomizedTimestepWrapper", "RandomizedPhasespaceFingersWrapper",
             "RandomizedJointLimitWrapper", "RandomizedTendonRangeWrapper", "FingersOccludedPhasespaceMarkers", "FingersFreezingPhasespaceMarkers",
             "CubeFreezingPhasespaceBody", "ActionNoiseWrapper", "FixedWristWrapper"]

DEFAULT_OBSERVATION_NOISE_LEVELS = {   # locked.py:232-237
    "fingertip_pos": {"uncorrelated": 0.002, "additive": 0.001},
    "hand_angle": {"additive": 0.1, "uncorrelated": 0.1},
    "cube_pos": {"additive": 0.005, "uncorrelated": 0.001},
    "cube_quat": {"additive": 0.1, "uncorrelated": 0.09},
}
NO_NOISE_LEVELS = {"fingertip_pos": {}, "hand_angle": {}, "cube_pos": {}, "cube_quat": {}}   # locked.py:239-244
QUAT_NOISE_CORRECTION = 1.96   # wrappers/randomizations.py:309-311


def _loguniform(gen, low, high, shape, device):
    lo, hi = float(np.log(low)), float(np.log(high))
    return torch.exp(lo + (hi - lo) * torch.rand(shape, generator=gen, device=device))


class BatchedDactylCubeWrappers:
    def __init__(self, env, randomize: bool = False, n_action_bins: Optional[int] = None, relative_goal_wrapper: bool = True, drop_reward: float = -20.0,
                 min_episode_length: int = -1, noise_levels: Optional[dict] = None, smooth_alpha: float = 0.0, clip: float = 100.0, fixed_wrist: bool = False):
        if fixed_wrist:
            raise NotImplementedError("FixedWristWrapper is not built (wrappers/dactyl.py:173-189)")
        self.env = env
        self.unwrapped = env
        self.randomize = bool(randomize)
        self.B, self.device, self.nu = env.batch_size, env.device, env.num_actions
        self.batch_size = self.B
        nb = 11 if n_action_bins is None else int(n_action_bins)          # DiscretizeActionWrapper.DEFAULT_BINS
        self.n_action_bins = nb
        self._bins = torch.linspace(-1.0, 1.0, nb, device=self.device)     # BinSpacing.LINEAR over Box(-1, 1)
        self.relative_goal_wrapper = relative_goal_wrapper
        self.drop_reward, self.min_episode_length, self.clip = float(drop_reward), int(min_episode_length), float(clip)
        self.levels = noise_levels if noise_levels is not None else (DEFAULT_OBSERVATION_NOISE_LEVELS if randomize else NO_NOISE_LEVELS)
        self.smooth_alpha = float(smooth_alpha)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed((env._seed * 7919 + 13) & 0x7FFFFFFF)
        B, dev = self.B, self.device
        z = lambda *s: torch.zeros(s, device=dev)
        self._steps = torch.zeros(B, dtype=torch.int32, device=dev)
        self._drops_so_far = torch.zeros(B, dtype=torch.int32, device=dev)
        self._first_drop = torch.zeros(B, dtype=torch.int32, device=dev)
        self._previous_action = z(B, self.nu)
        self._ema_value, self._ema_t = z(B, self.nu), torch.zeros(B, dtype=torch.int32, device=dev)
        self._additive_bias: Dict[str, torch.Tensor] = {}
        self._multiplicative_bias: Dict[str, torch.Tensor] = {}
        self._key_len = {"fingertip_pos": 15, "hand_angle": 24, "cube_pos": 3, "cube_quat": 1}   # key_length(): quaternions get ONE angle
        self._cube_center_z0 = env.mujoco_simulation.cube_body_z
        self._physics = []
        self._wind_hit_prob = z(B)
        if randomize:
            self._build_physics_randomizers()
        self.action_space = {"nvec": [nb] * self.nu, "dtype": "int64"}   # gym.spaces.MultiDiscrete([n_action_bins] * nu)

    # ------------------------------------------------------------------ physics randomizations (pre_obsnoise_randomizations, locked.py:264-279)
    def _build_physics_randomizers(self):
        sim = self.env.mujoco_simulation
        m = sim.model
        P = sim.params
        robot_geoms = [g for g, n in enumerate(m.names["geom"]) if n.startswith("robot0:")]
        cube_geoms = [g for g, n in enumerate(m.names["geom"]) if n.startswith("cube:")]
        robot_dofs = [d for d in range(sim.nv) if m.names["joint"][int(m.dof_jntid[d])].startswith("robot0:")]
        robot_acts = [u for u, n in enumerate(m.names["actuator"]) if n.startswith("robot0:")]
        self._orig = {k: P[k][0].clone() for k in ("body_inertia", "geom_friction", "gravity", "dof_damping", "actuator_gainprm")}
        dev, gen = self.device, self._gen

        def body_inertia(mask):    # RandomizedBodyInertiaWrapper (randomizations.py:72-92): one multiplier U(0.5, 1.5) per body
            mult = 0.5 + torch.rand((self.B, self._orig["body_inertia"].shape[0], 1), generator=gen, device=dev)
            P["body_inertia"].copy_(torch.where(mask[:, None, None], self._orig["body_inertia"] * mult, P["body_inertia"]))

        def friction(geoms, ranges):   # RandomizedFrictionBaseWrapper (:95-153): one multiplier per friction type for the whole geom set
            idx = torch.as_tensor(geoms, device=dev)

            def apply(mask):
                f = P["geom_friction"]
                for col, (lo, hi) in enumerate(ranges):
                    mult = lo + (hi - lo) * torch.rand((self.B, 1), generator=gen, device=dev)
                    new = self._orig["geom_friction"][idx, col][None, :] * mult
                    f[:, idx, col] = torch.where(mask[:, None], new, f[:, idx, col])
            return apply

        def gravity(mask):         # RandomizedGravityWrapper (:176-191): g + 0.4 * randn(3)
            g = self._orig["gravity"] + 0.4 * torch.randn((self.B, 3), generator=gen, device=dev)
            P["gravity"].copy_(torch.where(mask[:, None], g, P["gravity"]))

        def damping(mask):         # RandomizedRobotDampingWrapper (dactyl.py:153-160): loguniform(1/1.5, 1.5) per robot dof
            idx = torch.as_tensor(robot_dofs, device=dev)
            new = self._orig["dof_damping"][idx][None, :] * _loguniform(gen, 1 / 1.5, 1.5, (self.B, len(robot_dofs)), dev)
            P["dof_damping"][:, idx] = torch.where(mask[:, None], new, P["dof_damping"][:, idx])

        def kp(mask):              # RandomizedRobotKpWrapper (dactyl.py:163-170): loguniform(0.5, 2.0) per actuator
            idx = torch.as_tensor(robot_acts, device=dev)
            new = self._orig["actuator_gainprm"][idx, 0][None, :] * _loguniform(gen, 0.5, 2.0, (self.B, len(robot_acts)), dev)
            P["actuator_gainprm"][:, idx, 0] = torch.where(mask[:, None], new, P["actuator_gainprm"][:, idx, 0])

        self._physics = [body_inertia, friction(robot_geoms, [[0.7, 1.3], [0.5, 1.5], [0.5, 1.5]]), friction(cube_geoms, [[0.5, 1.5], [0.2, 5.0], [0.2, 5.0]]),
                         gravity, damping, kp]
        self._cube_body = m.name2id("body", "cube:middle")

    def _randomize_physics(self, mask):
        for f in self._physics:
            f(mask)
        # RandomizedWindWrapper.reset (cube.py:62-73): per-episode hit probability, loguniform over [0.01, 1] * step length / 0.8 s
        sim = self.env.mujoco_simulation
        step_s = sim.n_substeps * float(sim.model.opt_timestep[0])
        hp = _loguniform(self._gen, 0.01 * step_s / 0.8, step_s / 0.8, (self.B,), self.device)
        self._wind_hit_prob = torch.where(mask, hp, self._wind_hit_prob)

    def _wind_step(self):          # RandomizedWindWrapper.step (cube.py:75-85)
        P = self.env.mujoco_simulation.params
        x = P["xfrc_applied"][:, self._cube_body, :3]
        x *= 0.99
        hit = torch.rand(self.B, generator=self._gen, device=self.device) < self._wind_hit_prob
        force = torch.randn((self.B, 3), generator=self._gen, device=self.device) * P["body_mass"][:, self._cube_body, None] * 1.0
        P["xfrc_applied"][:, self._cube_body, :3] = torch.where(hit[:, None], force, x)

    # ------------------------------------------------------------------ observation pipeline
    def _is_fallen(self, obs):     # StopOnFallWrapper._is_fallen (cube.py:153-156): site cube:center z < 0.04
        return (self._cube_center_z0 + obs["cube_pos"][:, 2]) < 0.04

    def _noisy(self, obs):         # RandomizeObservationWrapper.observation (randomizations.py:352-393)
        out = {}
        for key in sorted(self.levels):
            lv, n = self.levels[key], self._key_len[key]
            unc = torch.randn((self.B, n), generator=self._gen, device=self.device) * lv.get("uncorrelated", 0.0)
            add = self._additive_bias[key] + unc
            v = obs[key].clone()
            if not key.endswith("_quat"):
                v = v * self._multiplicative_bias[key] + add
            else:
                axis = torch.rand((self.B, 3), generator=self._gen, device=self.device) * 2 - 1
                axis = axis / axis.norm(dim=-1, keepdim=True)
                ang = add * QUAT_NOISE_CORRECTION
                nq = torch.cat([torch.cos(ang / 2), torch.sin(ang / 2) * axis], dim=-1)
                nq = nq / nq.norm(dim=-1, keepdim=True)
                v = rotation.quat_normalize(rotation.quat_mul(v, nq))
            out["noisy_" + key] = v
        return out

    def _observation(self, obs, action_ema, reward):
        o = OrderedDict(obs)
        o["fell_down"] = self._is_fallen(obs)[:, None]                                  # StopOnFallWrapper
        o.update(self._noisy(obs))                                                       # (ObservationDelayWrapper: no groups) + RandomizeObservationWrapper
        o["action_ema"] = action_ema                                                     # SmoothActionWrapper
        if self.relative_goal_wrapper:                                                   # RelativeGoalWrapper(obs_prefix="cube_") with LockedParallelGoal.relative_goal
            gq = self.env._goal_quat
            zero3 = torch.zeros((self.B, 3), device=self.device, dtype=o["cube_pos"].dtype)
            rel = {"pos": lambda cur: zero3, "quat": lambda cur: rotation.quat_difference(gq.to(cur.dtype), cur)}
            for name in ("pos", "quat"):      # the reference's key order: per goal part, achieved / relative / noisy achieved / noisy relative
                o["achieved_goal_" + name] = o["cube_" + name].clone()
                o["relative_goal_" + name] = rel[name](o["cube_" + name])
                o["noisy_achieved_goal_" + name] = o["noisy_cube_" + name].clone()
                o["noisy_relative_goal_" + name] = rel[name](o["noisy_cube_" + name])
        for key in list(o.keys()):                                                       # AngleObservationWrapper
            if key.endswith("_angle"):
                o[key] = torch.cat([torch.cos(o[key]), torch.sin(o[key])], dim=-1)
        for goal_key in ("relative_goal", "achieved_goal", "goal"):                      # UnifiedGoalObservationWrapper(goal_parts = pos, quat, face_angle)
            for pre in ("", "noisy_"):
                if pre and not any(k.startswith("noisy_" + goal_key + "_") for k in o):
                    continue
                parts = [o[pre + goal_key + "_" + p] for p in ("pos", "quat", "face_angle") if pre + goal_key + "_" + p in o]
                if parts:
                    o[pre + goal_key] = torch.cat(parts, dim=-1)
        for key in o:                                                                    # ClipObservationWrapper
            if o[key].dtype.is_floating_point:
                o[key] = o[key].clamp(-self.clip, self.clip)
        o["previous_action"] = self._previous_action.clone()                             # PreviousActionObservationWrapper
        o["reward"] = reward                                                             # RewardObservationWrapper(reward_inds=[1, 2])
        return o

    # ------------------------------------------------------------------ gym surface
    def reset(self, mask: Optional[torch.Tensor] = None):
        B, dev = self.B, self.device
        mask = torch.ones(B, dtype=torch.bool, device=dev) if mask is None else mask.to(dev).bool()
        if self.randomize:
            self._randomize_physics(mask)      # RandomizedBodyWrapper.reset: parameters first, THEN the env's reset recipe runs with them
        obs = self.env.reset(mask)
        self._steps.masked_fill_(mask, 0); self._drops_so_far.masked_fill_(mask, 0); self._first_drop.masked_fill_(mask, 0)
        self._previous_action.masked_fill_(mask[:, None], 0.0)
        self._ema_value.masked_fill_(mask[:, None], 0.0); self._ema_t.masked_fill_(mask, 0)
        for key in sorted(self.levels):                                                  # RandomizeObservationWrapper.reset
            lv, n = self.levels[key], self._key_len[key]
            add = torch.randn((B, n), generator=self._gen, device=dev) * lv.get("additive", 0.0)
            mul = 1.0 + torch.randn((B, n), generator=self._gen, device=dev) * lv.get("multiplicative", 0.0)
            self._additive_bias[key] = torch.where(mask[:, None], add, self._additive_bias.get(key, add))
            self._multiplicative_bias[key] = torch.where(mask[:, None], mul, self._multiplicative_bias.get(key, mul))
        return self._observation(obs, torch.zeros((B, self.nu), device=dev), torch.zeros((B, 2), device=dev))

    def step(self, action: torch.Tensor):
        """action: int64 [B, nu] bin indices in [0, n_action_bins).  Returns (obs dict, reward [B, 4] = env, goal, success, drop,
        done [B], info)."""
        a = self._bins[torch.as_tensor(action, device=self.device).long()]              # DiscretizeActionWrapper.action
        self._previous_action = a.clone()                                                # PreviousActionObservationWrapper.step
        # SmoothActionWrapper.step: IncrementalExpAvg with alpha adjusted to the step length (util.py:142-219)
        sim = self.env.mujoco_simulation
        alpha = float(np.power(self.smooth_alpha, (float(sim.model.opt_timestep[0]) * sim.n_substeps) / 0.08)) if self.smooth_alpha > 0 else 0.0
        self._ema_value = self._ema_value * alpha + (1 - alpha) * a
        self._ema_t += 1
        a = self._ema_value / (1 - torch.pow(torch.full_like(self._ema_value, alpha), self._ema_t[:, None].to(a.dtype)))
        a_ema = a
        a = a.clamp(-1.0, 1.0)                                                           # ClipActionWrapper
        obs, rew, done, info = self.env.step(a)
        if self.randomize:
            self._wind_step()
        # StopOnFallWrapper.step (cube.py:125-151)
        fallen = self._is_fallen(obs)
        done = done | fallen
        first = fallen & (self._first_drop == 0)
        self._drops_so_far += fallen.to(torch.int32)
        drop_rew = torch.where(first, torch.full((self.B,), self.drop_reward, device=self.device), torch.zeros(self.B, device=self.device))
        self._first_drop = torch.where(first, info["successes_so_far"] + 1, self._first_drop)
        if self.min_episode_length > 0:
            done = done & ~(self._steps < self.min_episode_length)
        reward = torch.cat([rew, drop_rew[:, None]], dim=1).clamp(-self.clip, self.clip)   # ... + ClipRewardWrapper
        info = dict(info)
        info.update({"fell_down": fallen, "drops_so_far": self._drops_so_far.clone(), "first_drop": self._first_drop.clone()})
        self._steps += 1
        return self._observation(obs, a_ema, reward[:, 1:3]), reward, done, info
