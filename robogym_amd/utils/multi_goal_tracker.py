"""Batched `MultiGoalTracker` (/root/reference/robogym/utils/multi_goal_tracker.py:24-277): the
per-env integer/boolean bookkeeping of successes, timeouts and goal resets as [B] tensor ops.

Settings that are fixed in the dactyl envs are folded in: `success_pause_range_s = (0, 0)` makes
`_success_steps_required` always 1 (`randint(1, 2)`, :88-93), `min_timesteps_per_goal = 0`,
`check_goal_reachable = False`, a single goal type."""
from typing import Dict

import torch


class BatchedMultiGoalTracker:
    def __init__(self, batch_size: int, device, max_timesteps_per_goal=400, success_reward=5.0, successes_needed=50, use_goal_distance_reward=True):
        self.max_timesteps_per_goal = max_timesteps_per_goal
        self.success_reward = success_reward
        self.successes_needed = successes_needed
        self.use_goal_distance_reward = use_goal_distance_reward
        z = lambda: torch.zeros(batch_size, dtype=torch.int32, device=device)
        self.steps, self.steps_since_last_goal, self.successes_so_far, self.goals_so_far, self.consecutive_success = z(), z(), z(), z(), z()

    def reset(self, mask: torch.Tensor):
        """MultiGoalTracker.reset (:83-113) for the selected envs."""
        for buf in (self.steps, self.steps_since_last_goal, self.successes_so_far, self.goals_so_far, self.consecutive_success):
            buf[mask] = 0

    def reset_goal_steps(self, mask: torch.Tensor):
        """:115-125 — a new goal starts counting."""
        self.goals_so_far += mask.to(torch.int32)
        zero = torch.zeros_like(self.steps)
        self.steps_since_last_goal = torch.where(mask, zero, self.steps_since_last_goal)
        self.consecutive_success = torch.where(mask, zero, self.consecutive_success)

    def process(self, is_successful: torch.Tensor, goal_distance_reward: torch.Tensor, live: torch.Tensor = None):
        """:157-241.  Returns (reward [B,3], done [B], new_goal [B], info).  The caller samples the new
        goals for `new_goal` (the reference calls `reset_goal_fn` from inside `process`).
        `live` (optional bool [B]): envs outside it are not stepped by the tracker this call (they are in the
        middle of a pipelined reset): counters untouched, zero reward, not done."""
        if live is None:
            live = torch.ones_like(is_successful)
        one = live.to(torch.int32)
        is_successful = is_successful & live
        self.steps += one
        self.steps_since_last_goal += one
        self.consecutive_success = torch.where(live, torch.where(is_successful, self.consecutive_success + 1, torch.zeros_like(self.consecutive_success)), self.consecutive_success)
        got = (self.consecutive_success >= 1) & live
        success_reward = got.to(torch.float32) * self.success_reward
        self.successes_so_far += got.to(torch.int32)
        timeout = (~got) & live & (self.steps_since_last_goal >= self.max_timesteps_per_goal)
        trial_success = got & (self.successes_so_far >= self.successes_needed)
        done = timeout | trial_success
        self.steps_since_last_goal = torch.where(trial_success, torch.zeros_like(self.steps_since_last_goal), self.steps_since_last_goal)
        new_goal = got & ~trial_success
        goal_reward = goal_distance_reward * live.to(goal_distance_reward.dtype) if self.use_goal_distance_reward else torch.zeros_like(goal_distance_reward)
        reward = torch.stack([torch.zeros_like(goal_reward), goal_reward, success_reward], dim=-1)
        info: Dict[str, torch.Tensor] = {
            "sub_goal_is_successful": got,
            "trial_success": trial_success,
            "goal_reset": new_goal,
            "successes_so_far": self.successes_so_far.clone(),
            # update_info runs after reset_goal_steps in the reference: a fresh goal reports 0
            "steps_since_last_goal": torch.where(new_goal, torch.zeros_like(self.steps_since_last_goal), self.steps_since_last_goal),
            "env_crash": torch.zeros_like(done),
        }
        return reward, done, new_goal, info
