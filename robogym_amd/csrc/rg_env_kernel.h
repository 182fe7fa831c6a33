// rg_env_kernel.h — the part of RobotEnv.step that follows the physics, as ONE launch for the whole batch:
// goal-distance reward, success flag, MultiGoalTracker bookkeeping, goal resampling, `done`, and — with pipelined
// resets — the reset recipe's per-env phase machine with its masked state writes.
//
// Replaces, for the dactyl cube envs (/root/reference/robogym/...):
//   robot_env.py:550-625    _calculate_goal_distance_reward, _is_successful, _get_goal_info
//   utils/multi_goal_tracker.py:157-241   MultiGoalTracker.process (+ reset / reset_goal_steps :83-125)
//   envs/dactyl/goals/locked_parallel.py:32-46   LockedParallelGoal.next_goal (z rotation U(-pi,pi) x one of 24 parallel quats)
//   robot_env.py:893-909    reset_goal (goal counter, new goal, _previous_goal_distance = None, re-observation)
//   envs/dactyl/common/cube_env.py:330-355, envs/dactyl/locked.py:197-225   the reset recipe (pipelined mode)
// One 64-lane workgroup per env: lane 0 does the scalar bookkeeping, the wave does the row writes.
// The host-side mirror (robogym_amd/envs/dactyl/locked.py) used to issue ~100 tiny tensor kernels per env.step for this.
#pragma once
#include "rg_types.h"

// RgPostArgs = rg_post_args of include/rgstep.h (the C ABI struct is the kernel argument)

// counter-based generator: one 32-bit hash per (seed, step, env, k)
__device__ __forceinline__ unsigned rg_hash(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= (c + 0x165667B1u) * 0x27D4EB2Fu; h ^= h >> 13; h *= 0x9E3779B1u;
  h ^= (d + 0xD6E8FEB8u) * 0x85EBCA77u; h ^= h >> 16; h *= 0xC2B2AE3Du; h ^= h >> 15; h *= 0x27D4EB2Fu; h ^= h >> 13;
  return h;
}
__device__ __forceinline__ float rg_u01(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

struct RgPostFlags { int crash, wiggle, restart, newgoal, ok; float gq[4]; };

__global__ void __launch_bounds__(RG_WAVE) rg_post_step_kernel(RgBatchDev bt, RgPostArgs a, int nq, int nv, int nu, int npair) {
#ifdef RG_EMUL
  RgPostFlags& F = *(RgPostFlags*)emul_lds();
#else
  __shared__ RgPostFlags Fs; RgPostFlags& F = Fs;
#endif
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= bt.B) return;
  float* obs = a.obs + (size_t)e * a.obs_dim;
  float* qrow = bt.qpos + (size_t)e * nq;
  // ---- draws of this (step, env)
  auto U = [&](int k) -> float { return a.draws ? a.draws[(size_t)e * RG_POST_NDRAW + k] : rg_u01(rg_hash(a.seed, a.step, (unsigned)e, (unsigned)k)); };
  auto N = [&](int k) -> float {   // standard normal (Box-Muller on two hashes) unless the caller supplied draws
    if (a.draws) return a.draws[(size_t)e * RG_POST_NDRAW + k];
    float u1 = fmaxf(rg_u01(rg_hash(a.seed, a.step, (unsigned)e, (unsigned)(64 + 2 * k))), 1e-7f), u2 = rg_u01(rg_hash(a.seed, a.step, (unsigned)e, (unsigned)(65 + 2 * k)));
    return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
  };
  if (lane == 0) {
    const int crash = (bt.status[e] & RG_STATUS_BAD_STATE) != 0;
    const int ph0 = a.pipelined ? a.phase[e] : 0;
    const int resetting = ph0 > 0, live = !resetting;
    float dist = crash ? 0.f : a.goal_dist[e];
    a.t[e] += live;
    // ---- _get_goal_info (robot_env.py:577-625)
    float gdr = (a.prev_valid[e] && live && !crash) ? a.prev_dist[e] - dist : 0.f;
    if (live) { a.prev_dist[e] = dist; a.prev_valid[e] = 1; }
    int succ = live && !crash && dist < a.success_threshold;
    // ---- MultiGoalTracker.process (multi_goal_tracker.py:157-241), dactyl settings (one successful step suffices)
    int got = 0, trial = 0, timeout = 0, newgoal = 0;
    if (live) {
      a.steps[e] += 1;
      int ssl = a.steps_since_last_goal[e] + 1;
      int cons = succ ? a.consecutive[e] + 1 : 0;
      got = cons >= 1;
      if (got) a.successes_so_far[e] += 1;
      timeout = !got && ssl >= a.max_timesteps_per_goal;
      trial = got && a.successes_so_far[e] >= a.successes_needed;
      if (trial) ssl = 0;
      newgoal = got && !trial;
      a.steps_since_last_goal[e] = ssl; a.consecutive[e] = cons;
    }
    const int fallen = a.stop_on_fall && live && !crash && (a.cube_body_z + obs[2]) < 0.04f;   // cube:center z = body z offset + cube_tz
    int done = timeout || trial || (crash && live) || fallen;
    float* rw = a.reward + 3 * (size_t)e;
    rw[0] = 0.f; rw[1] = (a.use_goal_distance_reward && live) ? gdr : 0.f; rw[2] = got ? a.success_reward : 0.f;
    a.goal_dist_before[e] = dist;
    // ---- reset recipe progression (pipelined mode)
    int wiggle = 0, restart = 0, ok = 0, phase = ph0;
    if (a.pipelined) {
      int ph = ph0 + resetting;
      const int n1 = a.reset_initial_steps, n2 = a.reset_initial_steps + a.n_random_initial_steps;
      wiggle = resetting && ph == n1 + 1 && !crash;
      int finished = resetting && ph == n2 + 1 && !crash;
      int on_palm = (a.cube_body_z + obs[2]) > 0.04f;
      ok = finished && (on_palm || a.tries[e] + 1 >= a.max_pose_resets);
      int retry = (finished && !ok) || (crash && resetting);
      int start = done && live;
      restart = retry || start;
      a.tries[e] = start ? 0 : a.tries[e] + retry;
      phase = restart ? 1 : (ok ? 0 : ph);
      a.phase[e] = phase;
      if (ok) {   // RobotEnv.reset tail (robot_env.py:787-792): tracker.reset, clock, goal generation reset
        a.steps[e] = 0; a.steps_since_last_goal[e] = 0; a.successes_so_far[e] = 0; a.goals_so_far[e] = 0; a.consecutive[e] = 0;
        a.t[e] = 0; a.prev_valid[e] = 0;
      }
      newgoal = newgoal || ok;
    }
    // ---- reset_goal (robot_env.py:893-909) for envs that get a new goal
    float gq[4] = {a.goal_quat[4 * (size_t)e], a.goal_quat[4 * (size_t)e + 1], a.goal_quat[4 * (size_t)e + 2], a.goal_quat[4 * (size_t)e + 3]};
    int achieved = succ;
    if (newgoal) {
      // LockedParallelGoal.next_goal: quat_mul(z rotation by U(-pi, pi), PARALLEL_QUATS[randint(24)])
      float ang = (2.f * U(0) - 1.f) * 3.14159265358979f, cz = cosf(0.5f * ang), sz = sinf(0.5f * ang);
      int ch = (int)(U(1) * 24.f); ch = ch < 0 ? 0 : (ch > 23 ? 23 : ch);
      const float* p = a.parallel_quats + 4 * ch;
      // (cz, 0, 0, sz) * (pw, px, py, pz)
      float w = cz * p[0] - sz * p[3], x = cz * p[1] - sz * p[2], y = cz * p[2] + sz * p[1], z = cz * p[3] + sz * p[0];
      if (a.goal_override) { const float* g = a.goal_override + 4 * (size_t)e; w = g[0]; x = g[1]; y = g[2]; z = g[3]; }
      // stored sign-normalised (w >= 0): that is what the goal_quat observation reports (rotation.quat_normalize), and the
      // goal distance does not see the sign
      if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
      gq[0] = w; gq[1] = x; gq[2] = y; gq[3] = z;
      a.goals_so_far[e] += 1; a.steps_since_last_goal[e] = 0; a.consecutive[e] = 0;
      // re-observation: distance to the new goal becomes the previous distance (update_goal_info), 2 state-less forwards owed
      float cw = qrow[a.cube_quat_col], cx = -qrow[a.cube_quat_col + 1], cy = -qrow[a.cube_quat_col + 2], cq = -qrow[a.cube_quat_col + 3];
      float dw = w * cw - x * cx - y * cy - z * cq;
      float nd = crash ? 0.f : 2.0f * acosf(fminf(fabsf(dw), 1.f));
      a.prev_dist[e] = nd; a.prev_valid[e] = 1;
      achieved = !crash && nd < a.success_threshold;
      a.preticks[e] += 2;
    }
    a.is_successful[e] = achieved;
    a.done[e] = done; a.goal_reset[e] = newgoal && !ok; a.trial_success[e] = trial; a.sub_goal_ok[e] = got; a.env_crash[e] = crash;
    a.resetting[e] = phase > 0; a.episode_started[e] = ok; a.info_ssl[e] = a.steps_since_last_goal[e];
    {   // what the env's next step launch needs to know (simulation_interface.py:176-189 has one forward per recipe step,
        // the forwards after the perturbation and inside on_palm make it two on recipe steps n1 and n2; env.step has three)
      const int n1 = a.reset_initial_steps, n2 = a.reset_initial_steps + a.n_random_initial_steps;
      a.nticks_next[e] = phase == 0 ? 3 : ((phase == n1 || phase == n2) ? 2 : 1);
      a.reset_mask[e] = phase > 0; a.live_mask[e] = phase == 0;
    }
    F.crash = crash; F.wiggle = wiggle; F.restart = restart; F.newgoal = newgoal; F.ok = ok;
    for (int k = 0; k < 4; k++) F.gq[k] = gq[k];
  }
  __syncthreads();
  // ---- row writes by the whole wave
  if (F.crash) for (int i = lane; i < a.obs_dim; i += RG_WAVE) obs[i] = 0.f;
  if (F.newgoal) {
    if (lane < 4) a.goal_quat[4 * (size_t)e + lane] = F.gq[lane];
    for (int i = lane; i < nq; i += RG_WAVE) {
      float v = 0.f;
      if (i >= a.cube_quat_col && i < a.cube_quat_col + 4) v = F.gq[i - a.cube_quat_col];
      if (i == a.cube_pos_col + 2) v = -0.025f;
      a.qpos_goal[(size_t)e * nq + i] = v;
    }
  }
  bool touched = false;
  if (F.wiggle) {   // locked.py:208-216: cube position += N(0, std^2), uniform random orientation, then the random action's ctrl
    if (lane < 3) qrow[a.cube_pos_col + lane] = obs[lane] + N(6 + lane) * a.wiggle_std;
    if (lane == 3) {
      float q0 = N(2), q1 = N(3), q2 = N(4), q3 = N(5), n = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
      float sgn = (q0 < 0 ? -1.f : 1.f) / fmaxf(n, 1e-30f);
      qrow[a.cube_quat_col] = q0 * sgn; qrow[a.cube_quat_col + 1] = q1 * sgn; qrow[a.cube_quat_col + 2] = q2 * sgn; qrow[a.cube_quat_col + 3] = q3 * sgn;
    }
    for (int u = lane; u < nu; u += RG_WAVE) {
      float lo = a.ctrl_lo[u], hi = a.ctrl_hi[u], act = 2.f * U(9 + u) - 1.f;
      if (bt.envprm) { lo = bt.envprm[(size_t)e * RG_NPRM + RG_PRM_ACT_CTRLRANGE + 2 * u]; hi = bt.envprm[(size_t)e * RG_NPRM + RG_PRM_ACT_CTRLRANGE + 2 * u + 1]; }   // the env's own actuator_ctrlrange
      if (a.draws) act = a.draws[(size_t)e * RG_POST_NDRAW + 9 + u];   // supplied draws are the actions themselves (in [-1, 1])
      bt.ctrl[(size_t)e * nu + u] = fminf(fmaxf(0.5f * (hi + lo) + act * 0.5f * (hi - lo), lo), hi);
    }
    touched = true;
  }
  if (F.restart) {   // MjSim.reset of this env + the recipe's first ctrl
    for (int i = lane; i < nq; i += RG_WAVE) qrow[i] = a.qpos0[i];
    for (int i = lane; i < nv; i += RG_WAVE) { bt.qvel[(size_t)e * nv + i] = 0.f; bt.qacc_warmstart[(size_t)e * nv + i] = 0.f; }
    for (int i = lane; i < 3 * nu; i += RG_WAVE) bt.pid[(size_t)e * 3 * nu + i] = 0.f;
    for (int u = lane; u < nu; u += RG_WAVE)
      bt.ctrl[(size_t)e * nu + u] = bt.envprm ? 0.5f * (bt.envprm[(size_t)e * RG_NPRM + RG_PRM_ACT_CTRLRANGE + 2 * u] + bt.envprm[(size_t)e * RG_NPRM + RG_PRM_ACT_CTRLRANGE + 2 * u + 1]) : a.zero_ctrl[u];
    if (lane == 0) { bt.time[e] = 0.f; bt.status[e] = 0; a.preticks[e] = 0; }
    if (bt.envprm) for (int i = lane; i < 6 * RG_MAXBODY; i += RG_WAVE) ((float*)bt.envprm)[(size_t)e * RG_NPRM + RG_PRM_XFRC + i] = 0.f;   // mj_resetData zeroes data.xfrc_applied
    touched = true;
  }
  if (touched && bt.pairlb) for (int i = lane; i < npair; i += RG_WAVE) bt.pairlb[(size_t)e * npair + i] = 0.f;   // qpos written from outside: cache void
  // ---- the packed row a replicated learner consumes (and the multi-GPU all-gather moves)
  if (a.packed) {
    __syncthreads();
    const int pd = a.obs_dim + 3 + 4 + nq + 1 + 3 + 1;
    float* o = a.packed + (size_t)e * pd;
    for (int i = lane; i < a.obs_dim; i += RG_WAVE) o[i] = obs[i];
    if (lane < 3) o[a.obs_dim + lane] = 0.f;
    if (lane < 4) o[a.obs_dim + 3 + lane] = F.newgoal ? F.gq[lane] : a.goal_quat[4 * (size_t)e + lane];   // (stored with w >= 0)
    for (int i = lane; i < nq; i += RG_WAVE) o[a.obs_dim + 7 + i] = a.qpos_goal[(size_t)e * nq + i];
    if (lane == 0) {
      int base = a.obs_dim + 7 + nq;
      o[base] = (float)a.is_successful[e];
      o[base + 1] = a.reward[3 * (size_t)e]; o[base + 2] = a.reward[3 * (size_t)e + 1]; o[base + 3] = a.reward[3 * (size_t)e + 2];
      o[base + 4] = (float)a.done[e];
    }
  }
}
