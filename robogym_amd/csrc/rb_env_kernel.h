// rb_env_kernel.h — the env-level half of RobotEnv.step for dactyl/full_perpendicular (BASELINE.json configs[2]) as ONE
// launch for the whole batch after rb_step_kernel: FaceFreeGoal's distances, goal-distance reward, success flag,
// MultiGoalTracker bookkeeping, `done`, goal generation for the envs that need a new goal — including the manipulation of the
// TARGET cube's 66 joints it entails — and the observation row.
//
// Replaces (/root/reference/robogym/...):
//   envs/dactyl/goals/face_free.py:61-189         FaceFreeGoal.next_goal / relative_goal / goal_distance
//   envs/dactyl/common/cube_utils.py:26-181       uniform_z_aligned_quat, face_up, rotated_face_with_angle, align_quat_up,
//                                                 up_axis_with_sign, distance_quat_from_being_up
//   envs/dactyl/common/cube_manipulator.py:148-187, 377-409   CubeManipulator.rotate_face, soft_align_faces
//   envs/dactyl/full_perpendicular.py:138-155     clone_target_from_cube, align_target_faces, rotate_target_face
//   utils/rotation.py:86-148, 372-384, 461-538    euler2mat, mat2euler, normalize_angles, round_to_straight_angles, vectors2quat, rot_xyz_aligned
//   robot_env.py:550-625, 893-909                 _get_goal_info (reward = sum over the threshold keys), reset_goal
//   utils/multi_goal_tracker.py:157-241           MultiGoalTracker.process
//   envs/dactyl/full_perpendicular.py:177-192     _default_observation_map (the keys the physics and the goal produce)
// One 64-lane workgroup per env: lane 0 decides, lanes 0..19 each own one cubelet of the 3x3x3 cube (its three Euler hinge
// joints), lane 20 the six face drivers; the 66-joint block being manipulated sits in LDS.  The two state-less forwards of
// reset_goal's re-observation only tick the PID state (rb_kernel.h): they are executed here for the envs that got a new goal.
// With `pipelined`, an env whose episode ended restarts by itself: the reset recipe (cube_env.py:330-355, full_perpendicular.py:286-345) as a
// per-env phase counter with its state writes done here (include/rgstep.h, rb_post_args).
// fp32 throughout; the gimbal-lock test of mat2euler uses 4 x FLT_EPSILON where the reference (float64) uses 4 x DBL_EPSILON.
#pragma once
#include "rb_types.h"

namespace rgb {

// RbPostArgs / RbCubeOpsArgs = rb_post_args / rb_cube_ops_args of include/rgstep.h

#define RBC_PI 3.14159265358979f
#define RBC_NCUBELET 20
#define RBC_BLOCK 66          // 6 drivers + 20 x 3 Euler hinges, contiguous in qpos (checked by the host)

__device__ __forceinline__ float rbc_wrap(float a) { return a - 2.f * RBC_PI * floorf((a + RBC_PI) * (0.5f / RBC_PI)); }            // rotation.normalize_angles
__device__ __forceinline__ float rbc_straight(float a) { return rbc_wrap(rintf(a * (2.f / RBC_PI)) * (0.5f * RBC_PI)); }           // rotation.round_to_straight_angles

__device__ __forceinline__ void rbc_euler2mat(float e0, float e1, float e2, float* M) {
  const float a = -e2, b = -e1, c = -e0;
  const float sa = sinf(a), sb = sinf(b), sc = sinf(c), ca = cosf(a), cb = cosf(b), cc = cosf(c);
  M[0] = cb * ca; M[1] = cb * sa; M[2] = -sb;
  M[3] = sb * ca * sc - sa * cc; M[4] = sb * sa * sc + ca * cc; M[5] = cb * sc;
  M[6] = sb * ca * cc + sa * sc; M[7] = sb * sa * cc - ca * sc; M[8] = cb * cc;
}
__device__ __forceinline__ void rbc_mat2euler(const float* M, float* e) {
  const float cy = sqrtf(M[8] * M[8] + M[5] * M[5]);
  e[1] = -atan2f(-M[2], cy);
  if (cy > 4.f * 1.1920929e-7f) { e[0] = -atan2f(M[5], M[8]); e[2] = -atan2f(M[1], M[0]); }
  else { e[0] = 0.f; e[2] = -atan2f(-M[3], M[4]); }
}
__device__ __forceinline__ void rbc_quat2mat(const float* q, float* M) {
  const float w = q[0], x = q[1], y = q[2], z = q[3], n = w * w + x * x + y * y + z * z;
  if (!(n > 1.1920929e-7f)) { for (int i = 0; i < 9; i++) M[i] = (i % 4 == 0) ? 1.f : 0.f; return; }
  const float s = 2.f / n;
  M[0] = 1 - s * (y * y + z * z); M[1] = s * (x * y - w * z); M[2] = s * (x * z + w * y);
  M[3] = s * (x * y + w * z); M[4] = 1 - s * (x * x + z * z); M[5] = s * (y * z - w * x);
  M[6] = s * (x * z - w * y); M[7] = s * (y * z + w * x); M[8] = 1 - s * (x * x + y * y);
}
__device__ __forceinline__ void rbc_qmul(const float* p, const float* q, float* r) {
  r[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3]; r[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
  r[2] = p[0] * q[2] + p[2] * q[0] + p[3] * q[1] - p[1] * q[3]; r[3] = p[0] * q[3] + p[3] * q[0] + p[1] * q[2] - p[2] * q[1];
}
__device__ __forceinline__ void rbc_qsign(float* q) { if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; } }
// rotation.quat_magnitude = 2 acos(w) of a unit quaternion with w >= 0, evaluated as 2 atan2(|v|, w): the same number, but resolved to ~1e-7
// near zero where fp32 acos(w -> 1) resolves only ~3e-4 (the reference computes in float64)
__device__ __forceinline__ float rbc_qmag(const float* q) { return 2.f * atan2f(sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), fmaxf(q[0], 0.f)); }
// rotation.vectors2quat(v, z_up) for a unit vector v: the shortest arc that takes v to +z
__device__ __forceinline__ void rbc_to_up(const float* v, float* q) {
  q[0] = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + v[2]; q[1] = v[1]; q[2] = -v[0]; q[3] = 0.f;
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n < 1e-6f) {   // v = -z: half a turn about anything orthogonal (rotation.any_orthogonal: cross with the basis vector of the smallest component)
    int k = fabsf(v[1]) < fabsf(v[0]) ? 1 : 0; if (fabsf(v[2]) < fabsf(v[k])) k = 2;
    float b[3] = {0, 0, 0}; b[k] = 1.f;
    q[0] = 0.f; q[1] = v[1] * b[2] - v[2] * b[1]; q[2] = v[2] * b[0] - v[0] * b[2]; q[3] = v[0] * b[1] - v[1] * b[0];
    n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  }
  const float r = 1.f / n;
  for (int i = 0; i < 4; i++) q[i] *= r;
  rbc_qsign(q);
}
// cube_utils.up_axis_with_sign: the cube axis closest to +z and its sign
__device__ __forceinline__ void rbc_up_axis(const float* M, int& k, float& sgn) {
  k = 0; float best = fabsf(M[6]);
  if (fabsf(M[7]) > best) { k = 1; best = fabsf(M[7]); }
  if (fabsf(M[8]) > best) k = 2;
  sgn = M[6 + k] > 0 ? 1.f : (M[6 + k] < 0 ? -1.f : 0.f);
}
// cube_utils.distance_quat_from_being_up
__device__ __forceinline__ void rbc_dist_up(const float* M, int k, float sgn, float* q) { float v[3] = {M[k] * sgn, M[3 + k] * sgn, M[6 + k] * sgn}; rbc_to_up(v, q); }

// CubeManipulator.rotate_face on the 66-joint block `c` in LDS.  tab[k] = {offset of rotx, roty, rotz in the block, coords x, y, z}.
// Every argument is wave-uniform.  Ends with a barrier.
__device__ __forceinline__ void rbc_rotate_face(float* c, const int* tab, int axis, int side, float angle, bool drivers, int lane) {
  angle = rbc_wrap(angle);
  if (fabsf(angle) < 1e-4f) return;
  const float sgn = (float)(2 * side - 1);
  if (lane < RBC_NCUBELET) {
    const int* t = tab + 6 * lane;
    float M[9], T[9], N[9], e[3];
    rbc_euler2mat(c[t[0]], c[t[1]], c[t[2]], M);
    const float cur = M[3 * axis] * (float)t[3] + M[3 * axis + 1] * (float)t[4] + M[3 * axis + 2] * (float)t[5];
    if (cur * sgn > 0.5f) {
      rbc_euler2mat(axis == 0 ? angle : 0.f, axis == 1 ? angle : 0.f, axis == 2 ? angle : 0.f, T);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) N[3 * i + j] = T[3 * i] * M[j] + T[3 * i + 1] * M[3 + j] + T[3 * i + 2] * M[6 + j];
      rbc_mat2euler(N, e);
      c[t[0]] = e[0]; c[t[1]] = e[1]; c[t[2]] = e[2];
    }
  } else if (lane == RBC_NCUBELET && drivers) c[2 * axis + side] += angle;
  __syncthreads();
}
// CubeManipulator.soft_align_faces: faces to the nearest straight angles, largest correction first, then every cubelet's matrix rounded
__device__ __forceinline__ void rbc_soft_align(float* c, const int* tab, int lane) {
  float diff[6]; bool used[6];
  for (int k = 0; k < 6; k++) { diff[k] = rbc_wrap(rbc_straight(c[k]) - c[k]); used[k] = false; }
  __syncthreads();
  for (int r = 0; r < 6; r++) {
    int pick = -1;
    for (int k = 0; k < 6; k++) if (!used[k] && (pick < 0 || fabsf(diff[k]) >= fabsf(diff[pick]))) pick = k;   // descending |diff|, ties: larger index first
    used[pick] = true;
    rbc_rotate_face(c, tab, pick >> 1, pick & 1, diff[pick], true, lane);
  }
  if (lane < RBC_NCUBELET) {
    const int* t = tab + 6 * lane;
    float M[9], e[3];
    rbc_euler2mat(c[t[0]], c[t[1]], c[t[2]], M);
    for (int i = 0; i < 9; i++) M[i] = rintf(M[i]);
    rbc_mat2euler(M, e);
    c[t[0]] = e[0]; c[t[1]] = e[1]; c[t[2]] = e[2];
  }
  __syncthreads();
}

// ---- test / reset hook: a list of CubeManipulator operations per env on one of the two cubes
// ops[e][k] = {axis, side, angle, code}: code 0 rotate_face, 1 rotate_face without the driver joint (scramble), 2 soft_align_faces, < 0 nothing
__global__ void __launch_bounds__(64) rb_cube_ops_kernel(RbBatchDev bt, int nq, int block_col, const int* tab, const float* ops, int nops) {
#ifdef RG_EMUL
  float* c = (float*)emul_lds();
#else
  __shared__ float cs[RBC_BLOCK + 2]; float* c = cs;
#endif
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= bt.B || (bt.active && !bt.active[e])) return;
  float* q = bt.qpos + (size_t)e * nq + block_col;
  for (int i = lane; i < RBC_BLOCK; i += 64) c[i] = q[i];
  __syncthreads();
  for (int k = 0; k < nops; k++) {
    const float* o = ops + ((size_t)e * nops + k) * 4;
    const int code = (int)o[3];
    if (code == 0 || code == 1) rbc_rotate_face(c, tab, (int)o[0], (int)o[1], o[2], code == 0, lane);
    else if (code == 2) rbc_soft_align(c, tab, lane);
  }
  for (int i = lane; i < RBC_BLOCK; i += 64) q[i] = c[i];
}

// counter-based generator (as rg_env_kernel.h)
__device__ __forceinline__ unsigned rbp_hash(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= (c + 0x165667B1u) * 0x27D4EB2Fu; h ^= h >> 13; h *= 0x9E3779B1u;
  h ^= (d + 0xD6E8FEB8u) * 0x85EBCA77u; h ^= h >> 16; h *= 0xC2B2AE3Du; h ^= h >> 15; h *= 0x27D4EB2Fu; h ^= h >> 13;
  return h;
}

struct RbPostLds {
  float c[RBC_BLOCK + 2];
  float gq[4], gf[6], delta;
  int crash, newgoal, rotate, face;
  int restart, wiggle;   // pipelined resets: this env restarts (MjSim.reset + zero-action ctrl) / gets the recipe's state writes
};

// FaceFreeGoal.goal_distance of the state (quat, face) to the goal row g: out[0] = cube_quat, out[1] = cube_face_angle
__device__ __forceinline__ void rbp_goal_distance(const float* g, const float* quat, const float* face, float* out, int goal_mode) {
  float M[9], dq[4] = {1.f, 0.f, 0.f, 0.f};
  if (goal_mode == 1) {}                                        // FullUnconstrainedGoal: no orientation objective
  else if (g[10] > 0.5f && goal_mode != 2) { rbc_quat2mat(quat, M); rbc_dist_up(M, (int)g[11], g[12], dq); }   // (goal_mode 2, FaceCurriculumGoal: always the plain difference)
  else { float cj[4] = {quat[0], -quat[1], -quat[2], -quat[3]}; rbc_qmul(g, cj, dq); rbc_qsign(dq); }
  out[0] = rbc_qmag(dq);
  float s2 = 0; for (int k = 0; k < 6; k++) { const float d = rbc_wrap(g[4 + k] - face[k]); s2 += d * d; }
  out[1] = sqrtf(s2);
}

__global__ void __launch_bounds__(64) rb_post_step_kernel(const RbModelDev* mp, RbBatchDev bt, RbPostArgs a) {
#ifdef RG_EMUL
  RbPostLds& F = *(RbPostLds*)emul_lds();
#else
  __shared__ RbPostLds Fs; RbPostLds& F = Fs;
#endif
  const RbModelDev& m = *mp;
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= bt.B) return;
  const int forced = a.force_new_goal != 0;
  if (forced && !a.force_new_goal[e]) return;
  const int nq = m.nq, nu = m.nu;
  float* qrow = bt.qpos + (size_t)e * nq;
  const float* S = bt.scratch + (size_t)e * m.scratch_words;
  float* g = a.goal + (size_t)e * RB_GOAL_WORDS;
  auto U = [&](int k) -> float { return a.draws ? a.draws[(size_t)e * RB_POST_NDRAW + k] : (float)(rbp_hash(a.seed, a.step, (unsigned)e, (unsigned)k) >> 8) * (1.0f / 16777216.0f); };
  if (lane == 0) {
    const int crash = (bt.status[e] & RG_STATUS_BAD_STATE) != 0;
    float quat[4], face[6], dist[2] = {0.f, 0.f};
    for (int k = 0; k < 4; k++) quat[k] = qrow[a.cube_quat_col + k];
    for (int k = 0; k < 6; k++) face[k] = qrow[a.cube_block_col + k];
    if (!crash) rbp_goal_distance(g, quat, face, dist, a.goal_mode);
    int got = 0, trial = 0, timeout = 0, newgoal = forced, succ = 0;
    const int ph0 = (a.pipelined && !forced) ? a.phase[e] : 0;
    const int resetting = ph0 > 0, live = !resetting;
    int done = 0;
    if (!forced && !live) {   // inside the reset recipe: between episodes
      float* rw = a.reward + 3 * (size_t)e; rw[0] = rw[1] = rw[2] = 0.f;
      a.goal_dist[2 * e] = a.goal_dist[2 * e + 1] = 0.f;
      a.done[e] = 0; a.trial_success[e] = 0; a.sub_goal_ok[e] = 0; a.env_crash[e] = crash;
    }
    if (!forced && live) {
      a.t[e] += 1;
      // ---- _get_goal_info (robot_env.py:577-625): reward = sum over the keys of success_threshold of (previous - current) distance
      float gdr = 0.f;
      if (a.prev_valid[e] && !crash) gdr = (a.prev_dist[2 * e] - dist[0]) + (a.prev_dist[2 * e + 1] - dist[1]);
      a.prev_dist[2 * e] = dist[0]; a.prev_dist[2 * e + 1] = dist[1]; a.prev_valid[e] = 1;
      succ = !crash && dist[0] < a.quat_threshold && dist[1] < a.face_threshold;
      // ---- MultiGoalTracker.process (multi_goal_tracker.py:157-241), dactyl settings
      a.steps[e] += 1;
      int ssl = a.steps_since_last_goal[e] + 1;
      const int cons = succ ? a.consecutive[e] + 1 : 0;
      got = cons >= 1;
      if (got) a.successes_so_far[e] += 1;
      timeout = !got && ssl >= a.max_timesteps_per_goal;
      trial = got && a.successes_so_far[e] >= a.successes_needed;
      if (trial) ssl = 0;
      newgoal = got && !trial;
      a.steps_since_last_goal[e] = ssl; a.consecutive[e] = cons;
      const int fallen = a.stop_on_fall && !crash && S[m.off[RB_O_SPOS] + 3 * a.center_site + 2] < 0.04f;   // cube_utils.on_palm
      float* rw = a.reward + 3 * (size_t)e;
      rw[0] = 0.f; rw[1] = a.use_goal_distance_reward ? gdr : 0.f; rw[2] = got ? a.success_reward : 0.f;
      a.goal_dist[2 * e] = dist[0]; a.goal_dist[2 * e + 1] = dist[1];
      done = timeout || trial || crash || fallen;
      a.done[e] = done;
      a.trial_success[e] = trial; a.sub_goal_ok[e] = got; a.env_crash[e] = crash;
    }
    // ---- reset recipe progression (pipelined mode)
    F.restart = 0; F.wiggle = 0;
    if (a.pipelined && !forced) {
      const int ph = ph0 + resetting;
      const int n1 = a.reset_initial_steps, n2 = a.reset_initial_steps + a.n_random_initial_steps;
      const int wiggle = resetting && ph == n1 + 1 && !crash;
      const int finished = resetting && ph == n2 + 1 && !crash;
      const int on_palm = S[m.off[RB_O_SPOS] + 3 * a.center_site + 2] > 0.04f;      // (the launch's last forward is the one inside on_palm)
      const int ok = finished && (on_palm || a.tries[e] + 1 >= a.max_pose_resets);
      const int retry = (finished && !ok) || (crash && resetting);
      const int start = done && live;
      const int restart = retry || start;
      a.tries[e] = start ? 0 : a.tries[e] + retry;
      const int phase = restart ? 1 : (ok ? 0 : ph);
      a.phase[e] = phase;
      if (ok) {   // RobotEnv.reset tail (robot_env.py:787-792): tracker.reset, clock, then reset_goal below
        a.steps[e] = 0; a.steps_since_last_goal[e] = 0; a.successes_so_far[e] = 0; a.goals_so_far[e] = 0; a.consecutive[e] = 0;
        a.t[e] = 0; a.prev_valid[e] = 0;
      }
      newgoal = newgoal || ok;
      a.resetting[e] = phase > 0; a.episode_started[e] = ok;
      a.hold_next[e] = phase > 0;
      a.nticks_next[e] = phase == 0 ? 3 : ((phase == n1 || phase == n2) ? 2 : 1);   // (the forwards after the state writes and inside on_palm: second ticks of recipe steps n1, n2)
      F.restart = restart; F.wiggle = wiggle;
    }
    int achieved = succ;
    F.rotate = 0; F.face = 0; F.delta = 0.f;
    if (newgoal) {
      // ---- FaceFreeGoal.next_goal (face_free.py:61-136)
      float M[9], s2 = 0.f, rounded[6];
      for (int k = 0; k < 6; k++) { rounded[k] = rbc_straight(face[k]); const float d = rbc_wrap(face[k] - rounded[k]); s2 += d * d; }
      const bool face_aligned = sqrtf(s2) < a.face_threshold;
      rbc_quat2mat(quat, M);
      int axis; float asgn, dq[4];
      rbc_up_axis(M, axis, asgn);
      rbc_dist_up(M, axis, asgn, dq);
      const bool z_aligned = rbc_qmag(dq) < a.quat_threshold;                 // rotation.rot_xyz_aligned
      const bool reorient = a.goal_mode == 1 ? false : U(0) < a.p_face_flip;
      const bool rotate = a.goal_mode == 1 || (face_aligned && z_aligned && !reorient);
      float gq[4], gf[6];
      if (rotate) {
        int f = 0; float zb = S[m.off[RB_O_GPOS] + 3 * a.face_geom[0] + 2];                // cube_utils.face_up: the face geom highest in z
        for (int k = 1; k < 6; k++) { const float z = S[m.off[RB_O_GPOS] + 3 * a.face_geom[k] + 2]; if (z > zb) { zb = z; f = k; } }
        if (a.goal_mode == 1) { f = a.draws ? (int)U(3) : (int)(U(3) * 6.f); f = f < 0 ? 0 : (f > 5 ? 5 : f); }   // FullUnconstrainedGoal: any face (full_unconstrained.py:62)
        const float cw = (f & 1) ? -1.f : 1.f;
        float dirs[2]; int nd = 0;
        if (a.directions & 1) dirs[nd++] = 0.5f * RBC_PI * cw;
        if (a.directions & 2) dirs[nd++] = -0.5f * RBC_PI * cw;
        for (int k = 0; k < 6; k++) gf[k] = face[k];
        float delta;
        if (U(1) < a.round_target_face) {                                    // cube_utils.rotated_face_with_angle
          int k = a.draws ? (int)U(2) : (int)(U(2) * nd); k = k < 0 ? 0 : (k >= nd ? nd - 1 : k);
          delta = dirs[k];
          gf[f] += delta;
          for (int i = 0; i < 6; i++) gf[i] = rbc_straight(rbc_wrap(gf[i]));
        } else {
          float lo = 0.f, hi = 0.f; for (int k = 0; k < nd; k++) { lo = fminf(lo, dirs[k]); hi = fmaxf(hi, dirs[k]); }
          const float u = a.draws ? U(2) : U(5);
          delta = lo + (hi - lo) * u;
          gf[f] += delta;
          for (int i = 0; i < 6; i++) gf[i] = rbc_wrap(gf[i]);
        }
        rbc_qmul(dq, quat, gq);                                              // cube_utils.align_quat_up
        if (a.goal_mode == 1) gq[0] = gq[1] = gq[2] = gq[3] = 0.f;           // (no orientation goal: np.zeros(4))
        if (a.goal_mode == 2) {                                              // FaceCurriculumGoal: rotation.round_to_straight_quat(cube_quat)
          float eu[3];
          rbc_mat2euler(M, eu);
          const float ai = 0.5f * rbc_straight(eu[2]), aj = -0.5f * rbc_straight(eu[1]), ak = 0.5f * rbc_straight(eu[0]);
          const float si = sinf(ai), sj = sinf(aj), sk = sinf(ak), ci = cosf(ai), cj = cosf(aj), ck = cosf(ak);
          gq[0] = cj * ci * ck + sj * si * sk; gq[1] = cj * ci * sk - sj * si * ck; gq[2] = -(cj * si * sk + sj * ci * ck); gq[3] = cj * si * ck - sj * ci * sk;
        }
        F.face = f; F.delta = delta;
      } else {
        for (int k = 0; k < 6; k++) gf[k] = rounded[k];
        int f = a.draws ? (int)U(3) : (int)(U(3) * 6.f); f = f < 0 ? 0 : (f > 5 ? 5 : f);
        const float ang = a.draws ? U(4) : (2.f * U(4) - 1.f) * RBC_PI;      // cube_utils.uniform_z_aligned_quat
        float zq[4] = {cosf(0.5f * ang), 0.f, 0.f, sinf(0.5f * ang)};
        rbc_qsign(zq);
        rbc_qmul(zq, a.face_up_quats + 4 * f, gq);
      }
      rbc_qsign(gq);
      for (int k = 0; k < 4; k++) g[k] = gq[k];
      for (int k = 0; k < 6; k++) g[4 + k] = gf[k];
      g[10] = rotate ? 1.f : 0.f; g[11] = (float)axis; g[12] = asgn;
      F.rotate = rotate;
      // ---- reset_goal (robot_env.py:893-909): goal counters, _previous_goal_distance = None -> the re-observation sets it to the current distance
      a.goals_so_far[e] += 1; a.steps_since_last_goal[e] = 0; a.consecutive[e] = 0;
      float nd2[2] = {0.f, 0.f};
      if (!crash) rbp_goal_distance(g, quat, face, nd2, a.goal_mode);
      a.prev_dist[2 * e] = nd2[0]; a.prev_dist[2 * e + 1] = nd2[1]; a.prev_valid[e] = 1;
      achieved = !crash && nd2[0] < a.quat_threshold && nd2[1] < a.face_threshold;
    }
    for (int k = 0; k < 4; k++) F.gq[k] = g[k];
    for (int k = 0; k < 6; k++) F.gf[k] = g[4 + k];
    a.is_successful[e] = achieved;
    a.goal_reset[e] = newgoal && !forced && live;
    a.info_ssl[e] = a.steps_since_last_goal[e];
    F.crash = crash; F.newgoal = newgoal;
  }
  __syncthreads();
  if (F.newgoal) {
    // clone_target_from_cube, align_target_faces, rotate_target_face: the target's joints follow the cube's, aligned, with the goal's turn
    for (int i = lane; i < RBC_BLOCK; i += 64) F.c[i] = qrow[a.cube_block_col + i];
    __syncthreads();
    rbc_soft_align(F.c, a.cube_tab, lane);
    if (F.rotate) rbc_rotate_face(F.c, a.cube_tab, F.face >> 1, F.face & 1, F.delta, true, lane);
    for (int i = lane; i < RBC_BLOCK; i += 64) qrow[a.target_block_col + i] = F.c[i];
    // the two state-less forwards of the re-observation (_observe_sync): PID ticks on the unchanged hand state
    if (lane < nu && !F.crash) {
      const int u = lane, id = m.actuator_trnid[u];
      const float len = m.actuator_trntype[u] == 0 ? qrow[m.jnt_qposadr[id]] : S[m.off[RB_O_TENLEN] + id];
      float* st = bt.pid + ((size_t)e * nu + u) * 3;
      // (the env's own gains when the model carries per-env parameter rows, rb_types.h RB_P_*)
      const float* gainprm = m.prm_on ? S + m.prm_off[RB_P_ACT_GAINPRM] : m.actuator_gainprm;
      const float* forcerange = m.prm_on ? S + m.prm_off[RB_P_ACT_FORCERANGE] : m.actuator_forcerange;
      for (int k = 0; k < 2; k++) rb_pid_tick(m, gainprm, forcerange, u, bt.ctrl[(size_t)e * nu + u], len, st);
    }
  }
  // ---- state part of the observation row: cube_pos 3 | cube_quat 4 (w >= 0) | cube_face_angle 6 (wrapped) | hand_angle | fingertip_pos 15 (| goal_pos 3 |
  // goal_quat 4 | goal_face_angle 6 below).  Written BEFORE the pipelined recipe's state writes (restart: qpos <- qpos0; wiggle), so that an env whose
  // episode ends in this step returns its TERMINAL observation, as RobotEnv.step does (robot_env.py:804-844) — and so that no lane reads a state row another
  // lane is rewriting.  (`observe()`'s qpos / qvel keys are views of the state rows themselves: for a restarted env they show the reset state.)
  {
    float* o = a.obs + (size_t)e * a.obs_dim;
    if (lane < 3) o[lane] = qrow[a.cube_pos_col + lane];
    if (lane < 4) o[3 + lane] = (qrow[a.cube_quat_col] < 0 ? -1.f : 1.f) * qrow[a.cube_quat_col + lane];
    if (lane < 6) o[7 + lane] = rbc_wrap(qrow[a.cube_block_col + lane]);
    for (int i = lane; i < a.n_hand; i += 64) o[13 + i] = qrow[a.hand_col + i];
    const int ot = 13 + a.n_hand;
    if (lane < 5) {   // fingertips relative to the three reference sites (hand_forward_kinematics.py:39-50)
      const float* sp = S + m.off[RB_O_SPOS];
      v3 r0 = ld3(sp + 3 * a.ref_site[0]), r1 = ld3(sp + 3 * a.ref_site[1]), r2 = ld3(sp + 3 * a.ref_site[2]);
      v3 ax = normalized(r0 - r1), cx = normalized(r2 - r1), bx = cross(ax, cx);
      v3 t = ld3(sp + 3 * a.tip_site[lane]) - r1;
      o[ot + 3 * lane] = dot(t, ax); o[ot + 3 * lane + 1] = dot(t, bx); o[ot + 3 * lane + 2] = dot(t, cx);
    }
  }
  __syncthreads();
  if (a.pipelined && !forced) {
    const int nv = m.nv;
    auto RD = [&](int k) -> float { return a.reset_draws[(size_t)e * RB_RESET_NDRAW + k]; };
    auto HU = [&](int k) -> float { return (float)(rbp_hash(a.seed ^ 0x5bd1e995u, a.step, (unsigned)e, (unsigned)k) >> 8) * (1.0f / 16777216.0f); };
    auto HN = [&](int k) -> float { const float u1 = fmaxf(HU(200 + 2 * k), 1e-7f), u2 = HU(201 + 2 * k); return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); };
    if (F.wiggle) {   // full_perpendicular.py:311-332 after the settling steps
      if (lane < 3) qrow[a.cube_pos_col + lane] += (a.reset_draws ? RD(lane) : HN(lane)) * a.wiggle_std;
      if (lane == 3) {   // rotation.uniform_quat: four normals, normalised, w >= 0
        float q[4], n2 = 0.f;
        for (int k = 0; k < 4; k++) { q[k] = a.reset_draws ? RD(3 + k) : HN(3 + k); n2 += q[k] * q[k]; }
        const float sc = (q[0] < 0 ? -1.f : 1.f) / sqrtf(fmaxf(n2, 1e-30f));
        for (int k = 0; k < 4; k++) qrow[a.cube_quat_col + k] = q[k] * sc;
      }
      // _scramble_cube + from_pycuber: face turns on signed permutation matrices, one cubelet per lane, then the hinge angles
      for (int i = lane; i < RBC_BLOCK; i += 64) F.c[i] = 0.f;
      __syncthreads();
      if (lane < RBC_NCUBELET) {
        const int* t = a.cube_tab + 6 * lane;
        int M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int it = 0; it < a.num_scramble_steps; it++) {
          int act = a.reset_draws ? (int)RD(7 + it) : (int)(HU(it) * 12.f); act = act < 0 ? 0 : (act > 11 ? 11 : act);
          const int axis = act >> 2, side = (act >> 1) & 1, prime = act & 1, sgn = 2 * side - 1;
          if ((M[3 * axis] * t[3] + M[3 * axis + 1] * t[4] + M[3 * axis + 2] * t[5]) * sgn <= 0) continue;
          const int quarter = -sgn * (prime ? -1 : 1), i = (axis + 1) % 3, j = (axis + 2) % 3;     // a clockwise turn seen from outside = -90 degrees about the outward normal
          for (int c = 0; c < 3; c++) { const int ri = M[3 * i + c], rj = M[3 * j + c]; M[3 * i + c] = -quarter * rj; M[3 * j + c] = quarter * ri; }
        }
        float Mf[9], eu[3];
        for (int k = 0; k < 9; k++) Mf[k] = (float)M[k];
        rbc_mat2euler(Mf, eu);
        F.c[t[0]] = eu[0]; F.c[t[1]] = eu[1]; F.c[t[2]] = eu[2];
      } else if (lane == RBC_NCUBELET && a.scramble_face_angles) {
        for (int k = 0; k < 6; k++) { const float mult = a.reset_draws ? RD(57 + k) : floorf(HU(60 + k) * 5.f) - 2.f; F.c[k] = mult * (0.5f * RBC_PI); }
      }
      __syncthreads();
      if (a.randomize_face_angles) {
        int axis = a.reset_draws ? (int)RD(65) : (int)(HU(70) * 3.f); axis = axis < 0 ? 0 : (axis > 2 ? 2 : axis);
        for (int side = 0; side < 2; side++)
          rbc_rotate_face(F.c, a.cube_tab, axis, side, a.reset_draws ? RD(63 + side) : (HU(71 + side) - 0.5f) * (0.5f * RBC_PI), true, lane);
      }
      for (int i = lane; i < RBC_BLOCK; i += 64) qrow[a.cube_block_col + i] = F.c[i];
      if (lane < nu) {   // the random action's ctrl, absolute (denormalize_position_control with its default relative_action = False)
        const float lo = a.ctrl_lo[lane], hi = a.ctrl_hi[lane], act = a.reset_draws ? RD(66 + lane) : 2.f * HU(80 + lane) - 1.f;
        bt.ctrl[(size_t)e * nu + lane] = fminf(fmaxf(0.5f * (hi + lo) + act * 0.5f * (hi - lo), lo), hi);
      }
    }
    if (F.restart) {   // mujoco_simulation.reset() of this env + the recipe's first ctrl (zero action)
      for (int i = lane; i < nq; i += 64) qrow[i] = a.qpos0[i];
      for (int i = lane; i < nv; i += 64) { bt.qvel[(size_t)e * nv + i] = 0.f; bt.qacc_warmstart[(size_t)e * nv + i] = 0.f; }
      for (int i = lane; i < 3 * nu; i += 64) bt.pid[(size_t)e * 3 * nu + i] = 0.f;
      for (int u = lane; u < nu; u += 64) bt.ctrl[(size_t)e * nu + u] = 0.5f * (a.ctrl_lo[u] + a.ctrl_hi[u]);
      if (lane == 0) { bt.time[e] = 0.f; bt.status[e] = 0; }
    }
  }
  // ---- goal part of the observation row (written last: a new goal of this step is part of it); a crashed env returns a zero row
  float* o = a.obs + (size_t)e * a.obs_dim;
  if (F.crash) { for (int i = lane; i < a.obs_dim; i += 64) o[i] = 0.f; return; }
  const int ot = 13 + a.n_hand;
  if (lane < 3) o[ot + 15 + lane] = 0.f;
  if (lane < 4) o[ot + 18 + lane] = F.gq[lane];
  if (lane < 6) o[ot + 22 + lane] = F.gf[lane];
}

}  // namespace rgb
