// ra_env_kernel.h — the env-level half of RearrangeEnv.step (rearrange/blocks and friends; BASELINE.json configs[3]) after the two physics
// launches (rb_batch_step_tcp on the TCP solver's world, rb_batch_step_ex on the main world with a full final forward): one 64-lane
// workgroup per env.  Reference call sites:
//   RobotEnv._observe_sync / get_observation / step_finalize     /root/reference/robogym/robot_env.py:672-688, 804-880
//   RearrangeEnv._observe_simple                                  envs/rearrange/common/base.py:376-421 (24 keys, 289 scalars at 5 objects)
//   object / robot read-outs                                      envs/rearrange/simulation/base.py:420-480, robot/ur16e/mujoco/joint_controlled_arm.py:20-85
//   contact scans                                                 envs/rearrange/simulation/base.py:592-635 (object - finger pads),
//                                                                 robot/ur16e/mujoco/simulation/base.py:142-167 (gripper - table plane)
//   check_objects_off_table                                       envs/rearrange/simulation/base.py:805-832
//   reward / done                                                 envs/rearrange/common/base.py:768-795, 824-848
//   ObjectStateGoal.relative_goal / goal_distance                 envs/rearrange/goals/object_state.py:492-599 (rot_dist_type "full", all objects distinct)
//   _get_goal_info, MultiGoalTracker.process                      robot_env.py:577-625, utils/multi_goal_tracker.py:157-241
//   JointControlledTcpArm.on_observations_updated                 robot/ur16e/mujoco/joint_controlled_tcp_arm.py:114-129 (gripper state -> solver world)
// Rotation helpers follow robogym/utils/rotation.py (mat2euler, quat2mat, normalize_angles) as rb_env_kernel.h's rbc_* do.
#pragma once
#include "rb_env_kernel.h"

namespace rgb {

__global__ void __launch_bounds__(64) ra_post_step_kernel(const RbModelDev* mp, RbBatchDev bt, RaPostArgs a) {
  const RbModelDev& m = *mp;
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= bt.B) return;
  if (a.frozen && a.frozen[e] == 2) return;      // not this env's turn (re-observation of selected envs only)
  const int nq = m.nq, N = a.num_objects;
  const float* qrow = bt.qpos + (size_t)e * nq;
  const float* vrow = bt.qvel + (size_t)e * m.nv;
  const float* crow = bt.ctrl + (size_t)e * m.nu;
  const float* S = bt.scratch + (size_t)e * m.scratch_words;
  const float *xpos = S + m.off[RB_O_XPOS], *xquat = S + m.off[RB_O_XQUAT], *cvel = S + m.off[RB_O_CVEL], *rootcom = S + m.off[RB_O_ROOTCOM];
  const float* sens = bt.sensordata + (size_t)e * m.nsensordata;
  float* row = a.obs + (size_t)e * (a.obs_dim + 4);
  const int crash = (bt.status[e] & RG_STATUS_BAD_STATE) != 0;
  // world velocity of a body frame's origin (mujoco-py body_xvelp / body_xvelr: Jacobian of the origin times qvel)
  auto body_vel = [&](int b, float* vp, float* vr) {
    const float* c = cvel + 6 * b; const float* o = rootcom + 3 * m.body_rootid[b];
    const float dx = xpos[3 * b] - o[0], dy = xpos[3 * b + 1] - o[1], dz = xpos[3 * b + 2] - o[2];
    vr[0] = c[0]; vr[1] = c[1]; vr[2] = c[2];
    vp[0] = c[3] + (c[1] * dz - c[2] * dy); vp[1] = c[4] + (c[2] * dx - c[0] * dz); vp[2] = c[5] + (c[0] * dy - c[1] * dx);
  };
  float tcp_vp[3], tcp_vr[3];
  body_vel(a.tcp_body, tcp_vp, tcp_vr);
  const float* tcp = xpos + 3 * a.tcp_body;
  // ---- per object (lane k < N): pose, velocities, relative goal, distances, finger contacts, off-table test
  int ok_obj = 0, off_obj = 0;
  float dpos = 0.f, drot = 0.f;
  if (lane < N) {
    const int b = a.obj_body[lane];
    float M[9], eul[3], vp[3], vr[3];
    rbc_quat2mat(xquat + 4 * b, M);
    rbc_mat2euler(M, eul);
    for (int k = 0; k < 3; k++) eul[k] = rbc_wrap(eul[k]);
    body_vel(b, vp, vr);
    const float* gp = a.goal + ((size_t)e * N + lane) * 7;
    float qc[4] = {xquat[4 * b], -xquat[4 * b + 1], -xquat[4 * b + 2], -xquat[4 * b + 3]}, qd[4], Md[9], rel[3];
    rbc_qmul(gp + 3, qc, qd);                 // subtract_euler(goal, current) = quat2euler(q_goal conj(q_obj))
    rbc_quat2mat(qd, Md); rbc_mat2euler(Md, rel);
    for (int k = 0; k < 3; k++) rel[k] = rbc_wrap(rel[k]);
    const float rx = gp[0] - xpos[3 * b], ry = gp[1] - xpos[3 * b + 1], rz = gp[2] - xpos[3 * b + 2];
    dpos = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz) + a.goal_pos_offset, 0.f);
    rbc_qsign(qd);
    { const float n = sqrtf(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]); for (int k = 0; k < 4; k++) qd[k] /= n; }
    drot = a.goal_rot_weight * rbc_qmag(qd);   // quat_magnitude(quat_normalize(euler2quat(rel))): the angle of the same rotation
    ok_obj = !crash && dpos < a.pos_threshold && drot < a.rot_threshold;
    off_obj = xpos[3 * b + 2] < a.table_height * 0.75f || xpos[3 * b] < a.table_min[0] || xpos[3 * b] > a.table_max[0] || xpos[3 * b + 1] < a.table_min[1] || xpos[3 * b + 1] > a.table_max[1];
    float* o = row;
    for (int k = 0; k < 3; k++) {
      o[3 * lane + k] = xpos[3 * b + k]; o[3 * N + 3 * lane + k] = xpos[3 * b + k] - tcp[k]; o[6 * N + 3 * lane + k] = vp[k] - tcp_vp[k];
      o[9 * N + 3 * lane + k] = eul[k]; o[12 * N + 3 * lane + k] = vr[k];
    }
    const int g0 = 15 * N + 15 + 2 * nq;
    for (int k = 0; k < 3; k++) {
      o[g0 + 3 * lane + k] = gp[k]; o[g0 + 3 * N + 3 * lane + k] = a.goal_rot[((size_t)e * N + lane) * 3 + k];
      o[g0 + 6 * N + 1 + 3 * lane + k] = k == 0 ? rx : (k == 1 ? ry : rz); o[g0 + 9 * N + 1 + 3 * lane + k] = rel[k];
    }
    // obj_gripper_contact: any contact (dist < 1e-5) between one of the object's geoms and the left / right finger pad
    const float* con = S + m.off[RB_O_CON];
    const int ncon = (int)S[m.off[RB_O_DBG] + 3];
    float cl = 0.f, cr = 0.f;
    for (int c = 0; c < ncon; c++) {
      const float* C = con + RB_CONREC * c;
      if ((int)C[RB_CR_KIND] == RB_KIND_EQUALITY || !(C[RB_CR_DIST] < 1.0e-5f)) continue;
      const int g1 = (int)C[RB_CR_G1], g2 = (int)C[RB_CR_G2];
      for (int f = 0; f < 2; f++) {
        const int other = g1 == a.finger_geom[f] ? g2 : (g2 == a.finger_geom[f] ? g1 : -1);
        if (other >= 0 && m.geom_bodyid[other] == b) { if (f == 0) cl = 1.f; else cr = 1.f; }
      }
    }
    o[g0 + 12 * N + 1 + 2 * lane] = cl; o[g0 + 12 * N + 1 + 2 * lane + 1] = cr;
    const float* so = a.static_obs + ((size_t)e * N + lane) * 7;
    for (int k = 0; k < 3; k++) o[g0 + 14 * N + 1 + 3 * lane + k] = so[k];
    for (int k = 0; k < 4; k++) o[g0 + 17 * N + 1 + 4 * lane + k] = so[3 + k];
  }
  const unsigned long long okmask = __ballot(ok_obj), offmask = __ballot(off_obj);
  const int nsucc = __popcll(okmask), any_off = offmask != 0;
  // sums of the distances over the objects (goal_info["goal_dist"])
  float sp = dpos, sr = drot;
  for (int o = 32; o > 0; o >>= 1) { sp += __shfl_xor(sp, o); sr += __shfl_xor(sr, o); }
  // ---- robot read-outs and the copied blocks
  {
    float* o = row + 15 * N;
    if (lane < 6) o[lane] = qrow[a.arm_qposadr[lane]];
    if (lane < 3) { o[6 + lane] = tcp[lane]; o[9 + lane] = tcp_vp[lane]; }
    if (lane == 0) { o[12] = crow[a.grip_act]; o[13] = qrow[a.grip_qposadr]; o[14] = vrow[a.grip_dofadr]; }
    for (int k = lane; k < nq; k += 64) { o[15 + k] = qrow[k]; o[15 + nq + k] = a.qpos_goal[(size_t)e * nq + k]; }
  }
  // gripper - table-plane contact (any gripper geom against the table's collision plane)
  int table_hit = 0;
  {
    const float* con = S + m.off[RB_O_CON];
    const int ncon = (int)S[m.off[RB_O_DBG] + 3];
    for (int c = lane; c < ncon; c += 64) {
      const float* C = con + RB_CONREC * c;
      if ((int)C[RB_CR_KIND] == RB_KIND_EQUALITY) continue;
      const int g1 = (int)C[RB_CR_G1], g2 = (int)C[RB_CR_G2];
      const bool in1 = g1 < 64 && ((a.gripper_geom_mask >> g1) & 1ull), in2 = g2 < 64 && ((a.gripper_geom_mask >> g2) & 1ull);
      const int other = in1 ? g2 : (in2 ? g1 : -1);
      if (other == a.table_plane_geom) table_hit = 1;
    }
  }
  const int table_contact = __ballot(table_hit) != 0;
  const float fx = sens[a.force_adr], fy = sens[a.force_adr + 1], fz = sens[a.force_adr + 2];
  const int safety = sqrtf(fx * fx + fy * fy + fz * fz) > a.safety_stop_force;
  // frozen: 1 = the first observation of a new episode (reset's _observe_sync): observation row, gripper hand-over, zeroed outputs, the success count the next step's
  // reward is measured from; 4 = an env INSIDE its reset recipe (pipelined resets): the recipe's steps are `_set_action + mujoco_simulation.step()`
  // (common/base.py:484-496), no _observe_sync -- observation row and zeroed outputs only, no hand-over, no goal bookkeeping
  const int frozen = a.frozen ? a.frozen[e] : 0;
  if (lane == 0 && frozen == 3) {                  // 3: observation entries only (a live env whose goal was just replaced): reward / done / flags / counters untouched
    const int g0 = 15 * N + 15 + 2 * nq;
    row[g0 + 6 * N] = (float)(!crash && nsucc == N);
    // reset_goal -> _observe_sync -> update_goal_info (robot_env.py:893-909, 586-593): the re-observation under the new goal is what the next step's
    // goal-distance reward is measured from
    a.prev_nsucc[e] = (float)nsucc * a.goal_reward_per_object; a.prev_valid[e] = 1;
    float* o = row + g0 + 21 * N + 1;
    o[0] = (float)safety;
    for (int k = 0; k < 3; k++) { o[1 + k] = sens[a.force_adr + k]; o[4 + k] = sens[a.torque_adr + k]; }
  }
  if (lane == 0 && (frozen == 1 || frozen == 4)) {
    float* rw = a.reward + 3 * (size_t)e;
    rw[0] = rw[1] = rw[2] = 0.f;
    a.goal_dist[2 * e] = sp; a.goal_dist[2 * e + 1] = sr;
    a.done[e] = 0; a.goal_reset[e] = 0; a.trial_success[e] = 0; a.sub_goal_ok[e] = 0; a.env_crash[e] = crash; a.objects_off_table[e] = any_off;
    a.info_ssl[e] = a.steps_since_last_goal[e];
    // RobotEnv.reset -> reset_goal_generation -> _observe_sync -> update_goal_info (robot_env.py:757-792, 586-593): the observation that ends a reset
    // establishes the success count the first step's goal-distance reward is measured from
    if (frozen == 1) { a.prev_nsucc[e] = (float)nsucc * a.goal_reward_per_object; a.prev_valid[e] = 1; }
    const int g0 = 15 * N + 15 + 2 * nq;
    row[g0 + 6 * N] = 0.f;
    float* o = row + g0 + 21 * N + 1;
    o[0] = (float)safety;
    for (int k = 0; k < 3; k++) { o[1 + k] = sens[a.force_adr + k]; o[4 + k] = sens[a.torque_adr + k]; }
    float* tail = row + a.obs_dim;
    tail[0] = tail[1] = tail[2] = tail[3] = 0.f;
    if (a.solver_qpos && frozen == 1) {
      a.solver_qpos[(size_t)e * a.solver_nq + a.solver_grip_qposadr] = qrow[a.grip_qposadr];
      a.solver_ctrl[(size_t)e * a.solver_nu + a.solver_grip_act] = crow[a.grip_act];
    }
  }
  if (lane == 0 && frozen == 0) {
    // ---- reward / done of the simulation (common/base.py:768-795)
    float env_reward = 0.f;
    int done = 0;
    if (table_contact) env_reward -= a.penalty_table_collision;
    if (any_off) { done = 1; env_reward -= a.penalty_objects_off_table; }
    if (safety) env_reward -= a.penalty_safety_stop;
    // ---- _get_goal_info: reward = change of the number of objects within both thresholds (common/base.py:824-848)
    a.t[e] += 1;
    const float ns = (float)nsucc * a.goal_reward_per_object;
    const float gdr = (a.prev_valid[e] && !crash) ? ns - a.prev_nsucc[e] : 0.f;
    a.prev_nsucc[e] = ns; a.prev_valid[e] = 1;
    const int succ = !crash && nsucc == N;
    // ---- MultiGoalTracker.process (multi_goal_tracker.py:157-241)
    a.steps[e] += 1;
    int ssl = a.steps_since_last_goal[e] + 1;
    const int cons = succ ? a.consecutive[e] + 1 : 0;
    const int got = cons >= 1;
    if (got) a.successes_so_far[e] += 1;
    const int timeout = !got && ssl >= a.max_timesteps_per_goal;
    const int trial = got && a.successes_so_far[e] >= a.successes_needed;
    if (trial) ssl = 0;
    const int newgoal = got && !trial;
    if (newgoal) { ssl = 0; a.prev_valid[e] = 0; }          // reset_goal: reset_goal_steps, _previous_goal_distance = None (robot_env.py:893-909)
    a.steps_since_last_goal[e] = ssl; a.consecutive[e] = cons;
    float* rw = a.reward + 3 * (size_t)e;
    rw[0] = env_reward; rw[1] = a.use_goal_distance_reward ? gdr : 0.f; rw[2] = got ? a.success_reward : 0.f;
    if (a.reward_clip > 0.f) for (int k = 0; k < 3; k++) rw[k] = fminf(fmaxf(rw[k], -a.reward_clip), a.reward_clip);   // ClipRewardWrapper
    a.goal_dist[2 * e] = sp; a.goal_dist[2 * e + 1] = sr;
    done = done || timeout || trial || crash;
    a.done[e] = done; a.goal_reset[e] = newgoal; a.trial_success[e] = trial; a.sub_goal_ok[e] = got; a.env_crash[e] = crash;
    a.objects_off_table[e] = any_off; a.info_ssl[e] = ssl;
    const int g0 = 15 * N + 15 + 2 * nq;
    row[g0 + 6 * N] = (float)succ;                          // is_goal_achieved
    float* o = row + g0 + 21 * N + 1;
    o[0] = (float)safety;
    for (int k = 0; k < 3; k++) { o[1 + k] = sens[a.force_adr + k]; o[4 + k] = sens[a.torque_adr + k]; }
    float* tail = row + a.obs_dim;
    tail[0] = rw[0]; tail[1] = rw[1]; tail[2] = rw[2]; tail[3] = (float)done;
    // ---- on_observations_updated: the solver world's gripper follows the main world's (joint position and control target)
    if (a.solver_qpos) {
      a.solver_qpos[(size_t)e * a.solver_nq + a.solver_grip_qposadr] = qrow[a.grip_qposadr];
      a.solver_ctrl[(size_t)e * a.solver_nu + a.solver_grip_act] = crow[a.grip_act];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// The reset recipe and the goal sampling on the device (ra_recipe_args, include/rgstep.h): what BatchedBlockRearrangeEnv._advance_recipes does on the host
// (robogym_amd/envs/rearrange/blocks.py) without its readback of the done / goal_reset flags.  Reference call sites:
//   RearrangeEnv._reset, _randomize_robot_initial_position        envs/rearrange/common/base.py:897-932, 498-510
//   place_objects_in_grid / place_objects_with_no_constraint       envs/rearrange/common/utils.py:719-829 / 829-880 (_place_objects :623-716)
//   get_placement_area                                             envs/rearrange/simulation/base.py:980-1010
//   ObjectStateGoal.next_goal                                      envs/rearrange/goals/object_state.py:355-418
struct RaRecipeLds { int started, ended, regoal; float pos[RA_MAXOBJ][3], gpos[RA_MAXOBJ][3], gyaw[RA_MAXOBJ]; };

// One placement of the N objects (rotated about z by yaw[i]) inside the placement area: body origins in world coordinates.  Returns false when the rejection
// sampling ran out of restarts (out = its last proposal).  `k`: this env's running draw index.
__device__ inline bool ra_place(const RaRecipeArgs& a, const float* yaw, int ystride, unsigned seed, unsigned step, unsigned e, unsigned& k, float (*out)[3]) {
  const int N = a.num_objects;
  auto U = [&]() -> float { return (float)(rbp_hash(seed, step, e, k++) >> 8) * (1.0f / 16777216.0f); };
  float hx[RA_MAXOBJ], hy[RA_MAXOBJ], xy[RA_MAXOBJ][2];
  float mx = 0.f, my = 0.f;
  for (int i = 0; i < N; i++) {      // rotate_bounding_box
    const float c = fabsf(cosf(yaw[i * ystride])), s_ = fabsf(sinf(yaw[i * ystride]));
    hx[i] = c * a.obj_half[i][0] + s_ * a.obj_half[i][1]; hy[i] = s_ * a.obj_half[i][0] + c * a.obj_half[i][1];
    mx = fmaxf(mx, hx[i]); my = fmaxf(my, hy[i]);
  }
  const float width = a.area_size[0], height = a.area_size[1];
  const int ncol = (int)floorf(width / (2.f * mx)), nrow = (int)floorf(height / (2.f * my));
  const int M = ncol * nrow;
  bool ok = true;
  if (M >= N) {
    // N distinct cells in random order = the first N of a random permutation of the M cells: a uniform random subset (Floyd), then shuffled
    int cell[RA_MAXOBJ];
    for (int idx = 0, j = M - N; j < M; j++, idx++) {
      int t = (int)(U() * (float)(j + 1)); t = t > j ? j : t;
      bool seen = false;
      for (int q = 0; q < idx; q++) seen = seen || cell[q] == t;
      cell[idx] = seen ? j : t;
    }
    for (int i = N - 1; i > 0; i--) { int j = (int)(U() * (float)(i + 1)); j = j > i ? i : j; const int tmp = cell[i]; cell[i] = cell[j]; cell[j] = tmp; }
    const float cw = width / (float)ncol, ch = height / (float)nrow;
    for (int i = 0; i < N; i++) { xy[i][0] = cw * (float)(cell[i] % ncol) + hx[i]; xy[i][1] = ch * (float)(cell[i] / ncol) + hy[i]; }
  } else {
    // fewer cells than objects (large mesh objects): uniform proposals, rejected while the object's box overlaps a placed one; largest objects first; a set
    // restarts when one of its objects runs out of trials
    int order[RA_MAXOBJ];
    for (int i = 0; i < N; i++) order[i] = i;
    for (int i = 1; i < N; i++) {      // stable insertion sort, area descending
      const int v = order[i]; const float av = a.obj_half[v][0] * a.obj_half[v][1];
      int j = i - 1;
      while (j >= 0 && a.obj_half[order[j]][0] * a.obj_half[order[j]][1] < av) { order[j + 1] = order[j]; j--; }
      order[j + 1] = v;
    }
    ok = false;
    for (int round = 0; round < 200 && !ok; round++) {
      ok = true;
      for (int oi = 0; oi < N && ok; oi++) {
        const int i = order[oi];
        bool placed = false;
        for (int trial = 0; trial < 100 && !placed; trial++) {
          const float cx = hx[i] + U() * (width - 2.f * hx[i]), cy = hy[i] + U() * (height - 2.f * hy[i]);
          bool free_ = true;
          for (int od = 0; od < oi; od++) { const int d = order[od]; free_ = free_ && (fabsf(cx - xy[d][0]) >= hx[i] + hx[d] || fabsf(cy - xy[d][1]) >= hy[i] + hy[d]); }
          xy[i][0] = cx; xy[i][1] = cy;
          placed = free_;
        }
        ok = placed;
      }
    }
  }
  for (int i = 0; i < N; i++) {      // the body origin, from the centre of its bounding box
    const float c = cosf(yaw[i * ystride]), s_ = sinf(yaw[i * ystride]);
    const float ccx = c * a.obj_center[i][0] - s_ * a.obj_center[i][1], ccy = s_ * a.obj_center[i][0] + c * a.obj_center[i][1];
    out[i][0] = xy[i][0] + a.area_offset[0] - a.table_size[0] + a.table_pos[0] - ccx;
    out[i][1] = xy[i][1] + a.area_offset[1] - a.table_size[1] + a.table_pos[1] - ccy;
    out[i][2] = a.obj_half[i][2] + 2.f * a.table_size[2] - a.table_size[2] + a.table_pos[2] - a.obj_center[i][2];
  }
  return ok;
}

__global__ void __launch_bounds__(64) ra_recipe_kernel(const RbModelDev* mp, RbBatchDev bt, const RbModelDev* sp, RbBatchDev sb, RaRecipeArgs a) {
#ifdef RG_EMUL
  RaRecipeLds& F = *(RaRecipeLds*)emul_lds();
#else
  __shared__ RaRecipeLds Fs; RaRecipeLds& F = Fs;
#endif
  const RbModelDev& m = *mp;
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= bt.B) return;
  const int N = a.num_objects, AD = a.action_dim, nq = m.nq;
  if (lane == 0) {
    int st = a.stage[e], lf = a.left[e];
    int started = 0, ended = 0, regoal = 0, stabilised = 0;
    unsigned k = 0;
    auto U = [&]() -> float { return (float)(rbp_hash(a.seed, a.step, (unsigned)e, k++) >> 8) * (1.0f / 16777216.0f); };
    a.episode_started[e] = 0;
    // ---- an env inside the recipe: this step counted
    if (st > 0) {
      lf -= 1;
      if (lf <= 0) {
        if (st == 1) {
          stabilised = 1;                                              // stabilize_objects restores the objects' damping (host tensor op on this mask)
          if (a.n_random_initial_steps >= 1) {                         // -> one random action for n_random_initial_steps steps
            st = 2; lf = a.n_random_initial_steps;
            for (int d = 0; d < AD; d++) a.scripted[(size_t)e * AD + d] = 2.f * U() - 1.f;
            a.solver_active[e] = 1; a.hold_ctrl[e] = 0;
          } else started = 1;
        } else if (st == 2) {                                          // -> zero action while everything settles
          st = 3; lf = a.settle_steps;
          for (int d = 0; d < AD; d++) a.scripted[(size_t)e * AD + d] = 0.f;
        } else started = 1;
      }
    }
    const float* gy = 0; int gstride = 1;
    if (started) {                                                     // -> the episode starts: tracker reset, a fresh smoothing filter, the first goal
      st = 0; lf = 0;
      a.t[e] = 0; a.steps[e] = 0; a.steps_since_last_goal[e] = 0; a.successes_so_far[e] = 0; a.consecutive[e] = 0; a.ema_t[e] = 0;
      a.hold[e] = 0; a.hold_ctrl[e] = 0; a.frozen[e] = 0; a.solver_active[e] = 1; a.resetting[e] = 0; a.episode_started[e] = 1;
      for (int d = 0; d < AD; d++) { a.scripted[(size_t)e * AD + d] = 0.f; a.ema_value[(size_t)e * AD + d] = 0.f; a.action_ema[(size_t)e * AD + d] = 0.f; }
      gy = a.yaw + (size_t)e * N; gstride = 1;
    } else if (st == 0 && a.goal_reset[e] && !a.done[e]) {             // a live env that reached its goal: ObjectStateGoal.next_goal keeps the goal's yaw
      // (not when the episode ends on the same step -- objects off the table while every object sits within its thresholds --: the begin-of-episode state written
      //  below would be what the goal's re-observation reads; the terminal observation keeps the reached goal's entries, on this path and on the host path alike)
      regoal = 1;
      gy = a.goal_rot + (size_t)e * N * 3 + 2; gstride = 3;
    }
    if (gy) {
      k = 0u;     // (the goal's placement draws from its own stream: an env can get a new goal and end its episode on the same step)
      if (!ra_place(a, gy, gstride, a.seed ^ 0x9E3779B9u, a.step, (unsigned)e, k, F.gpos)) a.placement_failed[e] += 1;
      for (int i = 0; i < N; i++) F.gyaw[i] = gy[i * gstride];
    }
    // ---- an episode that ended on this step: its recipe begins (the returned observation / reward / done are the terminal ones)
    if (!started && st == 0 && a.done[e]) {
      ended = 1;
      k = 1000u;
      float* yw = a.yaw + (size_t)e * N;
      for (int i = 0; i < N; i++) yw[i] = 2.f * RBC_PI * U();
      if (!ra_place(a, yw, 1, a.seed, a.step, (unsigned)e, k, F.pos)) a.placement_failed[e] += 1;
      st = 1; lf = a.stabilize_steps > 0 ? a.stabilize_steps : 1;
      a.hold[e] = 1; a.hold_ctrl[e] = 1; a.frozen[e] = 4; a.solver_active[e] = 0; a.resetting[e] = 1;
      for (int d = 0; d < AD; d++) a.scripted[(size_t)e * AD + d] = 0.f;
    }
    // controller ticks of the NEXT step's main-world launch: two for live envs, one inside the recipe, two on the recipe's last step
    const int last_stage = a.n_random_initial_steps >= 1 ? 3 : 1;
    a.nticks[e] = st > 0 ? ((st == last_stage && lf <= 1) ? 2 : 1) : 2;
    a.stage[e] = st; a.left[e] = lf;
    a.reobserve[e] = started ? 1 : (regoal ? 3 : 2);
    a.ended[e] = (unsigned char)ended; a.stabilised[e] = (unsigned char)stabilised;
    F.started = started; F.ended = ended; F.regoal = regoal;
  }
  __syncthreads();
  // ---- goal rows (goals/object_state.py:381-418): position, orientation (a z rotation, as Euler angles and as a quaternion), qpos_goal = the current qpos with
  // the objects at their goals; the previous success count is void
  if (F.started || F.regoal) {
    if (lane < N) {
      const float ez = rbc_wrap(F.gyaw[lane]);        // mat2euler of a z rotation, normalised
      float* g = a.goal + ((size_t)e * N + lane) * 7;
      g[0] = F.gpos[lane][0]; g[1] = F.gpos[lane][1]; g[2] = F.gpos[lane][2];
      g[3] = cosf(0.5f * ez); g[4] = 0.f; g[5] = 0.f; g[6] = sinf(0.5f * ez);
      float* gr = a.goal_rot + ((size_t)e * N + lane) * 3;
      gr[0] = 0.f; gr[1] = 0.f; gr[2] = ez;
    }
    for (int i = lane; i < nq; i += 64) {
      float v = bt.qpos[(size_t)e * nq + i];
      for (int o = 0; o < N; o++) {
        const int d = i - a.obj_qposadr[o];
        if (d >= 0 && d < 7) { const float ez = rbc_wrap(F.gyaw[o]); v = d < 3 ? F.gpos[o][d] : (d == 3 ? cosf(0.5f * ez) : (d == 6 ? sinf(0.5f * ez) : 0.f)); }
      }
      a.qpos_goal[(size_t)e * nq + i] = v;
    }
    if (lane == 0) a.prev_valid[e] = 0;
  }
  __syncthreads();
  // ---- what RearrangeEnv._reset writes before anything is simulated: both worlds as freshly made, the arm's start pose, the objects placed
  if (F.ended) {
    const float* yw = a.yaw + (size_t)e * N;
    for (int i = lane; i < nq; i += 64) {
      float v = m.qpos0[i];
      for (int j = 0; j < 6; j++) if (i == a.arm_qposadr[j]) v = a.arm_start[j];
      for (int o = 0; o < N; o++) {
        const int d = i - a.obj_qposadr[o];
        if (d >= 0 && d < 7) v = d < 3 ? F.pos[o][d] : (d == 3 ? cosf(0.5f * yw[o]) : (d == 6 ? sinf(0.5f * yw[o]) : 0.f));
      }
      bt.qpos[(size_t)e * nq + i] = v;
    }
    for (int i = lane; i < m.nv; i += 64) { bt.qvel[(size_t)e * m.nv + i] = 0.f; bt.qacc_warmstart[(size_t)e * m.nv + i] = 0.f; }
    for (int i = lane; i < m.nu; i += 64) bt.ctrl[(size_t)e * m.nu + i] = i < 6 ? a.arm_start[i] : 0.f;
    for (int i = lane; i < 3 * m.nu; i += 64) bt.pid[(size_t)e * 3 * m.nu + i] = 0.f;
    if (lane == 0) { bt.time[e] = 0.f; bt.status[e] = 0; }
    if (sp) {
      const RbModelDev& ms = *sp;
      for (int i = lane; i < ms.nq; i += 64) {
        float v = ms.qpos0[i];
        for (int j = 0; j < 6; j++) if (i == a.solver_arm_qposadr[j]) v = a.arm_start[j];
        sb.qpos[(size_t)e * ms.nq + i] = v;
      }
      for (int i = lane; i < ms.nv; i += 64) { sb.qvel[(size_t)e * ms.nv + i] = 0.f; sb.qacc_warmstart[(size_t)e * ms.nv + i] = 0.f; }
      for (int i = lane; i < ms.nu; i += 64) sb.ctrl[(size_t)e * ms.nu + i] = 0.f;
      for (int i = lane; i < 3 * ms.nu; i += 64) sb.pid[(size_t)e * 3 * ms.nu + i] = 0.f;
      if (lane == 0) { sb.time[e] = 0.f; sb.status[e] = 0; }
      if (ms.neq > 0 && lane < 7) sb.eq_data[(size_t)e * 7 * ms.neq + lane] = lane == 3 ? 1.f : 0.f;      // reset_mocap_welds
    }
    if (lane < N) {      // the static observation: bounding box of the yawed object, a random colour
      const float c = fabsf(cosf(yw[lane])), s_ = fabsf(sinf(yw[lane]));
      float* so = a.static_obs + ((size_t)e * N + lane) * 7;
      so[0] = c * a.obj_half[lane][0] + s_ * a.obj_half[lane][1]; so[1] = s_ * a.obj_half[lane][0] + c * a.obj_half[lane][1]; so[2] = a.obj_half[lane][2];
      for (int q = 0; q < 3; q++) so[3 + q] = (float)(rbp_hash(a.seed, a.step, (unsigned)e, 500u + 3u * lane + q) >> 8) * (1.0f / 16777216.0f);
      so[6] = 1.f;
    }
  }
}

}  // namespace rgb
