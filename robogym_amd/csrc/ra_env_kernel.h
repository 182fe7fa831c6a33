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

}  // namespace rgb
