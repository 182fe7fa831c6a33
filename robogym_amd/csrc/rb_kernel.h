// rb_kernel.h — the LARGE-MODEL stepper for gfx950: env.step (action -> ctrl, nsubsteps x mj_step, the PID ticks of the
// reference's state-less forwards) for models beyond the compile-time layout of rg_kernel.h — dactyl/full_perpendicular
// (BASELINE.json configs[2]: Shadow hand + Rubik's cube, nv 168, 135 bodies, 26 condim-6 mesh hulls;
// /root/reference/robogym/envs/dactyl/full_perpendicular.py:92-136, cube_env.py:239-242, simulation_interface.py:176-189).
//
// Execution plan: ONE 256-thread workgroup (4 waves) per env, every size a run-time number of the model.  Stage arrays
// (body / geom frames, spatial inertias, motion axes, M in MuJoCo's tree-sparse form, contacts, contact Jacobians,
// constraint rows) live in the env's HBM scratch row under their mjData names and stay L2-resident for the launch; the
// vectors of the solver and ONE dense block (the Newton Hessian / M + h B of the largest constraint-coupled group of trees:
// 96 x 96 for hand + cube) live in LDS.  Tree recursions are level sweeps, reductions DPP + a 4-entry LDS exchange, all
// appends are ballot-prefix compactions in index order and all scatters are owner-computes loops: a launch is run-to-run
// deterministic.  Collision: static pair list -> bounding spheres -> libccd-style MPR in quads (rg_kernel.h's routine,
// 64 queries per trip).  Constraints: friction loss, joint / tendon limits, pyramidal contacts of condim 1 / 3 / 4 / 6 as
// six basis Jacobians per contact on the union of the two bodies' dof chains.  Solver: primal Newton with exact line
// search (mj_solNewton), dense Cholesky per group.  Arithmetic follows oracle/rg_oracle.c stage by stage; this first
// version is written for clarity and parity, not yet for speed.
// The file is included once per CONFIGURATION (rg_api.hip): RB_NS = namespace, RB_T = threads per workgroup (a multiple of 64),
// RB_MAXGROUP / RB_MAXNV / RB_MAXNQ = LDS capacities, RB_WG_PER_CU = waves per SIMD the registers are budgeted for.
#include "rb_types.h"
#if !defined(RB_NS) || !defined(RB_T) || !defined(RB_MAXGROUP) || !defined(RB_MAXNV) || !defined(RB_MAXNQ)
#error "rb_kernel.h: define RB_NS, RB_T, RB_MAXGROUP, RB_MAXNV, RB_MAXNQ before including"
#endif
#ifndef RB_MAXNU
#define RB_MAXNU 32      /* actuators (LDS rows ctrl, controller state, lengths, forces) */
#endif

namespace RB_NS {
using namespace rgl;   // small math, wave collectives and the MPR / support routines of rg_kernel.h

#ifdef RG_EMUL
typedef const RbModelDev& RbM;
typedef const RbLaunch& RbLRef;
#define RB_S() (*(RbLds*)emul_lds())
#define RB_ARENA() ((float*)((char*)emul_lds() + RB_ARENA_BASE))
#else
typedef const RG_AS4 RbModelDev& RbM;
typedef const RG_AS4 RbLaunch& RbLRef;
#define RB_S() (*(RbLds*)rg_lds_raw)
#define RB_ARENA() ((float*)(rg_lds_raw + RB_ARENA_BASE))   /* as a generic pointer: the stage functions reach their arrays through flat loads either way */
#endif
#define TID ((int)threadIdx.x)
#define WID (TID >> 6)
#define WL (TID & 63)
#define BFOR(i, n) for (int i = TID; i < (n); i += RB_T)
#define BSYNC() __syncthreads()
#define RB_NWAVE (RB_T / 64)
#define RB_TS (RB_T >= 256 ? 16 : 8)   /* side of the thread tile in the trailing update of the Cholesky (RB_TS^2 <= RB_T) */
#define RB_MINVAL 1e-15f
#ifndef RB_WG_PER_CU
#define RB_WG_PER_CU 4   /* resident workgroups per CU the kernel is compiled for (register budget 512 / RB_WG_PER_CU per lane) */
#endif
#define RB_CST (7 * RB_CONW + RB_NW + 2)   /* words of a staged contact; <= RB_T */
#ifndef RB_COST_EPS
#define RB_COST_EPS 1e-7f   /* relative rounding noise of the fp32 cost sum: improvements below it are not resolvable */
#endif

struct RbLds {
  alignas(16) float A[RB_MAXGROUP * (RB_MAXGROUP + 1) / 2 + 8];   // the dense block of the moment: LOWER triangle, packed by rows (entry (i, j <= i) at i (i + 1) / 2 + j)
  float Dinv[RB_MAXGROUP * 8];   // inverses of the 8 x 8 diagonal blocks of the factor
  float prow[8];
  float yb[2 * 8];   // rb_chol_solve: the current block's solution, double-buffered
  float cst[2 * RB_CST];   // rb_hessian_add: the contact being added and the next one (basis Jacobians, block rows, weights, nnz, dim)
  float sc[RB_MAXGROUP];
  float qpos[RB_MAXNQ], qvel[RB_MAXNV], warm[RB_MAXNV], ctrl[RB_MAXNU], pid[3 * RB_MAXNU], actlen[RB_MAXNU], actfrc[RB_MAXNU];
  // the force terms of the smooth dynamics are dead once qfrc_smooth is formed (sb_smooth; their stage dump sits right behind it), the solver's work vectors
  // are born in sb_solve: they share storage (round 5: the small configuration fits 8 kB of LDS, the medium one 13 kB)
  union { float qfrc_passive[RB_MAXNV]; float grad[RB_MAXNV]; };
  union { float qfrc_bias[RB_MAXNV]; float search[RB_MAXNV]; };
  union { float qfrc_act[RB_MAXNV]; float Mv[RB_MAXNV]; };
  float qfrc_smooth[RB_MAXNV], qacc_smooth[RB_MAXNV];
  float qa[RB_MAXNV], Ma[RB_MAXNV], qfrc_con[RB_MAXNV], x[RB_MAXNV];
  float red[16];
  float prof[16];
  int wcnt[RB_NWAVE < 4 ? 4 : RB_NWAVE];
  int ncand, ncand2, ncon, nefc, nlim, stop;   // (ncand2: box - box / plane - box candidates, listed from the END of the candidate array)
  int neqcon;            // equality constraints of this mj_step: they are the first records of the contact list
  int env;               // this workgroup's env in its batch (= blockIdx.x in a one-batch launch; a multi-batch launch, rb_step_multi_kernel, splits blockIdx.x into batch and env)
  float mocap[14];       // pose of the mocap bodies (mjData.mocap_pos / mocap_quat), at most two
  float time;            // mjData.time (the cascaded-PI controller warm-starts its smoothed set-point at time 0)
  unsigned status;
};

// ------------------------------------------------------------------------------------------------- block collectives
__device__ __forceinline__ float rb_sum(RbLds& s, float v) {
  float w = wave_sum(v);
  if (RB_NWAVE == 1) { BSYNC(); return w; }   // (one wave: the barrier is only the memory fence callers count on)
  BSYNC();
  if (WL == 0) s.red[WID] = w;
  BSYNC();
  if (RB_NWAVE == 4) return (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]);
  float t = 0.f; for (int k = 0; k < RB_NWAVE; k++) t += s.red[k];
  return t;
}
__device__ __forceinline__ void rb_sum3(RbLds& s, float& a, float& b, float& c) {
  float wa = wave_sum(a), wb = wave_sum(b), wc = wave_sum(c);
  if (RB_NWAVE == 1) { BSYNC(); a = wa; b = wb; c = wc; return; }
  BSYNC();
  if (WL == 0) { s.red[WID] = wa; s.red[4 + WID] = wb; s.red[8 + WID] = wc; }
  BSYNC();
  if (RB_NWAVE == 4) { a = (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]); b = (s.red[4] + s.red[5]) + (s.red[6] + s.red[7]); c = (s.red[8] + s.red[9]) + (s.red[10] + s.red[11]); return; }
  a = b = c = 0.f;
  for (int k = 0; k < RB_NWAVE; k++) { a += s.red[k]; b += s.red[4 + k]; c += s.red[8 + k]; }
}
// slot of this thread's item in a list that grows in thread order (all threads call; -1: no item, or the list is full)
__device__ __forceinline__ int rb_slot(RbLds& s, bool pred, int* cnt, int cap, unsigned full_bit) {
  unsigned long long bal = __ballot(pred);
  BSYNC();
  if (WL == 0) s.wcnt[WID] = __popcll(bal);
  BSYNC();
  int base = *cnt;
  for (int k = 0; k < WID; k++) base += s.wcnt[k];
  int slot = base + __popcll(bal & ((1ull << WL) - 1ull));
  if (pred && slot >= cap) { s.status |= full_bit; }
  BSYNC();
  if (TID == 0) { int n = *cnt; for (int k = 0; k < RB_NWAVE; k++) n += s.wcnt[k]; *cnt = n < cap ? n : cap; }
  BSYNC();
  return (pred && slot < cap) ? slot : -1;
}
// a stage array of this env.  Default build: in the HBM scratch row.  With -DRB_LDS_ARENA (an experiment of round 5, kept as a build option: tools/gpu_call_r05b.sh,
// DESIGN.md section 3.4c): in the workgroup's LDS arena if the model's placement (RB_LDS_PLACE) puts it there -- one scalar select per use of the pointer; the stage
// functions reach their arrays through generic pointers (flat loads) either way, so nothing else changes.  Measured: per-workgroup latency falls by 1.7x, occupancy
// by 2.3x (LDS), throughput by 17-27 %: what binds these kernels at 4 waves per SIMD is not the scratch row's latency alone.
// a randomisable model field of this env: from the env's parameter block when the model carries per-env rows (rb_types.h RB_P_*), else the model's own array
#define PRM(field, K) (m.prm_on ? (const float*)(S + m.prm_off[K]) : m.field)
#define RB_ARENA_BASE ((sizeof(RbLds) + 15) & ~(size_t)15)
#ifdef RB_LDS_ARENA
#define SC(name) (m.lds_off[RB_O_##name] >= 0 ? RB_ARENA() + m.lds_off[RB_O_##name] : S + m.off[RB_O_##name])
#else
#define SC(name) (S + m.off[RB_O_##name])
#endif

// ------------------------------------------------------------------------------------------------- position stage
// engine_core_smooth.c mj_kinematics: body frames top-down (level sweep), joint anchors / axes, geoms, sites
__device__ __forceinline__ void rb_kinematics(RbM m, RbLds& s, float* S) {
  float *xpos = SC(XPOS), *xquat = SC(XQUAT), *xipos = SC(XIPOS), *xiquat = SC(XIQUAT), *xanchor = SC(XANCHOR), *xaxis = SC(XAXIS);
  if (TID == 0) {
    st3(xpos, mk3(0, 0, 0)); st3(xipos, mk3(0, 0, 0));
    q4 id; id.w = 1; id.x = id.y = id.z = 0; stq(xquat, id); stq(xiquat, id);
  }
  BSYNC();
  // One wave per env and at most 64 bodies: body b lives in lane b, a level's bodies take their parent's frame from the parent's lane (lane exchange)
  // instead of from the scratch row, so the sweep has no memory round trip between levels; the arithmetic is the level loop's below, word for word.
  const bool wave_sweep = RB_NWAVE == 1 && m.nbody <= 64;
  if (wave_sweep) {
    const int b = TID; const bool on = b > 0 && b < m.nbody;
    const int p = on ? m.body_parentid[b] : 0, lvl = on ? m.b_body_level[b] : -1;
    v3 pos = mk3(0, 0, 0); q4 quat; quat.w = 1; quat.x = quat.y = quat.z = 0;
    // the body's own constants and its first joint's are fetched BEFORE the level sweep (round 5): a level's iteration is then arithmetic and lane exchange only,
    // instead of a chain of model loads per level (the arm is ten levels deep); further joints of a body are read in the sweep as before
    const int bb = on ? b : 0;
    const v3 bpos = ld3(PRM(body_pos, RB_P_BODY_POS) + 3 * bb); const q4 bquat = ldq(m.body_quat + 4 * bb);
    const int mocapid = m.nmocap > 0 ? m.body_mocapid[bb] : -1, jn = on ? m.body_jntnum[bb] : 0, j0 = m.body_jntadr[bb];
    const int jj = jn > 0 ? j0 : 0;
    const int t0 = m.jnt_type[jj], qa0 = m.jnt_qposadr[jj];
    const v3 jpos0 = ld3(m.jnt_pos + 3 * jj), jaxis0 = ld3(m.jnt_axis + 3 * jj);
    const float q00 = m.qpos0[qa0];
    for (int L = 0; L < m.nlevel; L++) {
      q4 pq; pq.w = __shfl(quat.w, p); pq.x = __shfl(quat.x, p); pq.y = __shfl(quat.y, p); pq.z = __shfl(quat.z, p);
      const v3 pp = mk3(__shfl(pos.x, p), __shfl(pos.y, p), __shfl(pos.z, p));
      if (lvl != L) continue;
      pos = pp + qrot(pq, bpos);
      quat = qmul(pq, bquat);
      if (mocapid >= 0) { const float* mc = s.mocap + 7 * mocapid; pos = ld3(mc); quat = ldq(mc + 3); }
      for (int k = 0; k < jn; k++) {
        const int j = j0 + k, qa = k == 0 ? qa0 : m.jnt_qposadr[j], t = k == 0 ? t0 : m.jnt_type[j];
        if (t == RG_JNT_FREE) {
          pos = ld3(s.qpos + qa); quat = qnormalize(ldq(s.qpos + qa + 3));
          st3(xanchor + 3 * j, pos); st3(xaxis + 3 * j, mk3(0, 0, 1));
          continue;
        }
        const v3 jpos = k == 0 ? jpos0 : ld3(m.jnt_pos + 3 * j), jaxis = k == 0 ? jaxis0 : ld3(m.jnt_axis + 3 * j);
        const v3 anchor = pos + qrot(quat, jpos), axis = qrot(quat, jaxis);
        st3(xanchor + 3 * j, anchor); st3(xaxis + 3 * j, axis);
        const float q0 = k == 0 ? q00 : m.qpos0[qa];
        if (t == RG_JNT_SLIDE) pos = pos + axis * (s.qpos[qa] - q0);
        else {
          const q4 ql = (t == RG_JNT_BALL) ? qnormalize(ldq(s.qpos + qa)) : axisangle(jaxis, s.qpos[qa] - q0);
          quat = qmul(quat, ql);
          pos = anchor - qrot(quat, jpos);
        }
      }
      quat = qnormalize(quat);
    }
    if (on) {
      st3(xpos + 3 * b, pos); stq(xquat + 4 * b, quat);
      st3(xipos + 3 * b, pos + qrot(quat, ld3(m.body_ipos + 3 * b)));
      stq(xiquat + 4 * b, qmul(quat, ldq(m.body_iquat + 4 * b)));
    }
    BSYNC();
  }
  for (int L = 0; L < (wave_sweep ? 0 : m.nlevel); L++) {
    for (int q = m.b_lvl_adr[L] + TID; q < m.b_lvl_adr[L + 1]; q += RB_T) {
      const int b = m.b_lvl_body[q], p = m.body_parentid[b];
      const q4 pq = ldq(xquat + 4 * p);
      v3 pos = ld3(xpos + 3 * p) + qrot(pq, ld3(PRM(body_pos, RB_P_BODY_POS) + 3 * b));
      q4 quat = qmul(pq, ldq(m.body_quat + 4 * b));
      if (m.nmocap > 0 && m.body_mocapid[b] >= 0) { const float* mc = s.mocap + 7 * m.body_mocapid[b]; pos = ld3(mc); quat = ldq(mc + 3); }   // mj_kinematics: mocap pose (normalised below)
      for (int k = 0; k < m.body_jntnum[b]; k++) {
        const int j = m.body_jntadr[b] + k, qa = m.jnt_qposadr[j], t = m.jnt_type[j];
        if (t == RG_JNT_FREE) {
          pos = ld3(s.qpos + qa); quat = qnormalize(ldq(s.qpos + qa + 3));
          st3(xanchor + 3 * j, pos); st3(xaxis + 3 * j, mk3(0, 0, 1));
          continue;
        }
        const v3 jpos = ld3(m.jnt_pos + 3 * j), jaxis = ld3(m.jnt_axis + 3 * j);
        const v3 anchor = pos + qrot(quat, jpos), axis = qrot(quat, jaxis);
        st3(xanchor + 3 * j, anchor); st3(xaxis + 3 * j, axis);
        if (t == RG_JNT_SLIDE) pos = pos + axis * (s.qpos[qa] - m.qpos0[qa]);
        else {
          const q4 ql = (t == RG_JNT_BALL) ? qnormalize(ldq(s.qpos + qa)) : axisangle(jaxis, s.qpos[qa] - m.qpos0[qa]);
          quat = qmul(quat, ql);
          pos = anchor - qrot(quat, jpos);   // the anchor stays where it is
        }
      }
      quat = qnormalize(quat);
      st3(xpos + 3 * b, pos); stq(xquat + 4 * b, quat);
      st3(xipos + 3 * b, pos + qrot(quat, ld3(m.body_ipos + 3 * b)));
      stq(xiquat + 4 * b, qmul(quat, ldq(m.body_iquat + 4 * b)));
    }
    BSYNC();
  }
  BFOR(g, m.ngeom) {
    const int b = m.geom_bodyid[g]; const q4 xq = ldq(xquat + 4 * b);
    st3(SC(GPOS) + 3 * g, ld3(xpos + 3 * b) + qrot(xq, ld3(PRM(geom_pos, RB_P_GEOM_POS) + 3 * g)));
    stq(SC(GQUAT) + 4 * g, qmul(xq, ldq(m.geom_quat + 4 * g)));
  }
  BFOR(i, m.nsite) {
    const int b = m.site_bodyid[i];
    st3(SC(SPOS) + 3 * i, ld3(xpos + 3 * b) + qrot(ldq(xquat + 4 * b), ld3(m.site_pos + 3 * i)));
  }
  BSYNC();
}

// mj_comPos: subtree com of every tree root, body inertias (cinert) and motion axes (cdof) in the com-based frame of the tree
__device__ __forceinline__ void rb_com_pos(RbM m, RbLds& s, float* S) {
  float *xipos = SC(XIPOS), *rootcom = SC(ROOTCOM), *cinert = SC(CINERT), *cdof = SC(CDOF);
  const float *body_mass = PRM(body_mass, RB_P_BODY_MASS), *body_inertia = PRM(body_inertia, RB_P_BODY_INERTIA);
  BFOR(r, m.nroot) {
    const int root = m.b_root_list[r];
    v3 acc = mk3(0, 0, 0);
    float msum = 0.f;      // (per-env masses: the subtree's mass is summed here, in the list's order; the model's own body_subtreemass otherwise)
    // (four bodies' loads in flight at a time, the sums in list order as before: a one-at-a-time loop waits for every body's round trip in turn)
    const int q1 = m.b_subtree_adr[root + 1];
    for (int q = m.b_subtree_adr[root]; q < q1; q += 4) {
      int bb[4]; float mm[4]; v3 xx[4];
#pragma unroll
      for (int u = 0; u < 4; u++) bb[u] = m.b_subtree[q + u < q1 ? q + u : q1 - 1];
#pragma unroll
      for (int u = 0; u < 4; u++) { mm[u] = body_mass[bb[u]]; xx[u] = ld3(xipos + 3 * bb[u]); }
#pragma unroll
      for (int u = 0; u < 4; u++) if (q + u < q1) { acc = acc + xx[u] * mm[u]; msum += mm[u]; }
    }
    const float sm = m.prm_on ? msum : m.body_subtreemass[root];
    st3(rootcom + 3 * root, sm < RB_MINVAL ? ld3(xipos + 3 * root) : acc * (1.0f / sm));
  }
  BSYNC();
  if (TID < 10) cinert[TID] = 0.f;
  for (int b = 1 + TID; b < m.nbody; b += RB_T) {
    float R[9], I[9];
    q2mat(R, ldq(SC(XIQUAT) + 4 * b));
    const float* in = body_inertia + 3 * b;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[3 * i + j] = R[3 * i] * in[0] * R[3 * j] + R[3 * i + 1] * in[1] * R[3 * j + 1] + R[3 * i + 2] * in[2] * R[3 * j + 2];
    const v3 d = ld3(xipos + 3 * b) - ld3(rootcom + 3 * m.body_rootid[b]);
    const float mass = body_mass[b], d2 = dot(d, d);
    float* ci = cinert + 10 * b;
    ci[0] = I[0] + mass * (d2 - d.x * d.x); ci[1] = I[4] + mass * (d2 - d.y * d.y); ci[2] = I[8] + mass * (d2 - d.z * d.z);
    ci[3] = I[1] - mass * d.x * d.y; ci[4] = I[2] - mass * d.x * d.z; ci[5] = I[5] - mass * d.y * d.z;
    ci[6] = mass * d.x; ci[7] = mass * d.y; ci[8] = mass * d.z; ci[9] = mass;
  }
  BFOR(j, m.njnt) {
    const int b = m.jnt_bodyid[j], t = m.jnt_type[j]; int da = m.jnt_dofadr[j];
    const v3 off = ld3(rootcom + 3 * m.body_rootid[b]) - ld3(SC(XANCHOR) + 3 * j);
    if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
      float R[9]; q2mat(R, ldq(SC(XQUAT) + 4 * b));
      if (t == RG_JNT_FREE) {
        for (int k = 0; k < 3; k++) { float* c = cdof + 6 * (da + k); for (int e = 0; e < 6; e++) c[e] = 0; c[3 + k] = 1; }
        da += 3;
      }
      for (int k = 0; k < 3; k++) { const v3 ax = mk3(R[k], R[3 + k], R[6 + k]); st3(cdof + 6 * (da + k), ax); st3(cdof + 6 * (da + k) + 3, cross(ax, off)); }
    } else if (t == RG_JNT_SLIDE) {
      st3(cdof + 6 * da, mk3(0, 0, 0)); st3(cdof + 6 * da + 3, ld3(SC(XAXIS) + 3 * j));
    } else {
      const v3 ax = ld3(SC(XAXIS) + 3 * j); st3(cdof + 6 * da, ax); st3(cdof + 6 * da + 3, cross(ax, off));
    }
  }
  BSYNC();
}

// translational Jacobian column of dof i for a point at offset `off` from its tree's com (mj_jac with cdof)
__device__ __forceinline__ v3 rb_jacp(const float* cdof, int i, v3 off) { return ld3(cdof + 6 * i + 3) + cross(ld3(cdof + 6 * i), off); }

// mj_tendon: lengths and Jacobians on the static dof supports (b_ten_dofs); actuator lengths (mj_transmission)
__device__ __forceinline__ void rb_tendon(RbM m, RbLds& s, float* S) {
  const float *spos = SC(SPOS), *cdof = SC(CDOF), *rootcom = SC(ROOTCOM);
  BFOR(t, m.ntendon) {
    const int adr = m.tendon_adr[t], num = m.tendon_num[t];
    const int* td = m.b_ten_dofs + RB_TENW * t;
    float J[RB_TENW], L = 0;
    for (int e = 0; e < RB_TENW; e++) J[e] = 0;
    if (m.wrap_type[adr] == RG_WRAP_JOINT) {
      for (int w = adr; w < adr + num; w++) {
        const int j = m.wrap_objid[w], d = m.jnt_dofadr[j];
        L += m.wrap_prm[w] * s.qpos[m.jnt_qposadr[j]];
        for (int e = 0; e < RB_TENW; e++) if (td[e] == d) J[e] = m.wrap_prm[w];
      }
    } else {
      float divisor = 1.f;
      int w = adr;
      while (w < adr + num - 1) {
        const int t0 = m.wrap_type[w], t1 = m.wrap_type[w + 1];
        if (t0 == RG_WRAP_PULLEY || t1 == RG_WRAP_PULLEY) { if (t0 == RG_WRAP_PULLEY) divisor = m.wrap_prm[w]; w++; continue; }
        v3 pnt[4]; int body[4], cnt;
        const int s0 = m.wrap_objid[w];
        pnt[0] = ld3(spos + 3 * s0); body[0] = m.site_bodyid[s0];
        float wlen = -1.f;
        if (t1 == RG_WRAP_SPHERE || t1 == RG_WRAP_CYLINDER) {
          const int g = m.wrap_objid[w + 1], s1 = m.wrap_objid[w + 2], sid = (int)lrintf(m.wrap_prm[w + 1]);
          float gm[9]; q2mat(gm, ldq(SC(GQUAT) + 4 * g));
          v3 w0 = mk3(0, 0, 0), w1 = mk3(0, 0, 0);
          const v3 x1 = ld3(spos + 3 * s1);
          wlen = rg_wrap(w0, w1, pnt[0], x1, ld3(SC(GPOS) + 3 * g), gm, m.geom_size[3 * g], t1, sid >= 0, sid >= 0 ? ld3(spos + 3 * sid) : mk3(0, 0, 0));
          if (wlen < 0) { pnt[1] = x1; body[1] = m.site_bodyid[s1]; cnt = 2; }
          else { pnt[1] = w0; pnt[2] = w1; pnt[3] = x1; body[1] = body[2] = m.geom_bodyid[g]; body[3] = m.site_bodyid[s1]; cnt = 4; }
          w += 2;
        } else {
          const int s1 = m.wrap_objid[w + 1];
          pnt[1] = ld3(spos + 3 * s1); body[1] = m.site_bodyid[s1]; cnt = 2;
          w += 1;
        }
        const float idiv = 1.0f / divisor;
        if (wlen >= 0) L += wlen * idiv;
        for (int k = 0; k < cnt - 1; k++) {
          if (cnt == 4 && k == 1) continue;   // the arc lies on the wrapping geom
          v3 dif = pnt[k + 1] - pnt[k];
          const float dist = sqrtf(dot(dif, dif));
          L += dist * idiv;
          if (body[k] != body[k + 1] && dist > RB_MINVAL) {
            dif = dif * (1.0f / dist);
            for (int side = 0; side < 2; side++) {
              const int bb = body[k + side]; const float sg = side ? idiv : -idiv;
              const v3 off = pnt[k + side] - ld3(rootcom + 3 * m.body_rootid[bb]);
              for (int i = m.b_body_lastdof[bb]; i >= 0; i = m.dof_parentid[i]) {
                const float v = sg * dot(dif, rb_jacp(cdof, i, off));
                for (int e = 0; e < RB_TENW; e++) if (td[e] == i) J[e] += v;
              }
            }
          }
        }
      }
    }
    SC(TENLEN)[t] = L;
    for (int e = 0; e < RB_TENW; e++) SC(TENJ)[RB_TENW * t + e] = J[e];
  }
  BSYNC();
  BFOR(u, m.nu) {
    const int id = m.actuator_trnid[u];
    s.actlen[u] = m.actuator_gear[u] * (m.actuator_trntype[u] == 0 ? s.qpos[m.jnt_qposadr[id]] : SC(TENLEN)[id]);
  }
  BSYNC();
}

// mj_crb: composite inertias (subtree sums, owner computes), M in tree-sparse form: entry e = (i, j = i or an ancestor of i)
__device__ __forceinline__ void rb_crb(RbM m, RbLds& s, float* S) {
  const float* cinert = SC(CINERT); float* crb = SC(CRB);
  const float* dof_armature = PRM(dof_armature, RB_P_DOF_ARMATURE);
  BFOR(w, 10 * m.nbody) {
    const int b = w / 10, k = w - 10 * b;
    float acc = 0;
    // (the world body's composite inertia is never read -- M's entries take their dof's body --: its subtree, every body of the model, is not summed;
    //  four loads in flight at a time, the sums in list order as before)
    const int q1 = b == 0 ? m.b_subtree_adr[0] : m.b_subtree_adr[b + 1];
    for (int q = m.b_subtree_adr[b]; q < q1; q += 4) {
      int bb[4]; float vv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) bb[u] = m.b_subtree[q + u < q1 ? q + u : q1 - 1];
#pragma unroll
      for (int u = 0; u < 4; u++) vv[u] = cinert[10 * bb[u] + k];
#pragma unroll
      for (int u = 0; u < 4; u++) if (q + u < q1) acc += vv[u];
    }
    crb[w] = acc;
  }
  BSYNC();
  BFOR(e, m.nM) {
    const int i = m.b_M_i[e], j = m.b_M_j[e];
    float buf[6];
    mul_inert_vec(buf, crb + 10 * m.dof_bodyid[i], SC(CDOF) + 6 * i);
    const float* c = SC(CDOF) + 6 * j;
    float v = c[0] * buf[0] + c[1] * buf[1] + c[2] * buf[2] + c[3] * buf[3] + c[4] * buf[4] + c[5] * buf[5];
    if (i == j) v += dof_armature[i];
    SC(MSP)[e] = v;
  }
  BSYNC();
}
// y = M x (all dofs): row i = its own entries (ancestors) + the entries of its descendants that name i.  The descendant lists of
// tree roots are long (the cube's root dofs: ~70 entries): those are summed by a whole wave each (b_Mlong), the others by the row's thread
__device__ __forceinline__ void rb_M_mul(RbM m, const float* Msp, const float* x, float* y) {
  BFOR(i, m.nv) {
    float acc = 0;
    for (int e = m.b_M_adr[i]; e < m.b_M_adr[i + 1]; e++) acc += Msp[e] * x[m.b_M_j[e]];
    const int q0 = m.b_Mdesc_adr[i], q1 = m.b_Mdesc_adr[i + 1];
    if (q1 - q0 <= RB_MLONG) for (int q = q0; q < q1; q++) acc += Msp[m.b_Mdesc_ent[q]] * x[m.b_Mdesc_dof[q]];   // descendants' entries that name i
    y[i] = acc;
  }
  BSYNC();
  const int nlong = m.b_Mlong[0];
  for (int k = WID; k < nlong; k += RB_T / 64) {
    const int i = m.b_Mlong[1 + k];
    float acc = 0;
    for (int q = m.b_Mdesc_adr[i] + WL; q < m.b_Mdesc_adr[i + 1]; q += 64) acc += Msp[m.b_Mdesc_ent[q]] * x[m.b_Mdesc_dof[q]];
    acc = wave_sum(acc);
    if (WL == 0) y[i] += acc;
  }
  BSYNC();
}
// s.A <- the block of group g of M (+ diag): lower triangle, packed by rows
#define RB_TRI(i, j) ((((i) * ((i) + 1)) >> 1) + (j))   /* i >= j */
__device__ __forceinline__ void rb_M_block(RbM m, RbLds& s, const float* Msp, int g, const float* diag, float dscale) {
  const int g0 = m.b_group_adr[g], n = m.b_group_adr[g + 1] - g0;
  BFOR(w, RB_TRI(n, 0)) s.A[w] = 0.f;
  BSYNC();
  BFOR(e, m.nM) {
    const int i = m.b_M_i[e], j = m.b_M_j[e];
    if (m.b_dof_group[i] != g) continue;
    const int li = m.b_dof_local[i], lj = m.b_dof_local[j];   // (j = i or an ancestor of i: lj <= li)
    float v = Msp[e];
    if (i == j && diag) v += dscale * diag[i];
    s.A[RB_TRI(li, lj)] = v;
  }
  BSYNC();
}
// In-place Cholesky of the n x n block in s.A (lower triangle, packed rows), blocked by RB_NB columns; false on a
// non-positive pivot.  Per block three workgroup barriers: (a1) wave 0 factors the RB_NB x RB_NB diagonal block with one row per lane
// in registers, pivots and multipliers by v_readlane, and leaves the block's inverse in s.Dinv (the substitutions and (a2) multiply by
// it instead of dividing serially); (a2) every row below multiplies its RB_NB entries by inv(L)': independent dot products; (b) the
// trailing block, a 16 x 16 thread tile over the tiles that still hold entries of the lower triangle, rank-RB_NB update per pass.
// (The first version factored the diagonal block redundantly in every thread, 36 words of a register array per thread: with the
// register budget of four resident workgroups that array lived in scratch and the step cost 6.9 k cycles per block.)
#define RB_NB 8
__device__ __forceinline__ bool rb_chol(RbLds& s, int n) {
  const int ty = TID / RB_TS, tx = TID % RB_TS;   // (threads beyond RB_TS^2 do not exist: RB_T = 256 / 16 or 64 / 8)
  bool ok = true;
  for (int kb = 0; kb < n; kb += RB_NB) {
    const int nb = n - kb < RB_NB ? n - kb : RB_NB;
#ifdef RB_CHOL_PROBE
    const long long tprobe = rg_clock();
#endif
    // (a1) the diagonal block, by wave 0 alone: lane r < 8 holds row r in registers, pivots and multipliers travel by v_readlane
    // (no LDS round trip, no barrier inside); lane c then builds column c of inv(L) the same way
    if (WID == 0) {
      const int r = WL;
      float Lr[RB_NB], y[RB_NB], idg[RB_NB];
#pragma unroll
      for (int q = 0; q < RB_NB; q++) Lr[q] = (r < nb && q <= r) ? s.A[RB_TRI(kb + r, kb + q)] : ((r < RB_NB && q == r) ? 1.f : 0.f);
#pragma unroll
      for (int c = 0; c < RB_NB; c++) {
        const float d = lane_bcast(Lr[c], c);
        if (!(d > RB_MINVAL)) ok = false;
        idg[c] = rg_rsqrt(fmaxf(d, RB_MINVAL));
        Lr[c] = (r == c) ? fmaxf(d, RB_MINVAL) * idg[c] : Lr[c] * idg[c];
#pragma unroll
        for (int q = 0; q < RB_NB; q++) if (q > c) { const float lqc = lane_bcast(Lr[c], q); Lr[q] -= Lr[c] * lqc; }
      }
#pragma unroll
      for (int rr = 0; rr < RB_NB; rr++) {          // y[rr] of lane c = inv(L)[rr][c]
        float acc = (rr == r) ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < RB_NB; q++) if (q < rr) acc -= lane_bcast(Lr[q], rr) * y[q];
        y[rr] = acc * idg[rr];
      }
      if (r < nb) {
#pragma unroll
        for (int q = 0; q < RB_NB; q++) if (q <= r) s.A[RB_TRI(kb + r, kb + q)] = Lr[q];
      }
      if (r < RB_NB) {
#pragma unroll
        for (int rr = 0; rr < RB_NB; rr++) s.Dinv[(kb / RB_NB) * RB_NB * RB_NB + rr * RB_NB + r] = (rr >= r && rr < nb && r < nb) ? y[rr] : 0.f;   // Dinv[rr][c]
      }
    }
    BSYNC();
    // (a2) every row below the block: its RB_NB entries times inv(L)' (independent dot products)
    {
      const int i = kb + nb + TID;
      if (i < n) {
        float p[RB_NB], o[RB_NB];
#pragma unroll
        for (int c = 0; c < RB_NB; c++) p[c] = c < nb ? s.A[RB_TRI(i, kb + c)] : 0.f;
        const float* D = s.Dinv + (kb / RB_NB) * RB_NB * RB_NB;
#pragma unroll
        for (int c = 0; c < RB_NB; c++) {
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < RB_NB; q++) if (q <= c) v += p[q] * D[c * RB_NB + q];
          o[c] = v;
        }
#pragma unroll
        for (int c = 0; c < RB_NB; c++) if (c < nb) s.A[RB_TRI(i, kb + c)] = o[c];
      }
    }
    BSYNC();
#ifdef RB_CHOL_PROBE
    if (TID == 0) s.prof[15] += (float)(rg_clock() - tprobe);
#endif
    // (b) trailing update: A[i][j] -= sum_c L[i][kb + c] L[j][kb + c], i, j >= kb + RB_NB
    const int t0 = kb + RB_NB;
    if (t0 < n) {
      constexpr int NT = (RB_MAXGROUP + RB_TS - 1) / RB_TS;
      float acc[NT][NT];
#pragma unroll
      for (int qa = 0; qa < NT; qa++)
#pragma unroll
        for (int qb = 0; qb < NT; qb++) acc[qa][qb] = 0.f;
#pragma unroll
      for (int half = 0; half < 2; half++) {
        float li[NT][RB_NB / 2], lj[NT][RB_NB / 2];
#pragma unroll
        for (int q = 0; q < NT; q++) if (t0 + RB_TS * q < n) {   // (uniform: tile rows / columns beyond the block hold nothing)
          const int ii = t0 + ty + RB_TS * q, jj = t0 + tx + RB_TS * q;
#pragma unroll
          for (int c = 0; c < RB_NB / 2; c++) {
            li[q][c] = ii < n ? s.A[RB_TRI(ii, kb + half * (RB_NB / 2) + c)] : 0.f;
            lj[q][c] = jj < n ? s.A[RB_TRI(jj, kb + half * (RB_NB / 2) + c)] : 0.f;
          }
        }
#pragma unroll
        for (int qa = 0; qa < NT; qa++) if (t0 + RB_TS * qa < n)
#pragma unroll
          for (int qb = 0; qb < NT; qb++) if (qb <= qa)       // lower triangle of tiles only
#pragma unroll
            for (int c = 0; c < RB_NB / 2; c++) acc[qa][qb] += li[qa][c] * lj[qb][c];
      }
      // (the operands sit in columns kb .. kb + RB_NB - 1, the stores go to columns >= t0: no barrier in between)
#pragma unroll
      for (int qa = 0; qa < NT; qa++) if (t0 + RB_TS * qa < n)
#pragma unroll
        for (int qb = 0; qb < NT; qb++) if (qb <= qa) {
          const int ii = t0 + ty + RB_TS * qa, jj = t0 + tx + RB_TS * qb;
          if (ii < n && jj <= ii) s.A[RB_TRI(ii, jj)] -= acc[qa][qb];
        }
    }
    BSYNC();
  }
  return ok;
}
#if RB_T == 256 && RB_MAXGROUP == 96 && !defined(RB_LDS_CHOL)
// ---- Round 6, the large configuration: the same factorisation on the MATRIX PIPE, by ONE wave.  A 96-dof Hessian is a dense matrix and its right-looking Cholesky is
// rank-k updates of the trailing block -- the one place of this code base where the work IS a small GEMM.  The UPPER triangle of the (scaled) matrix lives in six
// 32 x 32 f32 accumulator tiles of wave 0 (96 registers; rgl::rg_mfma32: lane l holds column l % 32, register r row 8 (r / 4) + 4 (l / 32) + r % 4): the part of pivot
// row j right of the diagonal is then one register across 32 lanes of each tile of its row block, which IS column j of L below the diagonal -- no transposes, no LDS,
// no barriers.  Step (j, j + 1): pivots and the multiplier by v_readlane, w = row / root on the VALU (row j + 1 takes step j's correction there), the two k-slots of
// the operand put together with one v_permlane32_swap per 32-column piece, then one v_mfma_f32_32x32x2_f32 per remaining tile (6 / 3 / 1 by row block: 160 in all),
// A = -w masked to the rows that are not final, B = w; columns j, j + 1 of L go to the packed block as they are formed.  (The blocked version above: per 8 columns a
// serial diagonal block, a panel pass and a 16 x 16 VALU tile update with three workgroup barriers, 127 k cycles per factorisation.  A first MFMA version with the
// nine tiles spread over the four waves and one barrier per column pair measured no faster than that: 48 four-wave barriers cost what the arithmetic saved.)
// (No inverses of diagonal blocks: this configuration's substitutions are rb_chol_solve_wave's.)
__device__ __forceinline__ bool rb_chol_mfma(RbLds& s, int n) {
  bool ok = true;
  if (WID == 0) {
    const int l = WL, col = l & 31, hi = l >> 5;
    rgacc T[3][3];   // tiles (R, C), R <= C
#pragma unroll
    for (int R = 0; R < 3; R++)
#pragma unroll
      for (int C = R; C < 3; C++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int i = 32 * R + 8 * g + 4 * hi + q, c = 32 * C + col;
            float v = (i == c) ? 1.f : 0.f;                                   // padding beyond n: identity
            if (i < n && c < n) v = i >= c ? s.A[RB_TRI(i, c)] : s.A[RB_TRI(c, i)];
            T[R][C][4 * g + q] = v;
          }
    const int nstep = (n + 1) & ~1;
#pragma unroll
    for (int Rj = 0; Rj < 3; Rj++) {
#pragma unroll
      for (int jl = 0; jl < 32; jl += 2) {
        const int j = 32 * Rj + jl;
        if (j < nstep) {
          constexpr int dummy = 0; (void)dummy;
          const int rj = 4 * (jl >> 3) + (jl & 3), hj = (jl >> 2) & 1, lj = 32 * hj + jl;
          float w0[3], w1[3];
          const float p = lane_bcast(T[Rj][Rj][rj], lj);
          if (!(p > RB_MINVAL)) ok = false;
          const float inv = rg_rsqrt(fmaxf(p, RB_MINVAL));
#pragma unroll
          for (int C = Rj; C < 3; C++) w0[C] = T[Rj][C][rj] * inv;
          const float mlt = lane_bcast(w0[Rj], lj + 1);
#pragma unroll
          for (int C = Rj; C < 3; C++) w1[C] = __builtin_fmaf(-mlt, w0[C], T[Rj][C][rj + 1]);
          const float p2 = lane_bcast(w1[Rj], lj + 1);
          if (!(p2 > RB_MINVAL)) ok = false;
          const float inv2 = rg_rsqrt(fmaxf(p2, RB_MINVAL));
          float WW[3];
#pragma unroll
          for (int C = Rj; C < 3; C++) {
            w1[C] *= inv2;
            const int x = 32 * C + col;
            if (hi == hj && x < n) {                                           // columns j, j + 1 of L (rows x >= j / j + 1)
              if (x >= j) s.A[RB_TRI(x, j)] = w0[C];
              if (x >= j + 1 && j + 1 < n) s.A[RB_TRI(x, j + 1)] = w1[C];
            }
            WW[C] = hj == 0 ? rg_halves<0>(w0[C], w1[C]) : rg_halves<1>(w0[C], w1[C]);
          }
#pragma unroll
          for (int R = Rj; R < 3; R++) {
            const float Aop = (R > Rj || col > jl + 1) ? -WW[R] : 0.f;         // rows <= j + 1 are final
#pragma unroll
            for (int C = R; C < 3; C++) rg_mfma32(Aop, WW[C], T[R][C]);
          }
        }
      }
    }
  }
  BSYNC();
  // (`ok` is wave 0's; the other waves learn it through LDS)
  if (TID == 0) s.prow[0] = ok ? 1.f : 0.f;
  BSYNC();
  return s.prow[0] != 0.f;
}
#endif
// x <- inv(L L') x for the group's local vector x[0..n), block by block: the diagonal block through its inverse (RB_NB threads, one
// dot product each), the rest of the column panel by everybody.  (Every thread doing the 8 x 8 product redundantly, which saves a
// barrier per block, measured 3 x slower: profiles/r03_ab.txt.)
__device__ __forceinline__ void rb_chol_solve(RbLds& s, int n, float* x) {
  // the block's solution travels through s.yb (two buffers, alternating: a block's readers may still be at it while the next block's
  // writers are done), which makes it TWO barriers per block
  int par = 0;
  for (int kb = 0; kb < n; kb += RB_NB, par ^= 1) {          // forward: L y = x
    const int nb = n - kb < RB_NB ? n - kb : RB_NB;
    float* yb = s.yb + RB_NB * par;
    if (TID < nb) { const float* D = s.Dinv + (kb / RB_NB) * RB_NB * RB_NB + TID * RB_NB; float v = 0.f; for (int c = 0; c <= TID; c++) v += D[c] * x[kb + c]; yb[TID] = v; }
    BSYNC();
    if (TID < nb) x[kb + TID] = yb[TID];
    const int i = kb + nb + TID;
    if (i < n) { float a = x[i]; for (int c = 0; c < nb; c++) a -= s.A[RB_TRI(i, kb + c)] * yb[c]; x[i] = a; }
    BSYNC();
  }
  const int last = ((n - 1) / RB_NB) * RB_NB;
  for (int kb = last; kb >= 0; kb -= RB_NB, par ^= 1) {      // backward: L' z = y
    const int nb = n - kb < RB_NB ? n - kb : RB_NB;
    float* yb = s.yb + RB_NB * par;
    if (TID < nb) { const float* D = s.Dinv + (kb / RB_NB) * RB_NB * RB_NB; float v = 0.f; for (int c = TID; c < nb; c++) v += D[c * RB_NB + TID] * x[kb + c]; yb[TID] = v; }   // inv(L')[r][c] = Dinv[c][r]
    BSYNC();
    if (TID < nb) x[kb + TID] = yb[TID];
    const int i = TID;
    if (i < kb) { float a = x[i]; for (int c = 0; c < nb; c++) a -= s.A[RB_TRI(kb + c, i)] * yb[c]; x[i] = a; }
    BSYNC();
  }
}
#if RB_T == 256 && !defined(RB_LDS_CHOL)
// Round 6: the same substitutions by ONE wave, no barriers.  rb_chol_solve above runs n / 8 blocks with two workgroup barriers each, twice (49 k cycles for the
// full cube's 96 dofs, 11 times per mj_step).  Here lane i of wave 0 owns rows i and i + 64 of the right-hand side; column j's solution y_j is formed by its lane and
// travels by v_readlane, every lane subtracts L[i][j] y_j from its rows -- the factor's entries come from the packed block in LDS, fetched ONE COLUMN AHEAD so that
// their latency sits beside the arithmetic.  2 n dependent steps of a multiply, a broadcast and a fused multiply-add.
__device__ __forceinline__ void rb_chol_solve_wave(RbLds& s, int n, float* x) {
  if (WID == 0) {
    const int i0 = WL, i1 = WL + 64;
    const bool v0 = i0 < n, v1 = i1 < n;
    float b0 = v0 ? x[i0] : 0.f, b1 = v1 ? x[i1] : 0.f;
    const float d0 = v0 ? rg_rcp(s.A[RB_TRI(i0, i0)]) : 0.f, d1 = v1 ? rg_rcp(s.A[RB_TRI(i1, i1)]) : 0.f;
    const int nblk = (n + 7) >> 3;
    float c0[8], c1[8], n0[8], n1[8];
    // forward: L y = b, eight columns per block; c0 / c1: this lane's entries of the block's columns (rows i0, i1), zero on and above the diagonal
#pragma unroll
    for (int q = 0; q < 8; q++) { c0[q] = (v0 && i0 > q && q < n) ? s.A[RB_TRI(i0, q)] : 0.f; c1[q] = (v1 && q < n) ? s.A[RB_TRI(i1, q)] : 0.f; }
    for (int jb = 0; jb < nblk; jb++) {
      const int j0 = 8 * jb, jn = j0 + 8;
#pragma unroll
      for (int q = 0; q < 8; q++) { const int j = jn + q; n0[q] = (v0 && i0 > j && j < n) ? s.A[RB_TRI(i0, j)] : 0.f; n1[q] = (v1 && i1 > j && j < n) ? s.A[RB_TRI(i1, j)] : 0.f; }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int j = j0 + q;   // (columns >= n: c = 0 and the broadcast value is unused)
        const float yl = j0 < 64 ? b0 * d0 : b1 * d1;
        const float yj = lane_bcast(yl, j & 63);
        if (j0 < 64) { if (i0 == j) b0 = yj; } else { if (i1 == j) b1 = yj; }
        b0 = __builtin_fmaf(-c0[q], yj, b0); b1 = __builtin_fmaf(-c1[q], yj, b1);
      }
#pragma unroll
      for (int q = 0; q < 8; q++) { c0[q] = n0[q]; c1[q] = n1[q]; }
    }
    // backward: L' z = y, eight rows of L per block (descending); c0 / c1: this lane's entries of those rows (columns i0, i1), zero on and right of the diagonal
    const int ktop = 8 * nblk - 1;
#pragma unroll
    for (int q = 0; q < 8; q++) { const int k = ktop - q; c0[q] = (v0 && k < n && i0 < k) ? s.A[RB_TRI(k, i0)] : 0.f; c1[q] = (v1 && k < n && i1 < k) ? s.A[RB_TRI(k, i1)] : 0.f; }
    for (int kb = nblk - 1; kb >= 0; kb--) {
      const int k0 = 8 * kb + 7, kn = k0 - 8;
#pragma unroll
      for (int q = 0; q < 8; q++) { const int k = kn - q; n0[q] = (v0 && k >= 0 && i0 < k) ? s.A[RB_TRI(k >= 0 ? k : 0, i0)] : 0.f; n1[q] = (v1 && k >= 0 && i1 < k) ? s.A[RB_TRI(k >= 0 ? k : 0, i1)] : 0.f; }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int k = k0 - q;
        const float zl = k0 - 7 < 64 ? b0 * d0 : b1 * d1;   // (a block of eight rows lies on one side of 64)
        const float zk = lane_bcast(zl, k & 63);
        if (k < n) { if (k0 - 7 < 64) { if (i0 == k) b0 = zk; } else { if (i1 == k) b1 = zk; } }
        b0 = __builtin_fmaf(-c0[q], zk, b0); b1 = __builtin_fmaf(-c1[q], zk, b1);
      }
#pragma unroll
      for (int q = 0; q < 8; q++) { c0[q] = n0[q]; c1[q] = n1[q]; }
    }
    if (v0) x[i0] = b0;
    if (v1) x[i1] = b1;
  }
  BSYNC();
}
#endif
// Symmetric diagonal scaling of the block in s.A before it is factored: A <- S A S with S = diag(1 / sqrt(A_ii)) (s.sc).  The
// blocks mix hand links with 5-gram cubelets and constraint weights D ~ 1e6: in fp32 the unscaled factor is accurate to a few
// digits only and the Newton directions it gives converge linearly (20 iterations instead of 6); scaled, the condition number is
// within a small factor of the best diagonal scaling can do (van der Sluis).
__device__ __forceinline__ void rb_scale_block(RbLds& s, int n) {
  BFOR(l, n) s.sc[l] = rg_rsqrt(fmaxf(s.A[RB_TRI(l, l)], RB_MINVAL));
  BSYNC();
  for (int w = TID; w < n * n; w += RB_T) { const int i = w / n, j = w - i * n; if (j <= i) s.A[RB_TRI(i, j)] *= s.sc[i] * s.sc[j]; }
  BSYNC();
}
// dst[dofs of group g] = inv(block) applied to src[dofs of group g] (block scaled by rb_scale_block, then factored), through s.x
__device__ __forceinline__ void rb_group_solve(RbM m, RbLds& s, int g, const float* src, float* dst, float scale) {
  const int g0 = m.b_group_adr[g], n = m.b_group_adr[g + 1] - g0;
  BFOR(l, n) s.x[l] = s.sc[l] * src[m.b_group_dofs[g0 + l]];
  BSYNC();
#if RB_T == 256 && !defined(RB_LDS_CHOL)
  rb_chol_solve_wave(s, n, s.x);
#else
  rb_chol_solve(s, n, s.x);
#endif
  BFOR(l, n) dst[m.b_group_dofs[g0 + l]] = scale * s.sc[l] * s.x[l];
  BSYNC();
}

#if RB_T == 64
// ---- Round 6, one-wave configurations: the group's Newton step x = inv(S A S) (S grad) in REGISTERS (the construction of rg_kernel.h's rg_chol_inv_solve_n).
// rb_chol + rb_chol_solve above are built for 256 threads: per 8 columns three workgroup barriers, the diagonal block through LDS, the substitutions two barriers per
// block -- 43 k + 22 k cycles for the rearrange main world's 38-dof group on ONE wave that waits 72 % of its cycles.  Here lane i < n loads row i of the packed block
// (scaled by the diagonal, as rb_scale_block does) into NP registers, lane NP holds the right-hand side as one more row, lanes n .. NP - 1 pad with the identity, and
// the wave runs the right-looking elimination with pivots and multipliers by v_readlane: no LDS, no barrier.  The right-hand side row comes out as y = inv(L) g;
// the backward substitution runs four columns per step: four independent wave sums of L[k][c] x_k over the lanes k that are done, then the 4 x 4 triangle on
// wave-uniform numbers.  NP = the padded size (template: straight-line code).
template <int NP> __device__ __forceinline__ bool rb_reg_solve_n(RbM m, RbLds& s, int g0, int n, const float* src, float* dst, float scale) {
  static_assert(NP % 4 == 0 && NP < 64, "row NP = the right-hand side");
  const int i = LANE;
  const float sci = i < n ? s.sc[i] : 1.f;
  const int dof_own = i < n ? m.b_group_dofs[g0 + i] : 0;                    // (one vector load; the right-hand side reaches its row's registers lane by lane)
  const float rhs_own = i < n ? sci * src[dof_own] : 0.f;
  float a[NP];
  {
    const bool isrow = i < n, isrhs = i == NP;
    const int rbase = isrow ? RB_TRI(i, 0) : 0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const float sck = lane_bcast(sci, k), rk = lane_bcast(rhs_own, k);
      float v = (k == i) ? 1.f : 0.f;                                        // padding rows: identity
      if (isrow) v = k <= i ? s.A[rbase + k] * sci * sck : 0.f;              // (entries right of the diagonal are never read)
      if (isrhs) v = rk;
      a[k] = v;
    }
  }
  bool ok = true;
  float dv = 1.f;   // lane j: 1 / L[j][j]
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const float p = lane_bcast(a[j], j);
    if (!(p > RB_MINVAL)) ok = false;
    const float inv = rg_rsqrt(fmaxf(p, RB_MINVAL));
    a[j] *= inv;
    if (i == j) dv = inv;
    const float naj = -a[j];
#pragma unroll
    for (int k = j + 1; k < NP; k++) a[k] = __builtin_fmaf(naj, lane_bcast(a[j], k), a[k]);
  }
  // backward substitution L' x = y, four columns per step (lane k holds x_k once it is known)
  float xk = 0.f;
#pragma unroll
  for (int c0 = NP - 4; c0 >= 0; c0 -= 4) {
    const bool below = i > c0 + 3 && i < NP;
    float p0 = below ? a[c0] * xk : 0.f, p1 = below ? a[c0 + 1] * xk : 0.f, p2 = below ? a[c0 + 2] * xk : 0.f, p3 = below ? a[c0 + 3] * xk : 0.f;
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2); p3 = wave_sum(p3);
    const float i0 = lane_bcast(dv, c0), i1 = lane_bcast(dv, c0 + 1), i2 = lane_bcast(dv, c0 + 2), i3 = lane_bcast(dv, c0 + 3);
    const float x3 = (lane_bcast(a[c0 + 3], NP) - p3) * i3;
    const float x2 = (lane_bcast(a[c0 + 2], NP) - p2 - lane_bcast(a[c0 + 2], c0 + 3) * x3) * i2;
    const float x1 = (lane_bcast(a[c0 + 1], NP) - p1 - lane_bcast(a[c0 + 1], c0 + 3) * x3 - lane_bcast(a[c0 + 1], c0 + 2) * x2) * i1;
    const float x0 = (lane_bcast(a[c0], NP) - p0 - lane_bcast(a[c0], c0 + 3) * x3 - lane_bcast(a[c0], c0 + 2) * x2 - lane_bcast(a[c0], c0 + 1) * x1) * i0;
    if (i == c0) xk = x0; else if (i == c0 + 1) xk = x1; else if (i == c0 + 2) xk = x2; else if (i == c0 + 3) xk = x3;
  }
  if (i < n) dst[dof_own] = scale * sci * xk;
  BSYNC();
  return ok;
}
// dst[dofs of group g] = scale * inv(block in s.A) src[dofs of group g]; false on a non-positive pivot
__device__ __forceinline__ bool rb_reg_solve(RbM m, RbLds& s, int g, const float* src, float* dst, float scale) {
  const int g0 = m.b_group_adr[g], n = m.b_group_adr[g + 1] - g0;
  BFOR(l, n) s.sc[l] = rg_rsqrt(fmaxf(s.A[RB_TRI(l, l)], RB_MINVAL));
  BSYNC();
  if (n <= 8) return rb_reg_solve_n<8>(m, s, g0, n, src, dst, scale);
  if (n <= 16) return rb_reg_solve_n<16>(m, s, g0, n, src, dst, scale);
  if (n <= 24) return rb_reg_solve_n<24>(m, s, g0, n, src, dst, scale);
#if RB_MAXGROUP <= 40
  return rb_reg_solve_n<40>(m, s, g0, n, src, dst, scale);
#else
  if (n <= 40) return rb_reg_solve_n<40>(m, s, g0, n, src, dst, scale);
  return rb_reg_solve_n<RB_MAXGROUP>(m, s, g0, n, src, dst, scale);
#endif
}
#endif

// "Star" trees (big_tables.py b_tree_*: a chain of <= 6 root dofs with simple chains of <= RB_STARB dofs hanging off its last dof -- the cubes, the
// hand): M and M + h B of such a tree, and the Newton Hessian M + a diagonal of a tree no contact or tendon touches (the target cube), have the
// tree's sparsity, [[R, C'], [C, blockdiag(B_k)]] in (root | chains) order.  Block elimination instead of a dense factorisation: thread k
// factors its chain's B_k = L L' (<= 5 x 5, registers), forms Z_k = inv(L) C_k and u_k = inv(L) y_k and leaves its contribution Z_k' Z_k,
// Z_k' u_k in LDS; the contributions are summed in chain order (deterministic), every thread then solves the <= 6 x 6 Schur system
// (R - sum Z'Z) x_r = y_r - sum Z'u redundantly in registers and back-substitutes its own chain.
// dst[dofs of tree t] = scale * inv(M_t + dscale * diag(diag)) src[dofs of tree t]; a non-positive pivot sets RG_STATUS_BAD_FACTOR.
__device__ __forceinline__ void rb_star_solve(RbM m, RbLds& s, const float* Msp, int t, const float* diag, float dscale, const float* src, float* dst, float scale) {
  const int r0 = m.b_tree_desc[4 * t + 1], nr = m.b_tree_desc[4 * t + 2];
  const int k0 = m.b_tree_desc[4 * t + 3], T = m.b_tree_brn_end[t] - k0;
  float* W = s.A;                     // [T][28]: lower triangle of Z'Z (21), Z'u (6)
  float* red = s.A + 28 * T;          // the 27 sums (big_tables.py: T <= 128)
  bool ok = true;
  float L[RB_STARB][RB_STARB], idg[RB_STARB], Z[RB_STARB][6], uu[RB_STARB];
  int f = 0, b = 0;
#pragma unroll
  for (int q = 0; q < RB_STARB; q++) {
    uu[q] = 0.f; idg[q] = 1.f;
#pragma unroll
    for (int a = 0; a < 6; a++) Z[q][a] = 0.f;
#pragma unroll
    for (int c = 0; c < RB_STARB; c++) L[q][c] = q == c ? 1.f : 0.f;
  }
  if (TID < T) {
    f = m.b_tree_branch[2 * (k0 + TID)]; b = m.b_tree_branch[2 * (k0 + TID) + 1];
#pragma unroll
    for (int q = 0; q < RB_STARB; q++) if (q < b) {
      const int i = f + q, e0 = m.b_M_adr[i];   // entries (i, i), (i, i - 1), ..., (i, chain top), then (i, last root dof) ... (i, first root dof)
#pragma unroll
      for (int c = 0; c < RB_STARB; c++) if (c <= q) L[q][q - c] = Msp[e0 + c];
      if (diag) L[q][q] += dscale * diag[i];
#pragma unroll
      for (int a = 0; a < 6; a++) if (a < nr) Z[q][nr - 1 - a] = Msp[e0 + q + 1 + a];
      uu[q] = src[i];
    }
#pragma unroll
    for (int c = 0; c < RB_STARB; c++) {           // Cholesky of the chain's block (rows beyond the chain: identity)
      float d = L[c][c];
#pragma unroll
      for (int q = 0; q < RB_STARB; q++) if (q < c) d -= L[c][q] * L[c][q];
      ok = ok && d > RB_MINVAL;
      idg[c] = rg_rsqrt(fmaxf(d, RB_MINVAL));
      L[c][c] = fmaxf(d, RB_MINVAL) * idg[c];
#pragma unroll
      for (int r = 0; r < RB_STARB; r++) if (r > c) {
        float v = L[r][c];
#pragma unroll
        for (int q = 0; q < RB_STARB; q++) if (q < c) v -= L[r][q] * L[c][q];
        L[r][c] = v * idg[c];
      }
    }
#pragma unroll
    for (int r = 0; r < RB_STARB; r++) {           // Z <- inv(L) Z, u <- inv(L) u
#pragma unroll
      for (int q = 0; q < RB_STARB; q++) if (q < r) {
#pragma unroll
        for (int a = 0; a < 6; a++) Z[r][a] -= L[r][q] * Z[q][a];
        uu[r] -= L[r][q] * uu[q];
      }
#pragma unroll
      for (int a = 0; a < 6; a++) Z[r][a] *= idg[r];
      uu[r] *= idg[r];
    }
    float* w = W + 28 * TID;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int c = 0; c < 6; c++) if (c <= a) { float v = 0.f; for (int r = 0; r < RB_STARB; r++) v += Z[r][a] * Z[r][c]; w[a * (a + 1) / 2 + c] = v; }
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < RB_STARB; r++) v += Z[r][a] * uu[r];
      w[21 + a] = v;
    }
  }
  BSYNC();
  if (TID < 27) { float acc = 0.f; for (int k = 0; k < T; k++) acc += W[28 * k + TID]; red[TID] = acc; }
  BSYNC();
  // the Schur system, redundantly
  float Sm[6][6], xr[6];
#pragma unroll
  for (int p = 0; p < 6; p++) {
#pragma unroll
    for (int c = 0; c < 6; c++) if (c <= p) Sm[p][c] = (p == c) ? 1.f : 0.f;
    xr[p] = 0.f;
    if (p < nr) {
      const int i = r0 + p, e0 = m.b_M_adr[i];
#pragma unroll
      for (int c = 0; c < 6; c++) if (c <= p) Sm[p][p - c] = Msp[e0 + c] - red[p * (p + 1) / 2 + p - c];
      if (diag) Sm[p][p] += dscale * diag[i];
      xr[p] = src[i] - red[21 + p];
    }
  }
  float sdg[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    float d = Sm[c][c];
#pragma unroll
    for (int q = 0; q < 6; q++) if (q < c) d -= Sm[c][q] * Sm[c][q];
    ok = ok && d > RB_MINVAL;
    sdg[c] = rg_rsqrt(fmaxf(d, RB_MINVAL));
    Sm[c][c] = fmaxf(d, RB_MINVAL) * sdg[c];
#pragma unroll
    for (int p = 0; p < 6; p++) if (p > c) {
      float v = Sm[p][c];
#pragma unroll
      for (int q = 0; q < 6; q++) if (q < c) v -= Sm[p][q] * Sm[c][q];
      Sm[p][c] = v * sdg[c];
    }
  }
#pragma unroll
  for (int p = 0; p < 6; p++) {   // L y = t
    float v = xr[p];
#pragma unroll
    for (int q = 0; q < 6; q++) if (q < p) v -= Sm[p][q] * xr[q];
    xr[p] = v * sdg[p];
  }
#pragma unroll
  for (int p = 5; p >= 0; p--) {  // L' x = y
    float v = xr[p];
#pragma unroll
    for (int q = 0; q < 6; q++) if (q > p) v -= Sm[q][p] * xr[q];
    xr[p] = v * sdg[p];
  }
  if (TID < T) {
    float x[RB_STARB];
#pragma unroll
    for (int r = 0; r < RB_STARB; r++) { float v = uu[r]; for (int a = 0; a < 6; a++) v -= Z[r][a] * xr[a]; x[r] = v; }
#pragma unroll
    for (int r = RB_STARB - 1; r >= 0; r--) {    // L' x = w
      float v = x[r];
#pragma unroll
      for (int q = 0; q < RB_STARB; q++) if (q > r) v -= L[q][r] * x[q];
      x[r] = v * idg[r];
    }
#pragma unroll
    for (int r = 0; r < RB_STARB; r++) if (r < b) dst[f + r] = scale * x[r];
  }
  if (TID == 0) {
#pragma unroll
    for (int p = 0; p < 6; p++) if (p < nr) dst[r0 + p] = scale * xr[p];
  }
  if (!ok) s.status |= RG_STATUS_BAD_FACTOR;   // (every writer ORs the same bit into the same word; nobody else writes it in this phase)
  BSYNC();
}
// every tree of group g through rb_star_solve (groups whose trees are all stars: b_star_grp[4 g + 3])
__device__ __forceinline__ void rb_star_group_solve(RbM m, RbLds& s, const float* Msp, int g, const float* diag, float dscale, const float* src, float* dst, float scale) {
  for (int t = m.b_tree_adr[g]; t < m.b_tree_adr[g + 1]; t++) rb_star_solve(m, s, Msp, t, diag, dscale, src, dst, scale);
}

// Models whose trees all have <= 8 dofs in one contiguous range (b_tree8 = [count, (first dof, dofs) ...]: the rearrange worlds -- one arm +
// gripper tree, one free body per object): groups of 8 lanes factor one tree each, all trees at once.  Lane r of a group holds row r of the
// tree's block of M (+ dscale diag) in registers; pivots and multipliers travel by lane exchange inside the group (no barrier, no LDS), the factor
// is parked in LDS only for the transposed reads of the backward substitution.  dst[tree dofs] = scale * inv(M_t + dscale diag) src[tree dofs].
// (rb_star_solve, one tree after the other with a redundant 6 x 6 Schur system per thread, was 11 % of the rearrange step.)
// value of lane `src` (a per-lane index: ds_bpermute; lane_bcast = v_readlane needs a wave-uniform one)
__device__ __forceinline__ float grp8_get(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ void rb_trees8_solve(RbM m, RbLds& s, const float* Msp, const float* diag, float dscale, const float* src, float* dst, float scale) {
  const int nt = m.b_tree8[0];
  bool ok = true;
  for (int base = 0; base < nt; base += RB_T / 8) {
    const int t = base + (TID >> 3), r = TID & 7, g0 = WL & ~7;
    const bool on = t < nt;
    const int first = on ? m.b_tree8[1 + 2 * t] : 0, n = on ? m.b_tree8[2 + 2 * t] : 0;
    float Lr[8], idg[8];
#pragma unroll
    for (int q = 0; q < 8; q++) Lr[q] = (q == r && r >= n) ? 1.f : 0.f;   // (identity padding beyond the tree's dofs)
    if (r < n) {
      // row i of M in tree-sparse storage: (i, i), (i, parent), (i, grandparent), ... -- the columns between them are structural zeros
      const int i = first + r;
      int e = m.b_M_adr[i], j = i;
#pragma unroll
      for (int q = 7; q >= 0; q--) if (j == first + q) {
        float v = Msp[e];
        if (j == i && diag) v += dscale * diag[i];
        Lr[q] = v; e++; j = m.dof_parentid[j];
      }
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const float d = grp8_get(Lr[c], g0 + c);
      if (c < n && !(d > RB_MINVAL)) ok = false;
      idg[c] = rg_rsqrt(fmaxf(d, RB_MINVAL));
      Lr[c] = (r == c) ? fmaxf(d, RB_MINVAL) * idg[c] : Lr[c] * idg[c];
#pragma unroll
      for (int q = 0; q < 8; q++) if (q > c) { const float lqc = grp8_get(Lr[c], g0 + q); Lr[q] -= Lr[c] * lqc; }
    }
    float x = r < n ? src[first + r] : 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {          // L y = b
      const float yc = grp8_get(x, g0 + c) * idg[c];
      if (r == c) x = yc; else if (r > c) x -= Lr[c] * yc;
    }
    float* Lt = s.A + 64 * (TID >> 3);
    BSYNC();                               // (s.A may still be read as the previous trip's factor)
#pragma unroll
    for (int q = 0; q < 8; q++) Lt[8 * r + q] = q <= r ? Lr[q] : 0.f;
    BSYNC();
#pragma unroll
    for (int c = 7; c >= 0; c--) {         // L' z = y
      const float zc = grp8_get(x, g0 + c) * idg[c];
      if (r == c) x = zc; else if (r < c) x -= Lt[8 * c + r] * zc;
    }
    if (r < n) dst[first + r] = scale * x;
  }
  if (__ballot(!ok) != 0ull && WL == 0) s.status |= RG_STATUS_BAD_FACTOR;
  BSYNC();
}

// ------------------------------------------------------------------------------------------------- velocity stage
// mj_comVel, mj_passive, mj_rne (zero acceleration: Coriolis, centrifugal, gravity)
__device__ __forceinline__ void rb_velocity(RbM m, RbLds& s, float* S) {
  const float* cdof = SC(CDOF); float *cdofdot = SC(CDOFDOT), *cvel = SC(CVEL), *cacc = SC(CACC), *cfrc = SC(CFRC);
  if (TID < 6) { cvel[TID] = 0.f; cacc[TID] = TID < 3 ? 0.f : -PRM(opt_gravity, RB_P_GRAVITY)[TID - 3]; cfrc[TID] = 0.f; }
  BSYNC();
  const bool wave_sweep = RB_NWAVE == 1 && m.nbody <= 64;   // (as in rb_kinematics: body b in lane b, the parent's cvel / cacc by lane exchange)
  if (wave_sweep) {
    const int b = TID; const bool on = b > 0 && b < m.nbody;
    const int p = on ? m.body_parentid[b] : 0, lvl = on ? m.b_body_level[b] : -1;
    float cv[6], ca[6];
#pragma unroll
    for (int c = 0; c < 6; c++) { cv[c] = 0.f; ca[c] = (b == 0 && c >= 3) ? -PRM(opt_gravity, RB_P_GRAVITY)[c - 3] : 0.f; }
    // (as in rb_kinematics: a body with ONE hinge or slide joint -- every link of the arm's ten-level chain -- has its joint's motion axis and velocity fetched before
    //  the sweep and keeps the axis' time derivative in registers: no memory round trip inside a level for it; the arithmetic is unchanged)
    const int bb = on ? b : 0;
    const int jn = on ? m.body_jntnum[bb] : 0, j0 = m.body_jntadr[bb], dn = on ? m.body_dofnum[bb] : 0, d0 = m.body_dofadr[bb];
    const int t0 = m.jnt_type[jn > 0 ? j0 : 0], da0 = m.jnt_dofadr[jn > 0 ? j0 : 0];
    const bool single = jn == 1 && dn == 1 && (t0 == RG_JNT_HINGE || t0 == RG_JNT_SLIDE);
    float cd0[6]; const float qv0 = s.qvel[single ? da0 : 0];
#pragma unroll
    for (int c = 0; c < 6; c++) cd0[c] = cdof[6 * (single ? da0 : 0) + c];
    for (int L = 0; L < m.nlevel; L++) {
      float pv[6], pa[6];
#pragma unroll
      for (int c = 0; c < 6; c++) { pv[c] = __shfl(cv[c], p); pa[c] = __shfl(ca[c], p); }
      if (lvl != L) continue;
#pragma unroll
      for (int c = 0; c < 6; c++) { cv[c] = pv[c]; ca[c] = pa[c]; }
      if (single) {
        float dd[6];
        cross_motion(dd, cv, cd0);
#pragma unroll
        for (int c = 0; c < 6; c++) { cdofdot[6 * da0 + c] = dd[c]; cv[c] += cd0[c] * qv0; }
#pragma unroll
        for (int c = 0; c < 6; c++) ca[c] += dd[c] * qv0;
        continue;
      }
      for (int k = 0; k < jn; k++) {
        const int j = j0 + k, t = m.jnt_type[j]; int da = m.jnt_dofadr[j];
        if (t == RG_JNT_FREE) {
          for (int i = 0; i < 3; i++) { for (int c = 0; c < 6; c++) { cdofdot[6 * (da + i) + c] = 0.f; cv[c] += cdof[6 * (da + i) + c] * s.qvel[da + i]; } }
          da += 3;
        }
        if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
          for (int i = 0; i < 3; i++) cross_motion(cdofdot + 6 * (da + i), cv, cdof + 6 * (da + i));
          for (int i = 0; i < 3; i++) for (int c = 0; c < 6; c++) cv[c] += cdof[6 * (da + i) + c] * s.qvel[da + i];
        } else {
          cross_motion(cdofdot + 6 * da, cv, cdof + 6 * da);
          for (int c = 0; c < 6; c++) cv[c] += cdof[6 * da + c] * s.qvel[da];
        }
      }
      for (int k = 0; k < dn; k++) { const int i = d0 + k; for (int c = 0; c < 6; c++) ca[c] += cdofdot[6 * i + c] * s.qvel[i]; }
    }
    if (on) {
      float t1[6], t2[6], t3[6];
      mul_inert_vec(t1, SC(CINERT) + 10 * b, ca);
      mul_inert_vec(t2, SC(CINERT) + 10 * b, cv);
      cross_force(t3, cv, t2);
      for (int c = 0; c < 6; c++) { cvel[6 * b + c] = cv[c]; cacc[6 * b + c] = ca[c]; cfrc[6 * b + c] = t1[c] + t3[c]; }
    }
    BSYNC();
  }
  for (int L = 0; L < (wave_sweep ? 0 : m.nlevel); L++) {
    for (int q = m.b_lvl_adr[L] + TID; q < m.b_lvl_adr[L + 1]; q += RB_T) {
      const int b = m.b_lvl_body[q], p = m.body_parentid[b];
      float cv[6], ca[6];
      for (int c = 0; c < 6; c++) { cv[c] = cvel[6 * p + c]; ca[c] = cacc[6 * p + c]; }
      for (int k = 0; k < m.body_jntnum[b]; k++) {
        const int j = m.body_jntadr[b] + k, t = m.jnt_type[j]; int da = m.jnt_dofadr[j];
        if (t == RG_JNT_FREE) {
          for (int i = 0; i < 3; i++) { for (int c = 0; c < 6; c++) { cdofdot[6 * (da + i) + c] = 0.f; cv[c] += cdof[6 * (da + i) + c] * s.qvel[da + i]; } }
          da += 3;
        }
        if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
          for (int i = 0; i < 3; i++) cross_motion(cdofdot + 6 * (da + i), cv, cdof + 6 * (da + i));
          for (int i = 0; i < 3; i++) for (int c = 0; c < 6; c++) cv[c] += cdof[6 * (da + i) + c] * s.qvel[da + i];
        } else {
          cross_motion(cdofdot + 6 * da, cv, cdof + 6 * da);
          for (int c = 0; c < 6; c++) cv[c] += cdof[6 * da + c] * s.qvel[da];
        }
      }
      for (int k = 0; k < m.body_dofnum[b]; k++) { const int i = m.body_dofadr[b] + k; for (int c = 0; c < 6; c++) ca[c] += cdofdot[6 * i + c] * s.qvel[i]; }
      float t1[6], t2[6], t3[6];
      mul_inert_vec(t1, SC(CINERT) + 10 * b, ca);
      mul_inert_vec(t2, SC(CINERT) + 10 * b, cv);
      cross_force(t3, cv, t2);
      for (int c = 0; c < 6; c++) { cvel[6 * b + c] = cv[c]; cacc[6 * b + c] = ca[c]; cfrc[6 * b + c] = t1[c] + t3[c]; }
    }
    BSYNC();
  }
  BFOR(t, m.ntendon) {
    float v = 0;
    for (int e = 0; e < RB_TENW; e++) { const int d = m.b_ten_dofs[RB_TENW * t + e]; if (d >= 0) v += SC(TENJ)[RB_TENW * t + e] * s.qvel[d]; }
    SC(TENVEL)[t] = v;
  }
  BSYNC();
  const float* jnt_stiffness = PRM(jnt_stiffness, RB_P_JNT_STIFFNESS);
  BFOR(i, m.nv) {
    // passive: joint spring, dof damping, tendon spring-dampers
    const int j = m.dof_jntid[i], jt = m.jnt_type[j];
    float f = -PRM(dof_damping, RB_P_DOF_DAMPING)[i] * s.qvel[i];
    if ((jt == RG_JNT_HINGE || jt == RG_JNT_SLIDE) && jnt_stiffness[j] != 0.f) { const int qa = m.jnt_qposadr[j]; f -= jnt_stiffness[j] * (s.qpos[qa] - m.qpos_spring[qa]); }
    for (int t = 0; t < m.ntendon; t++) {
      const float tf = m.tendon_stiffness[t] * (m.tendon_lengthspring[t] - SC(TENLEN)[t]) - m.tendon_damping[t] * SC(TENVEL)[t];
      if (tf != 0.f) for (int e = 0; e < RB_TENW; e++) if (m.b_ten_dofs[RB_TENW * t + e] == i) f += SC(TENJ)[RB_TENW * t + e] * tf;
    }
    s.qfrc_passive[i] = f;
    // bias: cdof_i . (sum of cfrc over the subtree of the dof's body)
    const int b = m.dof_bodyid[i];
    float acc[6] = {0, 0, 0, 0, 0, 0};
    for (int q = m.b_subtree_adr[b]; q < m.b_subtree_adr[b + 1]; q++) { const float* cf = cfrc + 6 * m.b_subtree[q]; for (int c = 0; c < 6; c++) acc[c] += cf[c]; }
    const float* c = cdof + 6 * i;
    s.qfrc_bias[i] = c[0] * acc[0] + c[1] * acc[1] + c[2] * acc[2] + c[3] * acc[3] + c[4] * acc[4] + c[5] * acc[5];
  }
  BSYNC();
}

// ------------------------------------------------------------------------------------------------- actuation
// mujoco-py's PID callback (mjpid.pyx semantics as restated by oracle ro_fwd_actuation); `apply`: also qfrc_actuator
// one tick of one actuator's controller: st = {integral, previous error, smoothed derivative}; returns the clamped force
// (gainprm / forcerange: the model's arrays or the env's own rows)
template <class Model>
__device__ __forceinline__ float rb_pid_tick(const Model& m, const float* gainprm, const float* forcerange, int u, float ctrl, float length, float* st) {
  const float dt = m.timestep;
  const float* gp = gainprm + 10 * u;
  const float kp = gp[0], ti = gp[1], iclamp = gp[2], td = gp[3], smooth = gp[4], deadband = gp[5];
  float err = ctrl - length;
  if (fabsf(err) < deadband) err = 0.f;
  const float integ = clampf(st[0] + err * dt, -iclamp, iclamp);
  const float deriv = (1.f - smooth) * st[2] + smooth * (err - st[1]) / dt;
  float force = kp * (err + (ti != 0.f ? integ / ti : 0.f) + td * deriv);
  st[0] = integ; st[1] = err; st[2] = deriv;
  const float lo = forcerange[2 * u], hi = forcerange[2 * u + 1];
  if (lo != 0.f || hi != 0.f) force = clampf(force, lo, hi);
  if (m.actuator_forcelimited[u]) force = clampf(force, lo, hi);
  return force;
}
// mujoco-py's cascaded PI controller (actuator user[0] == 1; ur16e/jointspec/calibrations/cascaded_pi/joint_actuations.xml:4-10), as restated by
// oracle ro_fwd_actuation: gainprm = [kp, ti, iclamp, td, dsmooth | kp_v, ti_v, iclamp_v, ema, max_vel]; st = {position integral, velocity
// integral, smoothed set-point}.  EMA-smoothed position set-point (warm start at time 0) -> P(I) on position -> velocity set-point clamped to
// +- max_vel -> PI on actuator_velocity -> plus `bias_ff`, the bias force (gravity + Coriolis) of the actuated dof over the gear -> clamped to forcerange.
// (The feed-forward is inferred from the reference's impulse-response pins, oracle/rg_oracle.c ro_cascade_bias_ff: the wrist joints' velocity loops are P-only.)
template <class Model>
__device__ __forceinline__ float rb_cascade_tick(const Model& m, const float* gainprm, const float* forcerange, int u, float ctrl, float length, float velocity, float bias_ff, bool time0, float* st) {
  const float dt = m.timestep;
  const float* gp = gainprm + 10 * u;
  const float setp = time0 ? ctrl : gp[8] * st[2] + (1.f - gp[8]) * ctrl;
  st[2] = setp;
  float des_vel;
  if (gp[0] != 0.f) {
    const float err = setp - length;
    const float integ = clampf(st[0] + err * dt, -gp[2], gp[2]);
    des_vel = gp[0] * (err + (gp[1] != 0.f ? integ / gp[1] : 0.f));
    st[0] = integ;
  } else des_vel = ctrl;
  des_vel = clampf(des_vel, -gp[9], gp[9]);
  const float errv = des_vel - velocity;
  const float integv = clampf(st[1] + errv * dt, -gp[7], gp[7]);
  float force = gp[5] * (errv + (gp[6] != 0.f ? integv / gp[6] : 0.f)) + bias_ff;
  st[1] = integv;
  const float lo = forcerange[2 * u], hi = forcerange[2 * u + 1];
  if (lo != 0.f || hi != 0.f) force = clampf(force, lo, hi);
  if (m.actuator_forcelimited[u]) force = clampf(force, lo, hi);
  return force;
}
__device__ __forceinline__ void rb_pid(RbM m, RbLds& s, float* S, bool apply) {
  const float *gainprm = PRM(actuator_gainprm, RB_P_ACT_GAINPRM), *forcerange = PRM(actuator_forcerange, RB_P_ACT_FORCERANGE);
  BFOR(u, m.nu) {
    if (m.actuator_user[u] == 1.f) {
      const int id = m.actuator_trnid[u];
      const bool joint = m.actuator_trntype[u] == 0;
      const float vel = m.actuator_gear[u] * (joint ? s.qvel[m.jnt_dofadr[id]] : SC(TENVEL)[id]);   // mj_transmission: moment . qvel
      // (a state-less tick, apply == false, keeps only the controller state, which the feed-forward never enters: it is not read there — the TCP
      // hook's sync tick runs before any stage of the launch has written qfrc_bias)
      const float ff = (apply && joint) ? s.qfrc_bias[m.jnt_dofadr[id]] / m.actuator_gear[u] : 0.f;
      s.actfrc[u] = rb_cascade_tick(m, gainprm, forcerange, u, s.ctrl[u], s.actlen[u], vel, ff, s.time == 0.f, s.pid + 3 * u);
    } else s.actfrc[u] = rb_pid_tick(m, gainprm, forcerange, u, s.ctrl[u], s.actlen[u], s.pid + 3 * u);
  }
  BSYNC();
  if (!apply) return;
  BFOR(i, m.nv) {
    float f = 0;
    for (int u = 0; u < m.nu; u++) {
      const int id = m.actuator_trnid[u]; const float g = m.actuator_gear[u];
      if (m.actuator_trntype[u] == 0) { if (m.jnt_dofadr[id] == i) f += g * s.actfrc[u]; }
      else for (int e = 0; e < RB_TENW; e++) if (m.b_ten_dofs[RB_TENW * id + e] == i) f += g * SC(TENJ)[RB_TENW * id + e] * s.actfrc[u];
    }
    s.qfrc_act[i] = f;
    s.qfrc_smooth[i] = s.qfrc_passive[i] - s.qfrc_bias[i] + f;
  }
  BSYNC();
}

// ------------------------------------------------------------------------------------------------- collision
__device__ __forceinline__ void rb_geom(RbM m, const float* S, int g, MprGeom& G) {
  G.type = m.geom_type[g]; G.quat = SC(GQUAT) + 4 * g; G.size = ld3(m.geom_size + 3 * g); G.mesh = -1; G.vertadr = 0; G.nvert = 0;
  if (G.type == RG_GEOM_MESH) { G.mesh = m.geom_dataid[g]; G.vertadr = m.mesh_vertadr[G.mesh]; G.nvert = m.mesh_vertnum[G.mesh]; }
}
__device__ __forceinline__ void rb_make_frame(float* f) {   // mju_makeFrame: f[0..2] given, tangents completed (as oracle make_frame)
  make_frame(f);
}
// mj_instantiateEquality (oracle ro_make_constraint, equality block): every active equality becomes one record at the head of the contact
// list — weld (6 rows: position residual (xpos1 + R1 relpos) - xpos2, rotation residual = vector part of conj(q2) q1 relquat) or joint
// coupling (1 row: q1 - q1_0 - poly(q2 - q2_0)).  Residuals in frame[0..5], diagApprox (translational, rotational) in friction[0..1].
__device__ __forceinline__ void rb_equality(RbM m, RbLds& s, float* S, const float* eq_data, const int* eq_active) {
  float* con = SC(CON);
  const float *biw = PRM(body_invweight0, RB_P_BODY_INVWEIGHT0), *diw = PRM(dof_invweight0, RB_P_DOF_INVWEIGHT0);
  if (TID == 0) {
    int n = 0;
    for (int e = 0; e < m.neq; e++) {
      if (!eq_active[e] || n >= m.maxcon) continue;
      float* C = con + RB_CONREC * n;
      const float* data = eq_data + 7 * e;
      for (int k = 0; k < RB_CONREC; k++) C[k] = 0.f;
      C[RB_CR_KIND] = (float)RB_KIND_EQUALITY; C[RB_CR_G1] = (float)e; C[RB_CR_G2] = 0.f; C[RB_CR_ADR] = -1.f;
      C[RB_CR_SOLREF] = m.eq_solref[2 * e]; C[RB_CR_SOLREF + 1] = m.eq_solref[2 * e + 1];
      for (int k = 0; k < 5; k++) C[RB_CR_SOLIMP + k] = m.eq_solimp[5 * e + k];
      const int o1 = m.eq_obj1id[e], o2 = m.eq_obj2id[e];
      if (m.eq_type[e] == 1) {          // mjEQ_WELD
        const q4 q1 = ldq(SC(XQUAT) + 4 * o1), q2 = ldq(SC(XQUAT) + 4 * o2);
        const v3 p1 = ld3(SC(XPOS) + 3 * o1) + qrot(q1, ld3(data));
        st3(C + RB_CR_POS, p1);
        st3(C + RB_CR_FRAME, p1 - ld3(SC(XPOS) + 3 * o2));
        q4 qc; qc.w = q2.w; qc.x = -q2.x; qc.y = -q2.y; qc.z = -q2.z;
        const q4 r = qmul(qc, qmul(q1, ldq(data + 3)));
        C[RB_CR_FRAME + 3] = r.x; C[RB_CR_FRAME + 4] = r.y; C[RB_CR_FRAME + 5] = r.z;
        C[RB_CR_FRIC] = biw[2 * o1] + biw[2 * o2]; C[RB_CR_FRIC + 1] = biw[2 * o1 + 1] + biw[2 * o2 + 1];
        C[RB_CR_DIM] = 6.f;
      } else {                          // mjEQ_JOINT
        const int qa1 = m.jnt_qposadr[o1];
        float pos = s.qpos[qa1] - m.qpos0[qa1] - data[0], deriv = 0.f, diag = diw[m.jnt_dofadr[o1]];
        if (o2 >= 0) {
          const int qa2 = m.jnt_qposadr[o2]; const float dif = s.qpos[qa2] - m.qpos0[qa2];
          pos -= data[1] * dif + data[2] * dif * dif + data[3] * dif * dif * dif + data[4] * dif * dif * dif * dif;
          deriv = data[1] + 2.f * data[2] * dif + 3.f * data[3] * dif * dif + 4.f * data[4] * dif * dif * dif;
          diag += diw[m.jnt_dofadr[o2]];
        }
        C[RB_CR_FRAME] = pos; C[RB_CR_FRAME + 1] = deriv; C[RB_CR_FRIC] = diag; C[RB_CR_DIM] = 1.f;
      }
      n++;
    }
    s.neqcon = n;
  }
  BSYNC();
}
// mjc_PlaneBox (oracle collide_pair, plane - box): corner l of the box (l < 8) against the plane through the origin of the pair-local frame with
// normal pn; the box centre is at bp
__device__ __forceinline__ bool rb_plane_box_lane(v3 pn, v3 bp, q4 bq, v3 sz, float margin, int l, float& dist, v3& pos) {
  if (l >= 8) return false;
  const v3 lc = mk3((l & 1) ? sz.x : -sz.x, (l & 2) ? sz.y : -sz.y, (l & 4) ? sz.z : -sz.z);
  const v3 c = bp + qrot(bq, lc);
  dist = dot(c, pn);
  pos = c - pn * (0.5f * dist);
  return dist <= margin;
}
// Oriented boxes (b_geom_aabb: centre and half extents in the geom's frame, the hull's for a mesh) further apart than `margin`: the
// separating-axis test on the 15 axes (faces of A, faces of B, edge x edge).  Conservative -- eps keeps near-parallel edge pairs from
// producing a false separation, the slack absorbs the rounding of the frames --, so a pruned pair cannot hold a contact.
__device__ __forceinline__ bool rb_obb_apart(v3 ca, q4 qa, v3 ha, v3 cb, q4 qb, v3 hb, float margin) {
  const v3 ax[3] = {qrot(qa, mk3(1, 0, 0)), qrot(qa, mk3(0, 1, 0)), qrot(qa, mk3(0, 0, 1))};
  const v3 bx[3] = {qrot(qb, mk3(1, 0, 0)), qrot(qb, mk3(0, 1, 0)), qrot(qb, mk3(0, 0, 1))};
  const v3 d = cb - ca;
  const float t[3] = {dot(d, ax[0]), dot(d, ax[1]), dot(d, ax[2])};
  const float a[3] = {ha.x, ha.y, ha.z}, b[3] = {hb.x, hb.y, hb.z};
  float R[3][3], Q[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { R[i][j] = dot(ax[i], bx[j]); Q[i][j] = fabsf(R[i][j]) + 1e-6f; }
  const float mg = margin + 1e-5f;
  bool apart = false;
#pragma unroll
  for (int i = 0; i < 3; i++) apart = apart || fabsf(t[i]) > a[i] + b[0] * Q[i][0] + b[1] * Q[i][1] + b[2] * Q[i][2] + mg;
#pragma unroll
  for (int j = 0; j < 3; j++) apart = apart || fabsf(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]) > a[0] * Q[0][j] + a[1] * Q[1][j] + a[2] * Q[2][j] + b[j] + mg;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      apart = apart || fabsf(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > a[i1] * Q[i2][j] + a[i2] * Q[i1][j] + b[j1] * Q[i][j2] + b[j2] * Q[i][j1] + mg;
    }
  return apart;
}
// mj_collision in three parts (each a stage call of its own, RB_STAGE wrappers below): broadphase, the support-map narrowphase, the multi-point box routines
// mixed contact parameters of static pair p (mj_contactParam; kernel_tables.py collision_pairs is the host statement): out = margin, gap, friction 3, solref 2,
// solimp 5.  The model's precomputed table, or -- per-env geom rows -- the same mixing on the env's own geom_margin / gap / friction / solref / solimp.
__device__ __forceinline__ float rb_pair_margin(RbM m, const float* S, int p) {
  if (!m.prm_on) return m.b_pair_prm[12 * p];
  const float* gm = S + m.prm_off[RB_P_GEOM_MARGIN];
  return fmaxf(gm[m.b_pair_geom[3 * p]], gm[m.b_pair_geom[3 * p + 1]]);
}
__device__ __forceinline__ void rb_pair_prm(RbM m, const float* S, int p, float* out) {
  if (!m.prm_on) { for (int k = 0; k < 12; k++) out[k] = m.b_pair_prm[12 * p + k]; return; }
  const int a = m.b_pair_geom[3 * p], b = m.b_pair_geom[3 * p + 1];
  const float *gm = S + m.prm_off[RB_P_GEOM_MARGIN], *gg = S + m.prm_off[RB_P_GEOM_GAP], *gf = S + m.prm_off[RB_P_GEOM_FRICTION];
  const float *gr = S + m.prm_off[RB_P_GEOM_SOLREF], *gi = S + m.prm_off[RB_P_GEOM_SOLIMP];
  out[0] = fmaxf(gm[a], gm[b]); out[1] = fmaxf(gg[a], gg[b]);
  for (int k = 0; k < 3; k++) out[2 + k] = fmaxf(gf[3 * a + k], gf[3 * b + k]);
  const float m1 = m.geom_solmix[a], m2 = m.geom_solmix[b];
  const float mix = (m1 >= 1e-15f && m2 >= 1e-15f) ? m1 / (m1 + m2) : ((m1 < 1e-15f && m2 < 1e-15f) ? 0.5f : (m1 < 1e-15f ? 0.f : 1.f));
  if (gr[2 * a] > 0.f && gr[2 * b] > 0.f) { out[5] = mix * gr[2 * a] + (1.f - mix) * gr[2 * b]; out[6] = mix * gr[2 * a + 1] + (1.f - mix) * gr[2 * b + 1]; }
  else { out[5] = fminf(gr[2 * a], gr[2 * b]); out[6] = fminf(gr[2 * a + 1], gr[2 * b + 1]); }
  for (int k = 0; k < 5; k++) out[7 + k] = mix * gi[5 * a + k] + (1.f - mix) * gi[5 * b + k];
}
__device__ __forceinline__ void rb_broadphase(RbM m, RbLds& s, float* S, int flags) {
  int* cand = (int*)SC(CAND);
  const float *gpos = SC(GPOS), *gquat = SC(GQUAT);
  if (TID == 0) { s.ncand = 0; s.ncand2 = 0; s.ncon = s.neqcon; }
  BSYNC();
  const bool multipoint = !(flags & 16);   // box - box and plane - box pairs have their own multi-point routines (bit 4: everything through MPR / the support map)
  const bool obb = !(flags & 2048);        // (bit 11: bounding spheres only, the test switch of the oriented-box prune)
  const int halfcand = m.maxcand >> 1;
  // broadphase: the static pair list against bounding spheres (planes: distance of the sphere to the plane), then the geoms' oriented boxes.
  // Two lists: pairs for the support-map routines from the front of the candidate array, box - box / plane - box pairs from its end
  for (int base = 0; base < m.npair; base += RB_T) {
    const int p = base + TID;
    bool keep = false, special = false;
    if (p < m.npair) {
      const int g1 = m.b_pair_geom[3 * p], g2 = m.b_pair_geom[3 * p + 1], t1 = m.geom_type[g1];
      const float margin = rb_pair_margin(m, S, p);
      const v3 P1 = ld3(gpos + 3 * g1), P2 = ld3(gpos + 3 * g2), dif = P2 - P1;
      const float* bb = m.b_geom_aabb + 6 * g2;
      if (t1 == RG_GEOM_PLANE) {
        const v3 pn = qrot(ldq(gquat + 4 * g1), mk3(0, 0, 1));
        keep = dot(dif, pn) <= m.geom_rbound[g2] + margin;
        if (keep && obb) {
          const q4 q2 = ldq(gquat + 4 * g2);
          const v3 c2 = dif + qrot(q2, ld3(bb)), nl = qrotT(q2, pn);
          keep = dot(c2, pn) - (bb[3] * fabsf(nl.x) + bb[4] * fabsf(nl.y) + bb[5] * fabsf(nl.z)) <= margin + 1e-5f;
        }
      } else {
        const float bound = m.geom_rbound[g1] + m.geom_rbound[g2] + margin;
        keep = dot(dif, dif) <= bound * bound;
        if (keep && obb) {
          const float* ba = m.b_geom_aabb + 6 * g1;
          const q4 q1 = ldq(gquat + 4 * g1), q2 = ldq(gquat + 4 * g2);
          keep = !rb_obb_apart(qrot(q1, ld3(ba)), q1, ld3(ba + 3), dif + qrot(q2, ld3(bb)), q2, ld3(bb + 3), margin);
        }
      }
      special = multipoint && m.geom_type[g2] == RG_GEOM_BOX && (t1 == RG_GEOM_BOX || t1 == RG_GEOM_PLANE);
    }
    const int slot = rb_slot(s, keep && !special, &s.ncand, halfcand, RG_STATUS_CAND_FULL);
    if (slot >= 0) cand[slot] = p;
    const int slot2 = rb_slot(s, keep && special, &s.ncand2, halfcand, RG_STATUS_CAND_FULL);
    if (slot2 >= 0) cand[m.maxcand - 1 - slot2] = p;
  }
  BSYNC();   // (the last trip's candidates are written after rb_slot's barrier: the narrowphase below reads them from other waves)
}
__device__ __forceinline__ void rb_narrow_convex(RbM m, RbLds& s, float* S, int flags) {
  const int* cand = (const int*)SC(CAND);
  const float *gpos = SC(GPOS), *gquat = SC(GQUAT);
  const bool multipoint = !(flags & 16);
  (void)multipoint;
  // narrowphase: one quad per candidate, 64 candidates per trip
  MprEnv E; E.mesh_vert = m.b_mesh_rec; E.cell_adr = m.b_cell_adr; E.cell_blk = (const rgf4*)m.b_cell_blk; E.cell_ovf = (const rgf4*)m.b_cell_ovf; E.prof = 0; E.cells = !(flags & 8); E.plane_depth = (flags & 16) != 0;
  const int ncand = s.ncand;
  float* con = SC(CON);
  for (int base = 0; base < ncand; base += RB_T / 4) {
    const int ci = base + (TID >> 2);
    const bool active = ci < ncand;
    bool hit = false; float dist = 0; v3 pos = mk3(0, 0, 0), nrm = mk3(0, 0, 1); int p = 0;
    MprGeom A, B; A.type = RG_GEOM_SPHERE; B.type = RG_GEOM_SPHERE; A.quat = B.quat = gquat; A.size = B.size = mk3(0, 0, 0); A.pos = B.pos = mk3(0, 0, 0);
    A.margin = B.margin = 0; A.mesh = B.mesh = -1; A.vertadr = B.vertadr = 0; A.nvert = B.nvert = 0;
    float margin = 0;
    v3 p1 = mk3(0, 0, 0);
    bool plane = false; const bool special = false;   // (box pairs are on the second list)
    if (active) {
      p = cand[ci];
      const int g1 = m.b_pair_geom[3 * p], g2 = m.b_pair_geom[3 * p + 1];
      margin = rb_pair_margin(m, S, p);
      rb_geom(m, S, g1, A); rb_geom(m, S, g2, B);
      p1 = ld3(gpos + 3 * g1);
      A.pos = mk3(0, 0, 0); B.pos = ld3(gpos + 3 * g2) - p1;
      plane = A.type == RG_GEOM_PLANE;
      A.margin = B.margin = plane ? 0.f : 0.5f * margin;
    }
    v3 sep, dir = mk3(0, 0, 0); float depth = 0;
    // (the quads of a wave take the two branches with their own lanes: the group collectives inside only need the quad)
    const bool mh = rg_mpr<4>(E, A, B, m.mpr_iterations, m.mpr_tolerance, depth, dir, pos, sep, active && !plane && !special);
    if (active && !plane && !special) {
      hit = mh && dot(dir, dir) > 0.25f;
      dist = margin - depth; nrm = normalized(dir); pos = pos + p1;
    }
    const v3 pn = qrot(ldq(A.quat), mk3(0, 0, 1));
    MprGeom Bs = B;   // plane - convex: the deepest point of the convex geom (quads on other pairs scan nothing)
    if (!(active && plane && !special)) { Bs.type = RG_GEOM_SPHERE; Bs.mesh = -1; Bs.size = mk3(0, 0, 0); }
    const v3 sp = rg_support<4>(E, Bs, (active && plane && !special) ? pn * -1.0f : mk3(0, 0, 1));
    if (active && plane && !special) {
      dist = dot(sp, pn);
      hit = dist <= margin;
      pos = sp + p1 - pn * (0.5f * dist); nrm = pn;
    }
    const int slot = rb_slot(s, hit && (TID & 3) == 0, &s.ncon, m.maxcon, RG_STATUS_CON_FULL);
    if (slot >= 0) {
      float* c = con + RB_CONREC * slot;
      float pr[12]; rb_pair_prm(m, S, p, pr);
      c[RB_CR_DIST] = dist; st3(c + RB_CR_POS, pos);
      float fr[9]; fr[0] = nrm.x; fr[1] = nrm.y; fr[2] = nrm.z; rb_make_frame(fr);
      for (int k = 0; k < 9; k++) c[RB_CR_FRAME + k] = fr[k];
      c[RB_CR_INCL] = pr[0] - pr[1];
      c[RB_CR_FRIC] = pr[2]; c[RB_CR_FRIC + 1] = pr[2]; c[RB_CR_FRIC + 2] = pr[3]; c[RB_CR_FRIC + 3] = pr[4]; c[RB_CR_FRIC + 4] = pr[4];
      c[RB_CR_SOLREF] = pr[5]; c[RB_CR_SOLREF + 1] = pr[6];
      for (int k = 0; k < 5; k++) c[RB_CR_SOLIMP + k] = pr[7 + k];
      c[RB_CR_DIM] = (float)m.b_pair_geom[3 * p + 2]; c[RB_CR_G1] = (float)m.b_pair_geom[3 * p]; c[RB_CR_G2] = (float)m.b_pair_geom[3 * p + 1];
      c[RB_CR_ADR] = -1.f; c[RB_CR_NNZ] = 0.f; c[RB_CR_KIND] = (float)((m.cone == 1 && m.b_pair_geom[3 * p + 2] > 1) ? RB_KIND_ELLIPTIC : RB_KIND_PYRAMID);
    }
  }
  BSYNC();
}
__device__ __forceinline__ void rb_narrow_box(RbM m, RbLds& s, float* S, int flags) {
  const int* cand = (const int*)SC(CAND);
  const float *gpos = SC(GPOS), *gquat = SC(GQUAT);
  float* con = SC(CON);
  const bool multipoint = !(flags & 16);
  // box - box (mjc_BoxBox, up to 8 contacts) and plane - box (up to 4 corners): 32 lanes per pair, 8 pairs per trip
  if (multipoint) {
    const int ncand2 = s.ncand2;
    for (int base = 0; base < ncand2; base += RB_T / 32) {
      const int ci = base + (TID >> 5), l = TID & 31;
      bool hit = false, planebox = false; float dist = 0; v3 pos = mk3(0, 0, 0), nrm = mk3(0, 0, 1); int p = 0;
      if (ci < ncand2) {
        p = cand[m.maxcand - 1 - ci];
        const int g1 = m.b_pair_geom[3 * p], g2 = m.b_pair_geom[3 * p + 1], t1 = m.geom_type[g1];
        if (m.geom_type[g2] == RG_GEOM_BOX && (t1 == RG_GEOM_BOX || t1 == RG_GEOM_PLANE)) {
          const float margin = rb_pair_margin(m, S, p);
          const v3 P1 = ld3(gpos + 3 * g1), t = ld3(gpos + 3 * g2) - P1;
          if (t1 == RG_GEOM_BOX) {
            const float A[3] = {m.geom_size[3 * g1], m.geom_size[3 * g1 + 1], m.geom_size[3 * g1 + 2]}, B[3] = {m.geom_size[3 * g2], m.geom_size[3 * g2 + 1], m.geom_size[3 * g2 + 2]};
            hit = box_box_lane(A, B, ldq(gquat + 4 * g1), ldq(gquat + 4 * g2), P1, t, margin, l, dist, pos, nrm);
          } else {
            nrm = qrot(ldq(gquat + 4 * g1), mk3(0, 0, 1));
            hit = rb_plane_box_lane(nrm, t, ldq(gquat + 4 * g2), ld3(m.geom_size + 3 * g2), margin, l, dist, pos);
            pos = pos + P1; planebox = true;
          }
        }
      }
      // plane - box: the first four corners (in corner order) within the margin, as the reference routine's early exit does
      const unsigned long long balpb = __ballot(hit && planebox);
      if (hit && planebox && __popc((unsigned)((balpb >> (TID & 32)) & 0xffull) & ((1u << l) - 1u)) >= 4) hit = false;
      const int slot = rb_slot(s, hit, &s.ncon, m.maxcon, RG_STATUS_CON_FULL);
      if (slot >= 0) {
        float* c = con + RB_CONREC * slot;
        float pr[12]; rb_pair_prm(m, S, p, pr);
        c[RB_CR_DIST] = dist; st3(c + RB_CR_POS, pos);
        const v3 nn = normalized(nrm);
        float fr[9]; fr[0] = nn.x; fr[1] = nn.y; fr[2] = nn.z; rb_make_frame(fr);
        for (int k = 0; k < 9; k++) c[RB_CR_FRAME + k] = fr[k];
        c[RB_CR_INCL] = pr[0] - pr[1];
        c[RB_CR_FRIC] = pr[2]; c[RB_CR_FRIC + 1] = pr[2]; c[RB_CR_FRIC + 2] = pr[3]; c[RB_CR_FRIC + 3] = pr[4]; c[RB_CR_FRIC + 4] = pr[4];
        c[RB_CR_SOLREF] = pr[5]; c[RB_CR_SOLREF + 1] = pr[6];
        for (int k = 0; k < 5; k++) c[RB_CR_SOLIMP + k] = pr[7 + k];
        c[RB_CR_DIM] = (float)m.b_pair_geom[3 * p + 2]; c[RB_CR_G1] = (float)m.b_pair_geom[3 * p]; c[RB_CR_G2] = (float)m.b_pair_geom[3 * p + 1];
        c[RB_CR_ADR] = -1.f; c[RB_CR_NNZ] = 0.f; c[RB_CR_KIND] = (float)((m.cone == 1 && m.b_pair_geom[3 * p + 2] > 1) ? RB_KIND_ELLIPTIC : RB_KIND_PYRAMID);
      }
    }
  }
  BSYNC();
}

// ------------------------------------------------------------------------------------------------- constraints
__device__ __forceinline__ float rb_impedance(const float* si, float pos, float margin) { return impedance(si, pos, margin); }
__device__ __forceinline__ void rb_KB(float timestep, const float* solref, const float* solimp, float& K, float& B) {
  const float dmax = clampf(solimp[1], 1e-4f, 0.9999f);
  if (solref[0] > 0) {
    const float tc = fmaxf(solref[0], 2.f * timestep), dr = solref[1];
    K = 1.f / fmaxf(RB_MINVAL, dmax * dmax * tc * tc * dr * dr); B = 2.f / fmaxf(RB_MINVAL, dmax * tc);
  } else { K = -solref[0] / fmaxf(RB_MINVAL, dmax * dmax); B = -solref[1] / fmaxf(RB_MINVAL, dmax); }
}
__device__ __forceinline__ int rb_npyr(int dim) { return dim == 1 ? 1 : 2 * (dim - 1); }
__device__ __forceinline__ int rb_nrows(int kind, int dim) { return kind == RB_KIND_PYRAMID ? rb_npyr(dim) : dim; }   // elliptic contacts and equalities: one row per dimension
// J_row . x for a static row
__device__ __forceinline__ float rb_srow_dot(RbM m, const float* S, int type, int id, float aux, const float* x) {
  if (type == 0) return x[id];
  if (type == 2) return aux * x[m.jnt_dofadr[id]];
  float v = 0;
  for (int e = 0; e < RB_TENW; e++) { const int d = m.b_ten_dofs[RB_TENW * id + e]; if (d >= 0) v += SC(TENJ)[RB_TENW * id + e] * x[d]; }
  return type == 1 ? v : aux * v;
}
// mj_makeConstraint + mj_makeImpedance: rows in MuJoCo's order (friction dofs, friction tendons, joint limits, tendon limits,
// contacts); per contact the six basis Jacobian rows (3 translational, 3 rotational, contact frame) on the union of the dof chains
__device__ __forceinline__ void rb_make_constraint(RbM m, RbLds& s, float* S, const float* L_eq_data) {
  float* row = SC(ROW); float* con = SC(CON);
  const float *biw = PRM(body_invweight0, RB_P_BODY_INVWEIGHT0), *diw = PRM(dof_invweight0, RB_P_DOF_INVWEIGHT0), *tiw = PRM(tendon_invweight0, RB_P_TENDON_INVWEIGHT0);
  const int nf = m.nfric_dof + m.nfric_ten;
  // friction-loss rows
  BFOR(r, nf) {
    float* R = row + RB_ROWREC * r;
    const bool ten = r >= m.nfric_dof;
    const int id = ten ? m.b_fric_ten[r - m.nfric_dof] : m.b_fric_dof[r];
    const float* solref = ten ? m.tendon_solref_fri + 2 * id : m.dof_solref + 2 * id;
    const float* solimp = ten ? m.tendon_solimp_fri + 5 * id : m.dof_solimp + 5 * id;
    const float diag = ten ? tiw[id] : diw[id], floss = ten ? m.tendon_frictionloss[id] : PRM(dof_frictionloss, RB_P_DOF_FRICTIONLOSS)[id];
    const float imp = rb_impedance(solimp, 0.f, 0.f);
    const float Rr = fmaxf(RB_MINVAL, (1.f - imp) * diag / imp);
    float K, B; rb_KB(m.timestep, solref, solimp, K, B);
    R[RB_RR_TYPE] = ten ? 1.f : 0.f; R[RB_RR_ID] = (float)id; R[RB_RR_AUX] = 1.f; R[RB_RR_FLOSS] = floss; R[RB_RR_D] = 1.f / Rr;
    R[RB_RR_AREF] = -B * rb_srow_dot(m, S, ten ? 1 : 0, id, 1.f, s.qvel);
  }
  // limits: joints then tendons, compacted in order.  One wave per env (round 5): a lane per (joint or tendon, side) candidate, the active ones numbered by a ballot
  // prefix -- the same rows in the same order as the serial loop below, without its chain of dependent loads per candidate
  int nlim_par = -1;
  if (WID == 0) {      // (several waves: wave 0 does this while the others wait at the barrier below, as they did for the serial walk)
    int n = nf;
    const int ncand = 2 * (m.nlim_jnt + m.nlim_ten);
    for (int k0 = 0; k0 < ncand; k0 += 64) {
      const int k = k0 + WL, q = k >> 1, side = (k & 1) ? 1 : -1;
      bool act = false; float dist = 0.f, margin = 0.f; int id = 0; bool ten = false;
      if (k < ncand) {
        ten = q >= m.nlim_jnt;
        id = ten ? m.b_lim_ten[q - m.nlim_jnt] : m.b_lim_jnt[q];
        const float value = ten ? SC(TENLEN)[id] : s.qpos[m.jnt_qposadr[id]];
        const float* range = ten ? PRM(tendon_range, RB_P_TENDON_RANGE) + 2 * id : PRM(jnt_range, RB_P_JNT_RANGE) + 2 * id;
        margin = ten ? m.tendon_margin[id] : PRM(jnt_margin, RB_P_JNT_MARGIN)[id];
        dist = side * (range[(side + 1) / 2] - value);
        act = dist < margin;
      }
      const unsigned long long bal = __ballot(act);
      const int slot = n + __popcll(bal & ((1ull << WL) - 1ull));
      if (act && slot < m.maxrow) {
        float* R = row + RB_ROWREC * slot;
        const float* solref = ten ? m.tendon_solref_lim + 2 * id : m.jnt_solref + 2 * id;
        const float* solimp = ten ? m.tendon_solimp_lim + 5 * id : m.jnt_solimp + 5 * id;
        const float diag = ten ? tiw[id] : diw[m.jnt_dofadr[id]];
        const float imp = rb_impedance(solimp, dist, margin);
        const float Rr = fmaxf(RB_MINVAL, (1.f - imp) * diag / imp);
        float K, B; rb_KB(m.timestep, solref, solimp, K, B);
        R[RB_RR_TYPE] = ten ? 3.f : 2.f; R[RB_RR_ID] = (float)id; R[RB_RR_AUX] = (float)(-side); R[RB_RR_FLOSS] = 0.f; R[RB_RR_D] = 1.f / Rr;
        R[RB_RR_AREF] = -B * rb_srow_dot(m, S, ten ? 3 : 2, id, (float)(-side), s.qvel) - K * imp * (dist - margin);
      }
      n += __popcll(bal);
      if (n > m.maxrow) n = m.maxrow;
    }
    nlim_par = n - nf;
  }
  if (TID == 0) {
    int n = nf;
    for (int q = 0; q < (nlim_par >= 0 ? 0 : m.nlim_jnt + m.nlim_ten); q++) {
      const bool ten = q >= m.nlim_jnt;
      const int id = ten ? m.b_lim_ten[q - m.nlim_jnt] : m.b_lim_jnt[q];
      const float value = ten ? SC(TENLEN)[id] : s.qpos[m.jnt_qposadr[id]];
      const float* range = ten ? PRM(tendon_range, RB_P_TENDON_RANGE) + 2 * id : PRM(jnt_range, RB_P_JNT_RANGE) + 2 * id;
      const float margin = ten ? m.tendon_margin[id] : PRM(jnt_margin, RB_P_JNT_MARGIN)[id];
      for (int side = -1; side <= 1; side += 2) {
        const float dist = side * (range[(side + 1) / 2] - value);
        if (dist < margin && n < m.maxrow) {
          float* R = row + RB_ROWREC * n;
          const float* solref = ten ? m.tendon_solref_lim + 2 * id : m.jnt_solref + 2 * id;
          const float* solimp = ten ? m.tendon_solimp_lim + 5 * id : m.jnt_solimp + 5 * id;
          const float diag = ten ? tiw[id] : diw[m.jnt_dofadr[id]];
          const float imp = rb_impedance(solimp, dist, margin);
          const float Rr = fmaxf(RB_MINVAL, (1.f - imp) * diag / imp);
          float K, B; rb_KB(m.timestep, solref, solimp, K, B);
          R[RB_RR_TYPE] = ten ? 3.f : 2.f; R[RB_RR_ID] = (float)id; R[RB_RR_AUX] = (float)(-side); R[RB_RR_FLOSS] = 0.f; R[RB_RR_D] = 1.f / Rr;
          R[RB_RR_AREF] = -B * rb_srow_dot(m, S, ten ? 3 : 2, id, (float)(-side), s.qvel) - K * imp * (dist - margin);
          n++;
        }
      }
    }
    if (nlim_par >= 0) n = nf + nlim_par;
    s.nlim = n - nf;
    s.nefc = n;      // (the contacts' rows follow: below)
  }
  BSYNC();
  // contacts: row addresses = running sum of the contacts' row counts.  One wave: a lane per contact and a wave prefix sum when everything fits (the usual case);
  // else -- and with several waves -- the serial walk with its skip-on-overflow rule
  {
    int n = s.nefc;
    bool done = false;
    if (s.ncon <= 64) {      // (several waves: each computes the same prefix from its own 64 lanes, wave 0 writes)
      const int np = WL < s.ncon ? rb_nrows((int)con[RB_CONREC * WL + RB_CR_KIND], (int)con[RB_CONREC * WL + RB_CR_DIM]) : 0;
      int incl = np;
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl(incl, WL >= o ? WL - o : WL); if (WL >= o) incl += t; }
      const int total = __shfl(incl, 63);
      if (n + total <= m.maxrow) {
        if (WID == 0 && WL < s.ncon) con[RB_CONREC * WL + RB_CR_ADR] = (float)(n + incl - np);
        BSYNC();
        if (TID == 0) s.nefc = n + total;
        done = true;
      }
    }
    if (!done) {
      BSYNC();
      if (TID == 0) {
        for (int c = 0; c < s.ncon; c++) {
          const int np = rb_nrows((int)con[RB_CONREC * c + RB_CR_KIND], (int)con[RB_CONREC * c + RB_CR_DIM]);
          if (n + np <= m.maxrow) { con[RB_CONREC * c + RB_CR_ADR] = (float)n; n += np; }
          else { con[RB_CONREC * c + RB_CR_ADR] = -1.f; s.status |= RG_STATUS_ROW_FULL; }
        }
        s.nefc = n;
      }
    }
  }
  BSYNC();
  // contact Jacobians: one thread per contact builds the dof list and the six basis rows
  float* cj = SC(CONJ); int* cidx = (int*)SC(CONIDX);
  const float *cdof = SC(CDOF), *rootcom = SC(ROOTCOM);
  BFOR(c, s.ncon) {
    float* C = con + RB_CONREC * c;
    int* idx = cidx + RB_CONW * c; float* J = cj + 6 * RB_CONW * c;
    if ((int)C[RB_CR_KIND] == RB_KIND_EQUALITY) {
      // equality rows: the basis Jacobian rows ARE the constraint rows (oracle ro_make_constraint, equality block)
      const int e = (int)C[RB_CR_G1], dim = (int)C[RB_CR_DIM], o1 = m.eq_obj1id[e], o2 = m.eq_obj2id[e];
      int nnz = 0;
      if (dim == 1) {
        idx[0] = m.jnt_dofadr[o1]; J[0] = 1.f; nnz = 1;
        if (o2 >= 0) { idx[1] = m.jnt_dofadr[o2]; J[1] = -C[RB_CR_FRAME + 1]; nnz = 2; }
      } else {
        const q4 q1 = ldq(SC(XQUAT) + 4 * o1), q2 = ldq(SC(XQUAT) + 4 * o2);
        q4 qc; qc.w = q2.w; qc.x = -q2.x; qc.y = -q2.y; qc.z = -q2.z;
        const q4 qr = qmul(q1, ldq(L_eq_data + 7 * e + 3));
#ifndef RB_ROWS_LEGACY
        if (RB_NWAVE == 1) {
          // (as for the contacts below: the dof list in LDS first, then every column formed in registers -- side 0 = body 1, then body 2 -- and stored once)
          int* li = (int*)s.A + RB_CONW * TID;
          for (int side = 0; side < 2; side++) {
            const int bb = side ? o2 : o1;
            if (bb <= 0 || m.body_weldid[bb] == 0) continue;
            for (int i = m.b_body_lastdof[bb]; i >= 0; i = m.dof_parentid[i]) {
              int q = 0;
              while (q < nnz && (li[q] & 0xffff) != i) q++;
              if (q == nnz) { if (nnz >= RB_CONW) continue; li[nnz++] = i | (1 << (16 + side)); }
              else li[q] |= 1 << (16 + side);
            }
          }
          const v3 offA = ld3(C + RB_CR_POS) - ld3(rootcom + 3 * m.body_rootid[o1 > 0 ? o1 : 0]), offB = ld3(SC(XPOS) + 3 * o2) - ld3(rootcom + 3 * m.body_rootid[o2 > 0 ? o2 : 0]);
          for (int q = 0; q < nnz; q++) {
            const int i = li[q] & 0xffff, sides = li[q] >> 16;
            const v3 jr = ld3(cdof + 6 * i);
            float col[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int side = 0; side < 2; side++) {
              if (!((sides >> side) & 1)) continue;
              const float sg = side ? -1.f : 1.f;
              const v3 jp = rb_jacp(cdof, i, side ? offB : offA);
              col[0] += sg * jp.x; col[1] += sg * jp.y; col[2] += sg * jp.z;
              q4 ax; ax.w = 0.f; ax.x = sg * jr.x; ax.y = sg * jr.y; ax.z = sg * jr.z;
              const q4 t2 = qmul(qmul(qc, ax), qr);
              col[3] += 0.5f * t2.x; col[4] += 0.5f * t2.y; col[5] += 0.5f * t2.z;
            }
            idx[q] = i;
#pragma unroll
            for (int r = 0; r < 6; r++) J[r * RB_CONW + q] = col[r];
          }
        } else
#endif
        for (int side = 0; side < 2; side++) {
          const int bb = side ? o2 : o1; const float sg = side ? -1.f : 1.f;   // body1 - body2 ("opposite of contact")
          if (bb <= 0 || m.body_weldid[bb] == 0) continue;
          const v3 pnt = side ? ld3(SC(XPOS) + 3 * o2) : ld3(C + RB_CR_POS);
          const v3 off = pnt - ld3(rootcom + 3 * m.body_rootid[bb]);
          for (int i = m.b_body_lastdof[bb]; i >= 0; i = m.dof_parentid[i]) {
            int q = 0;
            while (q < nnz && idx[q] != i) q++;
            if (q == nnz) { if (nnz >= RB_CONW) continue; idx[nnz] = i; for (int r = 0; r < 6; r++) J[r * RB_CONW + nnz] = 0.f; nnz++; }
            const v3 jp = rb_jacp(cdof, i, off), jr = ld3(cdof + 6 * i);
            J[q] += sg * jp.x; J[RB_CONW + q] += sg * jp.y; J[2 * RB_CONW + q] += sg * jp.z;
            q4 ax; ax.w = 0.f; ax.x = sg * jr.x; ax.y = sg * jr.y; ax.z = sg * jr.z;
            const q4 t2 = qmul(qmul(qc, ax), qr);                                  // 0.5 conj(q2) (jac1 - jac2) q1 relquat
            J[3 * RB_CONW + q] += 0.5f * t2.x; J[4 * RB_CONW + q] += 0.5f * t2.y; J[5 * RB_CONW + q] += 0.5f * t2.z;
          }
        }
      }
      C[RB_CR_NNZ] = (float)nnz;
      for (int q = nnz; q < RB_CONW; q++) idx[q] = -1;
      const int adr = (int)C[RB_CR_ADR];
      if (adr < 0) continue;
      float K, B; rb_KB(m.timestep, C + RB_CR_SOLREF, C + RB_CR_SOLIMP, K, B);
      for (int k = 0; k < dim; k++) {
        const float pos = C[RB_CR_FRAME + (dim == 1 ? 0 : k)];
        const float imp = rb_impedance(C + RB_CR_SOLIMP, pos, 0.f);
        float vel = 0.f; for (int q = 0; q < nnz; q++) vel += J[k * RB_CONW + q] * s.qvel[idx[q]];
        float* R = row + RB_ROWREC * (adr + k);
        R[RB_RR_TYPE] = 5.f; R[RB_RR_ID] = (float)c; R[RB_RR_AUX] = (float)k; R[RB_RR_FLOSS] = 0.f; R[RB_RR_ZONE] = 3.f;
        R[RB_RR_D] = 1.f / fmaxf(RB_MINVAL, (1.f - imp) * C[RB_CR_FRIC + (k < 3 ? 0 : 1)] / imp);
        R[RB_RR_AREF] = -B * vel - K * imp * pos;
      }
      continue;
    }
    const int b1 = m.geom_bodyid[(int)C[RB_CR_G1]], b2 = m.geom_bodyid[(int)C[RB_CR_G2]];
    const v3 pos = ld3(C + RB_CR_POS);
    int nnz = 0;
    float bd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool bd_done = false;
#ifndef RB_ROWS_LEGACY
    if (RB_NWAVE == 1) {
      // One wave per env (round 5): the dof list is built in LDS first (the block's storage is free until the smooth stage: RB_CONW words per lane, dof | sides << 16),
      // then every column of the six basis rows is formed in registers and stored ONCE -- instead of searching the list in the scratch row and six
      // read-modify-write round trips per dof and side.  Same sums in the same order (side 0 = body 2 first), so the rows are unchanged.
      static_assert(sizeof(s.A) + sizeof(s.Dinv) >= sizeof(int) * RB_CONW * RB_T || RB_NWAVE != 1, "rb_make_constraint: the dof lists do not fit the block's storage");
      int* li = (int*)s.A + RB_CONW * TID;
      for (int side = 0; side < 2; side++) {
        const int bb = side ? b1 : b2;
        for (int i = (bb > 0 ? m.b_body_lastdof[bb] : -1); i >= 0; i = m.dof_parentid[i]) {
          int e = 0;
          while (e < nnz && (li[e] & 0xffff) != i) e++;
          if (e == nnz) { if (nnz >= RB_CONW) continue; li[nnz++] = i | (1 << (16 + side)); }
          else li[e] |= 1 << (16 + side);
        }
      }
      const v3 fr0 = ld3(C + RB_CR_FRAME), fr1 = ld3(C + RB_CR_FRAME + 3), fr2 = ld3(C + RB_CR_FRAME + 6);
      const v3 off2 = pos - ld3(rootcom + 3 * m.body_rootid[b2]), off1 = pos - ld3(rootcom + 3 * m.body_rootid[b1]);
      for (int e = 0; e < nnz; e++) {
        const int i = li[e] & 0xffff, sides = li[e] >> 16;
        const v3 jr = ld3(cdof + 6 * i);
        float col[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (sides & 1) { const v3 jp = rb_jacp(cdof, i, off2); col[0] += dot(fr0, jp); col[3] += dot(fr0, jr); col[1] += dot(fr1, jp); col[4] += dot(fr1, jr); col[2] += dot(fr2, jp); col[5] += dot(fr2, jr); }
        if (sides & 2) { const v3 jp = rb_jacp(cdof, i, off1); col[0] += -1.f * dot(fr0, jp); col[3] += -1.f * dot(fr0, jr); col[1] += -1.f * dot(fr1, jp); col[4] += -1.f * dot(fr1, jr); col[2] += -1.f * dot(fr2, jp); col[5] += -1.f * dot(fr2, jr); }
        const float qv = s.qvel[i];
        idx[e] = i;
#pragma unroll
        for (int r = 0; r < 6; r++) { J[r * RB_CONW + e] = col[r]; bd[r] += col[r] * qv; }
      }
      bd_done = true;
    } else
#endif
    for (int side = 0; side < 2; side++) {
      const int bb = side ? b1 : b2; const float sg = side ? -1.f : 1.f;   // difference body2 - body1
      const v3 off = pos - ld3(rootcom + 3 * m.body_rootid[bb]);
      for (int i = (bb > 0 ? m.b_body_lastdof[bb] : -1); i >= 0; i = m.dof_parentid[i]) {
        int e = 0;
        while (e < nnz && idx[e] != i) e++;
        if (e == nnz) { if (nnz >= RB_CONW) continue; idx[nnz] = i; for (int r = 0; r < 6; r++) J[r * RB_CONW + nnz] = 0.f; nnz++; }
        const v3 jp = rb_jacp(cdof, i, off), jr = ld3(cdof + 6 * i);
        for (int r = 0; r < 3; r++) {
          const v3 fr = ld3(C + RB_CR_FRAME + 3 * r);
          J[r * RB_CONW + e] += sg * dot(fr, jp);
          J[(3 + r) * RB_CONW + e] += sg * dot(fr, jr);
        }
      }
    }
    C[RB_CR_NNZ] = (float)nnz;
    for (int e = nnz; e < RB_CONW; e++) idx[e] = -1;
    // rows of this contact
    const int adr = (int)C[RB_CR_ADR];
    if (adr < 0) continue;
    const int dim = (int)C[RB_CR_DIM], np = rb_npyr(dim);
    const float tran = biw[2 * b1] + biw[2 * b2], rot = biw[2 * b1 + 1] + biw[2 * b2 + 1];
    const float dist = C[RB_CR_DIST], incl = C[RB_CR_INCL];
    const float imp = rb_impedance(C + RB_CR_SOLIMP, dist, incl);
    float K, B; rb_KB(m.timestep, C + RB_CR_SOLREF, C + RB_CR_SOLIMP, K, B);
    if (!bd_done) for (int r = 0; r < 6; r++) { float v = 0; for (int e = 0; e < nnz; e++) v += J[r * RB_CONW + e] * s.qvel[idx[e]]; bd[r] = v; }
    if (dim == 1) {
      float* R = row + RB_ROWREC * adr;
      R[RB_RR_TYPE] = 4.f; R[RB_RR_ID] = (float)c; R[RB_RR_AUX] = 0.f; R[RB_RR_FLOSS] = 0.f;
      R[RB_RR_D] = 1.f / fmaxf(RB_MINVAL, (1.f - imp) * tran / imp);
      R[RB_RR_AREF] = -B * bd[0] - K * imp * (dist - incl);
    } else if ((int)C[RB_CR_KIND] == RB_KIND_ELLIPTIC) {
      // mj_makeImpedance, elliptic: R_1 = R_0 / impratio, mu = friction_0 sqrt(R_1 / R_0), R_j mu_j^2 = R_1 mu_1^2; only the normal row
      // carries the distance.  mu is kept in the record's solref slot (its last reader was rb_KB above)
      const float R0 = fmaxf(RB_MINVAL, (1.f - imp) * tran / imp), R1 = R0 / fmaxf(RB_MINVAL, m.impratio), fri0 = C[RB_CR_FRIC];
      C[RB_CR_SOLREF] = fri0 * sqrtf(R1 / R0);
      for (int k = 0; k < dim; k++) {
        float* R = row + RB_ROWREC * (adr + k);
        const float fk = k ? C[RB_CR_FRIC + k - 1] : 1.f;
        const float Rk = k == 0 ? R0 : (k == 1 ? R1 : R1 * fri0 * fri0 / (fk * fk));
        R[RB_RR_TYPE] = 5.f; R[RB_RR_ID] = (float)c; R[RB_RR_AUX] = (float)k; R[RB_RR_FLOSS] = 0.f; R[RB_RR_ZONE] = 0.f;
        R[RB_RR_D] = 1.f / Rk;
        R[RB_RR_AREF] = -B * bd[k] - (k == 0 ? K * imp * (dist - incl) : 0.f);
      }
    } else {
      // all pyramid rows share R = 2 mu^2 R_first, R_first from the first edge's diagApprox
      const float fri0 = C[RB_CR_FRIC], diag0 = tran + fri0 * fri0 * tran;
      const float Rfirst = fmaxf(RB_MINVAL, (1.f - imp) * diag0 / imp);
      const float mu = fri0 * sqrtf(1.f / m.impratio);
      const float Rpy = 2.f * mu * mu * Rfirst;
      for (int q = 0; q < np; q++) {
        const int k = q >> 1; const float sgn = (q & 1) ? -1.f : 1.f, fri = C[RB_CR_FRIC + k];
        float* R = row + RB_ROWREC * (adr + q);
        R[RB_RR_TYPE] = 4.f; R[RB_RR_ID] = (float)c; R[RB_RR_AUX] = sgn * (float)(k + 1); R[RB_RR_FLOSS] = 0.f;
        R[RB_RR_D] = 1.f / Rpy;
        R[RB_RR_AREF] = -B * (bd[0] + sgn * fri * bd[k + 1]) - K * imp * (dist - incl);
      }
    }
  }
  BSYNC();
}

// every dof's contact entries (contact c, slot e with idx[c][e] == dof), in contact order: the owner-computes form of J' f
__device__ __forceinline__ void rb_dof_contact_lists(RbM m, RbLds& s, float* S) {
  const int* cidx = (const int*)SC(CONIDX); int* adr = (int*)SC(DOFCON_ADR); int* lst = (int*)SC(DOFCON);
  // every contact dof's row in its group's dense block, once per mj_step (the Hessian assembly stages it with the Jacobian rows: one load, not a
  // dependent pair of them per contact and Newton iteration)
  { float* loc = SC(CONLOC); BFOR(w, RB_CONW * s.ncon) { const int d = cidx[w]; loc[w] = (float)((d >= 0 && d < m.nv) ? m.b_dof_local[d] : 0); } }
  const float* con = SC(CON);
  const int nvw = (m.nv + 31) >> 5, estride = (m.nv + 3) & ~3;
  // LDS that is free between the solver calls: the dense block and everything up to the contact staging area
  unsigned char* area = (unsigned char*)s.A;
  const int cap = (int)((unsigned char*)(s.cst + 2 * RB_CST) - area);
  if (s.ncon * (4 * nvw + estride) <= cap) {
    // Per contact (one thread each): a bit mask of its dofs and a byte map dof -> entry, both in LDS, from ONE pass over its dof list (16-byte loads).
    // Per dof: count, prefix, fill -- LDS reads only; the list keeps contact order, as the scan of the dof lists below produced it.
    unsigned* mask = (unsigned*)area; unsigned char* emap = area + 4 * nvw * s.ncon;
    BFOR(w, (nvw * s.ncon)) mask[w] = 0u;
    BSYNC();
    BFOR(c, s.ncon) {
      if (con[RB_CONREC * c + RB_CR_ADR] < 0) continue;
      const int nnz = (int)con[RB_CONREC * c + RB_CR_NNZ];
      const rgf4* I4 = (const rgf4*)(cidx + RB_CONW * c);
      rgf4 iv[RB_CONW / 4];
#pragma unroll
      for (int q = 0; q < RB_CONW / 4; q++) iv[q] = I4[q];
#pragma unroll
      for (int q = 0; q < RB_CONW / 4; q++) {
        const int d4[4] = {__builtin_bit_cast(int, iv[q].x), __builtin_bit_cast(int, iv[q].y), __builtin_bit_cast(int, iv[q].z), __builtin_bit_cast(int, iv[q].w)};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e = 4 * q + k, d = d4[k];
          if (e < nnz && d >= 0 && d < m.nv) { mask[nvw * c + (d >> 5)] |= 1u << (d & 31); emap[estride * c + d] = (unsigned char)e; }
        }
      }
    }
    BSYNC();
    int* cnt = (int*)s.x;
    BFOR(i, m.nv) {
      int n = 0;
      for (int c = 0; c < s.ncon; c++) n += (mask[nvw * c + (i >> 5)] >> (i & 31)) & 1u;
      cnt[i] = n;
    }
    BSYNC();
    BFOR(i, m.nv) {
      int a0 = 0;
      for (int k = 0; k < i; k++) a0 += cnt[k];
      adr[i] = a0;
      if (i == m.nv - 1) adr[m.nv] = a0 + cnt[i];
      int n = a0;
      for (int c = 0; c < s.ncon; c++) if ((mask[nvw * c + (i >> 5)] >> (i & 31)) & 1u) lst[n++] = c * RB_CONW + (int)emap[estride * c + i];
    }
    BSYNC();
    return;
  }
  // (more contacts than the free LDS holds: the same lists from masks in the scratch row and a scan of the dof lists)
  // a bit mask of its dofs per contact first (6 words: nv <= 192), so that a dof looks at one word per contact instead of its whole dof list
  unsigned* mask = (unsigned*)SC(CONF);   // (the per-contact solver scratch is free until the solver starts)
  BFOR(c, s.ncon) {
    unsigned w[6] = {0, 0, 0, 0, 0, 0};
    if (!(con[RB_CONREC * c + RB_CR_ADR] < 0)) {
      const int nnz = (int)con[RB_CONREC * c + RB_CR_NNZ];
      for (int e = 0; e < nnz; e++) { const int d = cidx[RB_CONW * c + e]; if (d >= 0 && d < RB_MAXNV) w[d >> 5] |= 1u << (d & 31); }
    }
    for (int k = 0; k < 6; k++) mask[6 * c + k] = w[k];
  }
  BSYNC();
  int* cnt = (int*)s.x;
  BFOR(i, m.nv) {   // counts
    int n = 0;
    for (int c = 0; c < s.ncon; c++) n += (mask[6 * c + (i >> 5)] >> (i & 31)) & 1u;
    cnt[i] = n;
  }
  BSYNC();
  BFOR(i, m.nv) {   // exclusive prefix (every dof sums the counts before it: LDS reads, no serial pass through the scratch row)
    int a0 = 0;
    for (int k = 0; k < i; k++) a0 += cnt[k];
    adr[i] = a0;
    if (i == m.nv - 1) adr[m.nv] = a0 + cnt[i];
    int n = a0;
    for (int c = 0; c < s.ncon; c++) {
      if (!((mask[6 * c + (i >> 5)] >> (i & 31)) & 1u)) continue;
      const int nnz = (int)con[RB_CONREC * c + RB_CR_NNZ];
      for (int e = 0; e < nnz; e++) if (cidx[RB_CONW * c + e] == i) { lst[n++] = c * RB_CONW + e; break; }
    }
  }
  BSYNC();
}
// Elliptic cones (engine_core_constraint.c mj_constraintUpdate, oracle cone_eval): one thread per contact evaluates its zone from the rows'
// residuals and leaves force / zone / cost in the row records (the whole contact's cost on its first row).  In the scaled variables
// U0 = mu jar_0, Uj = friction_(j-1) jar_j, N = U0, T = |U_1..|: top zone (N >= mu T) no force; bottom zone (mu N + T <= 0) every row
// quadratic with its own D; middle zone cost = 1/2 Dm (N - mu T)^2, Dm = D_0 / (mu^2 (1 + mu^2)).
__device__ __forceinline__ void rb_cone_update(RbM m, RbLds& s, float* S) {
  float* row = SC(ROW); const float* con = SC(CON);
  BFOR(c, s.ncon) {
    const float* C = con + RB_CONREC * c;
    const int adr = (int)C[RB_CR_ADR], dim = (int)C[RB_CR_DIM];
    if ((int)C[RB_CR_KIND] != RB_KIND_ELLIPTIC || adr < 0 || dim < 2) continue;
    float* R0 = row + RB_ROWREC * adr;
    const float mu = C[RB_CR_SOLREF];
    float U[6], T = 0.f;
    U[0] = R0[RB_RR_JAR] * mu;
    for (int j = 1; j < dim; j++) { U[j] = R0[RB_ROWREC * j + RB_RR_JAR] * C[RB_CR_FRIC + j - 1]; T += U[j] * U[j]; }
    const float N = U[0]; T = sqrtf(T);
    if (N >= mu * T || (T <= 0.f && N >= 0.f)) {
      for (int j = 0; j < dim; j++) { float* R = R0 + RB_ROWREC * j; R[RB_RR_FORCE] = 0.f; R[RB_RR_ZONE] = 0.f; R[RB_RR_COST] = 0.f; }
    } else if (mu * N + T <= 0.f || (T <= 0.f && N < 0.f)) {
      for (int j = 0; j < dim; j++) { float* R = R0 + RB_ROWREC * j; const float x = R[RB_RR_JAR], D = R[RB_RR_D]; R[RB_RR_FORCE] = -D * x; R[RB_RR_ZONE] = 1.f; R[RB_RR_COST] = 0.5f * D * x * x; }
    } else {
      const float Dm = R0[RB_RR_D] / (mu * mu * (1.f + mu * mu)), NT = N - mu * T;
      const float f0 = -Dm * NT * mu;
      R0[RB_RR_FORCE] = f0; R0[RB_RR_ZONE] = 2.f; R0[RB_RR_COST] = 0.5f * Dm * NT * NT;
      for (int j = 1; j < dim; j++) { float* R = R0 + RB_ROWREC * j; R[RB_RR_FORCE] = -f0 / T * U[j] * C[RB_CR_FRIC + j - 1]; R[RB_RR_ZONE] = 2.f; R[RB_RR_COST] = 0.f; }
    }
  }
  BSYNC();
}
// jar (to_jv = false: J x - aref) or jv (to_jv = true: J x) of every row
__device__ __forceinline__ void rb_J_mul(RbM m, RbLds& s, float* S, const float* x, bool to_jv) {
  float* row = SC(ROW); const float* con = SC(CON); const float* cj = SC(CONJ); const int* cidx = (const int*)SC(CONIDX);
  // the six basis products of every contact first (its pyramid rows combine them: up to ten rows share them)
  float* bd = SC(CONF);
  BFOR(w, 6 * s.ncon) {
    const int c = w / 6, k = w - 6 * c;
    const int nnz = (int)con[RB_CONREC * c + RB_CR_NNZ];
    // the whole padded row and its dof list in 16-byte loads issued together (a loop over nnz waits for every element's round trip in turn); the sum
    // itself runs over the first nnz entries in order, as before
    const rgf4* J4 = (const rgf4*)(cj + 6 * RB_CONW * c + k * RB_CONW); const rgf4* I4 = (const rgf4*)(cidx + RB_CONW * c);
    rgf4 jv[RB_CONW / 4], iv[RB_CONW / 4];
#pragma unroll
    for (int q = 0; q < RB_CONW / 4; q++) { jv[q] = J4[q]; iv[q] = I4[q]; }
    float v = 0;
#pragma unroll
    for (int q = 0; q < RB_CONW / 4; q++) {
      if (4 * q + 0 < nnz) v += jv[q].x * x[__builtin_bit_cast(int, iv[q].x)];
      if (4 * q + 1 < nnz) v += jv[q].y * x[__builtin_bit_cast(int, iv[q].y)];
      if (4 * q + 2 < nnz) v += jv[q].z * x[__builtin_bit_cast(int, iv[q].z)];
      if (4 * q + 3 < nnz) v += jv[q].w * x[__builtin_bit_cast(int, iv[q].w)];
    }
    bd[w] = v;
  }
  BSYNC();
  BFOR(r, s.nefc) {
    float* R = row + RB_ROWREC * r;
    const int type = (int)R[RB_RR_TYPE], id = (int)R[RB_RR_ID];
    const float aux = R[RB_RR_AUX];
    float v;
    if (type < 4) v = rb_srow_dot(m, S, type, id, aux, x);
    else if (type == 5) v = bd[6 * id + (int)aux];
    else {
      const int a = (int)fabsf(aux);
      v = bd[6 * id] + (a ? (aux < 0 ? -1.f : 1.f) * con[RB_CONREC * id + RB_CR_FRIC + a - 1] * bd[6 * id + a] : 0.f);
    }
    if (to_jv) R[RB_RR_JV] = v; else R[RB_RR_JAR] = v - R[RB_RR_AREF];
  }
  BSYNC();
  if (!to_jv && m.cone == 1) rb_cone_update(m, s, S);
}
// force of one constraint row at its current residual (R[JAR]), whether it is in its quadratic zone, and its cost
__device__ __forceinline__ float rb_row_force(const float* R, bool& quad, float& cost) {
  const float D = R[RB_RR_D], x = R[RB_RR_JAR];
  if ((int)R[RB_RR_TYPE] == 5) {
    if (R[RB_RR_ZONE] == 3.f) { quad = true; cost = 0.5f * D * x * x; return -D * x; }   // equality row: always active, two-sided
    quad = R[RB_RR_ZONE] == 1.f; cost = R[RB_RR_COST]; return R[RB_RR_FORCE];              // elliptic contact row: what rb_cone_update left
  }
  if ((int)R[RB_RR_TYPE] < 2) {
    const float f = R[RB_RR_FLOSS], Rr = 1.f / D;
    if (x <= -Rr * f) { quad = false; cost = f * (-0.5f * Rr * f - x); return f; }
    if (x >= Rr * f) { quad = false; cost = f * (-0.5f * Rr * f + x); return -f; }
    quad = true; cost = 0.5f * D * x * x; return -D * x;
  }
  if (x >= 0) { quad = false; cost = 0.f; return 0.f; }
  quad = true; cost = 0.5f * D * x * x; return -D * x;
}
// dst[i] = sum over rows of J[r][i] * force[r]: the contacts' basis forces first (one (contact, basis row) per thread), then every
// dof collects its own friction row, the few limit rows and its contact entries (owner computes, fixed order)
__device__ __forceinline__ void rb_JT_force(RbM m, RbLds& s, float* S, float* dst) {
  const float* row = SC(ROW); const float* con = SC(CON); const float* cj = SC(CONJ);
  float* Fb = SC(CONF);
  BFOR(w, 6 * s.ncon) {
    const int c = w / 6, b = w - 6 * c;
    const float* C = con + RB_CONREC * c;
    const int adr = (int)C[RB_CR_ADR], np = rb_npyr((int)C[RB_CR_DIM]);
    float acc = 0;
    if ((int)C[RB_CR_KIND] != RB_KIND_PYRAMID) {   // rows = basis rows: the basis force is the row's force
      if (adr >= 0 && b < (int)C[RB_CR_DIM]) { bool qd; float cst; acc = rb_row_force(row + RB_ROWREC * (adr + b), qd, cst); }
    } else if (adr >= 0) {
      // (the pyramid rows of a contact are laid out normal-plus / normal-minus per friction axis k: rows 2 (k - 1) and 2 (k - 1) + 1 carry |aux| = k; basis row 0 sums
      //  every row, basis row b >= 1 exactly its two -- read directly instead of scanning all rows for them; a condim-1 contact has one row with aux 0)
      const int q0 = b == 0 ? 0 : 2 * (b - 1), q1 = b == 0 ? np : (2 * b <= np ? 2 * b : q0);
      for (int q = q0; q < q1; q++) {
        const float* R = row + RB_ROWREC * (adr + q);
        const int a = (int)fabsf(R[RB_RR_AUX]);
        if (b != 0 && a != b) continue;
        bool qd; float cst; const float f = rb_row_force(R, qd, cst);
        acc += b == 0 ? f : (R[RB_RR_AUX] < 0 ? -1.f : 1.f) * C[RB_CR_FRIC + a - 1] * f;
      }
    }
    Fb[w] = acc;
  }
  BSYNC();
  const int nf = m.nfric_dof + m.nfric_ten;
  const int* adr = (const int*)SC(DOFCON_ADR); const int* lst = (const int*)SC(DOFCON);
  // friction tendons and limits: one thread per row leaves (force, kind, dof or tendon, coefficient) in LDS; the dofs then scan LDS
  // (kind 0: nothing, 1: one dof, 2: a tendon's dofs)
  const int nst = nf + s.nlim - m.nfric_dof, cap = 2 * RB_CST / 4;
  BFOR(i, m.nv) {
    float acc = 0;
    const int fr = m.b_dof_fricrow[i];
    if (fr >= 0) { bool q; float c; acc += rb_row_force(row + RB_ROWREC * fr, q, c); }
    dst[i] = acc;
  }
  for (int r0 = 0; r0 < nst; r0 += cap) {
    const int nr = nst - r0 < cap ? nst - r0 : cap;
    BSYNC();
    BFOR(k, nr) {
      const float* R = row + RB_ROWREC * (m.nfric_dof + r0 + k);
      const int type = (int)R[RB_RR_TYPE], id = (int)R[RB_RR_ID];
      bool q; float c; const float f = rb_row_force(R, q, c);
      float* o = s.cst + 4 * k;
      o[0] = f; o[1] = f == 0.f ? 0.f : (type == 2 ? 1.f : 2.f); o[2] = (float)(type == 2 ? m.jnt_dofadr[id] : id); o[3] = type == 1 ? 1.f : R[RB_RR_AUX];
    }
    BSYNC();
    BFOR(i, m.nv) {
      float acc = 0;
      for (int k = 0; k < nr; k++) {
        const float* o = s.cst + 4 * k;
        if (o[1] == 0.f) continue;
        if (o[1] == 1.f) { if ((int)o[2] == i) acc += o[3] * o[0]; }
        else { const int id = (int)o[2]; for (int e = 0; e < RB_TENW; e++) if (m.b_ten_dofs[RB_TENW * id + e] == i) acc += o[3] * SC(TENJ)[RB_TENW * id + e] * o[0]; }
      }
      dst[i] += acc;
    }
  }
  BSYNC();
  BFOR(i, m.nv) {
    float acc = 0;
    for (int q = adr[i]; q < adr[i + 1]; q++) {
      const int ce = lst[q], c = ce / RB_CONW, e = ce - c * RB_CONW;
      const float* J = cj + 6 * RB_CONW * c; const float* F = Fb + 6 * c;
      acc += (J[e] * F[0] + J[RB_CONW + e] * F[1] + J[2 * RB_CONW + e] * F[2]) + (J[3 * RB_CONW + e] * F[3] + J[4 * RB_CONW + e] * F[4] + J[5 * RB_CONW + e] * F[5]);
    }
    dst[i] += acc;
  }
  BSYNC();
}
// out[d] = sum of D over the quadratic friction-loss row and the active limit row of dof d, for the dofs of group g (the diagonal a
// star group's Newton Hessian adds to M)
__device__ __forceinline__ void rb_row_diag(RbM m, RbLds& s, float* S, int g, float* out) {
  const float* row = SC(ROW);
  const int nstat = m.nfric_dof + m.nfric_ten + s.nlim;
  const int g0 = m.b_group_adr[g], n = m.b_group_adr[g + 1] - g0;
  BFOR(l, n) out[m.b_group_dofs[g0 + l]] = 0.f;
  BSYNC();
  for (int pass = 0; pass < 2; pass++) {
    BFOR(r, nstat) {
      const float* R = row + RB_ROWREC * r;
      const int type = (int)R[RB_RR_TYPE], id = (int)R[RB_RR_ID];
      if (type != (pass == 0 ? 0 : 2)) continue;
      const int d = type == 0 ? id : m.jnt_dofadr[id];
      if (m.b_dof_group[d] != g) continue;
      bool q; float c; rb_row_force(R, q, c);
      if (q) out[d] += R[RB_RR_D];
    }
    BSYNC();
  }
}

// s.A (holding the group's block of M) += J' diag(D, quadratic rows) J restricted to group g
#ifndef RB_HESS_MULTIWAVE_OFF
#define RB_HESS_MULTIWAVE_OFF 0      /* 1: configurations with several waves keep the staged one-contact-per-barrier assembly (A/B switch) */
#endif
// (-DRB_HESS_PROBE=k: cycles of section k of the assembly -- 1 static rows, 2 contact weights, 3 the contacts' entries -- accumulate in prof[15]; a profiling build)
#ifdef RB_HESS_PROBE
#define RB_HPROBE_BEGIN(k) long long tprobe##k = 0; if (RB_HESS_PROBE == k) { BSYNC(); tprobe##k = rg_clock(); }
#define RB_HPROBE_END(k) if (RB_HESS_PROBE == k) { BSYNC(); if (TID == 0) s.prof[15] += (float)(rg_clock() - tprobe##k); }
#else
#define RB_HPROBE_BEGIN(k)
#define RB_HPROBE_END(k)
#endif
__device__ __forceinline__ void rb_hessian_add(RbM m, RbLds& s, float* S, int g) {
  const float* row = SC(ROW); const float* con = SC(CON); const float* cj = SC(CONJ); const int* cidx = (const int*)SC(CONIDX);
  RB_HPROBE_BEGIN(1)
  const int nstat = m.nfric_dof + m.nfric_ten + s.nlim;
  // static rows: dof rows add to the diagonal (friction rows first, then the limit rows: a dof has one friction row and at most
  // one active limit side, so neither pass has two writers of an entry), tendon rows as small outer products one row at a time
  for (int pass = 0; pass < 2; pass++) {
    BFOR(r, nstat) {
      const float* R = row + RB_ROWREC * r;
      const int type = (int)R[RB_RR_TYPE], id = (int)R[RB_RR_ID];
      if (type != (pass == 0 ? 0 : 2)) continue;
      const int d = type == 0 ? id : m.jnt_dofadr[id];
      if (m.b_dof_group[d] != g) continue;
      bool q; float c; rb_row_force(R, q, c);
      if (q) { const int l = m.b_dof_local[d]; s.A[RB_TRI(l, l)] += R[RB_RR_D]; }
    }
    BSYNC();
  }
  for (int r = m.nfric_dof; r < nstat; r++) {   // (the friction-dof rows, the bulk of the static rows, were handled above)
    const float* R = row + RB_ROWREC * r;
    const int type = (int)R[RB_RR_TYPE], id = (int)R[RB_RR_ID];
    if (type != 1 && type != 3) continue;
    bool q; float c; rb_row_force(R, q, c);
    if (!q) continue;
    if (TID < RB_TENW * RB_TENW) {
      const int ea = TID / RB_TENW, eb = TID % RB_TENW;
      const int da = m.b_ten_dofs[RB_TENW * id + ea], db = m.b_ten_dofs[RB_TENW * id + eb];
      if (da >= 0 && db >= 0 && db <= da && m.b_dof_group[da] == g) s.A[RB_TRI(m.b_dof_local[da], m.b_dof_local[db])] += R[RB_RR_D] * SC(TENJ)[RB_TENW * id + ea] * SC(TENJ)[RB_TENW * id + eb];
    }
    BSYNC();
  }
  // contacts: A += Jc' W Jc with the 6 x 6 weight W of the contact's quadratic pyramid edges (w_row = e0 +- mu_k e_(k+1)):
  // the weights of all contacts first (one contact per thread), then the contacts one at a time, one (a, b) entry per thread
  RB_HPROBE_END(1)
  RB_HPROBE_BEGIN(2)
  float* Wc = SC(CONF);   // RB_NW words per contact.  Pyramidal: W00, W0k[5], Wkk[5] (mode 1).  Elliptic / equality: the lower triangle of the 6 x 6 weight
                          // of the basis rows (mode 2): diag(D) of the quadratic rows, or the cone's Hessian in its middle zone (engine_solver.c HessianCone)
  if (TID == 0) s.wcnt[0] = 0;
  BSYNC();
  BFOR(c, s.ncon) {
    const float* C = con + RB_CONREC * c;
    const int adr = (int)C[RB_CR_ADR], kind = (int)C[RB_CR_KIND], dim = (int)C[RB_CR_DIM];
    float* W = Wc + RB_NW * c;
    if (RB_NWAVE == 1) W[RB_NW - 1] = 0.f;      // (one wave: the consumer reads exactly the entries its mode defines, each of which is assigned below; no blanket zeroing)
    else for (int k = 0; k < RB_NW; k++) W[k] = 0.f;
    if (!(adr >= 0 && m.b_dof_group[cidx[RB_CONW * c]] == g)) continue;
    if (kind == RB_KIND_PYRAMID) {
      float W00 = 0, W0k[5] = {0, 0, 0, 0, 0}, Wkk[5] = {0, 0, 0, 0, 0};
      const int np = rb_npyr(dim);
      for (int q = 0; q < np; q++) {
        const float* R = row + RB_ROWREC * (adr + q);
        bool qd; float cst; rb_row_force(R, qd, cst);
        if (!qd) continue;
        const float D = R[RB_RR_D]; const int a = (int)fabsf(R[RB_RR_AUX]);
        W00 += D;
        if (a) { const float mu = (R[RB_RR_AUX] < 0 ? -1.f : 1.f) * C[RB_CR_FRIC + a - 1]; W0k[a - 1] += D * mu; Wkk[a - 1] += D * mu * mu; }
      }
      W[0] = W00; for (int k = 0; k < 5; k++) { W[1 + k] = W0k[k]; W[6 + k] = Wkk[k]; }
      if (W00 != 0.f) { W[RB_NW - 1] = 1.f; s.wcnt[0] = 1; }   // (benign race: every writer stores the same value)
    } else {
      const float* R0 = row + RB_ROWREC * adr;
      const float zone = R0[RB_RR_ZONE];
      bool any = false, diag = false;
      if (zone == 2.f) {   // middle zone of the cone: second derivatives of 1/2 Dm (N - mu T)^2 through U = diag(mu, friction) jar
        const float mu = C[RB_CR_SOLREF];
        float U[6], sc[6], T = 0.f;
        sc[0] = mu; U[0] = R0[RB_RR_JAR] * mu;
        for (int j = 1; j < dim; j++) { sc[j] = C[RB_CR_FRIC + j - 1]; U[j] = R0[RB_ROWREC * j + RB_RR_JAR] * sc[j]; T += U[j] * U[j]; }
        T = sqrtf(T);
        const float N = U[0], Dm = R0[RB_RR_D] / (mu * mu * (1.f + mu * mu)), iT = 1.f / T;
        for (int j = 0; j < dim; j++) for (int k = 0; k <= j; k++) {
          float h;
          if (j == 0) h = 1.f;
          else if (k == 0) h = -mu * U[j] * iT;
          else h = mu * N * iT * iT * iT * U[j] * U[k] + (j == k ? mu * mu - mu * N * iT : 0.f);
          W[j * (j + 1) / 2 + k] = Dm * sc[j] * sc[k] * h;
        }
        any = true;
      } else {
        for (int j = 0; j < dim; j++) { const float* R = R0 + RB_ROWREC * j; bool qd; float cst; rb_row_force(R, qd, cst); if (qd) { W[j * (j + 1) / 2 + j] = R[RB_RR_D]; any = true; } else if (RB_NWAVE == 1) W[j * (j + 1) / 2 + j] = 0.f; }
        diag = true;
      }
      if (any) { W[RB_NW - 1] = diag ? 3.f : 2.f; s.wcnt[0] = 1; }   // (mode 3: a diagonal weight -- equality rows, cone contacts outside the middle zone: the common case of objects at rest)
    }
  }
  BSYNC();
  const bool any = s.wcnt[0] != 0;
  BSYNC();
  if (TID == 0) s.wcnt[0] = 0;
  RB_HPROBE_END(2)
  if (!any) return;
  // One contact at a time (two contacts may share entries; the order of the sums is fixed), but its data -- six basis Jacobian rows,
  // the dofs' rows in the block, the weights -- is staged through LDS by one load per thread while the previous contact is being
  // added: a thread adding an entry then reads LDS only.  (Reading them from the scratch row cost ~30 global loads per thread and
  // contact: 40 % of the kernel's vector memory instructions.)
  // (one word per thread at 256 threads, three at 64; the load is issued before the current contact is added, the LDS store after it, so that the
  //  load's latency is covered by the adding instead of being waited for in front of it)
  const float* cloc = SC(CONLOC);
#ifndef RB_HESS_SERIAL
  if (RB_NWAVE == 1 || !(RB_HESS_MULTIWAVE_OFF)) {
    RB_HPROBE_BEGIN(3)
    // One wave per env (round 5): a lane per (contact, Jacobian column).  The columns of up to 64 contacts are numbered through a prefix sum of nnz; a round takes as
    // many whole contacts as fit 64 lanes.  A lane fetches ITS column of the six basis rows, its block row and the contact's weight once (one round of independent
    // loads), then walks the contact's columns k = 0 .. nnz - 1: the partner column comes by lane exchange from the lane that holds it, and the pair (e, k <= e) is
    // added into the block with an LDS atomic.  A single wave issues its LDS atomics in program order and resolves same-address lanes in lane order, so the sums
    // are run-to-run identical (rg_kernel.h relies on the same property).  Against the staged loop below: no barrier and no staging round trip per contact, no idle
    // lanes; against one entry per lane: a sixth of the loads.
    // Several waves (the large configuration): EVERY wave walks all columns, and a lane adds its entry only if the entry's block row belongs to its wave
    // (row mod RB_NWAVE): waves never touch the same address, each wave's own atomics are ordered as above -- still run-to-run identical, four waves' worth of
    // lanes on the pairs, at the price of every wave loading every column.
    int* offs = (int*)s.cst;           // 65 words: column offsets of this chunk's contacts
    int* cmeta = (int*)s.cst + 65;     // 64 words: dim | mode << 8 of this chunk's contacts
    for (int c0 = 0; c0 < s.ncon; c0 += 64) {
      const int c = c0 + WL;
      int cnt = 0, meta = 0;
      if (c < s.ncon) { const int mode = (int)Wc[RB_NW * c + RB_NW - 1]; if (mode != 0) { cnt = (int)con[RB_CONREC * c + RB_CR_NNZ]; meta = (int)con[RB_CONREC * c + RB_CR_DIM] | (mode << 8); } }
      int incl = cnt;
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl(incl, WL >= o ? WL - o : WL); if (WL >= o) incl += t; }
      BSYNC();
      if (WID == 0) { offs[WL + 1] = incl; cmeta[WL] = meta; }      // (every wave computed the same numbers: one writes)
      if (TID == 0) offs[0] = 0;
      BSYNC();
      int cs = 0;                       // first contact (chunk-local) of the round
      while (cs < 64 && offs[cs] < offs[64]) {
        // the round's contacts: cs .. ce - 1, the longest run whose columns fit the wave (a single contact has at most RB_CONW <= 64 columns)
        const unsigned long long fit = __ballot(WL >= cs && offs[WL + 1] - offs[cs] <= 64);
        const int ce = cs + __popcll(fit);
        const int t = offs[cs] + WL;
        const bool on = t < offs[ce];
        int lo = cs, hi = ce;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= t) lo = mid; else hi = mid; }
        const int cl = on ? lo : cs, cc = c0 + cl, e = on ? t - offs[cl] : 0;
        const int base = offs[cl] - offs[cs], nnz = offs[cl + 1] - offs[cl], dim = cmeta[cl] & 255, mode = cmeta[cl] >> 8;
        const float* K = cj + 6 * RB_CONW * cc; const float* W = Wc + RB_NW * cc;
        float ja[6], w[21];
#pragma unroll
        for (int q = 0; q < 6; q++) ja[q] = (on && q < dim) ? K[q * RB_CONW + e] : 0.f;
        const int la = on ? (int)cloc[RB_CONW * cc + e] : 0;
        // the weight: pyramid (mode 1) W00, W0k[5], Wkk[5]; diagonal (mode 3) the six diagonal entries; general (mode 2) the lower triangle of the 6 x 6
#pragma unroll
        for (int q = 0; q < 21; q++) w[q] = 0.f;
        if (on && mode == 1) { for (int q = 0; q < 11; q++) w[q] = W[q]; }
        else if (on && mode == 3) { for (int q = 0; q < 6; q++) if (q < dim) w[q] = W[q * (q + 1) / 2 + q]; }
        else if (on) { for (int q = 0; q < 21; q++) w[q] = W[q]; }
        int kmax = on ? nnz : 0;
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(kmax, o); kmax = other > kmax ? other : kmax; }
        for (int k = 0; k < kmax; k++) {
          const int src = (base + (k < nnz ? k : 0)) & 63;
          float jb[6];
#pragma unroll
          for (int q = 0; q < 6; q++) jb[q] = __shfl(ja[q], src);
          const int lb = __shfl(la, src);
          if (!(on && k <= e)) continue;
          float v;
          if (mode == 1) {
            v = w[0] * ja[0] * jb[0];
#pragma unroll
            for (int q = 0; q < 5; q++) if (q < dim - 1) v += w[1 + q] * (ja[0] * jb[q + 1] + ja[q + 1] * jb[0]) + w[6 + q] * ja[q + 1] * jb[q + 1];
          } else if (mode == 3) {
            v = 0.f;
#pragma unroll
            for (int q = 0; q < 6; q++) if (q < dim) v += w[q] * ja[q] * jb[q];
          } else {
            v = 0.f;
#pragma unroll
            for (int q = 0; q < 6; q++) if (q < dim) {
              v += w[q * (q + 1) / 2 + q] * ja[q] * jb[q];
#pragma unroll
              for (int r = 0; r < 5; r++) if (r < q) { const float wk = w[q * (q + 1) / 2 + r]; if (wk != 0.f) v += wk * (ja[q] * jb[r] + ja[r] * jb[q]); }
            }
          }
          const int rmax = la >= lb ? la : lb, rmin = la >= lb ? lb : la;
          if (RB_NWAVE == 1 || (rmax & (RB_NWAVE - 1)) == WID) atomicAdd(&s.A[RB_TRI(rmax, rmin)], v);
        }
        cs = ce;
      }
      BSYNC();
    }
    RB_HPROBE_END(3)
    return;
  }
#endif
  constexpr int NST = (RB_CST + RB_T - 1) / RB_T;   // words per thread (1 at 256 threads)
  struct Staged { float v[NST]; };
  auto stage_load = [&](int c) -> Staged {
    Staged r;
#pragma unroll
    for (int k = 0; k < NST; k++) {
      const int t = TID + k * RB_T;
      float v = 0.f;
      if (t < 6 * RB_CONW) v = cj[6 * RB_CONW * c + t];
      else if (t < 7 * RB_CONW) v = cloc[RB_CONW * c + t - 6 * RB_CONW];
      else if (t < 7 * RB_CONW + RB_NW) v = Wc[RB_NW * c + t - 7 * RB_CONW];
      else if (t < RB_CST) v = con[RB_CONREC * c + (t == 7 * RB_CONW + RB_NW ? RB_CR_NNZ : RB_CR_DIM)];
      r.v[k] = v;
    }
    return r;
  };
  auto stage_store = [&](int b, const Staged& r) {
#pragma unroll
    for (int k = 0; k < NST; k++) { const int t = TID + k * RB_T; if (t < RB_CST) s.cst[RB_CST * b + t] = r.v[k]; }
  };
  const Staged zero = {};
  int buf = 0;
  stage_store(0, stage_load(0));
  Staged next = s.ncon > 1 ? stage_load(1) : zero;   // three contacts in flight: the words stored at the end of a pass were requested two passes earlier
  Staged next2 = s.ncon > 2 ? stage_load(2) : zero;
  BSYNC();
  for (int c = 0; c < s.ncon; c++, buf ^= 1) {
    const Staged after = c + 3 < s.ncon ? stage_load(c + 3) : zero;
    const float* K = s.cst + RB_CST * buf;
    const float* W = K + 7 * RB_CONW;
    const float mode = W[RB_NW - 1];
    if (mode != 0.f) {
      const int nnz = (int)K[7 * RB_CONW + RB_NW], dim = (int)K[7 * RB_CONW + RB_NW + 1];
      for (int w = TID; w < nnz * (nnz + 1) / 2; w += RB_T) {
        int ea = (int)((sqrtf(8.f * (float)w + 1.f) - 1.f) * 0.5f);
        if (ea * (ea + 1) / 2 > w) ea--; else if ((ea + 1) * (ea + 2) / 2 <= w) ea++;
        const int eb = w - ea * (ea + 1) / 2;                    // every unordered pair of the contact's dofs once (eb <= ea)
        float v;
        if (mode == 1.f) {
          v = W[0] * K[ea] * K[eb];
          for (int k = 0; k < dim - 1; k++) {
            const float ja = K[(k + 1) * RB_CONW + ea], jb = K[(k + 1) * RB_CONW + eb];
            v += W[1 + k] * (K[ea] * jb + ja * K[eb]) + W[6 + k] * ja * jb;
          }
        } else if (mode == 3.f) {   // diagonal weight (the same sums as the general form with its zero terms left out)
          v = 0.f;
          for (int j = 0; j < dim; j++) v += W[j * (j + 1) / 2 + j] * K[j * RB_CONW + ea] * K[j * RB_CONW + eb];
        } else {   // general symmetric weight of the basis rows
          v = 0.f;
          for (int j = 0; j < dim; j++) {
            const float ja = K[j * RB_CONW + ea], jb = K[j * RB_CONW + eb];
            v += W[j * (j + 1) / 2 + j] * ja * jb;
            for (int k = 0; k < j; k++) { const float wk = W[j * (j + 1) / 2 + k]; if (wk != 0.f) v += wk * (ja * K[k * RB_CONW + eb] + K[k * RB_CONW + ea] * jb); }
          }
        }
        const int la = (int)K[6 * RB_CONW + ea], lb = (int)K[6 * RB_CONW + eb];
        s.A[la >= lb ? RB_TRI(la, lb) : RB_TRI(lb, la)] += v;
      }
    }
    stage_store(buf ^ 1, next);
    next = next2; next2 = after;
    BSYNC();
  }
}

// ------------------------------------------------------------------------------------------------- stage calls
// The stages are real function calls (as rg_kernel.h's): each gets its own register allocation, and the substep loop, the TCP hook's forward and the final
// full forward share ONE copy of every stage (inlined, the kernel was 445 kB of code with every stage three times over).  A stage finds the model, the launch
// descriptor and the env's scratch row through wave-uniform addresses (read back into SGPRs: everything loaded through them is a scalar load again).
struct RbCtx { const void* km; const void* kl; float* S; };
#ifdef RG_EMUL
#define RB_M(c) (*(const RbModelDev*)(c).km)
#define RB_L(c) (*(const RbLaunch*)(c).kl)
#define RB_SP(c) ((c).S)
#define RB_STAGE static inline
#else
#define RB_M(c) (*(const RG_AS4 RbModelDev*)rg_uniform((c).km))
#define RB_L(c) (*(const RG_AS4 RbLaunch*)rg_uniform((c).kl))
#define RB_SP(c) ((float*)rg_uniform((c).S))
#define RB_STAGE __device__ __attribute__((noinline))
#endif
#define RB_STAGE_ENTER() RbM m = RB_M(c); RbLds& s = RB_S(); float* S = RB_SP(c); (void)m; (void)s; (void)S
// ------------------------------------------------------------------------------------------------- solver
struct RbLs { float cost, grad, hess; };
// the rows a thread owns in the line search (r = TID and TID + RB_T; rows beyond 2 RB_T are read from the scratch row each time): loaded once
struct RbLsRows { float D[2], jar[2], jv[2], fl[2]; int fric[2]; };
// row classes of the line search: 0 one-sided (limits, pyramid edges), 1 friction loss, 2 always quadratic (equality), 3 row of an elliptic contact
// (skipped here: rb_ls_cones evaluates the contact as a whole)
__device__ __forceinline__ int rb_ls_class(const float* R) {
  const int type = (int)R[RB_RR_TYPE];
  return type < 2 ? 1 : (type == 5 ? (R[RB_RR_ZONE] == 3.f ? 2 : 3) : 0);
}
__device__ __forceinline__ void rb_ls_acc(float D, float jar, float jv, float fl, int cls, float alpha, float& cst, float& grd, float& hss) {
  const float x = jar + alpha * jv;
  if (cls == 3) return;
  if (cls == 2) { cst += 0.5f * D * x * x; grd += D * x * jv; hss += D * jv * jv; return; }
  if (cls == 1) {
    const float Rr = 1.f / D;
    if (x <= -Rr * fl) { cst += fl * (-0.5f * Rr * fl - x); grd += -fl * jv; }
    else if (x >= Rr * fl) { cst += fl * (-0.5f * Rr * fl + x); grd += fl * jv; }
    else { cst += 0.5f * D * x * x; grd += D * x * jv; hss += D * jv * jv; }
  } else if (x < 0) { cst += 0.5f * D * x * x; grd += D * x * jv; hss += D * jv * jv; }
}
// the elliptic contacts' cost along the search direction and its first two derivatives in alpha (oracle ls_eval, elliptic branch): one thread per contact
__device__ __forceinline__ void rb_ls_cones(const float* row, const float* con, int ncon, float alpha, float& cst, float& grd, float& hss) {
  for (int c = TID + RB_T; c < ncon; c += RB_T) {      // (contact TID itself: rb_ls_cone_eval on the thread's cached copy)
    const float* C = con + RB_CONREC * c;
    const int adr = (int)C[RB_CR_ADR], dim = (int)C[RB_CR_DIM];
    if ((int)C[RB_CR_KIND] != RB_KIND_ELLIPTIC || adr < 0 || dim < 2) continue;
    const float* R0 = row + RB_ROWREC * adr;
    const float mu = C[RB_CR_SOLREF];
    float T = 0.f, UV = 0.f, VV = 0.f;
    const float N = (R0[RB_RR_JAR] + alpha * R0[RB_RR_JV]) * mu, N1 = R0[RB_RR_JV] * mu;
    for (int j = 1; j < dim; j++) {
      const float* R = R0 + RB_ROWREC * j; const float f = C[RB_CR_FRIC + j - 1];
      const float u = (R[RB_RR_JAR] + alpha * R[RB_RR_JV]) * f, v = R[RB_RR_JV] * f;
      T += u * u; UV += u * v; VV += v * v;
    }
    T = sqrtf(T);
    if (N >= mu * T || (T <= 0.f && N >= 0.f)) continue;
    if (mu * N + T <= 0.f || (T <= 0.f && N < 0.f)) {
      for (int j = 0; j < dim; j++) { const float* R = R0 + RB_ROWREC * j; const float D = R[RB_RR_D], jv = R[RB_RR_JV], x = R[RB_RR_JAR] + alpha * jv; cst += 0.5f * D * x * x; grd += D * x * jv; hss += D * jv * jv; }
    } else {
      const float Dm = R0[RB_RR_D] / (mu * mu * (1.f + mu * mu)), NT = N - mu * T, T1 = UV / T, T2 = (VV - T1 * T1) / T, dN = N1 - mu * T1;
      cst += 0.5f * Dm * NT * NT; grd += Dm * NT * dN; hss += Dm * (dN * dN - NT * mu * T2);
    }
  }
}
// the elliptic contact a thread owns in the line search (contact TID; contacts beyond RB_T go through rb_ls_cones each time): its rows' D / jar / jv and its friction
// coefficients, loaded ONCE per line search instead of once per evaluation (the same arithmetic on the same numbers as rb_ls_cones)
struct RbLsCone { int dim; float mu, fr[5], D[6], jar[6], jv[6]; };
__device__ __forceinline__ void rb_ls_cone_load(const float* row, const float* con, int ncon, RbLsCone& k) {
  k.dim = 0;
  if (TID >= ncon) return;
  const float* C = con + RB_CONREC * TID;
  const int adr = (int)C[RB_CR_ADR], dim = (int)C[RB_CR_DIM];
  if ((int)C[RB_CR_KIND] != RB_KIND_ELLIPTIC || adr < 0 || dim < 2) return;
  k.dim = dim; k.mu = C[RB_CR_SOLREF];
  const float* R0 = row + RB_ROWREC * adr;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const float* R = R0 + RB_ROWREC * (j < dim ? j : 0);
    k.D[j] = R[RB_RR_D]; k.jar[j] = R[RB_RR_JAR]; k.jv[j] = R[RB_RR_JV];
    if (j >= 1) k.fr[j - 1] = C[RB_CR_FRIC + (j < dim ? j - 1 : 0)];
  }
}
__device__ __forceinline__ void rb_ls_cone_eval(const RbLsCone& k, float alpha, float& cst, float& grd, float& hss) {
  const int dim = k.dim;
  if (dim < 2) return;
  const float mu = k.mu;
  float T = 0.f, UV = 0.f, VV = 0.f;
  const float N = (k.jar[0] + alpha * k.jv[0]) * mu, N1 = k.jv[0] * mu;
#pragma unroll
  for (int j = 1; j < 6; j++) if (j < dim) {
    const float f = k.fr[j - 1];
    const float u = (k.jar[j] + alpha * k.jv[j]) * f, v = k.jv[j] * f;
    T += u * u; UV += u * v; VV += v * v;
  }
  T = sqrtf(T);
  if (N >= mu * T || (T <= 0.f && N >= 0.f)) return;
  if (mu * N + T <= 0.f || (T <= 0.f && N < 0.f)) {
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) { const float D = k.D[j], jv = k.jv[j], x = k.jar[j] + alpha * jv; cst += 0.5f * D * x * x; grd += D * x * jv; hss += D * jv * jv; }
  } else {
    const float Dm = k.D[0] / (mu * mu * (1.f + mu * mu)), NT = N - mu * T, T1 = UV / T, T2 = (VV - T1 * T1) / T, dN = N1 - mu * T1;
    cst += 0.5f * Dm * NT * NT; grd += Dm * NT * dN; hss += Dm * (dN * dN - NT * mu * T2);
  }
}
__device__ __forceinline__ RbLs rb_ls_eval(RbLds& s, const float* row, const float* con, int ncone, const RbLsRows& own, const RbLsCone& cone, int nefc, float alpha, float q0, float q1, float q2) {
  float cst = 0, grd = 0, hss = 0;
#pragma unroll
  for (int k = 0; k < 2; k++) if (TID + k * RB_T < nefc) rb_ls_acc(own.D[k], own.jar[k], own.jv[k], own.fl[k], own.fric[k], alpha, cst, grd, hss);
  for (int r = TID + 2 * RB_T; r < nefc; r += RB_T) {
    const float* R = row + RB_ROWREC * r;
    rb_ls_acc(R[RB_RR_D], R[RB_RR_JAR], R[RB_RR_JV], R[RB_RR_FLOSS], rb_ls_class(R), alpha, cst, grd, hss);
  }
  if (ncone > 0) {
    rb_ls_cone_eval(cone, alpha, cst, grd, hss);
    if (ncone > RB_T) rb_ls_cones(row, con, ncone, alpha, cst, grd, hss);
  }
  rb_sum3(s, cst, grd, hss);
  RbLs p; p.cost = alpha * alpha * q2 + alpha * q1 + q0 + cst; p.grad = 2.f * alpha * q2 + q1 + grd; p.hess = 2.f * q2 + hss;
  return p;
}
// exact minimiser of the convex piecewise-quadratic 1-D restriction (oracle line_search: safeguarded Newton on the derivative)
__device__ __forceinline__ float rb_line_search(RbLds& s, const float* row, const float* con, int ncone, int nefc, float q0, float q1, float q2, float gtol, int maxit) {
  RbLsRows own;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int r = TID + k * RB_T;
    const float* R = row + RB_ROWREC * (r < nefc ? r : 0);
    own.D[k] = R[RB_RR_D]; own.jar[k] = R[RB_RR_JAR]; own.jv[k] = R[RB_RR_JV]; own.fl[k] = R[RB_RR_FLOSS]; own.fric[k] = rb_ls_class(R);
  }
  RbLsCone cone;
  rb_ls_cone_load(row, con, ncone, cone);
  const RbLs p0 = rb_ls_eval(s, row, con, ncone, own, cone, nefc, 0.f, q0, q1, q2);
  if (p0.grad >= 0 || p0.hess <= 0) return 0.f;
  float lo = 0, hi = -1, glo = p0.grad, hlo = p0.hess, ghi = 0, hhi = 0;
  float a = -p0.grad / p0.hess;
  float best_a = 0.f, best_cost = p0.cost;   // the best point seen: what is returned when the iteration limit ends the search
  float wprev = 3.0e38f; int since = 0;
  for (int it = 0; it < maxit; it++) {
    const RbLs p = rb_ls_eval(s, row, con, ncone, own, cone, nefc, a, q0, q1, q2);
    if (p.cost < best_cost) { best_cost = p.cost; best_a = a; }
    if (fabsf(p.grad) < gtol) return a;
    if (p.grad < 0) { lo = a; glo = p.grad; hlo = p.hess; } else { hi = a; ghi = p.grad; hhi = p.hess; }
    float cand = lo - glo / hlo;
    if (hi >= 0) {
      if (!(cand > lo && cand < hi)) {
        cand = hi - ghi / hhi;
        if (!(cand > lo && cand < hi)) cand = 0.5f * (lo + hi);
      }
      // at a kink of the derivative (a stiff row switching on) the Newton steps from the two ends can alternate between two points on either side of it
      // for ever: three steps in a row that are not at least halving (a converging Newton iteration shrinks them much faster) -> bisect (oracle line_search)
      float step = fabsf(cand - a);
      if (step > 0.5f * wprev) since++; else since = 0;     // (wprev: the previous step's length)
      if (since >= 3) { cand = 0.5f * (lo + hi); since = 0; step = fabsf(cand - a); }
      wprev = step;
    }
    if (cand == a) return a;
    a = cand;
  }
  return best_a;
}
// mj_solNewton (oracle ro_solve): s.qa <- qacc, s.qfrc_con <- J' f; returns the iteration count
// the solver's sub-stages as calls of their own: one copy each of the products M x / J x (three call sites), J' f, the Hessian assembly, the factorisation and the line search
__device__ __forceinline__ const float* rb_vec_src(RbLds& s, int which) { return which == 0 ? s.warm : which == 1 ? s.qacc_smooth : which == 2 ? s.qa : s.search; }
RB_STAGE void sv_MJ_mul(RbCtx c, int which) {   // which: 0 warm start, 1 qacc_smooth, 2 qa (-> Ma, jar); 3 search (-> Mv, jv)
  RB_STAGE_ENTER();
  const float* a = rb_vec_src(s, which);
  rb_M_mul(m, SC(MSP), a, which == 3 ? s.Mv : s.Ma);
  rb_J_mul(m, s, S, a, which == 3);
}
// after the line search: Ma += alpha M v, jar += alpha J v on every row (mj_solNewton advances its products the same way), then the cones' zones / forces / costs
// at the new residuals -- instead of forming M qa and J qa again
RB_STAGE void sv_advance(RbCtx c, float alpha) {
  RB_STAGE_ENTER();
  float* row = SC(ROW);
  BFOR(i, m.nv) s.Ma[i] += alpha * s.Mv[i];
  BFOR(r, s.nefc) { float* R = row + RB_ROWREC * r; R[RB_RR_JAR] += alpha * R[RB_RR_JV]; }
  BSYNC();
  if (m.cone == 1) rb_cone_update(m, s, S);
}
RB_STAGE void sv_JT_force(RbCtx c) { RB_STAGE_ENTER(); rb_JT_force(m, s, S, s.qfrc_con); }
RB_STAGE void sv_hessian(RbCtx c, int grp) { RB_STAGE_ENTER(); rb_M_block(m, s, SC(MSP), grp, (const float*)0, 0.f); rb_hessian_add(m, s, S, grp); }
#if RB_T == 256 && RB_MAXGROUP == 96 && !defined(RB_LDS_CHOL)
RB_STAGE int sv_factor(RbCtx c, int grp) { RB_STAGE_ENTER(); const int n = m.b_group_adr[grp + 1] - m.b_group_adr[grp]; rb_scale_block(s, n); return rb_chol_mfma(s, n) ? 1 : 0; }
#else
RB_STAGE int sv_factor(RbCtx c, int grp) { RB_STAGE_ENTER(); const int n = m.b_group_adr[grp + 1] - m.b_group_adr[grp]; rb_scale_block(s, n); return rb_chol(s, n) ? 1 : 0; }
#endif
RB_STAGE void sv_direction(RbCtx c, int grp) { RB_STAGE_ENTER(); rb_group_solve(m, s, grp, s.grad, s.search, -1.f); }
#if RB_T == 64
RB_STAGE int sv_factor_direction(RbCtx c, int grp) { RB_STAGE_ENTER(); return rb_reg_solve(m, s, grp, s.grad, s.search, -1.f) ? 1 : 0; }   // one wave: factorisation and both substitutions in registers
#endif
RB_STAGE void sv_star_direction(RbCtx c, int grp, int part) {
  RB_STAGE_ENTER();
  if (part == 0) rb_row_diag(m, s, S, grp, s.Mv); else rb_star_group_solve(m, s, SC(MSP), grp, s.Mv, 1.f, s.grad, s.search, -1.f);
}
RB_STAGE float sv_line_search(RbCtx c, float gauss, float q1, float q2, float gtol) {
  RB_STAGE_ENTER();
  return rb_line_search(s, SC(ROW), SC(CON), m.cone == 1 ? s.ncon : 0, s.nefc, gauss, q1, q2, gtol, 40);
}
// dst = inv(M + h B) src of every group: mode 0 qacc_smooth = inv(M) qfrc_smooth (h = 0), mode 1 search = inv(M + h B) grad (the Euler step)
RB_STAGE void sv_M_solve(RbCtx c, int mode, int flags) {
  RB_STAGE_ENTER();
  const float* diag = mode ? PRM(dof_damping, RB_P_DOF_DAMPING) : (const float*)0; const float h = mode ? m.timestep : 0.f;
  const float* src = mode ? s.grad : s.qfrc_smooth; float* dst = mode ? s.search : s.qacc_smooth;
  if (m.b_tree8[0] > 0 && !(flags & 4)) { rb_trees8_solve(m, s, SC(MSP), diag, h, src, dst, 1.f); return; }
  for (int grp = 0; grp < m.ngroup; grp++) {
    if (m.b_star_grp[4 * grp + 3] && !(flags & 4)) { rb_star_group_solve(m, s, SC(MSP), grp, diag, h, src, dst, 1.f); continue; }   // (flags bit 2: dense path everywhere, test hook)
    rb_M_block(m, s, SC(MSP), grp, diag, h);
#if RB_T == 64 && !defined(RB_LDS_CHOL)
    if (!rb_reg_solve(m, s, grp, src, dst, 1.f) && TID == 0) s.status |= RG_STATUS_BAD_FACTOR;
#else
    rb_scale_block(s, m.b_group_adr[grp + 1] - m.b_group_adr[grp]);
    if (!rb_chol(s, m.b_group_adr[grp + 1] - m.b_group_adr[grp]) && TID == 0) s.status |= RG_STATUS_BAD_FACTOR;
    rb_group_solve(m, s, grp, src, dst, 1.f);
#endif
  }
}
__device__ __forceinline__ int rb_solve(RbCtx cx, RbM m, RbLds& s, float* S, int flags) {
  long long tq0 = rg_clock(), tq1;
#define RB_PROFS(k) do { if (flags & 2) { BSYNC(); tq1 = rg_clock(); if (TID == 0) s.prof[k] += (float)(tq1 - tq0); tq0 = tq1; } } while (0)
  const int nv = m.nv, ne = s.nefc;
  float* row = SC(ROW); const float* Msp = SC(MSP);
  if (ne == 0) { BFOR(i, nv) { s.qa[i] = s.qacc_smooth[i]; s.qfrc_con[i] = 0.f; } BSYNC(); return 0; }
  const float scale = 1.f / (m.meaninertia * (nv > 1 ? nv : 1));
  // warm start: the better of qacc_warmstart and qacc_smooth.  qacc_smooth is evaluated FIRST: the warm start wins on most mj_steps, and then Ma / jar / the cones
  // already hold the starting point's products when the iteration begins (round 5: one M x, J x evaluation less per solve; same numbers)
  float cost2[2];
  for (int pass = 1; pass >= 0; pass--) {
    const float* a = pass == 0 ? s.warm : s.qacc_smooth;
    sv_MJ_mul(cx, pass);
    float g = 0, c = 0;
    BFOR(i, nv) g += 0.5f * (s.Ma[i] - s.qfrc_smooth[i]) * (a[i] - s.qacc_smooth[i]);
    BFOR(r, ne) { bool q; float cc; rb_row_force(row + RB_ROWREC * r, q, cc); c += cc; }
    cost2[pass] = rb_sum(s, g + c);
  }
  const bool from_warm = cost2[0] < cost2[1];
  { const float* a = from_warm ? s.warm : s.qacc_smooth; BFOR(i, nv) s.qa[i] = a[i]; }
  BSYNC();
  // fp32 cannot resolve (scaled) cost improvements below ~3e-7 (rg_kernel.h RG_TOL_FLOOR): MuJoCo's 1e-8 is an fp64 number
  const float tol = fmaxf(m.tolerance, 3e-7f);
  float cost = 0, oldcost = 0;
  int iters = 0;
  for (int iter = 0;; iter++) {
    if (iter == 0 && !from_warm) sv_MJ_mul(cx, 2);      // (the last evaluation was the warm start's: redo qacc_smooth's; later iterations advance incrementally, sv_advance)
    float g = 0, c = 0;
    BFOR(i, nv) g += 0.5f * (s.Ma[i] - s.qfrc_smooth[i]) * (s.qa[i] - s.qacc_smooth[i]);
    BFOR(r, ne) { bool q; float cc; rb_row_force(row + RB_ROWREC * r, q, cc); c += cc; }
    const float gauss = rb_sum(s, g);
    const float ccost = rb_sum(s, c);
    oldcost = cost; cost = gauss + ccost;
    RB_PROFS(8);
    sv_JT_force(cx);
    float gn = 0;
    BFOR(i, nv) { const float gi = s.Ma[i] - s.qfrc_smooth[i] - s.qfrc_con[i]; s.grad[i] = gi; gn += gi * gi; }
    gn = sqrtf(rb_sum(s, gn)) * scale;
    // (fp32: the cost itself carries ~1e-6 of relative rounding noise, with ~450 rows far more than the absolute floor)
#if defined(RG_EMUL) && defined(RB_TRACE)
    if (TID == 0) printf("iter %d cost %.9g improvement*scale %.3e gn %.3e scale %.3e\n", iter, (double)cost, (double)(scale * (oldcost - cost)), (double)gn, (double)scale);
#endif
    if (iter > 0 && scale * (oldcost - cost) < fmaxf(tol, RB_COST_EPS * scale * fabsf(cost))) break;
    if (gn < tol || iter >= m.iterations) break;
    iters = iter + 1;
    RB_PROFS(9);
    // search = - inv(H) grad, group by group
    bool okf = true;
    for (int grp = 0; grp < m.ngroup; grp++) {
      if (m.b_star_grp[4 * grp] && !(flags & 4)) {   // H = M + diagonal on this group: block elimination along the tree
        sv_star_direction(cx, grp, 0); RB_PROFS(10);
        sv_star_direction(cx, grp, 1); RB_PROFS(11);
        continue;
      }
      sv_hessian(cx, grp); RB_PROFS(10);
#if RB_T == 64 && !defined(RB_LDS_CHOL)
      okf = (sv_factor_direction(cx, grp) != 0) && okf; RB_PROFS(11);
#else
      okf = (sv_factor(cx, grp) != 0) && okf; RB_PROFS(11);
      sv_direction(cx, grp); RB_PROFS(12);
#endif
    }
    if (!okf && TID == 0) s.status |= RG_STATUS_BAD_FACTOR;
    sv_MJ_mul(cx, 3);
    float q1 = 0, q2 = 0, sn = 0;
    BFOR(i, nv) { q1 += s.search[i] * (s.Ma[i] - s.qfrc_smooth[i]); q2 += 0.5f * s.search[i] * s.Mv[i]; sn += s.search[i] * s.search[i]; }
    rb_sum3(s, q1, q2, sn);
    sn = sqrtf(sn);
    if (sn < RB_MINVAL) break;
    const float gtol = tol * 0.01f * sn / scale * 1e-3f;   // (tolerance x ls_tolerance x |search| / scale x 1e-3: oracle's "exact" line search)
    RB_PROFS(13);
    const float alpha = sv_line_search(cx, gauss, q1, q2, gtol); RB_PROFS(14);
    if (alpha == 0.f) break;
    BFOR(i, nv) s.qa[i] += alpha * s.search[i];
    BSYNC();
    sv_advance(cx, alpha);
  }
  // forces at the solution (the loop's last gradient evaluation holds them: qfrc_con = J' f(qa))
  return iters;
}

// ------------------------------------------------------------------------------------------------- integration
// mj_Euler: implicit in joint damping, quaternion integration
__device__ __forceinline__ void rb_euler(RbCtx cx, RbM m, RbLds& s, float* S, int flags) {
  const float h = m.timestep;
  BFOR(i, m.nv) s.grad[i] = s.qfrc_smooth[i] + s.qfrc_con[i];
  BSYNC();
  sv_M_solve(cx, 1, flags);
  BFOR(i, m.nv) s.qvel[i] += h * s.search[i];
  BSYNC();
  BFOR(j, m.njnt) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j]; const int t = m.jnt_type[j];
    if (t == RG_JNT_FREE) { for (int k = 0; k < 3; k++) s.qpos[qa + k] += h * s.qvel[da + k]; qa += 3; da += 3; }
    if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
      v3 w = mk3(s.qvel[da], s.qvel[da + 1], s.qvel[da + 2]);
      const float ang = norm(w) * h;
      if (ang > 0) { const q4 qn = qnormalize(qmul(ldq(s.qpos + qa), axisangle(normalized(w), ang))); stq(s.qpos + qa, qn); }
    } else s.qpos[qa] += h * s.qvel[da];
  }
  BSYNC();
}

// ------------------------------------------------------------------------------------------------- sensors
// mj_sensorPos (jointpos) and mj_sensorAcc (force, torque) after a full forward pass at the final state: mj_rnePostConstraint — body
// accelerations with the solver's qacc, contact forces as external forces, the interaction force of the sensor site's body with its
// parent = the sum over its subtree — moved to the site and rotated into the site frame (oracle ro_rne_post_constraint / ro_sensor).
__device__ __forceinline__ void rb_sensors(RbM m, RbLds& s, float* S, float* out) {
  const float *cdof = SC(CDOF), *cdofdot = SC(CDOFDOT), *cvel = SC(CVEL), *rootcom = SC(ROOTCOM), *con = SC(CON), *row = SC(ROW);
  float *cacc = SC(CACC), *cfrc = SC(CFRC), *cext = SC(CFRCEXT);
  if (TID < 6) { cacc[TID] = TID < 3 ? 0.f : -PRM(opt_gravity, RB_P_GRAVITY)[TID - 3]; cfrc[TID] = 0.f; }
  BFOR(b, m.nbody) {
    float acc[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < s.ncon; c++) {
      const float* C = con + RB_CONREC * c;
      const int adr = (int)C[RB_CR_ADR], kind = (int)C[RB_CR_KIND];
      if (adr < 0 || kind == RB_KIND_EQUALITY) continue;
      const int b1 = m.geom_bodyid[(int)C[RB_CR_G1]], b2 = m.geom_bodyid[(int)C[RB_CR_G2]];
      if (b == 0 || (b != b1 && b != b2)) continue;
      const int dim = (int)C[RB_CR_DIM];
      float cf[6] = {0, 0, 0, 0, 0, 0};
      if (kind == RB_KIND_ELLIPTIC) { for (int j = 0; j < dim; j++) cf[j] = row[RB_ROWREC * (adr + j) + RB_RR_FORCE]; }
      else if (dim == 1) { bool q; float cc; cf[0] = rb_row_force(row + RB_ROWREC * adr, q, cc); }
      else for (int j = 0; j < dim - 1; j++) {
        bool q; float cc;
        const float fp = rb_row_force(row + RB_ROWREC * (adr + 2 * j), q, cc), fn = rb_row_force(row + RB_ROWREC * (adr + 2 * j + 1), q, cc);
        cf[0] += fp + fn; cf[1 + j] = (fp - fn) * C[RB_CR_FRIC + j];
      }
      const v3 f0 = ld3(C + RB_CR_FRAME), f1 = ld3(C + RB_CR_FRAME + 3), f2 = ld3(C + RB_CR_FRAME + 6);
      const v3 f = f0 * cf[0] + f1 * cf[1] + f2 * cf[2], t = f0 * cf[3] + f1 * cf[4] + f2 * cf[5];
      const v3 off = ld3(C + RB_CR_POS) - ld3(rootcom + 3 * m.body_rootid[b]);
      const v3 tq = t + cross(off, f);
      const float sg = (b == b2 ? 1.f : 0.f) - (b == b1 ? 1.f : 0.f);
      acc[0] += sg * tq.x; acc[1] += sg * tq.y; acc[2] += sg * tq.z; acc[3] += sg * f.x; acc[4] += sg * f.y; acc[5] += sg * f.z;
    }
    for (int k = 0; k < 6; k++) cext[6 * b + k] = acc[k];
  }
  BSYNC();
  for (int L = 0; L < m.nlevel; L++) {
    for (int q = m.b_lvl_adr[L] + TID; q < m.b_lvl_adr[L + 1]; q += RB_T) {
      const int b = m.b_lvl_body[q], p = m.body_parentid[b];
      float ca[6], cv[6];
      for (int c = 0; c < 6; c++) { ca[c] = cacc[6 * p + c]; cv[c] = cvel[6 * b + c]; }
      for (int k = 0; k < m.body_dofnum[b]; k++) { const int i = m.body_dofadr[b] + k; for (int c = 0; c < 6; c++) ca[c] += cdofdot[6 * i + c] * s.qvel[i] + cdof[6 * i + c] * s.qa[i]; }
      float t1[6], t2[6], t3[6];
      mul_inert_vec(t1, SC(CINERT) + 10 * b, ca);
      mul_inert_vec(t2, SC(CINERT) + 10 * b, cv);
      cross_force(t3, cv, t2);
      for (int c = 0; c < 6; c++) { cacc[6 * b + c] = ca[c]; cfrc[6 * b + c] = t1[c] + t3[c] - cext[6 * b + c]; }
    }
    BSYNC();
  }
  BFOR(k, m.nsensor) {
    const int type = m.sensor_type[k], obj = m.sensor_objid[k], adr = m.sensor_adr[k];
    if (type == 8) { out[adr] = s.qpos[m.jnt_qposadr[obj]]; continue; }   // mjSENS_JOINTPOS
    if (type != 4 && type != 5) { out[adr] = 0.f; continue; }              // (touch sensors: rg_kernel.h's sensor instantiation; not needed by this path)
    const int body = m.site_bodyid[obj];
    float cf[6] = {0, 0, 0, 0, 0, 0};
    for (int q = m.b_subtree_adr[body]; q < m.b_subtree_adr[body + 1]; q++) { const float* f = cfrc + 6 * m.b_subtree[q]; for (int c = 0; c < 6; c++) cf[c] += f[c]; }
    const v3 frc = mk3(cf[3], cf[4], cf[5]), off = ld3(SC(SPOS) + 3 * obj) - ld3(rootcom + 3 * m.body_rootid[body]);
    const v3 v = type == 4 ? frc : mk3(cf[0], cf[1], cf[2]) - cross(off, frc);
    q4 sq = qmul(ldq(SC(XQUAT) + 4 * body), ldq(m.site_quat + 4 * obj)); sq.x = -sq.x; sq.y = -sq.y; sq.z = -sq.z;
    st3(out + adr, qrot(sq, v));
  }
  BSYNC();
}

RB_STAGE void sb_position(RbCtx c) { RB_STAGE_ENTER(); rb_kinematics(m, s, S); rb_com_pos(m, s, S); }
RB_STAGE void sb_tendon(RbCtx c) { RB_STAGE_ENTER(); rb_tendon(m, s, S); }
RB_STAGE void sb_crb(RbCtx c) { RB_STAGE_ENTER(); rb_crb(m, s, S); }
RB_STAGE void sb_velocity(RbCtx c) { RB_STAGE_ENTER(); rb_velocity(m, s, S); }
RB_STAGE void sb_collision(RbCtx c, int flags) {   // equality pseudo-contacts + broadphase
  RB_STAGE_ENTER(); RbLRef L = RB_L(c);
  const int e = s.env;
  const float* eqd = m.neq > 0 ? L.bt.eq_data + (size_t)e * 7 * m.neq : (const float*)0;
  const int* eqa = m.neq > 0 ? L.bt.eq_active + (size_t)e * m.neq : (const int*)0;
  rb_equality(m, s, S, eqd, eqa);
  rb_broadphase(m, s, S, flags);
}
RB_STAGE void sb_narrow_convex(RbCtx c, int flags) { RB_STAGE_ENTER(); rb_narrow_convex(m, s, S, flags); }
RB_STAGE void sb_narrow_box(RbCtx c, int flags) { RB_STAGE_ENTER(); rb_narrow_box(m, s, S, flags); }
RB_STAGE void sb_rows(RbCtx c) {
  RB_STAGE_ENTER(); RbLRef L = RB_L(c);
  const float* eqd = m.neq > 0 ? L.bt.eq_data + (size_t)s.env * 7 * m.neq : (const float*)0;
  rb_make_constraint(m, s, S, eqd);
}
RB_STAGE void sb_pid(RbCtx c, int apply) { RB_STAGE_ENTER(); rb_pid(m, s, S, apply != 0); }
RB_STAGE void sb_smooth(RbCtx c, int flags) { RB_STAGE_ENTER(); rb_dof_contact_lists(m, s, S); rb_pid(m, s, S, true); sv_M_solve(c, 0, flags); }
RB_STAGE int sb_solve(RbCtx c, int flags) { RB_STAGE_ENTER(); return rb_solve(c, m, s, S, flags); }
RB_STAGE void sb_euler(RbCtx c, int flags) { RB_STAGE_ENTER(); rb_euler(c, m, s, S, flags); }
RB_STAGE void sb_sensors(RbCtx c) { RB_STAGE_ENTER(); RbLRef L = RB_L(c); rb_sensors(m, s, S, L.bt.sensordata + (size_t)s.env * m.nsensordata); }

// The kernel's body for env `e` of the batch described by the launch record at `klp` (in the kernel argument segment) and the model `mp`: one batch per launch
// (rb_step_kernel) or several batches of several models in ONE launch (rb_step_multi_kernel: heterogeneous object sets side by side at full occupancy instead of one
// launch chain per set on its own stream, which the GPU runs mostly one after the other -- profiles/r05_ycb_object_sets.txt).
#ifdef RG_EMUL
typedef const RbLaunch* RbKlp;
#define RB_MAKE_CTX() const RbModelDev& m = *mp; const RbLaunch& L = *klp
#else
typedef const RG_AS4 char* RbKlp;
#define RB_MAKE_CTX() RbM m = *(const RG_AS4 RbModelDev*)rg_uniform(mp); RbLRef L = *(const RG_AS4 RbLaunch*)klp
#endif
__device__ __forceinline__ void rb_step_body(const RbModelDev* mp, RbKlp klp, const int e) {
  RB_MAKE_CTX();
  RbLds& s = RB_S();
  if (e >= L.bt.B) return;
  if (L.bt.active && !L.bt.active[e]) return;
  float* S = L.bt.scratch + (size_t)e * m.scratch_words;
  const RbCtx c{(const void*)mp, (const void*)klp, S};
  const int nv = m.nv, nq = m.nq, nu = m.nu, flags = L.flags;
  BFOR(i, nq) s.qpos[i] = L.bt.qpos[(size_t)e * nq + i];
  BFOR(i, nv) { s.qvel[i] = L.bt.qvel[(size_t)e * nv + i]; s.warm[i] = L.bt.qacc_warmstart[(size_t)e * nv + i]; }
  BFOR(i, 3 * nu) s.pid[i] = L.bt.pid[(size_t)e * 3 * nu + i];
  if (TID == 0) { s.status = L.bt.status[e]; s.stop = 0; s.neqcon = 0; s.time = L.bt.time[e]; s.env = e; }
  if (TID < 16) s.prof[TID] = 0.f;
  if (TID < 7 * m.nmocap && TID < 14) s.mocap[TID] = L.bt.mocap[(size_t)e * 7 * m.nmocap + TID];
  const float* eqd = m.neq > 0 ? L.bt.eq_data + (size_t)e * 7 * m.neq : (const float*)0;
  const int* eqa = m.neq > 0 ? L.bt.eq_active + (size_t)e * m.neq : (const int*)0;
  BSYNC();
  // ---- action -> ctrl (robot_interface.py:247-278 with the hand's position -> control matrix)
  bool use_action = L.bt.action != 0 && !(L.bt.hold && L.bt.hold[e]);
  if (use_action) {
    float nf = 0; BFOR(u, nu) nf += (fabsf(L.bt.action[(size_t)e * nu + u]) <= 3.0e38f) ? 0.f : 1.f;
    if (rb_sum(s, nf) > 0) { use_action = false; if (TID == 0) s.status |= RG_STATUS_BAD_ACTION; }
  }
  BFOR(u, nu) {
    if (use_action) {
      const float lo = PRM(actuator_ctrlrange, RB_P_ACT_CTRLRANGE)[2 * u], hi = PRM(actuator_ctrlrange, RB_P_ACT_CTRLRANGE)[2 * u + 1];
      float centre, half = 0.5f * (hi - lo);
      const bool from_ctrl = (L.env.ctrl_centre_mask >> u) & 1u;
      if (L.env.relative_action) {
        if (from_ctrl) centre = L.bt.ctrl[(size_t)e * nu + u];
        else {
          centre = 0; for (int j = 0; j < L.env.n_hand_jnt; j++) centre += L.env.pos_to_ctrl[u * L.env.n_hand_jnt + j] * s.qpos[L.env.hand_qposadr + j];
          if (L.env.max_position_change > 0.f) half = fminf(half, L.env.max_position_change);   // Robot.actuation_range (robot_interface.py:220-231)
        }
      } else centre = 0.5f * (hi + lo);
      s.ctrl[u] = clampf(centre + clampf(L.bt.action[(size_t)e * nu + u], -1.f, 1.f) * half, lo, hi);
    } else s.ctrl[u] = L.bt.ctrl[(size_t)e * nu + u];
  }
  BSYNC();
  // ---- the TCP solver hook (RbTcpHook): arm joints <- main simulation, forward(), mocap target <- TCP pose + denormalised action
  float tcp_grip_action = 0.f;
  const bool tcp_on = L.tcp.enabled && !(L.tcp.skip && L.tcp.skip[e] != 0);
  if (tcp_on) {
    if (L.tcp.sync) { if (TID < 6) s.qpos[L.tcp.arm_q[TID]] = L.tcp.main_qpos[(size_t)e * L.tcp.main_nq + L.tcp.main_arm_q[TID]]; }
    BSYNC();
    sb_position(c); sb_tendon(c);
    if (L.tcp.sync) sb_pid(c, 0);   // the controller tick of sync_to's mj_forward (free_dof_tcp_arm.py:214-225)
    // the action that reaches the env: as given, or bin index -> value (DiscretizeActionWrapper.action, wrappers/util.py:66-70) -> exponential
    // moving average with bias correction (SmoothActionWrapper.step / IncrementalExpAvg, util.py:142-160, 213-218)
    const bool scripted = L.tcp.hold && L.tcp.hold[e] != 0;
    if (TID < 6 && scripted) {
      const float a = L.tcp.scripted[(size_t)e * 6 + TID];
      if (L.tcp.action_out) L.tcp.action_out[(size_t)e * 6 + TID] = a;
      s.prow[TID] = a;
    }
    if (TID < 6 && !scripted) {
      float a = L.tcp.action ? L.tcp.action[(size_t)e * 6 + TID] : 0.f;
      if (L.tcp.action_index) {
        int ix = L.tcp.action_index[(size_t)e * 6 + TID];
        ix = ix < 0 ? 0 : (ix >= L.tcp.nbins ? L.tcp.nbins - 1 : ix);
        a = L.tcp.bins[TID * L.tcp.nbins + ix];
      }
      if (L.tcp.ema_value) {
        const float al = L.tcp.ema_alpha;
        const float v = L.tcp.ema_value[(size_t)e * 6 + TID] * al + (1.f - al) * a;
        const int t = L.tcp.ema_t[e] + 1;
        L.tcp.ema_value[(size_t)e * 6 + TID] = v;
        a = v / (1.f - powf(al, (float)t));
      }
      if (L.tcp.action_out) L.tcp.action_out[(size_t)e * 6 + TID] = a;
      s.prow[TID] = a;
    }
    BSYNC();
    tcp_grip_action = s.prow[5];
    if (TID == 0 && L.tcp.ema_value && !scripted) L.tcp.ema_t[e] += 1;
    if (TID == 0) {
      const float* a = s.prow;
      const float mpc = L.tcp.max_position_change;
      const float a0 = clampf(a[0], -1.f, 1.f), a1 = clampf(a[1], -1.f, 1.f), a2 = clampf(a[2], -1.f, 1.f), a3 = clampf(a[3], -1.f, 1.f), a4 = clampf(a[4], -1.f, 1.f);
      const float roll = L.tcp.wrist_only ? 0.f : a3 * L.tcp.speed[0] * mpc;
      const float q6 = s.qpos[L.tcp.arm_q[5]], lo = PRM(jnt_range, RB_P_JNT_RANGE)[2 * L.tcp.wrist_jnt], hi = PRM(jnt_range, RB_P_JNT_RANGE)[2 * L.tcp.wrist_jnt + 1];
      const float pitch = clampf(a4 * L.tcp.speed[1] * mpc, lo + L.tcp.drift_threshold - q6, hi - L.tcp.drift_threshold - q6);   // FreeDOFTcpArm.constrain_quat_ctrl
      // MocapSolver.get_tcp_quat: euler = (roll, 0, pitch dimension) -> qx(roll) * qz(.), applied on the right of the TCP's orientation; mocap_set_action
      // adds the DIFFERENCE to the mocap quaternion, which reset_mocap2body_xpos has just set to the TCP's own pose
      const float cr = cosf(0.5f * roll), sr = sinf(0.5f * roll), cp = cosf(0.5f * pitch), sp = sinf(0.5f * pitch);
      q4 eq; eq.w = cr * cp; eq.x = sr * cp; eq.y = -sr * sp; eq.z = cr * sp;
      const q4 gq = ldq(SC(XQUAT) + 4 * L.tcp.tcp_body);
      q4 tq = qmul(gq, eq);
      if (L.tcp.wrist_only) {
        // MocapSolver.align_axis(cmd, PITCH = world z) (mocap_solver.py:58-74): of the commanded frame's axes the one closest to z (by |dot|), made to point up,
        // is rotated onto z along the shortest arc (rotation.vectors2quat; its antiparallel branch cannot occur: the dot product is >= 1 / sqrt 3)
        float R[9]; q2mat(R, tq);
        int k = 0; if (fabsf(R[7]) > fabsf(R[6 + k])) k = 1; if (fabsf(R[8]) > fabsf(R[6 + k])) k = 2;      // argmax |row 2| (first maximum, as numpy)
        const float sg = R[6 + k] > 0.f ? 1.f : (R[6 + k] < 0.f ? -1.f : 0.f);
        const float ax = sg * R[k], ay = sg * R[3 + k], az = sg * R[6 + k];
        // vectors2quat(axis, ez): w = |axis| |ez| + axis . ez, xyz = axis x ez = (ay, -ax, 0)
        float w = sqrtf(ax * ax + ay * ay + az * az) + az, x = ay, y = -ax;
        const float n = sqrtf(w * w + x * x + y * y);
        w /= n; x /= n; y /= n;
        if (w < 0.f) { w = -w; x = -x; y = -y; }       // quat_normalize
        q4 d; d.w = w; d.x = x; d.y = y; d.z = 0.f;
        tq = qmul(d, tq);
      }
      const v3 tp = ld3(SC(XPOS) + 3 * L.tcp.tcp_body);
      s.mocap[0] = tp.x + a0 * mpc; s.mocap[1] = tp.y + a1 * mpc; s.mocap[2] = tp.z + a2 * mpc;
      s.mocap[3] = gq.w + (tq.w - gq.w); s.mocap[4] = gq.x + (tq.x - gq.x); s.mocap[5] = gq.y + (tq.y - gq.y); s.mocap[6] = gq.z + (tq.z - gq.z);
      if (L.tcp.self_world) {   // MujocoRobotiqGripper.set_position_control on THIS world, before its mj_steps (CompositeRobot.set_position_control, then mujoco_simulation.step())
        float* gc = s.ctrl + L.tcp.main_grip_act;
        *gc = clampf(*gc + clampf(tcp_grip_action, -1.f, 1.f) * 0.5f * (L.tcp.grip_hi - L.tcp.grip_lo), L.tcp.grip_lo, L.tcp.grip_hi);
      }
    }
    BSYNC();
  }
  float st_ncon = 0, st_nefc = 0, st_iter = 0; int nsub_done = 0;
  for (int sub = 0; sub < L.nsubsteps; sub++) {
    float bd = 0; BFOR(i, nq) bd += (fabsf(s.qpos[i]) < 1e10f) ? 0.f : 1.f; BFOR(i, nv) bd += (fabsf(s.qvel[i]) < 1e10f) ? 0.f : 1.f;
    if (rb_sum(s, bd) > 0) { if (TID == 0) s.status |= RG_STATUS_BAD_STATE; break; }
    long long tp0 = rg_clock(), tp1;
#define RB_PROF(k) do { if (flags & 2) { BSYNC(); tp1 = rg_clock(); if (TID == 0) s.prof[k] += (float)(tp1 - tp0); tp0 = tp1; } } while (0)
    sb_position(c); RB_PROF(0);
    sb_tendon(c); sb_crb(c); RB_PROF(1);
    sb_velocity(c); RB_PROF(2);
    sb_collision(c, flags); sb_narrow_convex(c, flags); sb_narrow_box(c, flags); RB_PROF(3);
    sb_rows(c); RB_PROF(4);
    sb_smooth(c, flags);
    if ((flags & 1) && sub == 0) {   // stage dump of the first mj_step: ncon, nefc (the arrays themselves are read from the scratch row)
      if (TID == 0) { SC(DBG)[0] = (float)s.ncon; SC(DBG)[1] = (float)s.nefc; }
      BFOR(i, nv) { SC(DBG)[8 + i] = s.qfrc_bias[i]; SC(DBG)[8 + nv + i] = s.qfrc_passive[i]; SC(DBG)[8 + 2 * nv + i] = s.qfrc_act[i]; SC(DBG)[8 + 3 * nv + i] = s.qacc_smooth[i]; }
    }
    RB_PROF(5);
    const int iters = sb_solve(c, flags); RB_PROF(6);
    if ((flags & 1) && sub == 0) { if (TID == 0) SC(DBG)[2] = (float)iters; BFOR(i, nv) SC(DBG)[8 + 4 * nv + i] = s.qa[i]; }
    st_ncon += s.ncon; st_nefc += s.nefc; st_iter += iters; nsub_done++;
    bd = 0; BFOR(i, nv) bd += (fabsf(s.qa[i]) < 1e10f) ? 0.f : 1.f;
    if (rb_sum(s, bd) > 0) { if (TID == 0) s.status |= RG_STATUS_BAD_STATE; break; }
    BFOR(i, nv) s.warm[i] = s.qa[i];
    sb_euler(c, flags); RB_PROF(7);
    if (TID == 0) s.time += m.timestep;
    BSYNC();
  }
  if ((flags & 2) && TID < 16) SC(DBG)[8 + 5 * nv + TID] = s.prof[TID];   // stage cycle counters: frames+com, tendon+crb, velocity, collision, rows, smooth, Newton, Euler
  // ---- the state-less forward() calls of the reference: only their PID side effect touches the state
  const int nticks = L.bt.nticks ? L.bt.nticks[e] : L.nforward_ticks;
  const bool full_forward = (flags & 32) && nticks > 0;   // bit 5: the last state-less forward runs in full (mj_forward) and the sensors are read from it
  if (nticks > 0) {
    sb_position(c); sb_tendon(c);
    for (int k = 0; k < nticks - (full_forward ? 1 : 0); k++) sb_pid(c, 0);
  }
  if (full_forward) {
    sb_crb(c); sb_velocity(c);
    sb_collision(c, flags); sb_narrow_convex(c, flags); sb_narrow_box(c, flags);
    sb_rows(c);
    sb_smooth(c, flags);
    sb_solve(c, flags);
    if (m.nsensor > 0 && L.bt.sensordata) sb_sensors(c);
    if (TID == 0) { SC(DBG)[3] = (float)s.ncon; SC(DBG)[4] = (float)s.nefc; SC(DBG)[5] = (float)s.neqcon; }   // what the env kernel's contact scans read
  }
  if (tcp_on && !L.tcp.self_world) {
    // JointControlledArm.set_position_control: main ctrl[:6] <- the solver's joint angles; MujocoRobotiqGripper: a relative target around its current ctrl
    if (TID < 6) L.tcp.main_ctrl[(size_t)e * L.tcp.main_nu + TID] = s.qpos[L.tcp.arm_q[TID]];
    if (TID == 6) {
      float* gc = L.tcp.main_ctrl + (size_t)e * L.tcp.main_nu + L.tcp.main_grip_act;
      *gc = clampf(*gc + clampf(tcp_grip_action, -1.f, 1.f) * 0.5f * (L.tcp.grip_hi - L.tcp.grip_lo), L.tcp.grip_lo, L.tcp.grip_hi);
    }
  }
#ifdef RB_LDS_ARENA
  if (m.lds_words > 0) {   // LDS-resident stage arrays -> the env's scratch row (what the env kernel and the host's stage readers see)
    BSYNC();
#pragma unroll 1
    for (int k = 0; k < RB_NOFF; k++) {
      const int lo = m.lds_off[k];
      if (lo < 0) continue;
      const float* src = RB_ARENA() + lo; float* dst = S + m.off[k];
      BFOR(w, m.lds_len[k]) dst[w] = src[w];
    }
  }
#endif
  if (TID < 7 * m.nmocap && TID < 14) L.bt.mocap[(size_t)e * 7 * m.nmocap + TID] = s.mocap[TID];
  BFOR(i, nq) L.bt.qpos[(size_t)e * nq + i] = s.qpos[i];
  BFOR(i, nv) { L.bt.qvel[(size_t)e * nv + i] = s.qvel[i]; L.bt.qacc_warmstart[(size_t)e * nv + i] = s.warm[i]; }
  BFOR(i, 3 * nu) L.bt.pid[(size_t)e * 3 * nu + i] = s.pid[i];
  BFOR(u, nu) L.bt.ctrl[(size_t)e * nu + u] = s.ctrl[u];
  if (TID == 0) {
    L.bt.status[e] = s.status; L.bt.time[e] += nsub_done * m.timestep;
    float* st = L.bt.stats + 4 * (size_t)e; st[0] += st_ncon; st[1] += st_nefc; st[2] += st_iter; st[3] += nsub_done;
  }
}
__global__ void __launch_bounds__(RB_T, RB_WG_PER_CU) rb_step_kernel(const RbModelDev* mp, RbLaunch launch) {
#ifdef RG_EMUL
  rb_step_body(mp, &launch, (int)blockIdx.x);
#else
  rb_step_body(mp, (const RG_AS4 char*)__builtin_amdgcn_kernarg_segment_ptr() + 8, (int)blockIdx.x);
#endif
}
#if RB_NWAVE == 1
// several batches (same size, same kernel configuration, each with its own model) in one launch: workgroup k steps env k % group_size of batch k / group_size
__global__ void __launch_bounds__(RB_T, RB_WG_PER_CU) rb_step_multi_kernel(RbMultiLaunch ml) {
#ifdef RG_EMUL
  const int g = (int)blockIdx.x / ml.group_size, e = (int)blockIdx.x - g * ml.group_size;
  if (g >= ml.n) return;
  rb_step_body(ml.m[g], &ml.L[g], e);
#else
  const RG_AS4 char* ka = (const RG_AS4 char*)__builtin_amdgcn_kernarg_segment_ptr();
  const RG_AS4 RbMultiLaunch& K = *(const RG_AS4 RbMultiLaunch*)ka;
  const int g = (int)blockIdx.x / K.group_size, e = (int)blockIdx.x - g * K.group_size;
  if (g >= K.n) return;
  rb_step_body(K.m[g], ka + offsetof(RbMultiLaunch, L) + (size_t)g * sizeof(RbLaunch), e);
#endif
}
#endif
}  // namespace rgb
