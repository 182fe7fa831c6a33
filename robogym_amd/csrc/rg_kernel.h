// rg_kernel.h — the batched env.step kernel for gfx950 (MI355X).
//
// Execution plan: ONE 64-lane wavefront (= one workgroup) per environment.  The whole env step —
// action -> ctrl, nsubsteps x mj_step (kinematics, tendons, CRB into per-tree dense blocks, velocities,
// collision = cached distance bounds + boxes + MPR, constraint rows, PID actuation, blocked Cholesky,
// Newton solve, implicit-damping Euler), the three PID
// ticks of the reference's state-less forward() calls, observation readout and goal distance —
// runs inside one launch with all per-env state resident in LDS.  HBM traffic per env step is the
// state row in, state row + observation row out (row-major [B][n] buffers, contiguous per env so
// a wave's loads are coalesced) plus the stream of per-pair collision caches (pairlb, sepdir).  Tree recursions are level sweeps over precomputed level lists,
// reductions are DPP row shifts/broadcasts, sparse scatters are LDS atomics issued by a single wave
// (in-order, hence deterministic).  No MFMA: this is sparse articulated dynamics.
//
// Replaces, for the hot path, mujoco_py.MjSim.step + mjpid (reference call sites:
// /root/reference/robogym/mujoco/simulation_interface.py:176-189, :86-88) and the readout done by
// /root/reference/robogym/robot_env.py:804-844.  Semantics follow the CPU oracle stage by stage.
#ifndef RG_KERNEL_COMMON_H
#define RG_KERNEL_COMMON_H
#include "rg_types.h"

#ifdef RG_EMUL
#include "hip_emul.h"
#else
#include <hip/hip_runtime.h>
#endif

#ifdef RG_EMUL
typedef const RgModelDev& RgM;
#else
// the model descriptor lives in constant address space: every field read is a scalar load that the
// compiler may re-issue next to its use instead of keeping ~240 SGPRs of pointers alive
typedef const __attribute__((address_space(4))) RgModelDev& RgM;
#endif
#ifdef RG_FINE_PROF
#define RG_NPROF 56   /* analysis build (-DRG_FINE_PROF): sub-stage counters inside the Newton loop, slots 24.. (tools/stage_profile.py prints them) */
#else
#define RG_NPROF 24
#endif
struct RgAux {  // extra static tables (kept out of RgModelDev to keep the kernarg small)
  const int *subtree_adr, *subtree;
  const uint32_t* dof_velmask;
};

// ------------------------------------------------------------------------------------------------- launch context
// Everything a launch passes besides the model, as ONE by-value kernel argument that is read through the
// constant address space (the kernarg segment), so that its ~100 scalars are loaded where they are used
// instead of living in (or being spilled from) SGPRs for the whole kernel.
struct RgLaunch { RgAux x; RgEnvDev env; RgBatchDev bt; int nsubsteps, nforward_ticks, flags, nqueues; };

// The substep is a sequence of REAL function calls (not inlined): each stage gets its own register
// allocation, so loop invariants of one stage are not kept alive (or spilled) through all the others.
// A stage finds the model, the launch descriptor and the env's LDS image by itself: the kernarg segment
// pointer is an implicit argument of every device function, and the LDS image is the workgroup's only
// dynamic shared allocation.
struct RgCtx { const void* km; const void* kl; };   // device address of the model descriptor, address of the launch descriptor
#ifdef RG_EMUL
#define RG_M(c) (*(const RgModelDev*)(c).km)
#define RG_L(c) (*(const RgLaunch*)(c).kl)
#define RG_S() (*(RgLds*)emul_lds())
#define RG_STAGE static inline
typedef const RgLaunch& RgLRef;
#else
extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds_raw[];
#define RG_AS4 __attribute__((address_space(4)))
// function arguments arrive in VGPRs: read the (wave-uniform) addresses back into SGPRs so that everything
// loaded through them is a scalar load again
__device__ __forceinline__ unsigned long long rg_uniform(const void* p) {
  unsigned long long v = (unsigned long long)p;
  unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
#define RG_M(c) (*(const RG_AS4 RgModelDev*)rg_uniform((c).km))
#define RG_L(c) (*(const RG_AS4 RgLaunch*)rg_uniform((c).kl))
#define RG_S() (*(RgLds*)rg_lds_raw)
#ifdef RG_INLINE_STAGES
#define RG_STAGE __device__ __forceinline__
#else
#define RG_STAGE __device__ __attribute__((noinline))
#endif
typedef const RG_AS4 RgLaunch& RgLRef;
#endif
#ifdef RG_EMUL
#define RG_STAGE_BIG static inline
#else
#ifdef RG_BIG_STAGES_AS_CALLS
#define RG_STAGE_BIG __device__ __attribute__((noinline))
#else
#define RG_STAGE_BIG __device__ __forceinline__   /* the register-hungriest stages stay in the kernel body: a call would save/restore ~30 callee-saved VGPRs through scratch */
#endif
#endif
// env handled by this workgroup: the launch may carry a permutation (longest-expected-first dispatch order)
#ifdef RG_EMUL
static inline int rg_env_blk(RgLRef L) { return L.bt.order ? L.bt.order[blockIdx.x] : (int)blockIdx.x; }
#else
__device__ __forceinline__ int rg_env_blk(RgLRef L) { return L.bt.order ? __builtin_amdgcn_readfirstlane(L.bt.order[blockIdx.x]) : (int)blockIdx.x; }
#endif
// (rg_env / rg_prm, the env of this workgroup and its row of model parameters, are defined per kernel configuration below)
// ------------------------------------------------------------------------------------------------- small math
struct alignas(16) rgf4 { float x, y, z, w; };
struct v3 { float x, y, z; };
__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// hardware reciprocal / reciprocal square root (v_rcp_f32 / v_rsq_f32, 1 ulp): the IEEE-exact sqrtf and '/' expand to
// ~10 dependent instructions each, which is what the pivot chains of the factorisations would otherwise spend their time on
#ifdef RG_EMUL
static inline float rg_rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline float rg_rcp(float x) { return 1.0f / x; }
static inline float rg_sqrt(float x) { return sqrtf(x); }
#else
__device__ __forceinline__ float rg_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rg_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rg_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#endif
__device__ __forceinline__ float norm(v3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ v3 normalized(v3 a) { float n2 = dot(a, a); return n2 < 1e-37f ? mk3(1, 0, 0) : a * rg_rsqrt(n2); }
// M row-major 3x3
__device__ __forceinline__ v3 mulm(const float* M, v3 v) { return mk3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z); }
__device__ __forceinline__ v3 mulmT(const float* M, v3 v) { return mk3(M[0] * v.x + M[3] * v.y + M[6] * v.z, M[1] * v.x + M[4] * v.y + M[7] * v.z, M[2] * v.x + M[5] * v.y + M[8] * v.z); }
struct q4 { float w, x, y, z; };
__device__ __forceinline__ q4 ldq(const float* p) { q4 q; q.w = p[0]; q.x = p[1]; q.y = p[2]; q.z = p[3]; return q; }
__device__ __forceinline__ void stq(float* p, q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ q4 qmul(q4 a, q4 b) {
  q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}
__device__ __forceinline__ q4 qnormalize(q4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-30f) { q.w = 1; q.x = q.y = q.z = 0; return q; }
  float s = rg_rcp(n); q.w *= s; q.x *= s; q.y *= s; q.z *= s; return q;
}
__device__ __forceinline__ void q2mat(float* m, q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
// v rotated by the unit quaternion q (and by its inverse): v + w t + u x t with t = 2 u x v.  Frames are kept as
// quaternions in LDS (4 words instead of the 9 of a rotation matrix: LDS per env is what bounds the envs in flight)
__device__ __forceinline__ v3 qrot(q4 q, v3 v) { v3 u = mk3(q.x, q.y, q.z), t = cross(u, v) * 2.0f; return v + t * q.w + cross(u, t); }
__device__ __forceinline__ v3 qrotT(q4 q, v3 v) { v3 u = mk3(q.x, q.y, q.z), t = cross(u, v) * 2.0f; return v - t * q.w + cross(u, t); }
// sin / cos of a hinge half-angle: |x| <= 1.5 covers every joint range with margin; there the Taylor
// polynomials below are exact to fp32 rounding (remainders x^13/13! < 3e-8 relative, x^14/14! < 4e-9) and cost
// a dozen FMAs instead of libm's range-reduced sincosf
__device__ __forceinline__ void rg_sincos(float x, float& sn, float& cs) {
  if (fabsf(x) > 1.5f) { sincosf(x, &sn, &cs); return; }
  float x2 = x * x;
  sn = x * (1.f + x2 * (-1.6666667e-1f + x2 * (8.3333333e-3f + x2 * (-1.9841270e-4f + x2 * (2.7557319e-6f + x2 * (-2.5052108e-8f))))));
  cs = 1.f + x2 * (-0.5f + x2 * (4.1666667e-2f + x2 * (-1.3888889e-3f + x2 * (2.4801587e-5f + x2 * (-2.7557319e-7f + x2 * 2.0876757e-9f)))));
}
__device__ __forceinline__ q4 axisangle(v3 ax, float ang) {
  float s, c; rg_sincos(0.5f * ang, s, c);
  q4 q; q.w = c; q.x = ax.x * s; q.y = ax.y * s; q.z = ax.z * s; return q;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ------------------------------------------------------------------------------------------------- wave collectives
// gfx950: reductions run on the VALU with DPP row shifts / row broadcasts (no LDS crossbar round trips);
// the result is read back from lane 63 with v_readlane, i.e. it is wave-uniform (an SGPR).
#ifdef RG_EMUL
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_min_i(int v) { for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
__device__ __forceinline__ float lane_bcast(float v, int src) { return __shfl(v, src); }
#else
template <int CTRL, int RMASK> __device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false));
}
template <int CTRL, int RMASK> __device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RMASK, 0xf, false); }
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0x111, 0xf>(0.f, v); v += dpp_f<0x112, 0xf>(0.f, v); v += dpp_f<0x114, 0xf>(0.f, v); v += dpp_f<0x118, 0xf>(0.f, v);
  v += dpp_f<0x142, 0xa>(0.f, v); v += dpp_f<0x143, 0xc>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -3.0e38f;
  v = fmaxf(v, dpp_f<0x111, 0xf>(ninf, v)); v = fmaxf(v, dpp_f<0x112, 0xf>(ninf, v)); v = fmaxf(v, dpp_f<0x114, 0xf>(ninf, v)); v = fmaxf(v, dpp_f<0x118, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(ninf, v)); v = fmaxf(v, dpp_f<0x143, 0xc>(ninf, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int wave_min_i(int v) {
  const int big = 0x7fffffff;
  v = imin(v, dpp_i<0x111, 0xf>(big, v)); v = imin(v, dpp_i<0x112, 0xf>(big, v)); v = imin(v, dpp_i<0x114, 0xf>(big, v)); v = imin(v, dpp_i<0x118, 0xf>(big, v));
  v = imin(v, dpp_i<0x142, 0xa>(big, v)); v = imin(v, dpp_i<0x143, 0xc>(big, v));
  return __builtin_amdgcn_readlane(v, 63);
}
// value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ float lane_bcast(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
#endif
// all-reduce inside a group of G = 16 lanes (one DPP row: row rotations), G = 8 lanes (half a row: two quad
// permutes and the half-row mirror) or G = 4 lanes (a quad: two quad permutes); every lane of the group ends up
// with the result.  Only the
// group has to be convergent.
#ifdef RG_EMUL
template <int G> __device__ __forceinline__ float grp_max(float v) { for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
template <int G> __device__ __forceinline__ int grp_min_i(int v) { for (int o = G / 2; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
#else
template <int G> __device__ __forceinline__ float grp_max(float v) {
  if (G == 16) { v = fmaxf(v, dpp_f<0x128, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x124, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x122, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x121, 0xf>(v, v)); }
  else { v = fmaxf(v, dpp_f<0xB1, 0xf>(v, v)); if (G >= 4) v = fmaxf(v, dpp_f<0x4E, 0xf>(v, v)); if (G == 8) v = fmaxf(v, dpp_f<0x141, 0xf>(v, v)); }
  return v;
}
template <int G> __device__ __forceinline__ int grp_min_i(int v) {
  if (G == 16) { v = imin(v, dpp_i<0x128, 0xf>(v, v)); v = imin(v, dpp_i<0x124, 0xf>(v, v)); v = imin(v, dpp_i<0x122, 0xf>(v, v)); v = imin(v, dpp_i<0x121, 0xf>(v, v)); }
  else { v = imin(v, dpp_i<0xB1, 0xf>(v, v)); if (G >= 4) v = imin(v, dpp_i<0x4E, 0xf>(v, v)); if (G == 8) v = imin(v, dpp_i<0x141, 0xf>(v, v)); }
  return v;
}
#endif
// sum over a group of 16 lanes (one DPP row); every lane of the group ends up with the total of all 16 (on the device each lane adds them
// in its own rotation order: read the result from ONE designated lane where bit-identical copies matter)
#ifdef RG_EMUL
__device__ __forceinline__ float grp_sum16(float v) { for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
#else
__device__ __forceinline__ float grp_sum16(float v) { v += dpp_f<0x128, 0xf>(v, v); v += dpp_f<0x124, 0xf>(v, v); v += dpp_f<0x122, 0xf>(v, v); v += dpp_f<0x121, 0xf>(v, v); return v; }
#endif
// ---- the matrix pipe, used by the dense Newton step (rg_chol_mfma_n).  One 32 x 32 f32 accumulator tile = 16 registers per lane: lane l holds
// column l % 32, register r holds row 8 (r / 4) + 4 (l / 32) + r % 4.  rg_mfma32: acc += A B with A = 32 x 2 (lane l: A[l % 32][l / 32]) and
// B = 2 x 32 (lane l: B[l / 32][l % 32]) -- v_mfma_f32_32x32x2_f32, exact f32 (a chain of fused multiply-adds over k).
// rg_halves<H>(a, b): lanes 0..31 get half H of a, lanes 32..63 get half H of b (H = 0: lanes 0..31, H = 1: lanes 32..63): v_permlane32_swap.
#ifdef RG_EMUL
typedef float rgacc __attribute__((vector_size(64)));
static inline void rg_mfma32(float a, float b, rgacc& acc) {
  const int l = (int)(threadIdx.x & 63), col = l & 31, hi = l >> 5; const unsigned w0 = threadIdx.x & ~63u;
  float ab[2] = {a, b}; uint64_t raw; memcpy(&raw, ab, 8);
  emul_xchg[threadIdx.x] = raw;          // one exchange per instruction: every lane publishes (a, b), then reads what it needs
  emul_barrier(64);
  float av[32][2], bv[2];
  for (int k = 0; k < 2; k++) { float t[2]; memcpy(t, &emul_xchg[w0 + col + 32 * k], 8); bv[k] = t[1]; }
  for (int i = 0; i < 32; i++) for (int k = 0; k < 2; k++) { float t[2]; memcpy(t, &emul_xchg[w0 + i + 32 * k], 8); av[i][k] = t[0]; }
  emul_barrier(64);
  for (int r = 0; r < 16; r++) { const int i = 8 * (r >> 2) + 4 * hi + (r & 3); acc[r] = fmaf(av[i][1], bv[1], fmaf(av[i][0], bv[0], acc[r])); }
}
template <int H> static inline float rg_halves(float a, float b) {
  const int l = (int)(threadIdx.x & 63);
  const float ax = __shfl(a, (l & 31) + 32 * H), bx = __shfl(b, (l & 31) + 32 * H);
  return l < 32 ? ax : bx;
}
#else
typedef float rgacc __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void rg_mfma32(float a, float b, rgacc& acc) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); }
template <int H> __device__ __forceinline__ float rg_halves(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, H ? r[1] : r[0]);
}
#endif
// arg-max with smallest-index tie break (matches a first-max serial scan)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
  float vm = wave_max(v);
  i = wave_min_i(v == vm ? i : 0x7fffffff);
  v = vm;
}
#ifdef RG_EMUL
static inline long long rg_clock() { return 0; }
#else
#ifdef RG_CLOCK_REALTIME   /* analysis builds (tools/tail_analysis.py): the constant-rate counter instead of the shader clock */
__device__ __forceinline__ long long rg_clock() { return (long long)__builtin_amdgcn_s_memrealtime(); }
#else
__device__ __forceinline__ long long rg_clock() { return (long long)__builtin_readcyclecounter(); }
#endif
#endif
// ---- cross-workgroup plumbing of the substep-granular dispatch (rg_step_items_kernel).  All work items of an env run on ONE
// XCD (queue = XCC id), so the XCD's L2 is the point of coherence: the producer's plain stores are in L2 once its vmcnt
// has drained (the vector L1 is write-through), the consumer reads everything a previous item wrote with sc1 loads (agent-scope
// relaxed atomics: they bypass the CU's L1, which is never refreshed by another CU's stores).  No cache-wide fence anywhere.
#ifdef RG_EMUL
static inline int rg_xcc_id() { return 0; }
static inline int rg_ld_sc1(const int* p) { return *p; }
static inline float rg_ld_sc1(const float* p) { return *p; }
static inline unsigned rg_ld_sc1(const unsigned* p) { return *p; }
static inline void rg_st_sc1(int* p, int v) { *p = v; }
static inline void rg_st_sc1(unsigned* p, unsigned v) { *p = v; }
static inline void rg_st_sc1(float* p, float v) { *p = v; }
static inline int rg_ticket(int* p) { int o = *p; *p = o + 1; return o; }
static inline void rg_drain_stores() {}
static inline void rg_pause() {}
static inline int rg_first(int v) { return __shfl(v, 0); }
#else
__device__ __forceinline__ int rg_xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15); }   // hwreg(HW_REG_XCC_ID, 0, 4)
__device__ __forceinline__ int rg_ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rg_ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rg_ld_sc1(const float* p) { return __builtin_bit_cast(float, __hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void rg_st_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rg_st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rg_st_sc1(float* p, float v) { __hip_atomic_store((int*)p, __builtin_bit_cast(int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int rg_ticket(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rg_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void rg_pause() { __builtin_amdgcn_s_sleep(16); }
__device__ __forceinline__ int rg_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
#define LANE ((int)threadIdx.x)
#define SYNC() __syncthreads()
#define PFOR(i, n) for (int i = LANE; i < (n); i += RG_WAVE)

// spatial algebra in the com-based frame: 6-vectors are [rotational(3); translational(3)]
__device__ __forceinline__ void mul_inert_vec(float* r, const float* i, const float* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float* r, const float* vel, const float* v) {
  v3 w = ld3(vel), vt = ld3(vel + 3), a = ld3(v), b = ld3(v + 3);
  st3(r, cross(w, a)); st3(r + 3, cross(w, b) + cross(vt, a));
}
__device__ __forceinline__ void cross_force(float* r, const float* vel, const float* f) {
  v3 w = ld3(vel), vt = ld3(vel + 3), a = ld3(f), b = ld3(f + 3);
  st3(r, cross(w, a) + cross(vt, b)); st3(r + 3, cross(w, b));
}

#if defined(RG_WOODBURY_CHECK) && defined(RG_EMUL)
static float rg_wchk_max = 0.f; static int rg_wchk_n = 0, rg_wchk_rows = 0;   // emulation-harness self check of rg_cholinv_woodbury (see rg_solve)
#endif
#endif  // RG_KERNEL_COMMON_H

// =================================================================================================================
// Everything below depends on the compiled contact / candidate capacities and is instantiated once per kernel
// CONFIGURATION: the includer defines RG_NS (namespace), RG_MAXCON, RG_CPOOL, RG_MAXCAND, RG_MAXCAND2 and includes this
// file again (rg_api.hip: "rollout" capacities sized for the hot path, "large" ones for the reset recipe, where a hand
// closing around a freshly dropped cube makes 2-3 x more contacts).  LDS per env, hence envs in flight per CU, is what
// the capacities buy.
// =================================================================================================================
#if !defined(RG_NS) || !defined(RG_MAXCON) || !defined(RG_CPOOL) || !defined(RG_MAXCAND) || !defined(RG_MAXCAND2)
#error "define RG_NS, RG_MAXCON, RG_CPOOL, RG_MAXCAND, RG_MAXCAND2 before including rg_kernel.h"
#endif
#ifndef RG_ITEMS
#define RG_ITEMS 0     /* 1: this configuration is the substep-granular one: rg_step_items_kernel instead of rg_step_kernel, rows that a previous
                          work item of the env may have written are read with L1-bypassing loads, the env comes from the work item */
#endif
#undef RG_ROW_LD
#undef RG_ROW_ST
#if RG_ITEMS
#define RG_ROW_LD(p) rg_ld_sc1(p)
#else
#define RG_ROW_LD(p) (*(p))
#endif
// stores of the rows a later work item of the env reads.  Default: plain stores (made visible in the XCD's L2 by rg_item_publish's vmcnt drain: the vector L1 is
// write-through).  -DRG_ROW_ST_SC1: agent-scope relaxed atomic stores, the formal counterpart of RG_ROW_LD (A/B: profiles/r05_ab_items_sc1_stores.txt).
#undef RG_SEPDIR_ST
#if RG_ITEMS && defined(RG_ROW_ST_SC1)
#define RG_ROW_ST(p, v) rg_st_sc1((p), (v))
#define RG_SEPDIR_ST(p, cd) do { float* sp__ = (float*)(p); rg_st_sc1(sp__, (cd).x); rg_st_sc1(sp__ + 1, (cd).y); rg_st_sc1(sp__ + 2, (cd).z); } while (0)
#else
#define RG_ROW_ST(p, v) (*(p) = (v))
#define RG_SEPDIR_ST(p, cd) (*(p) = (cd))
#endif
#ifndef RG_SETCONST
#define RG_SETCONST 0  /* 1: this configuration also carries rg_setconst_kernel (one instantiation is enough) */
#endif
#ifndef RG_SENSORS
#define RG_SENSORS 0   /* 1: this configuration evaluates data.sensordata (launch flag bit 5); its own instantiation so that the hot configurations carry none of it */
#endif
namespace RG_NS {
#define RG_MAXPYR (RG_MAXCON * 6)
#define RG_PSLOTS ((RG_MAXPYR + RG_WAVE - 1) / RG_WAVE)   // pyramid rows a lane owns during a line search

// Per-env LDS image (LDS capacity is what bounds the number of envs in flight per CU).
// Arrays whose lifetimes inside a substep do not overlap share storage:
//   pos:  position/velocity/collision scratch, itself overlaid by phase
//           slot A: kinematics frames (T1-T3) | composite inertias (T4) | velocity-stage vectors (T6-T8)
//           slot B: geom frames (until the narrowphase)
//           slot C: cinert (com_pos .. velocity) | candidate lists + raw contacts (collision .. constraint rows)
//   slv:  solver scratch — H holds, in turn, the per-tree block expansion of M and its L'DL factor, the Newton
//         Hessian (compact nvc x hs) and M + hB; all needed only after `pos` died.
// The joint-space inertia M persists in MuJoCo's tree-sparse form (one word per (dof, ancestor) pair, 149 for the hand
// models) and is expanded into H's block layout where a factorisation needs it.
struct RgLds {
  // state
  alignas(16) float qpos[RG_MAXNQ], qvel[RG_MAXNV], ctrl[RG_MAXU];   // (the PID controller state of actuator u and qacc_warmstart[d] live in lane u's / lane d's registers)
  // persistent through the substep
  float org[4 * 3];   // com-frame origin per kinematic tree (slot via b2org)
  float Msp[RG_MAXNM];  // M[i][j] for the model's (i, j = i or an ancestor of i) list (M_i, M_j)
  float tenlen[RG_MAXTEN], tenJ[RG_MAXTEN * 4];
  float actlen[RG_MAXU], actfrc[RG_MAXU];
  float qfrc_smooth[RG_MAXNV], qacc_smooth[RG_MAXNV];
  unsigned char ten_cdof[RG_MAXTEN * 4], c2d[RG_MAXNVC], b2org[RG_MAXBODY];
  int cblk[RG_MAXNVC];   // per compact dof: inertia-block row word | compact index of its tree start << 16 | tree size << 24
  int ncand, ncand2, ncon;
  float c_D[RG_MAXCON], c_mu[RG_MAXCON * 2];   // friction coefficient of the two sliding directions | of the spin direction
  short c_pair[RG_MAXCON], c_off[RG_MAXCON];
  unsigned char c_dim[RG_MAXCON], c_nnz[RG_MAXCON];
#if RG_SENSORS
  unsigned char c_touch[RG_MAXCON];   // bit k: touch sensor k sees this contact (sensor pass only)
#endif
  unsigned char c_idx[RG_MAXCON * RG_W];
  float c_pool[RG_CPOOL];  // basis Jacobian rows (normal, tangent1, tangent2, spin) x nnz, packed per contact
  float c_aref0[RG_MAXCON], c_kb[RG_MAXCON];   // reference acceleration of the contact's pyramid rows = aref0 - kb * (row velocity); the rows' lanes hold the result (RowRegs::paref)
  unsigned int status;
  int has_xfrc;   // any non-zero entry in the env's xfrc_applied row (checked once per launch)
#if RG_ITEMS
  int cur_env;    // the env of the work item this workgroup is on
#endif
  union {
    struct {  // ---- pos
      union {  // slot A
        struct { alignas(16) float xquat[RG_MAXBODY * 4]; float xpos[RG_MAXBODY * 3], xanchor[RG_MAXJNT * 3], xaxis[RG_MAXJNT * 3], spos[RG_MAXSITE * 3]; };
        struct { float crb[RG_MAXBODY * 10]; };
        struct { float cdofdot[RG_MAXNV * 6], cacc[RG_MAXBODY * 6], tenfrc[RG_MAXTEN], tenvel[RG_MAXTEN], qfrc_passive[RG_MAXNV], qfrc_bias[RG_MAXNV], qfrc_act[RG_MAXNV]; };   // (cacc: per-body forces, then their subtree sums in place)
      };
      alignas(16) float gquat[RG_MAXGEOM * 4]; float gpos[RG_MAXGEOM * 3];  // slot B: geom frames (orientation as a quaternion)
      float cdof[RG_MAXNV * 6];   // alive from com_pos to the constraint rows
      float gspeed[RG_MAXGEOM];  // bound on the speed of any point of the geom (velocity stage -> broadphase)
      union {  // slot C
        float cinert[RG_MAXBODY * 10];   // com_pos .. velocity stage
        struct {   // collision .. constraint rows
          short cand[RG_MAXCAND], cand2[RG_MAXCAND2];
          union {
            short tlist[RG_TLIST];   // broadphase only
            struct { float c_dist[RG_MAXCON], c_pos[RG_MAXCON * 3], c_normal[RG_MAXCON * 3]; };   // narrowphase .. constraint rows
          };
        };
      };
    };
    struct {  // ---- slv
      alignas(16) float H[RG_HWORDS];
      float a[RG_MAXNVC], jtf[RG_MAXNVC], Ma[RG_MAXNVC];
      union { struct { float as[RG_MAXNVC], fs[RG_MAXNVC]; }; float qfrc_con[RG_MAXNV]; };   // (qfrc_con / qacc: the solver's results in dof order, written when its
      union { struct { float search[RG_MAXNVC], Mv[RG_MAXNVC]; }; float qacc[RG_MAXNV]; };   //  compact-space vectors are dead; read by the integrator)
      float dinv[RG_MAXNV], tmpv[RG_MAXNV];
      unsigned char p_quad[RG_MAXPYR];
      // per contact basis (normal, t1, t2, spin): J x on the way to the pyramid rows | the rows' forces 0..3 on their way to J' f (rows 4, 5: c_aref0 / c_kb)
      float c_bdot[RG_MAXCON * 4];
    };
  };
  float prof[RG_NPROF];   // LAST: launches without the profiling flag do not allocate it (rg_lds_launch_bytes)
};
// dynamic LDS bytes of a launch
static inline size_t rg_lds_launch_bytes(bool profiling) { return profiling ? sizeof(RgLds) : offsetof(RgLds, prof); }

// the env this workgroup works on: from the block index (through the dispatch permutation), or -- substep-granular
// configuration -- from the work item it holds; and the env's row of model parameters (RG_PRM_* layout): its own if the
// batch carries per-env overrides, else the model's
#ifdef RG_EMUL
#if RG_ITEMS
static inline int rg_env(RgLRef L) { return RG_S().cur_env; }
#else
static inline int rg_env(RgLRef L) { return rg_env_blk(L); }
#endif
static inline const float* rg_prm(const RgModelDev& m, RgLRef L) { return L.bt.envprm ? L.bt.envprm + (size_t)rg_env(L) * RG_NPRM : m.prm_default; }
#else
#if RG_ITEMS
__device__ __forceinline__ int rg_env(RgLRef L) { return __builtin_amdgcn_readfirstlane(RG_S().cur_env); }
#else
__device__ __forceinline__ int rg_env(RgLRef L) { return rg_env_blk(L); }
#endif
__device__ __forceinline__ const float* rg_prm(const RG_AS4 RgModelDev& m, RgLRef L) { return L.bt.envprm ? L.bt.envprm + (size_t)rg_env(L) * RG_NPRM : m.prm_default; }
#endif
// ------------------------------------------------------------------------------------------------- position stage
// data.xipos of body b (com of the body in the world): not stored, its three readers derive it from the body frame
__device__ __forceinline__ v3 rg_xipos(RgM m, const RgLds& s, int b) { return ld3(s.xpos + 3 * b) + qrot(ldq(s.xquat + 4 * b), ld3(m.body_ipos + 3 * b)); }
__device__ __forceinline__ void rg_kinematics(RgM m, RgLds& s, const float* P) {
  // Pass A, one lane per moving body, all at once: the body's frame RELATIVE TO ITS PARENT (body offset, then its joints in
  // turn: engine_core_smooth.c mj_kinematics with the parent at the origin) and its joints' anchors / axes in that frame.
  // Pass B, one lane per moving body: compose the relative frames down the body's own chain of moving ancestors (a lane
  // recomputes its ancestors' world frames with the arithmetic their own lanes use, so parent and child agree to the bit):
  // 7 x (rotate + quaternion product) per lane instead of 7 level barriers with the joint trigonometry inside each.
  // (relative frames wait in the geom-frame area, which is written only at the end of this stage)
  float* lp = s.gpos; float* lq = s.gquat;
  PFOR(b, m.nbody) {
    if ((m.body_depth[b] & 255) == 0) {   // static body (the world included): its constant frame
      st3(s.xpos + 3 * b, ld3(m.static_xpos + 3 * b)); stq(s.xquat + 4 * b, ldq(m.static_xquat + 4 * b));
      continue;
    }
    // everything static about this body in one 80-byte record (one load latency, not a chain of four)
    const rgf4* R = (const rgf4*)m.body_rec + (RG_KINREC / 4) * b;
    rgf4 r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
    int w0 = __builtin_bit_cast(int, r0.x), w1 = __builtin_bit_cast(int, r0.y);
    int jn = (w0 >> 16) & 15, ja = w1 & 0xFFFF;
    v3 pos = mk3(r0.z, r0.w, r1.x);
    q4 quat; quat.w = r1.y; quat.x = r1.z; quat.y = r1.w; quat.z = r2.x;
    for (int jj = 0; jj < jn; jj++) {
      int j = ja + jj, t, qa; v3 jpos, jaxis; float q0;
      if (jj == 0) { t = (w0 >> 20) & 15; qa = (w1 >> 16) & 0xFFFF; jpos = mk3(r3.x, r3.y, r3.z); jaxis = mk3(r3.w, r4.x, r4.y); q0 = r4.z; }
      else { t = m.jnt_type[j]; qa = m.jnt_qposadr[j]; jpos = ld3(m.jnt_pos + 3 * j); jaxis = ld3(m.jnt_axis + 3 * j); q0 = m.qpos0[qa]; }
      if (t == RG_JNT_FREE) {   // a free joint states the world frame itself (body_depth flags the chain element as absolute)
        pos = ld3(s.qpos + qa); quat = qnormalize(ldq(s.qpos + qa + 3));
        st3(s.xanchor + 3 * j, pos); st3(s.xaxis + 3 * j, mk3(0, 0, 1));
        continue;
      }
      v3 anchor = pos + qrot(quat, jpos);
      v3 axis = qrot(quat, jaxis);
      st3(s.xanchor + 3 * j, anchor); st3(s.xaxis + 3 * j, axis);
      if (t == RG_JNT_SLIDE) pos = pos + axis * (s.qpos[qa] - q0);
      else {
        q4 ql = (t == RG_JNT_BALL) ? qnormalize(ldq(s.qpos + qa)) : axisangle(jaxis, s.qpos[qa] - q0);
        quat = qmul(quat, ql);
        pos = anchor - qrot(quat, jpos);
      }
    }
    st3(lp + 3 * b, pos); stq(lq + 4 * b, quat);
  }
  SYNC();
  for (int b = 1 + LANE; b < m.nbody; b += RG_WAVE) {
    const int dw = m.body_depth[b], depth = dw & 255, absmask = (dw >> 8) & 255, p0 = (dw >> 16) & 255;
    if (depth == 0) continue;
    const unsigned c0 = (unsigned)m.body_chain[2 * b], c1 = (unsigned)m.body_chain[2 * b + 1];
    const int w0 = __builtin_bit_cast(int, m.body_rec[RG_KINREC * b]), w1 = __builtin_bit_cast(int, m.body_rec[RG_KINREC * b + 1]);
    v3 pos = ld3(m.static_xpos + 3 * p0), ppos = pos; q4 quat = ldq(m.static_xquat + 4 * p0), pq = quat;   // the chain hangs off the world or a static body
    for (int i = 0; i < depth; i++) {
      const int a = (i < 4 ? c0 >> (8 * i) : c1 >> (8 * (i - 4))) & 255;
      ppos = pos; pq = quat;
      if ((absmask >> i) & 1) { pos = ld3(lp + 3 * a); quat = ldq(lq + 4 * a); }
      else { pos = ppos + qrot(pq, ld3(lp + 3 * a)); quat = qnormalize(qmul(pq, ldq(lq + 4 * a))); }
    }
    const bool absolute = (absmask >> (depth - 1)) & 1;
    const int jn = (w0 >> 16) & 15, ja = w1 & 0xFFFF;
    if (!absolute) for (int jj = 0; jj < jn; jj++) {   // the body's joints: parent frame -> world
      const int j = ja + jj;
      st3(s.xanchor + 3 * j, ppos + qrot(pq, ld3(s.xanchor + 3 * j))); st3(s.xaxis + 3 * j, qrot(pq, ld3(s.xaxis + 3 * j)));
    }
    st3(s.xpos + 3 * b, pos); stq(s.xquat + 4 * b, quat);
  }
  SYNC();
  for (int g0 = 0; g0 < m.ngeom; g0 += 2 * RG_WAVE) {   // two geoms per lane and trip, their constants requested together (65 geoms: one trip)
    const int gA = g0 + LANE, gB = gA + RG_WAVE, last = m.ngeom - 1, iA = gA < last ? gA : last, iB = gB < last ? gB : last;
    const int bA = m.geom_bodyid[iA], bB = m.geom_bodyid[iB];
    const v3 pA = ld3(m.geom_pos + 3 * iA), pB = ld3(m.geom_pos + 3 * iB); const q4 qA = ldq(m.geom_quat + 4 * iA), qB = ldq(m.geom_quat + 4 * iB);
    if (gA < m.ngeom) { q4 xq = ldq(s.xquat + 4 * bA); st3(s.gpos + 3 * gA, ld3(s.xpos + 3 * bA) + qrot(xq, pA)); stq(s.gquat + 4 * gA, qmul(xq, qA)); }
    if (gB < m.ngeom) { q4 xq = ldq(s.xquat + 4 * bB); st3(s.gpos + 3 * gB, ld3(s.xpos + 3 * bB) + qrot(xq, pB)); stq(s.gquat + 4 * gB, qmul(xq, qB)); }
  }
  PFOR(i, m.nsite) {
    int b = m.site_bodyid[i];
    st3(s.spos + 3 * i, ld3(s.xpos + 3 * b) + qrot(ldq(s.xquat + 4 * b), ld3(P + RG_PRM_SITE_POS + 3 * i)));
  }
  if (LANE < 4) {   // com-frame origin of every kinematic tree: the com of its origin body, or a constant (one record per slot, rg_api.hip)
    const rgf4 o = ((const rgf4*)m.org_rec)[LANE];
    const int ob = __builtin_bit_cast(int, o.x);
    if (ob != -2) st3(s.org + 3 * LANE, ob >= 0 ? ld3(s.xpos + 3 * ob) + qrot(ldq(s.xquat + 4 * ob), mk3(o.y, o.z, o.w)) : mk3(o.y, o.z, o.w));
  }
  SYNC();
}

// translational Jacobian column of dof d for a point with offset `off` from the com-frame origin
__device__ __forceinline__ v3 jac_col(const RgLds& s, int d, v3 off) { return ld3(s.cdof + 6 * d + 3) + cross(ld3(s.cdof + 6 * d), off); }
__device__ __forceinline__ bool in_chain(RgM m, int body, int d) { return (m.body_dofmask[2 * body + (d >> 5)] >> (d & 31)) & 1u; }

__device__ __forceinline__ void rg_com_pos(RgM m, RgLds& s, const float* P) {
  for (int b = 1 + LANE; b < m.nbody; b += RG_WAVE) {
    float R[9], I[9];
    q2mat(R, qmul(ldq(s.xquat + 4 * b), ldq(m.body_iquat + 4 * b)));   // orientation of the inertial frame: body frame x iquat
    const float* in = P + RG_PRM_BODY_INERTIA + 3 * b;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[3 * i + j] = R[3 * i] * in[0] * R[3 * j] + R[3 * i + 1] * in[1] * R[3 * j + 1] + R[3 * i + 2] * in[2] * R[3 * j + 2];
    v3 d = rg_xipos(m, s, b) - ld3(s.org + 3 * s.b2org[b]);
    float mass = P[RG_PRM_BODY_MASS + b], d2 = dot(d, d);
    float* ci = s.cinert + 10 * b;
    ci[0] = I[0] + mass * (d2 - d.x * d.x); ci[1] = I[4] + mass * (d2 - d.y * d.y); ci[2] = I[8] + mass * (d2 - d.z * d.z);
    ci[3] = I[1] - mass * d.x * d.y; ci[4] = I[2] - mass * d.x * d.z; ci[5] = I[5] - mass * d.y * d.z;
    ci[6] = mass * d.x; ci[7] = mass * d.y; ci[8] = mass * d.z; ci[9] = mass;
  }
  PFOR(j, m.njnt) {
    int b = m.jnt_bodyid[j], da = m.jnt_dofadr[j], t = m.jnt_type[j];
    v3 off = ld3(s.org + 3 * s.b2org[b]) - ld3(s.xanchor + 3 * j);
    float R[9]; q2mat(R, ldq(s.xquat + 4 * b));
    if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
      if (t == RG_JNT_FREE) {
        for (int k = 0; k < 3; k++) { float* c = s.cdof + 6 * (da + k); for (int e = 0; e < 6; e++) c[e] = 0; c[3 + k] = 1; }
        da += 3;
      }
      for (int k = 0; k < 3; k++) { v3 ax = mk3(R[k], R[3 + k], R[6 + k]); st3(s.cdof + 6 * (da + k), ax); st3(s.cdof + 6 * (da + k) + 3, cross(ax, off)); }
    } else if (t == RG_JNT_SLIDE) {
      st3(s.cdof + 6 * da, mk3(0, 0, 0)); st3(s.cdof + 6 * da + 3, ld3(s.xaxis + 3 * j));
    } else {
      v3 ax = ld3(s.xaxis + 3 * j); st3(s.cdof + 6 * da, ax); st3(s.cdof + 6 * da + 3, cross(ax, off));
    }
  }
  SYNC();
  if (s.has_xfrc) {
    // data.xfrc_applied (mj_xfrcAccumulate): force and torque at the com of every body whose chain holds the dof.  cdof is
    // expressed in the com-based frame of the body's tree: J_point(d) = cdof_lin + cdof_ang x (point - origin).  The result
    // waits in qfrc_smooth for rg_smooth (the body frames are gone by then).
    PFOR(d, m.nv) {
      float xf = 0;
      for (int b = 1; b < m.nbody; b++) {
        if (!in_chain(m, b, d)) continue;
        const float* w = P + RG_PRM_XFRC + 6 * b;
        v3 off = rg_xipos(m, s, b) - ld3(s.org + 3 * s.b2org[b]);
        xf += dot(ld3(w), jac_col(s, d, off)) + dot(ld3(w + 3), ld3(s.cdof + 6 * d));
      }
      s.qfrc_smooth[d] = xf;
    }
    SYNC();
  }
}


// ---- tendon wrapping (see oracle: wrap_circle / ro_wrap)
__device__ __forceinline__ bool seg_intersect(float p1x, float p1y, float p2x, float p2y, float p3x, float p3y, float p4x, float p4y) {
  float det = (p4y - p3y) * (p2x - p1x) - (p4x - p3x) * (p2y - p1y);
  if (fabsf(det) < 1e-15f) return false;
  float idet = rg_rcp(det);
  float a = ((p4x - p3x) * (p1y - p3y) - (p4y - p3y) * (p1x - p3x)) * idet;
  float b = ((p2x - p1x) * (p1y - p3y) - (p2y - p1y) * (p1x - p3x)) * idet;
  return a >= 0 && a <= 1 && b >= 0 && b <= 1;
}
__device__ __forceinline__ float wrap_circle(float* pnt, const float* d, bool has_sd, float sd0, float sd1, float rad) {
  float sq0 = d[0] * d[0] + d[1] * d[1], sq1 = d[2] * d[2] + d[3] * d[3], sqr = rad * rad;
  float dx = d[2] - d[0], dy = d[3] - d[1], dd = dx * dx + dy * dy;
  if (sq0 < sqr || sq1 < sqr || rad < 1e-15f || dd < 1e-15f) return -1;
  float a = clampf(-(dx * d[0] + dy * d[1]) * rg_rcp(dd), 0.f, 1.f);
  float nx = a * dx + d[0], ny = a * dy + d[1];
  if (nx * nx + ny * ny > sqr && (!has_sd || sd0 * nx + sd1 * ny >= 0)) return -1;
  float sol[2][4], good[2];
  float r0 = rg_sqrt(sq0 - sqr), r1 = rg_sqrt(sq1 - sqr), isq0 = rg_rcp(sq0), isq1 = rg_rcp(sq1);
  for (int i = 0; i < 2; i++) {
    float sgn = i == 0 ? 1.f : -1.f;
    sol[i][0] = (d[0] * sqr + sgn * rad * d[1] * r0) * isq0; sol[i][1] = (d[1] * sqr - sgn * rad * d[0] * r0) * isq0;
    sol[i][2] = (d[2] * sqr - sgn * rad * d[3] * r1) * isq1; sol[i][3] = (d[3] * sqr + sgn * rad * d[2] * r1) * isq1;
    if (has_sd) {
      float mx = sol[i][0] + sol[i][2], my = sol[i][1] + sol[i][3], n = fmaxf(rg_sqrt(mx * mx + my * my), 1e-15f);
      good[i] = (mx * sd0 + my * sd1) * rg_rcp(n);
    } else {
      float tx = sol[i][0] - sol[i][2], ty = sol[i][1] - sol[i][3];
      good[i] = -(tx * tx + ty * ty);
    }
    if (seg_intersect(d[0], d[1], sol[i][0], sol[i][1], d[2], d[3], sol[i][2], sol[i][3])) good[i] = -10000.f;
  }
  int k = good[0] > good[1] ? 0 : 1;
  for (int e = 0; e < 4; e++) pnt[e] = sol[k][e];
  if (seg_intersect(d[0], d[1], pnt[0], pnt[1], d[2], d[3], pnt[2], pnt[3])) return -1;
  return rad * acosf(clampf((pnt[0] * pnt[2] + pnt[1] * pnt[3]) * rg_rcp(sqr), -1.f, 1.f));
}
__device__ __forceinline__ float rg_wrap(v3& w0, v3& w1, v3 x0, v3 x1, v3 gpos, const float* gmat, float radius, int type, bool has_side, v3 side) {
  v3 p0 = mulmT(gmat, x0 - gpos), p1 = mulmT(gmat, x1 - gpos);
  if (norm(p0) < 1e-15f || norm(p1) < 1e-15f) return -1;
  v3 ax0, ax1;
  if (type == RG_WRAP_SPHERE) {
    ax0 = normalized(p0);
    v3 nrm = cross(p0, p1);
    if (norm(nrm) < 1e-15f) nrm = cross(ax0, fabsf(ax0.x) < 0.9f ? mk3(1, 0, 0) : mk3(0, 1, 0));
    nrm = normalized(nrm); ax1 = normalized(cross(nrm, ax0));
  } else { ax0 = mk3(1, 0, 0); ax1 = mk3(0, 1, 0); }
  float dd[4] = {dot(p0, ax0), dot(p0, ax1), dot(p1, ax0), dot(p1, ax1)}, sd0 = 0, sd1 = 0;
  if (has_side) {
    v3 sl = mulmT(gmat, side - gpos);
    sd0 = dot(sl, ax0); sd1 = dot(sl, ax1);
    float n = fmaxf(rg_sqrt(sd0 * sd0 + sd1 * sd1), 1e-15f), rn = radius * rg_rcp(n);
    sd0 *= rn; sd1 *= rn;
  }
  float pnt[4], wlen = wrap_circle(pnt, dd, has_side, sd0, sd1, radius);
  if (wlen < 0) return -1;
  v3 r0 = ax0 * pnt[0] + ax1 * pnt[1], r1 = ax0 * pnt[2] + ax1 * pnt[3];
  if (type == RG_WRAP_CYLINDER) {
    float L0 = rg_sqrt((dd[0] - pnt[0]) * (dd[0] - pnt[0]) + (dd[1] - pnt[1]) * (dd[1] - pnt[1]));
    float L1 = rg_sqrt((dd[2] - pnt[2]) * (dd[2] - pnt[2]) + (dd[3] - pnt[3]) * (dd[3] - pnt[3]));
    float iL = rg_rcp(L0 + wlen + L1);
    r0.z = p0.z + (p1.z - p0.z) * L0 * iL;
    r1.z = p0.z + (p1.z - p0.z) * (L0 + wlen) * iL;
    wlen = rg_sqrt(wlen * wlen + (r1.z - r0.z) * (r1.z - r0.z));
  }
  w0 = mulm(gmat, r0) + gpos; w1 = mulm(gmat, r1) + gpos;
  return wlen;
}

// tendon lengths and Jacobians on their static dof supports; actuator lengths
__device__ __forceinline__ void rg_tendon(RgM m, RgLds& s) {
  PFOR(t, m.ntendon) {
    // the tendon's path as static 8-word records (kernel_tables.py k_ten_path): one load per stretch, no index chains
    int r0 = m.ten_path_adr[t], r1 = m.ten_path_adr[t + 1];
    const int* td = m.ten_dofs + 4 * t;
    int d0 = td[0], d1 = td[1], d2 = td[2], d3 = td[3];
    float J[4] = {0, 0, 0, 0}, L = 0;
    for (int r = r0; r < r1; r++) {
      const rgf4* R = (const rgf4*)m.ten_path + 2 * r;
      rgf4 ra = R[0], rb = R[1];
      int kind = __builtin_bit_cast(int, ra.x), w1 = __builtin_bit_cast(int, ra.y), w2 = __builtin_bit_cast(int, ra.z);
      if ((kind & 15) == 0) {   // fixed tendon: coef * joint position
        float coef = ra.w;
        L += coef * s.qpos[w1];
        for (int e = 0; e < 4; e++) if (e == w2) J[e] += coef;
        continue;
      }
      float idiv = ra.w, radius = rb.x; int bits = __builtin_bit_cast(int, rb.y);
      // straight segments of this stretch: site -> site, or site -> wrap entry and wrap exit -> site
      // (named points, no indexed private arrays: those live in scratch memory)
      int s0 = w1 & 255, s1 = (w1 >> 8) & 255, g = (w1 >> 16) & 255, sid = (w1 >> 24) & 255;
      int ba = w2 & 255, bs = (w2 >> 8) & 255, bg = (w2 >> 16) & 255;
      v3 pa = ld3(s.spos + 3 * s0), x1 = ld3(s.spos + 3 * s1), pb = x1, pc = mk3(0, 0, 0), pd = mk3(0, 0, 0);
      bool two = false;
      float wlen = -1;
      if ((kind & 15) == 2) {
        v3 w0 = mk3(0, 0, 0), w1p = mk3(0, 0, 0);
        float gm[9]; q2mat(gm, ldq(s.gquat + 4 * g));
        wlen = rg_wrap(w0, w1p, pa, x1, ld3(s.gpos + 3 * g), gm, radius, kind >> 4, sid != 255, sid != 255 ? ld3(s.spos + 3 * sid) : mk3(0, 0, 0));
        if (wlen >= 0) { pb = w0; pc = w1p; pd = x1; two = true; L += wlen * idiv; }
      }
      for (int seg = 0; seg < (two ? 2 : 1); seg++) {
        v3 q0 = seg ? pc : pa, q1 = seg ? pd : pb;
        int k0 = seg ? 2 : 0, k1 = two ? (seg ? 1 : 2) : 1;      // which of (ba, bs, bg) the segment's end points ride on
        int b0 = seg ? bg : ba, b1 = two ? (seg ? bs : bg) : bs;
        v3 dif = q1 - q0;
        float dist = rg_sqrt(dot(dif, dif));
        L += dist * idiv;
        if (b0 != b1 && dist > 1e-15f) {
          dif = dif * rg_rcp(dist);
          v3 o0 = q0 - ld3(s.org + 3 * s.b2org[b0]), o1 = q1 - ld3(s.org + 3 * s.b2org[b1]);
          for (int e = 0; e < 4; e++) {
            int d = e == 0 ? d0 : (e == 1 ? d1 : (e == 2 ? d2 : d3));
            if (d < 0) continue;
            float v = 0;
            if ((bits >> (3 * e + k1)) & 1) v += dot(dif, jac_col(s, d, o1));
            if ((bits >> (3 * e + k0)) & 1) v -= dot(dif, jac_col(s, d, o0));
            J[e] += v * idiv;
          }
        }
      }
    }
    s.tenlen[t] = L;
    for (int e = 0; e < 4; e++) s.tenJ[4 * t + e] = J[e];
  }
  SYNC();
  PFOR(u, m.nu) {
    int id = m.actuator_trnid[u];
    s.actlen[u] = m.actuator_gear[u] * (m.actuator_trntype[u] == 0 ? s.qpos[m.jnt_qposadr[id]] : s.tenlen[id]);
  }
  SYNC();
}

// composite inertias (subtree gathers), sparse M, tree-sparse L'DL factorisation
__device__ __forceinline__ void rg_crb(RgM m, RgLds& s, const float* P, const int* subtree_adr, const int* subtree) {
  for (int w = LANE; w < m.nbody * 10; w += RG_WAVE) {
    int b = w / 10, k = w - 10 * b;
    float acc = 0;
    for (uint32_t bits = (uint32_t)m.subtree_mask[b]; bits; bits &= bits - 1) acc += s.cinert[10 * __builtin_ctz(bits) + k];
    s.crb[w] = acc;
  }
  SYNC();
  PFOR(e, m.nM) {
    const int ijb = m.M_ijb[e], i = ijb & 255, j = (ijb >> 8) & 255;
    float buf[6];
    mul_inert_vec(buf, s.crb + 10 * (ijb >> 16), s.cdof + 6 * i);
    const float* c = s.cdof + 6 * j;
    float v = c[0] * buf[0] + c[1] * buf[1] + c[2] * buf[2] + c[3] * buf[3] + c[4] * buf[4] + c[5] * buf[5];
    if (i == j) v += P[RG_PRM_DOF_ARMATURE + i];
    s.Msp[e] = v;
  }
  SYNC();
}
// s.H <- M in the per-tree block layout (both triangles; M_ent[2e] = word of (i,j) | word of (j,i) << 16)
__device__ __forceinline__ void rg_M_to_blocks(RgM m, RgLds& s) {
  { rgf4 z; z.x = z.y = z.z = z.w = 0.f; rgf4* H4 = (rgf4*)s.H; for (int w = LANE; w < ((m.blkwords + 3) >> 2); w += RG_WAVE) H4[w] = z; }
  SYNC();
  PFOR(e, m.nM) { int a = m.M_ent[2 * e]; float v = s.Msp[e]; s.H[a & 0xFFFF] = v; s.H[(a >> 16) & 0xFFFF] = v; }
  SYNC();
}

// ------------------------------------------------------------------------------------------------- collision
struct SupPt { v3 v, s; };  // v = v1 - v2 (Minkowski difference), s = v1 + v2 (all the contact position needs)
// per-query (lane-varying) description of one geom, and the wave-uniform tables every query shares: keeping the
// 64-bit table pointers out of MprGeom keeps them in SGPRs (the narrowphase is the register-hungriest stage)
struct MprGeom { int type; const float* quat; v3 pos; v3 size; int vertadr, nvert, mesh; float margin; };   // quat: the geom's orientation in LDS (16-byte aligned)   // mesh: id, -1 for primitives
struct MprEnv { const float* mesh_vert; const int* cell_adr; const rgf4 *cell_blk, *cell_ovf; float* prof; bool cells; bool plane_depth; };

// per-lane scan of a hull's vertices: 16-byte records (one dwordx4 load per vertex), four independent
// loads in flight per lane; out-of-range slots re-read the last vertex (harmless for a max).
// G lanes cooperate on one hull (G = 64: the wave; G = 16 / 8: a DPP row / half row, 4 / 8 MPR queries per wave).
// one fixed fused sequence, so that every scan variant ranks the vertices identically
__device__ __forceinline__ float vdot(v3 d, const rgf4& a) { return __builtin_fmaf(d.z, a.z, __builtin_fmaf(d.y, a.y, d.x * a.x)); }
template <int G> __device__ __forceinline__ void scan_batch(const rgf4* vert, int nvert, int base, v3 ld, float& bv, int& bi, v3& bp) {
  int i0 = base + (LANE & (G - 1)), last = nvert - 1;
  int j0 = i0 < last ? i0 : last, j1 = i0 + G < last ? i0 + G : last, j2 = i0 + 2 * G < last ? i0 + 2 * G : last, j3 = i0 + 3 * G < last ? i0 + 3 * G : last;
  rgf4 a = vert[j0], b = vert[j1], c = vert[j2], d = vert[j3];
  float da = vdot(ld, a), db = vdot(ld, b), dc = vdot(ld, c), dd = vdot(ld, d);
  if (da > bv) { bv = da; bi = j0; bp = mk3(a.x, a.y, a.z); }
  if (db > bv) { bv = db; bi = j1; bp = mk3(b.x, b.y, b.z); }
  if (dc > bv) { bv = dc; bi = j2; bp = mk3(c.x, c.y, c.z); }
  if (dd > bv) { bv = dd; bi = j3; bp = mk3(d.x, d.y, d.z); }
}
// full scan of a hull: the reference path behind the cell lists (flags bit 3, the plane-convex pairs)
template <int G> __device__ __forceinline__ void scan_verts(const rgf4* vert, int nvert, v3 ld, float& bv, int& bi, v3& bp) {
  for (int base = 0; base < nvert; base += 4 * G) scan_batch<G>(vert, nvert, base, ld, bv, bi, bp);
}
// direction cell of a (hull-local) direction: cube-map face and RG_CELLN x RG_CELLN grid on it
// (kernel_tables.py _direction_cells is the host-side statement of the same arithmetic)
__device__ __forceinline__ int dir_cell(v3 ld) {
  float ax = fabsf(ld.x), ay = fabsf(ld.y), az = fabsf(ld.z);
  int axis = ax >= ay ? (ax >= az ? 0 : 2) : (ay >= az ? 1 : 2);
  float mj = axis == 0 ? ld.x : (axis == 1 ? ld.y : ld.z), a = axis == 0 ? ld.y : (axis == 1 ? ld.z : ld.x), b = axis == 0 ? ld.z : (axis == 1 ? ld.x : ld.y);
  float inv = (0.5f * RG_CELLN) * rg_rcp(fmaxf(fabsf(mj), 1e-30f));
  int iu = (int)(a * inv + 0.5f * RG_CELLN), iv = (int)(b * inv + 0.5f * RG_CELLN);
  iu = iu < 0 ? 0 : (iu > RG_CELLN - 1 ? RG_CELLN - 1 : iu); iv = iv < 0 ? 0 : (iv > RG_CELLN - 1 ? RG_CELLN - 1 : iv);
  return ((2 * axis + (mj < 0 ? 1 : 0)) * RG_CELLN + iu) * RG_CELLN + iv;
}
// the same arg-max restricted to the cell's candidate list (typically 3-6 records instead of 60-300
// vertices; the list provably contains every vertex that can win, so the result is bit-identical)
template <int G> __device__ __forceinline__ void scan_cell(const int* celladr, const rgf4* cellblk, const rgf4* cellovf, v3 ld, float& bv, int& bi, v3& bp) {
  int cell = dir_cell(ld), l = LANE & (G - 1);
  int e = celladr[cell];
  if (G >= 4) {
    rgf4 a = cellblk[4 * cell + (l & 3)];   // the cell's first four candidates sit at a computable address: one load latency
    float da = vdot(ld, a);
    if (da > bv) { bv = da; bi = __builtin_bit_cast(int, a.w); bp = mk3(a.x, a.y, a.z); }
  } else {   // pairs of lanes: two of the four each
    rgf4 a = cellblk[4 * cell + 2 * (l & 1)], a2 = cellblk[4 * cell + 2 * (l & 1) + 1];
    float da = vdot(ld, a), da2 = vdot(ld, a2);
    if (da > bv) { bv = da; bi = __builtin_bit_cast(int, a.w); bp = mk3(a.x, a.y, a.z); }
    int i2 = __builtin_bit_cast(int, a2.w);
    if (da2 > bv || (da2 == bv && i2 < bi)) { bv = da2; bi = i2; bp = mk3(a2.x, a2.y, a2.z); }
  }
  int cnt = (e & 255) - 4, last = cnt - 1;
  if (cnt > 0) {
    const rgf4* rec = cellovf + (e >> 8);
    for (int base = 0; base < cnt; base += 2 * G) {
      int i0 = base + l, j0 = i0 < last ? i0 : last, j1 = i0 + G < last ? i0 + G : last;
      rgf4 b = rec[j0], c = rec[j1];
      float db = vdot(ld, b), dc = vdot(ld, c);
      int ib = __builtin_bit_cast(int, b.w), ic = __builtin_bit_cast(int, c.w);
      if (db > bv || (db == bv && ib < bi)) { bv = db; bi = ib; bp = mk3(b.x, b.y, b.z); }
      if (dc > bv || (dc == bv && ic < bi)) { bv = dc; bi = ic; bp = mk3(c.x, c.y, c.z); }
    }
  }
}
template <int G> __device__ __forceinline__ void scan_hull(const MprEnv& E, const MprGeom& g, v3 ld, float& bv, int& bi, v3& bp) {
  if (E.cells) scan_cell<G>(E.cell_adr + g.mesh * RG_NCELL, E.cell_blk + (size_t)g.mesh * (RG_NCELL * 4), E.cell_ovf, ld, bv, bi, bp);
  else scan_verts<G>((const rgf4*)E.mesh_vert + g.vertadr, g.nvert, ld, bv, bi, bp);
}
// arg-max over the G cooperating lanes (lowest vertex index on ties, as a serial first-max scan); the
// winner's coordinates come from the registers of a lane that scanned it
template <int G> __device__ __forceinline__ v3 pick_vert(float bv, int bi, v3 bp) {
  if (G == 64) {
    float vm = wave_max(bv);
    int wi = wave_min_i(bv == vm ? bi : 0x7fffffff);
    int wl = wave_min_i((bv == vm && bi == wi) ? LANE : 0x7fffffff);
    return mk3(lane_bcast(bp.x, wl), lane_bcast(bp.y, wl), lane_bcast(bp.z, wl));
  } else {
    float vm = grp_max<G>(bv);
    int wi = grp_min_i<G>(bv == vm ? bi : 0x7fffffff);
    bool win = bv == vm && bi == wi;  // several lanes may hold the same (clamped) vertex: identical coordinates
    return mk3(grp_max<G>(win ? bp.x : -3.0e38f), grp_max<G>(win ? bp.y : -3.0e38f), grp_max<G>(win ? bp.z : -3.0e38f));
  }
}
__device__ __forceinline__ v3 support_primitive(const MprGeom& g, v3 ld) {
  v3 lr;
  if (g.type == RG_GEOM_BOX) lr = mk3(ld.x >= 0 ? g.size.x : -g.size.x, ld.y >= 0 ? g.size.y : -g.size.y, ld.z >= 0 ? g.size.z : -g.size.z);
  else if (g.type == RG_GEOM_SPHERE) lr = ld * g.size.x;
  else if (g.type == RG_GEOM_CAPSULE) { lr = ld * g.size.x; lr.z += ld.z >= 0 ? g.size.y : -g.size.y; }
  else if (g.type == RG_GEOM_CYLINDER) {
    float n = sqrtf(ld.x * ld.x + ld.y * ld.y);
    lr = n > 1e-15f ? mk3(ld.x / n * g.size.x, ld.y / n * g.size.x, 0) : mk3(0, 0, 0);
    lr.z = ld.z >= 0 ? g.size.y : -g.size.y;
  } else {  // ellipsoid
    v3 t = mk3(ld.x * g.size.x, ld.y * g.size.y, ld.z * g.size.z);
    float n = fmaxf(norm(t), 1e-15f);
    lr = mk3(t.x * g.size.x / n, t.y * g.size.y / n, t.z * g.size.z / n);
  }
  return lr;
}
// support point of one geom (relative to the MPR reference origin), cooperative for meshes
template <int G> __device__ __forceinline__ v3 rg_support(const MprEnv& E, const MprGeom& g, v3 dir) {
  q4 gq = ldq(g.quat);
  v3 ld = qrotT(gq, dir), lr;
  if (g.type == RG_GEOM_MESH) {
    float bv = -3.0e38f; int bi = 0x7fffffff; v3 bp = mk3(0, 0, 0);
    scan_hull<G>(E, g, ld, bv, bi, bp);
    lr = pick_vert<G>(bv, bi, bp);
  } else lr = support_primitive(g, ld);
  lr = lr + ld * g.margin;
  return qrot(gq, lr) + g.pos;
}
// Minkowski-difference support A(dir) - B(-dir); the two hull scans are issued back to back so their
// vertex loads overlap
template <int G> __device__ __forceinline__ void mpr_support(const MprEnv& E, const MprGeom& a, const MprGeom& b, v3 dir, SupPt& p) {
  long long tt0 = E.prof ? rg_clock() : 0;
  q4 qa = ldq(a.quat), qb = ldq(b.quat);
  v3 la = qrotT(qa, dir), lb = qrotT(qb, dir * -1.0f), ra, rb;
  float av = -3.0e38f, bvv = -3.0e38f; int ai = 0x7fffffff, bi = 0x7fffffff; v3 ap = mk3(0, 0, 0), bp = mk3(0, 0, 0);
  bool am = a.type == RG_GEOM_MESH, bm = b.type == RG_GEOM_MESH;
  if (am) scan_hull<G>(E, a, la, av, ai, ap);
  if (bm) scan_hull<G>(E, b, lb, bvv, bi, bp);
  long long tt1 = E.prof ? rg_clock() : 0;
  ra = am ? pick_vert<G>(av, ai, ap) : support_primitive(a, la);
  rb = bm ? pick_vert<G>(bvv, bi, bp) : support_primitive(b, lb);
  v3 w1 = qrot(qa, ra + la * a.margin) + a.pos;
  v3 w2 = qrot(qb, rb + lb * b.margin) + b.pos;
  p.v = w1 - w2; p.s = w1 + w2;
  if (E.prof && LANE == 0) { long long tt2 = rg_clock(); E.prof[20] += 1.f; E.prof[21] += (float)(tt1 - tt0); E.prof[22] += (float)(tt2 - tt1); }
}
#define MPR_EPS 1.0e-7f  /* plays the role of libccd's CCD_EPS at fp32 (coordinates are pair-local, |x| ~ 0.1) */
__device__ __forceinline__ bool mz(float x) { return fabsf(x) < MPR_EPS * 1e-3f; }
// (the portal is four named points, not an array: an indexed private array ends up in scratch memory)
__device__ __forceinline__ v3 portal_dir(const SupPt& p1, const SupPt& p2, const SupPt& p3) { return normalized(cross(p2.v - p1.v, p3.v - p1.v)); }
__device__ __forceinline__ bool portal_reach_tol(const SupPt& p1, const SupPt& p2, const SupPt& p3, const SupPt& v4, v3 dir, float tol) {
  float dv4 = dot(v4.v, dir);
  float mn = fminf(fminf(dv4 - dot(p1.v, dir), dv4 - dot(p2.v, dir)), dv4 - dot(p3.v, dir));
  return mn <= tol;
}
__device__ __forceinline__ v3 sel3(bool c, v3 a, v3 b) { return mk3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
// (value selects per component, not "store through a selected pointer": the latter keeps the points in scratch memory)
__device__ __forceinline__ void expand_portal(const SupPt& p0, SupPt& p1, SupPt& p2, SupPt& p3, const SupPt& v4) {
  v3 c = cross(v4.v, p0.v);
  bool a = dot(p1.v, c) > 0, b = dot(p2.v, c) > 0, d = dot(p3.v, c) > 0;
  bool to1 = a ? b : !d, to3 = a && !b, to2 = !a && d;
  p1.v = sel3(to1, v4.v, p1.v); p1.s = sel3(to1, v4.s, p1.s);
  p2.v = sel3(to2, v4.v, p2.v); p2.s = sel3(to2, v4.s, p2.s);
  p3.v = sel3(to3, v4.v, p3.v); p3.s = sel3(to3, v4.s, p3.s);
}
__device__ __forceinline__ float origin_tri_dist2(v3 a, v3 b, v3 c, v3& w) {
  v3 ab = b - a, ac = c - a, ap = a * -1.0f;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { w = a; return dot(w, w); }
  v3 bp = b * -1.0f; float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { w = b; return dot(w, w); }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { w = a + ab * (d1 / (d1 - d3)); return dot(w, w); }
  v3 cp = c * -1.0f; float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { w = c; return dot(w, w); }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { w = a + ac * (d2 / (d2 - d6)); return dot(w, w); }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { w = b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6))); return dot(w, w); }
  float den = 1.0f / (va + vb + vc);
  w = a + ab * (vb * den) + ac * (vc * den);
  return dot(w, w);
}
// depth, direction and position of a penetration from the final portal (the tail of libccd's __ccdMPRFindPenetr / ccdMPRPenetration)
__device__ __forceinline__ void mpr_result(bool plane_depth, const SupPt& p0, const SupPt& p1, const SupPt& p2, const SupPt& p3, v3 dir, float& depth, v3& dir_out, v3& pos) {
  // depth / direction from the portal PLANE (not libccd's closest point on the final portal triangle,
  // whose choice among the triangles of a flat supporting plane is rounding noise; see DESIGN.md "MPR")
  if (plane_depth) {
    depth = fmaxf((dot(p1.v, dir) + dot(p2.v, dir) + dot(p3.v, dir)) * (1.0f / 3.0f), 0.f);
    dir_out = dir;
  } else {   // libccd verbatim (what MuJoCo 2.0's mjc_Convex runs): the closest point of the final portal triangle to the origin
    v3 w; depth = sqrtf(origin_tri_dist2(p1.v, p2.v, p3.v, w));
    dir_out = mz(depth) ? mk3(0, 0, 0) : w * (1.0f / depth);
  }
  // contact position from the barycentric coordinates of the origin in the portal tetrahedron
  float b0 = dot(cross(p1.v, p2.v), p3.v), b1 = dot(cross(p3.v, p2.v), p0.v);
  float b2 = dot(cross(p0.v, p1.v), p3.v), b3 = dot(cross(p2.v, p1.v), p0.v);
  float sum = b0 + b1 + b2 + b3;
  if (sum <= 0) {
    v3 dd = portal_dir(p1, p2, p3);
    b0 = 0; b1 = dot(cross(p2.v, p3.v), dd); b2 = dot(cross(p3.v, p1.v), dd); b3 = dot(cross(p1.v, p2.v), dd);
    sum = b1 + b2 + b3;
  }
  float inv = 0.5f * rg_rcp(sum);
  pos = p0.s * (b0 * inv) + p1.s * (b1 * inv) + p2.s * (b2 * inv) + p3.s * (b3 * inv);
}
// MPR penetration query (libccd ccdMPRPenetration) of one pair per group of G lanes.  Returns true on contact.
// `sep`: on a "no contact" exit that PROVES separation (support of the Minkowski difference along `sep` is <= 0)
// the unit direction that proves it, else zero — cached per pair and tried first on the next substep.
//
// All groups of the wave run ONE loop whose body is "support point along dir, then the state's bookkeeping":
// every step of the algorithm (the two initial supports, portal discovery, portal refinement, penetration
// search) is "pick a direction, take a support point, test / update the portal", so the expensive common
// part — the support — executes convergently for all groups and only the few dozen scalar instructions
// of bookkeeping diverge.  (With one nested loop per phase the groups drift apart and the wave
// serialises their different phases.)  `active` false: the group idles; call from convergent code.
enum { MPR_DONE = 0, MPR_FIRST, MPR_SECOND, MPR_DISCOVER, MPR_REFINE, MPR_PENETR };
template <int G> __device__ __forceinline__ bool rg_mpr(const MprEnv& E, const MprGeom& A, const MprGeom& B, int max_iter, float tol, float& depth, v3& dir_out, v3& pos, v3& sep, bool active) {
  SupPt p0, p1, p2, p3;
  p0.v = p0.s = p1.v = p1.s = p2.v = p2.s = p3.v = p3.s = mk3(0, 0, 0);
  int state = active ? MPR_FIRST : MPR_DONE, guard = 0;
  bool result = false, finish = false;
  v3 dir = mk3(1, 0, 0);
  sep = mk3(0, 0, 0);
  if (active) {
    p0.s = A.pos + B.pos; p0.v = A.pos - B.pos;
    if (mz(p0.v.x) && mz(p0.v.y) && mz(p0.v.z)) p0.v.x += 1e-6f;
    dir = normalized(p0.v * -1.0f);
  }
  while (__ballot(state != MPR_DONE)) {
    if (state == MPR_DONE) continue;
    SupPt q;
    mpr_support<G>(E, A, B, dir, q);
    float dq = dot(q.v, dir);
    bool to_refine = false;   // portal complete or expanded: decide between refinement and the penetration search
    if (state == MPR_FIRST) {
      p1 = q;
      if (dq <= 0) { sep = dir; state = MPR_DONE; }
      else {
        dir = cross(p0.v, p1.v);
        if (dot(dir, dir) < 1e-30f) {   // centre, origin and support point on one line
          pos = p1.s * 0.5f; result = true; state = MPR_DONE;
          if (dot(p1.v, p1.v) < 1e-30f) { depth = 0; dir_out = mk3(0, 0, 0); }
          else { depth = norm(p1.v); dir_out = p1.v * (1.0f / depth); }
        } else { dir = normalized(dir); state = MPR_SECOND; }
      }
    } else if (state == MPR_SECOND) {
      p2 = q;
      if (dq <= 0) { sep = dir; state = MPR_DONE; }
      else {
        dir = normalized(cross(p1.v - p0.v, p2.v - p0.v));
        if (dot(dir, p0.v) > 0) { SupPt t = p1; p1 = p2; p2 = t; dir = dir * -1.0f; }
        state = MPR_DISCOVER; guard = 0;
      }
    } else if (state == MPR_DISCOVER) {
      p3 = q;
      if (dq <= 0) { sep = dir; state = MPR_DONE; }
      else {
        bool cont = false;
        if (dot(cross(p1.v, p3.v), p0.v) < 0) { p2 = p3; cont = true; }
        if (!cont && dot(cross(p3.v, p2.v), p0.v) < 0) { p1 = p3; cont = true; }
        if (cont) { dir = normalized(cross(p1.v - p0.v, p2.v - p0.v)); if (++guard > 64) state = MPR_DONE; }
        else { to_refine = true; guard = 0; }
      }
    } else {
      // refinement and penetration search share "did the portal reach the surface / expand it / new direction":
      // one copy of that code for the groups in either state
      bool penetr = state == MPR_PENETR;
      bool reach = portal_reach_tol(p1, p2, p3, q, dir, tol);
      bool expand = false;
      if (!penetr) {
        if (dq < 0) { sep = dir; state = MPR_DONE; }
        else if (reach) state = MPR_DONE;
        else expand = true;
      } else if (reach || guard > max_iter) {
        // the penetration is found: its depth / direction / position are formed ONCE after the loop from the portal this group keeps (round 6: inside the loop the
        // block ran once per iteration in which some group finished -- the groups finish at different iterations --, ~250 instructions each time for the whole wave)
        finish = true; state = MPR_DONE;
      } else expand = true;
      if (expand) {
        expand_portal(p0, p1, p2, p3, q);
        guard++;
        if (!penetr && guard > 128) state = MPR_DONE;
        else to_refine = true;
      }
    }
    if (to_refine) {   // new portal: its normal is the next direction; during refinement the origin side decides the state
      dir = portal_dir(p1, p2, p3);
      if (state != MPR_PENETR) { if (dot(dir, p1.v) >= 0) { state = MPR_PENETR; guard = 0; } else state = MPR_REFINE; }
    }
  }
  if (finish) { mpr_result(E.plane_depth, p0, p1, p2, p3, dir, depth, dir_out, pos); result = true; }
  return result;
}

// separating-axis test of two oriented boxes (half extents ea, eb; rotations Ra, Rb; centre offset t in world).
// Returns -1 when no axis separates them, otherwise a lower bound (>= 0) on their distance: the gap along
// the separating face normal (unit axes), 0 when only an edge-edge axis separates.
__device__ __forceinline__ float obb_gap(const float* Ra, v3 ea, const float* Rb, v3 eb, v3 tw) {
  float R[9], AR[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    R[3 * i + j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
    AR[3 * i + j] = fabsf(R[3 * i + j]) + 1e-6f;
  }
  v3 t = mulmT(Ra, tw);
  float ta[3] = {t.x, t.y, t.z}, a[3] = {ea.x, ea.y, ea.z}, b[3] = {eb.x, eb.y, eb.z};
  float gap = -1.f;
  for (int i = 0; i < 3; i++) gap = fmaxf(gap, fabsf(ta[i]) - (a[i] + b[0] * AR[3 * i] + b[1] * AR[3 * i + 1] + b[2] * AR[3 * i + 2]));
  for (int j = 0; j < 3; j++) gap = fmaxf(gap, fabsf(ta[0] * R[j] + ta[1] * R[3 + j] + ta[2] * R[6 + j]) - (a[0] * AR[j] + a[1] * AR[3 + j] + a[2] * AR[6 + j] + b[j]));
  if (gap > 0) return gap;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    float ra = a[i1] * AR[3 * i2 + j] + a[i2] * AR[3 * i1 + j];
    float rb = b[j1] * AR[3 * i + j2] + b[j2] * AR[3 * i + j1];
    if (fabsf(ta[i2] * R[3 * i1 + j] - ta[i1] * R[3 * i2 + j]) > ra + rb) return 0.f;
  }
  return -1.f;
}

__device__ __forceinline__ void make_frame(float* f) {
  v3 n = normalized(ld3(f));
  v3 y = (n.y < 0.5f && n.y > -0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
  y = normalized(y - n * dot(n, y));
  st3(f, n); st3(f + 3, y); st3(f + 6, cross(n, y));
}
__device__ __forceinline__ void add_contact(RgLds& s, int pair, float dist, v3 pos, v3 normal, int dim) {
  // called with wave-uniform arguments; lane 0 writes
  int c = s.ncon;
  SYNC();   // every lane holds the count before lane 0 advances it (the cap test below must stay wave-uniform)
  if (c >= RG_MAXCON) { if (LANE == 0) s.status |= RG_STATUS_CON_FULL; return; }
  if (LANE == 0) {
    s.c_dist[c] = dist; st3(s.c_pos + 3 * c, pos); st3(s.c_normal + 3 * c, normalized(normal));
    s.c_pair[c] = pair; s.c_dim[c] = dim; s.ncon = c + 1;
  }
  SYNC();
}

// the two geoms of candidate pair p in pair-local coordinates (origin at geom1's centre: fp32 resolution ~1e-9 m),
// each inflated by margin/2
__device__ __forceinline__ MprEnv rg_mpr_env(RgM m, float* prof, bool cells) {
  MprEnv E;
  E.mesh_vert = m.mesh_vert; E.cell_adr = m.mesh_cell_adr; E.cell_blk = (const rgf4*)m.mesh_cell_blk; E.cell_ovf = (const rgf4*)m.mesh_cell_ovf;
  E.prof = prof; E.cells = cells; E.plane_depth = false;
  return E;
}
__device__ __forceinline__ void rg_mpr_geoms(RgM m, const RgLds& s, int p, float gscale, MprGeom& A, MprGeom& B, int& dim, float& margin) {
  const rgf4* R = (const rgf4*)m.pair_rec + (RG_PAIRREC / 4) * p;
  rgf4 r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3];
  int hdr = __builtin_bit_cast(int, r0.x), g1 = hdr & 255, g2 = (hdr >> 8) & 255;
  dim = (hdr >> 16) & 15; margin = r0.y;
  A.type = (hdr >> 20) & 15; A.quat = s.gquat + 4 * g1; A.size = mk3(r1.x, r1.y, r1.z); A.margin = 0.5f * margin; A.pos = mk3(0, 0, 0);
  B.type = (hdr >> 24) & 15; B.quat = s.gquat + 4 * g2; B.size = mk3(r2.x, r2.y, r2.z); B.margin = 0.5f * margin; B.pos = ld3(s.gpos + 3 * g2) - ld3(s.gpos + 3 * g1);
  if (hdr & RG_PAIR_SCALED1) A.size = A.size * gscale;   // per-env size factor of the flagged geoms (RG_PRM_GEOM_SCALE)
  if (hdr & RG_PAIR_SCALED2) B.size = B.size * gscale;
  A.mesh = __builtin_bit_cast(int, r0.z); A.vertadr = __builtin_bit_cast(int, r3.x); A.nvert = __builtin_bit_cast(int, r1.w);
  B.mesh = __builtin_bit_cast(int, r0.w); B.vertadr = __builtin_bit_cast(int, r3.y); B.nvert = __builtin_bit_cast(int, r2.w);
}
// ------------------------------------------------------------------------------------------------- box - box
// engine_collision_box.c: mjc_BoxBox (oracle: collide_box_box) — multi-point contacts of two boxes: separating-axis search
// over the 15 candidate axes; edge-edge: one contact at the closest points of the two edges; face: the incident face clipped
// against the reference face, one contact per vertex of the clipped polygon that lies within the margin.
// Two pairs per pass, 32 lanes each.  The axis search is scalar per pair (every lane of the group runs it); the clipped
// polygon's vertices are enumerated one candidate per lane — 4 incident corners inside the reference rectangle, 4 reference
// corners inside the incident parallelogram, 16 edge crossings — which is the point set Sutherland-Hodgman produces,
// without its serial polygon lists.  Launch flag bit 4 (round-1 contact variant) sends these pairs through MPR instead.
__device__ __forceinline__ v3 pick3(v3 a0, v3 a1, v3 a2, int k) { return k == 0 ? a0 : (k == 1 ? a1 : a2); }
__device__ __forceinline__ float pick3(float a0, float a1, float a2, int k) { return k == 0 ? a0 : (k == 1 ? a1 : a2); }
// lane l of the group's work on one pair of boxes: half sizes A, B, orientations q1, q2, centre P1 of box 1 and t = centre of box 2 - P1
__device__ __forceinline__ bool box_box_lane(const float* A, const float* B, q4 q1, q4 q2, v3 P1, v3 t, float margin, int l, float& dist, v3& pos, v3& n) {
  float R1[9], R2[9]; q2mat(R1, q1); q2mat(R2, q2);
  v3 ax1[3], ax2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { ax1[k] = mk3(R1[k], R1[3 + k], R1[6 + k]); ax2[k] = mk3(R2[k], R2[3 + k], R2[6 + k]); }
  float ta[3], tb[3], Q[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++) { ta[i] = dot(t, ax1[i]); tb[i] = dot(t, ax2[i]);
#pragma unroll
    for (int j = 0; j < 3; j++) Q[i][j] = fabsf(dot(ax1[i], ax2[j])); }
  float best = -3.0e38f; int code = 0; bool sep = false; n = mk3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; i++) {   // face normals of box 1
    float sv = fabsf(ta[i]) - (A[i] + B[0] * Q[i][0] + B[1] * Q[i][1] + B[2] * Q[i][2]);
    sep = sep || sv > margin;
    if (sv > best) { best = sv; code = 1 + i; n = ax1[i] * (ta[i] < 0 ? -1.f : 1.f); }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {   // face normals of box 2
    float sv = fabsf(tb[j]) - (B[j] + A[0] * Q[0][j] + A[1] * Q[1][j] + A[2] * Q[2][j]);
    sep = sep || sv > margin;
    if (sv > best) { best = sv; code = 4 + j; n = ax2[j] * (tb[j] < 0 ? -1.f : 1.f); }
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {   // edge x edge
      v3 cv = cross(ax1[i], ax2[j]);
      float ln = norm(cv);
      if (ln < 1e-4f) continue;   // parallel edges: covered by the face axes (fp32: below this the direction is rounding noise)
      cv = cv * rg_rcp(ln);
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float ra = A[i1] * fabsf(dot(cv, ax1[i1])) + A[i2] * fabsf(dot(cv, ax1[i2]));
      float rb = B[j1] * fabsf(dot(cv, ax2[j1])) + B[j2] * fabsf(dot(cv, ax2[j2]));
      float tc = dot(t, cv), sv = fabsf(tc) - (ra + rb);
      sep = sep || sv > margin;
      // a face axis is preferred unless the edge axis is clearly better (5 % hysteresis)
      if (sv > best + 0.05f * fabsf(best) + 1e-12f && sv > best) { best = sv; code = 7 + 3 * i + j; n = cv * (tc < 0 ? -1.f : 1.f); }
    }
  }
  if (sep) return false;
  if (code >= 7) {   // edge - edge: one contact midway between the closest points of the two edges
    const int i = (code - 7) / 3, j = (code - 7) - 3 * i;
    v3 pa = mk3(0, 0, 0), pb = t;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k != i) pa = pa + ax1[k] * ((dot(n, ax1[k]) > 0 ? 1.f : -1.f) * A[k]);
      if (k != j) pb = pb + ax2[k] * ((dot(n, ax2[k]) > 0 ? -1.f : 1.f) * B[k]);
    }
    const v3 ui = pick3(ax1[0], ax1[1], ax1[2], i), uj = pick3(ax2[0], ax2[1], ax2[2], j), w = pb - pa;
    const float uaub = dot(ui, uj), q1 = dot(ui, w), q2 = -dot(uj, w), den = 1.f - uaub * uaub;
    float alpha = den > 1e-6f ? (q1 + uaub * q2) / den : 0.f, beta = den > 1e-6f ? (uaub * q1 + q2) / den : 0.f;
    const float Ai = pick3(A[0], A[1], A[2], i), Bj = pick3(B[0], B[1], B[2], j);
    alpha = clampf(alpha, -Ai, Ai); beta = clampf(beta, -Bj, Bj);
    pa = pa + ui * alpha; pb = pb + uj * beta;
    dist = best; pos = (pa + pb) * 0.5f + P1;
    return l == 0;
  }
  // face contact: the reference box owns the axis; nr = its outward normal towards the other box
  const bool ref1 = code <= 3; const int ia = ref1 ? code - 1 : code - 4;
  v3 rax[3], iax[3]; float rs[3], is[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { rax[k] = ref1 ? ax1[k] : ax2[k]; iax[k] = ref1 ? ax2[k] : ax1[k]; rs[k] = ref1 ? A[k] : B[k]; is[k] = ref1 ? B[k] : A[k]; }
  const v3 rp = ref1 ? mk3(0, 0, 0) : t, ip = ref1 ? t : mk3(0, 0, 0), nr = n * (ref1 ? 1.f : -1.f);
  int ib = 0; float bd = 3.0e38f;   // incident face: the one facing the reference face most directly
#pragma unroll
  for (int k = 0; k < 3; k++) { float dk = -fabsf(dot(nr, iax[k])); if (dk < bd) { bd = dk; ib = k; } }
  const v3 iab = pick3(iax[0], iax[1], iax[2], ib);
  const float isgn = dot(nr, iab) > 0 ? -1.f : 1.f;
  const v3 ic = ip + iab * (isgn * pick3(is[0], is[1], is[2], ib));
  const int u = (ia + 1) % 3, v = (ia + 2) % 3, iu = (ib + 1) % 3, iv = (ib + 2) % 3;
  const v3 ru = pick3(rax[0], rax[1], rax[2], u), rv = pick3(rax[0], rax[1], rax[2], v);
  const float lu = pick3(rs[0], rs[1], rs[2], u), lv = pick3(rs[0], rs[1], rs[2], v), la = pick3(rs[0], rs[1], rs[2], ia);
  const v3 eu = pick3(iax[0], iax[1], iax[2], iu) * pick3(is[0], is[1], is[2], iu), ev = pick3(iax[0], iax[1], iax[2], iv) * pick3(is[0], is[1], is[2], iv);
  // the reference face's plane coordinates: incident parallelogram = c2 + a e1 + b e2, |a|, |b| <= 1; reference rectangle |x| <= lu, |y| <= lv
  const v3 rel = ic - rp;
  const float c2x = dot(rel, ru), c2y = dot(rel, rv), e1x = dot(eu, ru), e1y = dot(eu, rv), e2x = dot(ev, ru), e2y = dot(ev, rv);
  float x = 0, y = 0; bool ok = false;
  if (l < 4) {          // incident corner l (order (+,+), (-,+), (-,-), (+,-)) inside the rectangle
    const float su = (l == 0 || l == 3) ? 1.f : -1.f, sv = l < 2 ? 1.f : -1.f;
    x = c2x + su * e1x + sv * e2x; y = c2y + su * e1y + sv * e2y;
    ok = fabsf(x) <= lu && fabsf(y) <= lv;
  } else if (l < 8) {   // reference corner inside the parallelogram
    x = (l & 1) ? lu : -lu; y = (l & 2) ? lv : -lv;
    const float det = e1x * e2y - e1y * e2x, dx = x - c2x, dy = y - c2y;
    if (fabsf(det) > 1e-12f) { const float a = (dx * e2y - dy * e2x) / det, b = (e1x * dy - e1y * dx) / det; ok = fabsf(a) < 1.f && fabsf(b) < 1.f; }
  } else if (l < 24) {  // incident edge (l-8)/4 crossing reference edge line (l-8)%4, within that edge's extent
    const int ea = (l - 8) >> 2, eb = (l - 8) & 3, k0 = ea, k1 = (ea + 1) & 3;
    const float su0 = (k0 == 0 || k0 == 3) ? 1.f : -1.f, sv0 = k0 < 2 ? 1.f : -1.f, su1 = (k1 == 0 || k1 == 3) ? 1.f : -1.f, sv1 = k1 < 2 ? 1.f : -1.f;
    const float ax_ = c2x + su0 * e1x + sv0 * e2x, ay_ = c2y + su0 * e1y + sv0 * e2y, bx_ = c2x + su1 * e1x + sv1 * e2x, by_ = c2y + su1 * e1y + sv1 * e2y;
    const bool xedge = eb < 2; const float sg = (eb & 1) ? -1.f : 1.f, lim = xedge ? lu : lv, olim = xedge ? lv : lu;
    const float da = sg * (xedge ? ax_ : ay_) - lim, db = sg * (xedge ? bx_ : by_) - lim;
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      const float tt = da / (da - db);
      x = ax_ + tt * (bx_ - ax_); y = ay_ + tt * (by_ - ay_);
      ok = fabsf(xedge ? y : x) <= olim;
    }
  }
  if (!ok) return false;
  // lift onto the incident face's plane: X = rp + x ru + y rv + h nr with (X - ic) . inorm = 0
  const v3 base = rp + ru * x + rv * y, inorm = iab * isgn;
  const float dn = dot(nr, inorm), hh = fabsf(dn) > 1e-12f ? dot(ic - base, inorm) / dn : dot(ic - base, nr);
  dist = hh - la;   // signed distance of the incident-face point from the reference face
  if (dist > margin) return false;
  pos = base + nr * (la + 0.5f * dist) + P1;
  return true;
}
__device__ __forceinline__ bool rg_box_box_lane(RgM m, const RgLds& s, int p, float gscale, int l, float& dist, v3& pos, v3& n, int& dim) {
  const rgf4* Rc = (const rgf4*)m.pair_rec + (RG_PAIRREC / 4) * p;
  rgf4 r0 = Rc[0], r1 = Rc[1], r2 = Rc[2];
  const int hdr = __builtin_bit_cast(int, r0.x), g1 = hdr & 255, g2 = (hdr >> 8) & 255;
  dim = (hdr >> 16) & 15;
  const float margin = r0.y;
  const float sc1 = (hdr & RG_PAIR_SCALED1) ? gscale : 1.f, sc2 = (hdr & RG_PAIR_SCALED2) ? gscale : 1.f;
  const float A[3] = {r1.x * sc1, r1.y * sc1, r1.z * sc1}, B[3] = {r2.x * sc2, r2.y * sc2, r2.z * sc2};
  const v3 P1 = ld3(s.gpos + 3 * g1), t = ld3(s.gpos + 3 * g2) - P1;   // pair-local coordinates: origin at box 1's centre
  return box_box_lane(A, B, ldq(s.gquat + 4 * g1), ldq(s.gquat + 4 * g2), P1, t, margin, l, dist, pos, n);
}
RG_STAGE void rg_narrow_boxbox(RgCtx c, int ncand) {
  RgM m = RG_M(c); RgLRef L = RG_L(c); RgLds& s = RG_S();
  const float gscale = rg_prm(m, L)[RG_PRM_GEOM_SCALE];
  const int l = LANE & 31, grp = LANE >> 5;
  for (int cbase = 0; cbase < ncand; cbase += RG_WAVE) {
    bool is_bb = false;
    if (cbase + LANE < ncand) { const int hdr = __builtin_bit_cast(int, m.pair_rec[RG_PAIRREC * s.cand[cbase + LANE]]); is_bb = ((hdr >> 20) & 15) == RG_GEOM_BOX && ((hdr >> 24) & 15) == RG_GEOM_BOX; }
    unsigned long long bits = __ballot(is_bb);
    while (bits) {
      const int ci0 = cbase + __builtin_ctzll(bits); bits &= bits - 1;
      int ci1 = -1; if (bits) { ci1 = cbase + __builtin_ctzll(bits); bits &= bits - 1; }
      const int ci = grp == 0 ? ci0 : ci1;
      bool hit = false; float dist = 0; v3 pos = mk3(0, 0, 0), nrm = mk3(0, 0, 1); int p = 0, dim = 3;
      if (ci >= 0) { p = s.cand[ci]; hit = rg_box_box_lane(m, s, p, gscale, l, dist, pos, nrm, dim); }
      unsigned long long bal = __ballot(hit);
      int cb = s.ncon;
      SYNC();
      if (hit) {
        int cc = cb + __popcll(bal & ((1ull << LANE) - 1ull));
        if (cc < RG_MAXCON) { s.c_dist[cc] = dist; st3(s.c_pos + 3 * cc, pos); st3(s.c_normal + 3 * cc, normalized(nrm)); s.c_pair[cc] = p; s.c_dim[cc] = dim; }
        else s.status |= RG_STATUS_CON_FULL;
      }
      if (LANE == 0) { int nn = cb + __popcll(bal); s.ncon = nn < RG_MAXCON ? nn : RG_MAXCON; }
      SYNC();
    }
  }
}
template <int G> RG_STAGE void rg_narrow_phase1(RgCtx c, int ncand) {
  RgM m = RG_M(c); RgLRef L = RG_L(c); RgLds& s = RG_S();
  bool cells = !(L.flags & 8);
  rgf4* sepdir = L.bt.sepdir ? (rgf4*)L.bt.sepdir + (size_t)rg_env(L) * m.npair : (rgf4*)0;
  float* pairlb = (L.bt.pairlb && !(L.flags & 4)) ? L.bt.pairlb + (size_t)rg_env(L) * m.npair : (float*)0;
  const float gscale = rg_prm(m, L)[RG_PRM_GEOM_SCALE];
  MprEnv E = rg_mpr_env(m, (float*)0, cells);
  const bool boxbox = !(L.flags & 16);   // box-box pairs have their own routine (rg_narrow_boxbox)
  for (int base = 0; base < ncand; base += RG_WAVE / G) {
    int ci = base + LANE / G;
    bool keep = false; int p = 0;
    if (ci < ncand) {
      p = s.cand[ci];
      MprGeom A, B; int dim; float margin;
      rg_mpr_geoms(m, s, p, gscale, A, B, dim, margin);
      if (A.type != RG_GEOM_PLANE && !(boxbox && A.type == RG_GEOM_BOX && B.type == RG_GEOM_BOX)) {
        v3 c0 = A.pos - B.pos;
        if (mz(c0.x) && mz(c0.y) && mz(c0.z)) c0.x += 1e-6f;
        v3 dir = normalized(c0 * -1.0f);
        if (sepdir) { rgf4 cd; const float* sp_ = (const float*)(sepdir + p); cd.x = RG_ROW_LD(sp_); cd.y = RG_ROW_LD(sp_ + 1); cd.z = RG_ROW_LD(sp_ + 2); cd.w = 0.f; if (cd.x * cd.x + cd.y * cd.y + cd.z * cd.z > 0.5f) dir = mk3(cd.x, cd.y, cd.z); }  // last substep's separating direction first
        SupPt p1; mpr_support<G>(E, A, B, dir, p1);
        float d = dot(p1.v, dir);
        keep = d > 0;
        // separated by -d along dir: a lower bound on the distance of the inflated shapes
        if (!keep && pairlb && (LANE & (G - 1)) == 0) RG_ROW_ST(pairlb + p, fmaxf(-d - 1e-6f, 0.f));
      }
    }
    bool lead = keep && (LANE & (G - 1)) == 0;
    unsigned long long bal = __ballot(lead);
    int cbase = s.ncand2;
    SYNC();
    if (lead) {
      int slot = cbase + __popcll(bal & ((1ull << LANE) - 1ull));
      if (slot < RG_MAXCAND2) s.cand2[slot] = p; else s.status |= RG_STATUS_CAND_FULL;
    }
    if (LANE == 0) { int n = cbase + __popcll(bal); s.ncand2 = n < RG_MAXCAND2 ? n : RG_MAXCAND2; }
    SYNC();
  }
}
template <int G> RG_STAGE_BIG void rg_narrow_phase2(RgCtx c, int ncand2) {
  RgM m = RG_M(c); RgLRef L = RG_L(c); RgLds& s = RG_S();
  bool cells = !(L.flags & 8);
  float* prof = (L.flags & 2) ? s.prof : (float*)0;
  rgf4* sepdir = L.bt.sepdir ? (rgf4*)L.bt.sepdir + (size_t)rg_env(L) * m.npair : (rgf4*)0;
  const float gscale = rg_prm(m, L)[RG_PRM_GEOM_SCALE];
  MprEnv E = rg_mpr_env(m, prof, cells);
  E.plane_depth = (L.flags & 16) != 0;
  for (int base = 0; base < ncand2; base += RG_WAVE / G) {
    int ci = base + LANE / G;
    bool hit = false;
    float depth = 0, margin = 0; v3 dir = mk3(0, 0, 0), pos = mk3(0, 0, 0); int p = 0, dim = 3;
    bool active = ci < ncand2;
    MprGeom A, B;
    if (active) {
      p = s.cand2[ci];
      rg_mpr_geoms(m, s, p, gscale, A, B, dim, margin);
    }
    v3 sep;
    hit = rg_mpr<G>(E, A, B, m.mpr_iterations, m.mpr_tolerance, depth, dir, pos, sep, active);
    if (active) {
      if (sepdir && (LANE & (G - 1)) == 0) { rgf4 cd; cd.x = hit ? 0.f : sep.x; cd.y = hit ? 0.f : sep.y; cd.z = hit ? 0.f : sep.z; cd.w = 0.f; RG_SEPDIR_ST(sepdir + p, cd); }
      hit = hit && dot(dir, dir) > 0.25f;
      pos = pos + ld3(s.gpos + 3 * (m.pair_gg[p] & 255));
    }
    // append the groups' contacts in candidate order (group leaders hold the result)
    bool lead = hit && (LANE & (G - 1)) == 0;
    unsigned long long bal = __ballot(lead);
    int cbase = s.ncon;
    SYNC();
    if (lead) {
      int c = cbase + __popcll(bal & ((1ull << LANE) - 1ull));
      if (c < RG_MAXCON) {
        s.c_dist[c] = margin - depth; st3(s.c_pos + 3 * c, pos); st3(s.c_normal + 3 * c, normalized(dir));
        s.c_pair[c] = p; s.c_dim[c] = dim;
      } else s.status |= RG_STATUS_CON_FULL;
    }
    if (LANE == 0) { int n = cbase + __popcll(bal); s.ncon = n < RG_MAXCON ? n : RG_MAXCON; }
    SYNC();
  }
}
__device__ __forceinline__ void rg_collision(RgCtx c, RgM m, RgLds& s, const float* P, float* prof, rgf4* sepdir, float* pairlb, bool cells) {
  long long tb0 = rg_clock();
  const float gscale = P[RG_PRM_GEOM_SCALE];
  if (LANE == 0) { s.ncand = 0; s.ncon = 0; }
  SYNC();
  // Broadphase over the static pair list.
  // Pass 1 (every pair, a few instructions): the cached lower bound on the pair's distance minus this
  // substep's motion bound (velocity stage: gspeed) — pairs that are still certainly apart stop here.
  // Pass 2 (pairs whose bound ran out, compacted so that only full rows pay for it): bounding spheres,
  // then oriented boxes; both conservative, both refresh the bound when they separate the pair.
  int nround = (m.npair + RG_WAVE - 1) / RG_WAVE;
  float hb = 1.5f * P[RG_PRM_TIMESTEP];
  int nt = 0; bool bbany = false, plany = false;   // (bbany: this lane queued a pair of two boxes — those have their own narrowphase routine)
  int gg_next[4]; float lb_next[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { int p = k * RG_WAVE + LANE, pc = p < m.npair ? p : m.npair - 1; gg_next[k] = m.pair_gg[pc]; lb_next[k] = pairlb ? RG_ROW_LD(pairlb + pc) : 0.f; }
  for (int r0 = 0; r0 < nround; r0 += 4) {   // four rounds per trip: eight independent loads in flight per lane, requested one trip ahead (a trip only writes the bounds of its own pairs)
    int gg[4]; float lbv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { gg[k] = gg_next[k]; lbv[k] = lb_next[k]; }
    if (r0 + 4 < nround) {
#pragma unroll
      for (int k = 0; k < 4; k++) { int p = (r0 + 4 + k) * RG_WAVE + LANE, pc = p < m.npair ? p : m.npair - 1; gg_next[k] = m.pair_gg[pc]; lb_next[k] = pairlb ? RG_ROW_LD(pairlb + pc) : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int p = (r0 + k) * RG_WAVE + LANE;
      bool need = false;
      if (p < m.npair) {
        int g1 = gg[k] & 255, g2 = gg[k] >> 8;
        float lb = lbv[k] - (hb * (s.gspeed[g1] + s.gspeed[g2]) + 1e-7f);
        need = !(lb > 0.f);
        if (!need) RG_ROW_ST(pairlb + p, lb);
      }
      unsigned long long nb = __ballot(need);
      if (need) s.tlist[nt + __popcll(nb & ((1ull << LANE) - 1ull))] = (short)p;
      nt += __popcll(nb);
    }
    if (nt <= RG_TLIST - 4 * RG_WAVE && r0 + 4 < nround) continue;
    SYNC();
#ifdef RG_FINE_PROF
    const long long tref0 = rg_clock();
    if (prof && LANE == 0) { prof[42] += (float)nt; prof[43] += (float)((nt + RG_WAVE - 1) / RG_WAVE); }
#endif
    for (int i0 = 0; i0 < nt; i0 += RG_WAVE) {
      int i = i0 + LANE, q = 0;
      bool hit = false, isbb = false, ispl = false;
      if (i < nt) {
        q = s.tlist[i];
        const rgf4* R = (const rgf4*)m.pair_rec + (RG_PAIRREC / 4) * q;
        rgf4 r0 = R[0], r3 = R[3], r4 = R[4], r5 = R[5];
        int hdr = __builtin_bit_cast(int, r0.x), g1 = hdr & 255, g2 = (hdr >> 8) & 255;
        float margin = r0.y, newlb = 0.f;
        const float sc1 = (hdr & RG_PAIR_SCALED1) ? gscale : 1.f, sc2 = (hdr & RG_PAIR_SCALED2) ? gscale : 1.f;
        r3.z *= sc1; r3.w *= sc2;
        v3 p1 = ld3(s.gpos + 3 * g1), p2 = ld3(s.gpos + 3 * g2), dif = p2 - p1;
        isbb = ((hdr >> 20) & 255) == (RG_GEOM_BOX | (RG_GEOM_BOX << 4)); ispl = ((hdr >> 20) & 15) == RG_GEOM_PLANE;
        if (((hdr >> 20) & 15) == RG_GEOM_PLANE) {
          float d = dot(dif, qrot(ldq(s.gquat + 4 * g1), mk3(0, 0, 1))) - (r3.w + margin);
          hit = d <= 0; newlb = fmaxf(d, 0.f);
        } else {
          float d = sqrtf(dot(dif, dif)) - (r3.z + r3.w + margin);
          if (d <= 0) {
            float hm = 0.5f * margin + 1e-6f;
            v3 ea = mk3(r4.x * sc1 + hm, r4.y * sc1 + hm, r4.z * sc1 + hm), eb = mk3(r5.x * sc2 + hm, r5.y * sc2 + hm, r5.z * sc2 + hm);
            float Ra[9], Rb[9]; q2mat(Ra, ldq(s.gquat + 4 * g1)); q2mat(Rb, ldq(s.gquat + 4 * g2));
            float og = obb_gap(Ra, ea, Rb, eb, dif);
            hit = og < 0; newlb = fmaxf(og, 0.f);
          } else newlb = d;
        }
        if (pairlb) RG_ROW_ST(pairlb + q, fmaxf(newlb - 1e-6f, 0.f));
      }
      unsigned long long bal = __ballot(hit);
      bbany = bbany || (hit && isbb); plany = plany || (hit && ispl);
      int base = s.ncand;
      SYNC();
      if (hit) {
        int slot = base + __popcll(bal & ((1ull << LANE) - 1ull));
        if (slot < RG_MAXCAND) s.cand[slot] = q; else s.status |= RG_STATUS_CAND_FULL;
      }
      if (LANE == 0) { int n = base + __popcll(bal); s.ncand = n < RG_MAXCAND ? n : RG_MAXCAND; }
      SYNC();
    }
#ifdef RG_FINE_PROF
    if (prof && LANE == 0) prof[44] += (float)(rg_clock() - tref0);
#endif
    nt = 0;
  }
  if (prof && LANE == 0) prof[5] += (float)(rg_clock() - tb0);
  // narrowphase.  Convex pairs run in groups of G lanes, 64/G queries per wave (the portal algebra is scalar
  // per query; a whole wave per query would execute it 64-fold redundantly).  Wider groups scan a hull
  // faster, narrower groups run more queries at once (G = 16 / 8 / 4: 4 / 8 / 16 per wave): picked by queue length;
  // with the cell lists a support is ~4 records, so the one-support phase 1 runs in quads once the queue is long.
  // Phase 1 (uniform cost): one support test per candidate, along the cached separating direction of the
  // pair or else the centre line — "separated along that direction" rejects most box-overlapping
  // neighbours.  Survivors are compacted in order so that phase 2 (full MPR) only runs groups of
  // genuinely close pairs.
  int ncand = s.ncand;
  if (LANE == 0) s.ncand2 = 0;
  SYNC();
  if (ncand > 16) rg_narrow_phase1<2>(c, ncand); else rg_narrow_phase1<4>(c, ncand);   // pairs of lanes once the queue is long: 32 one-support tests per trip
  int ncand2 = s.ncand2;
  if (prof && LANE == 0) { prof[16] += (float)(rg_clock() - tb0); prof[17] += ncand; prof[18] += ncand2; }
  rg_narrow_phase2<4>(c, ncand2);   // quads whatever the queue length: a support's first four candidates are one load round, 16 queries per trip (measured against 8- and 16-lane groups for short queues: quads win everywhere, profiles/r02_ab.txt)   // (quads: a support's first four candidates are one load round, and a crowded queue gets through in half the trips)
  if (prof && LANE == 0) prof[19] += (float)(rg_clock() - tb0);
  if (!(RG_L(c).flags & 16) && __ballot(bbany) != 0) rg_narrow_boxbox(c, ncand);
  // plane pairs (rare: something near the floor), whole wave cooperating, one pair at a time.  Which candidates are plane
  // pairs is found lane-parallel from the pair headers (a serial scan of ~25 candidates cost two dependent loads each).
  if (__ballot(plany) != 0)   // (no plane pair among the candidates: nothing to look for)
  for (int cbase = 0; cbase < ncand; cbase += RG_WAVE) {
  bool is_plane = false;
  if (cbase + LANE < ncand) is_plane = ((__builtin_bit_cast(int, m.pair_rec[RG_PAIRREC * s.cand[cbase + LANE]]) >> 20) & 15) == RG_GEOM_PLANE;
  unsigned long long plane_bits = __ballot(is_plane);
  while (plane_bits) {
    int ci = cbase + __builtin_ctzll(plane_bits); plane_bits &= plane_bits - 1;
    int p = s.cand[ci];
    int g1 = m.pair_geom[3 * p], g2 = m.pair_geom[3 * p + 1], dim = m.pair_geom[3 * p + 2];
    float margin = m.pair_prm[12 * p];
    int t2 = m.geom_type[g2];
    v3 p1 = ld3(s.gpos + 3 * g1), p2 = ld3(s.gpos + 3 * g2);
    MprGeom B;
    B.type = t2; B.quat = s.gquat + 4 * g2; B.size = ld3(m.geom_size + 3 * g2); B.margin = 0; B.mesh = -1; B.vertadr = 0; B.nvert = 0;
    if (__builtin_bit_cast(int, m.pair_rec[RG_PAIRREC * p]) & RG_PAIR_SCALED2) B.size = B.size * gscale;
    if (t2 == RG_GEOM_MESH) { B.mesh = m.geom_dataid[g2]; B.vertadr = m.mesh_vertadr[B.mesh]; B.nvert = m.mesh_vertnum[B.mesh]; }
    v3 n = qrot(ldq(s.gquat + 4 * g1), mk3(0, 0, 1));
    if (t2 == RG_GEOM_BOX) {
      int cnt = 0;
      for (int i = 0; i < 8 && cnt < 4; i++) {
        v3 lc = mk3((i & 1) ? B.size.x : -B.size.x, (i & 2) ? B.size.y : -B.size.y, (i & 4) ? B.size.z : -B.size.z);
        v3 c = qrot(ldq(B.quat), lc) + p2;
        float dist = dot(c - p1, n);
        if (dist > margin) continue;
        add_contact(s, p, dist, c - n * (0.5f * dist), n, dim); cnt++;
      }
    } else {
      B.pos = p2 - p1;
      v3 sp = rg_support<64>(rg_mpr_env(m, (float*)0, false), B, n * -1.0f);   // whole wave on one hull: full scan
      float dist = dot(sp, n);
      if (dist <= margin) add_contact(s, p, dist, sp + p1 - n * (0.5f * dist), n, dim);
    }
  }
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------------- velocity stage
__device__ __forceinline__ void rg_velocity(RgM m, RgLds& s, const float* P, const uint32_t* dof_velmask, const int* subtree_adr, const int* subtree) {
  // cdof_dot: spatial velocity accumulated over the dofs "before" d on its chain, crossed with cdof
  PFOR(d, m.nv) {
    float cv[6] = {0, 0, 0, 0, 0, 0};
    for (int h = 0; h < 2; h++) {
      uint32_t bits = dof_velmask[2 * d + h];
      while (bits) { int a = __builtin_ctz(bits) + 32 * h; bits &= bits - 1; float q = s.qvel[a]; for (int e = 0; e < 6; e++) cv[e] += s.cdof[6 * a + e] * q; }
    }
    int j = m.dof_jntid[d];
    if (m.jnt_type[j] == RG_JNT_FREE && d - m.jnt_dofadr[j] < 3) { for (int e = 0; e < 6; e++) s.cdofdot[6 * d + e] = 0; }
    else cross_motion(s.cdofdot + 6 * d, cv, s.cdof + 6 * d);
  }
  PFOR(t, m.ntendon) {
    float v = 0;
    for (int e = 0; e < 4; e++) { int d = m.ten_dofs[4 * t + e]; if (d >= 0) v += s.tenJ[4 * t + e] * s.qvel[d]; }
    s.tenvel[t] = v;
    s.tenfrc[t] = m.tendon_stiffness[t] * (m.tendon_lengthspring[t] - s.tenlen[t]) - m.tendon_damping[t] * v;
  }
  SYNC();
  // body velocities / bias accelerations by chain gathers, then body forces
  PFOR(k, m.body_geomnum[0]) s.gspeed[m.body_geomadr[0] + k] = 0;
  for (int b = 1 + LANE; b < m.nbody; b += RG_WAVE) {
    float cv[6] = {0, 0, 0, 0, 0, 0}, ca[6] = {0, 0, 0, -P[RG_PRM_GRAVITY], -P[RG_PRM_GRAVITY + 1], -P[RG_PRM_GRAVITY + 2]};
    for (int h = 0; h < 2; h++) {
      uint32_t bits = m.body_dofmask[2 * b + h];
      while (bits) {
        int a = __builtin_ctz(bits) + 32 * h; bits &= bits - 1; float q = s.qvel[a];
        for (int e = 0; e < 6; e++) { cv[e] += s.cdof[6 * a + e] * q; ca[e] += s.cdofdot[6 * a + e] * q; }
      }
    }
    {  // speed bound of the body's geoms: |v(geom centre)| + |omega| * bounding radius (radius of a size-scaled geom: conservative factor for all)
      const float gsmax = fmaxf(P[RG_PRM_GEOM_SCALE], 1.f);
      v3 w = mk3(cv[0], cv[1], cv[2]), vo = mk3(cv[3], cv[4], cv[5]), og = ld3(s.org + 3 * s.b2org[b]);
      float wn = norm(w);
      int ga = m.body_geomadr[b], gn = m.body_geomnum[b];
      for (int k = 0; k < gn; k++) s.gspeed[ga + k] = norm(vo + cross(w, ld3(s.gpos + 3 * (ga + k)) - og)) + wn * (m.geom_rbound[ga + k] * gsmax);
    }
    float t1[6], t2[6], t3[6];
    mul_inert_vec(t1, s.cinert + 10 * b, ca);
    mul_inert_vec(t2, s.cinert + 10 * b, cv);
    cross_force(t3, cv, t2);
    for (int e = 0; e < 6; e++) s.cacc[6 * b + e] = t1[e] + t3[e];  // per-body force before subtree accumulation
  }
  SYNC();
  {  // subtree sums in place: every lane gathers its (<= RG_MAXBODY * 6 / 64 = 3) sums first, then the wave overwrites
    float acc[(RG_MAXBODY * 6 + RG_WAVE - 1) / RG_WAVE];
#pragma unroll
    for (int k0 = 0; k0 < (RG_MAXBODY * 6 + RG_WAVE - 1) / RG_WAVE; k0++) {
      int w = LANE + RG_WAVE * k0, b = w / 6, k = w - 6 * b; float a = 0;
      if (w < m.nbody * 6 && b > 0) for (uint32_t bits = (uint32_t)m.subtree_mask[b]; bits; bits &= bits - 1) a += s.cacc[6 * __builtin_ctz(bits) + k];
      acc[k0] = a;
    }
    SYNC();
#pragma unroll
    for (int k0 = 0; k0 < (RG_MAXBODY * 6 + RG_WAVE - 1) / RG_WAVE; k0++) { int w = LANE + RG_WAVE * k0; if (w < m.nbody * 6) s.cacc[w] = acc[k0]; }
  }
  SYNC();
  PFOR(d, m.nv) {
    const rgf4 dr0 = ((const rgf4*)m.dof_rec)[2 * d]; const float spring_ref = m.dof_rec[8 * d + 4];   // body, tendon entries, joint stiffness, qposadr | spring reference
    const int w0 = __builtin_bit_cast(int, dr0.x);
    const float *c = s.cdof + 6 * d, *f = s.cacc + 6 * (w0 & 255);
    s.qfrc_bias[d] = c[0] * f[0] + c[1] * f[1] + c[2] * f[2] + c[3] * f[3] + c[4] * f[4] + c[5] * f[5];
    float pas = -P[RG_PRM_DOF_DAMPING + d] * s.qvel[d];
    if (dr0.z != 0) pas -= dr0.z * (s.qpos[__builtin_bit_cast(int, dr0.w)] - spring_ref);
    for (int q = (w0 >> 8) & 255; q < ((w0 >> 16) & 255); q++) { int tt = m.dof_ten[2 * q], sl = m.dof_ten[2 * q + 1]; pas += s.tenJ[4 * tt + sl] * s.tenfrc[tt]; }
    s.qfrc_passive[d] = pas;
  }
  SYNC();
}

// PID actuators (mjpid.pyx semantics, see oracle ro_fwd_actuation); updates controller state
static_assert(RG_MAXU <= RG_WAVE && RG_MAXNV <= RG_WAVE, "controller state / warm start: one lane per actuator / dof");
__device__ __forceinline__ void rg_pid(RgM m, RgLds& s, const float* P, float* st /* this lane's actuator: integral, previous error, filtered derivative */) {
  float dt = P[RG_PRM_TIMESTEP];
  PFOR(u, m.nu) {
    const float* gp = P + RG_PRM_ACT_GAINPRM + 10 * u;
    float force;
    float lo = P[RG_PRM_ACT_FORCERANGE + 2 * u], hi = P[RG_PRM_ACT_FORCERANGE + 2 * u + 1];
    if (m.actuator_biastype[u] == 2) {
      float err = s.ctrl[u] - s.actlen[u];
      if (fabsf(err) < gp[5]) err = 0;
      float integ = clampf(st[0] + err * dt, -gp[2], gp[2]);
      float deriv = (1 - gp[4]) * st[2] + gp[4] * (err - st[1]) * rg_rcp(dt);
      force = gp[0] * (err + (gp[1] != 0 ? integ * rg_rcp(gp[1]) : 0.f) + gp[3] * deriv);
      st[0] = integ; st[1] = err; st[2] = deriv;
      if (lo != 0 || hi != 0) force = clampf(force, lo, hi);
    } else {
      float c = s.ctrl[u];
      if (m.actuator_ctrllimited[u]) c = clampf(c, P[RG_PRM_ACT_CTRLRANGE + 2 * u], P[RG_PRM_ACT_CTRLRANGE + 2 * u + 1]);
      force = gp[0] * c;
    }
    if (m.actuator_forcelimited[u]) force = clampf(force, lo, hi);
    s.actfrc[u] = force;
  }
  SYNC();
}
__device__ __forceinline__ void rg_smooth(RgM m, RgLds& s, const float* P) {
  PFOR(d, m.nv) {
    float f = 0;
    const int aw = __builtin_bit_cast(int, m.dof_rec[8 * d + 1]);
    for (int q = aw & 255; q < ((aw >> 8) & 255); q++) {
      int u = m.dof_act[2 * q], sl = m.dof_act[2 * q + 1];
      f += m.actuator_gear[u] * (sl < 0 ? 1.0f : s.tenJ[4 * m.actuator_trnid[u] + sl]) * s.actfrc[u];
    }
    s.qfrc_act[d] = f;
    float v = s.qfrc_passive[d] - s.qfrc_bias[d] + f + (s.has_xfrc ? s.qfrc_smooth[d] : 0.f);   // (+ J' xfrc_applied, left there by rg_com_pos)
    s.qfrc_smooth[d] = v; s.qacc_smooth[d] = v;
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------------- constraints
__device__ __forceinline__ float impedance(const float* si, float pos, float margin) {
  float dmin = clampf(si[0], 1e-4f, 0.9999f), dmax = clampf(si[1], 1e-4f, 0.9999f), width = fmaxf(si[2], 1e-15f);
  float mid = clampf(si[3], 1e-4f, 0.9999f), power = fmaxf(si[4], 1.0f);
  if (dmin == dmax || width <= 1e-15f) return 0.5f * (dmin + dmax);
  float x = fabsf((pos - margin) * rg_rcp(width));
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  float y;
  if (power == 1) y = x;
  else if (power == 2) y = x <= mid ? x * x * rg_rcp(mid) : 1 - (1 - x) * (1 - x) * rg_rcp(1 - mid);   // MuJoCo's default solimp: no powf
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
  else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
__device__ __forceinline__ void kb(float timestep, const float* solref, const float* solimp, float& K, float& B) {
  float dmax = clampf(solimp[1], 1e-4f, 0.9999f);
  if (solref[0] > 0) {
    float tc = fmaxf(solref[0], 2 * timestep), dr = solref[1];
    K = rg_rcp(fmaxf(1e-15f, dmax * dmax * tc * tc * dr * dr)); B = 2.0f * rg_rcp(fmaxf(1e-15f, dmax * tc));
  } else { K = -solref[0] / fmaxf(1e-15f, dmax * dmax); B = -solref[1] / fmaxf(1e-15f, dmax); }
}
// static row slot layout: [friction dofs][friction tendons][joint limits x2][tendon limits x2].
// Each slot has a packed descriptor (RowRegs.desc, derived by the owner lane at the start of a solve): bits 0-5 compact dof
// (single-dof rows), bits 6-10 tendon id (31: none), bit 11: negative sign.  Tendon rows take their (<=4) compact dofs from
// s.ten_cdof and their coefficients from s.tenJ.
__device__ __forceinline__ int nsrow(RgM m) { return m.nfric_dof + m.nfric_ten + 2 * m.nlim_jnt + 2 * m.nlim_ten; }
__device__ __forceinline__ void rg_build_row_desc(RgM m, RgLds& s) {
  int ns = nsrow(m);
  (void)ns;
  PFOR(w, m.ntendon * 4) { int d = m.ten_dofs[w]; s.ten_cdof[w] = (unsigned char)(d >= 0 ? m.d2c[d] : 255); }
  PFOR(i, m.nvc) { s.c2d[i] = (unsigned char)m.c2d[i]; s.cblk[i] = m.c_blk[i]; }
  PFOR(b, m.nbody) s.b2org[b] = (unsigned char)m.body_orgslot[b];
  SYNC();
}
// J_r . x for a static row; x is indexed by compact dof unless `full` (then through c2d)
template <bool FULL> __device__ __forceinline__ float srow_dot(const RgLds& s, int desc, const float* x) {
  int t = (desc >> 6) & 31; float v;
  if (t == 31) { int d = desc & 63; v = x[FULL ? s.c2d[d] : d]; }
  else { v = 0; for (int e = 0; e < 4; e++) { int d = s.ten_cdof[4 * t + e]; if (d != 255) v += s.tenJ[4 * t + e] * x[FULL ? s.c2d[d] : d]; } }
  return (desc >> 11) & 1 ? -v : v;
}
// dst[compact dofs] += coef * J_r  (LDS atomics: several rows may touch one dof)
__device__ __forceinline__ void srow_scatter(RgLds& s, int desc, float coef, float* dst) {
  int t = (desc >> 6) & 31;
  if ((desc >> 11) & 1) coef = -coef;
  if (t == 31) { atomicAdd(dst + (desc & 63), coef); return; }
  for (int e = 0; e < 4; e++) { int d = s.ten_cdof[4 * t + e]; if (d != 255) atomicAdd(dst + d, coef * s.tenJ[4 * t + e]); }
}
// H += D * J_r^T J_r
__device__ __forceinline__ void srow_hess(RgM m, RgLds& s, int desc, float D) {
  int t = (desc >> 6) & 31;
  if (t == 31) { int d = desc & 63; atomicAdd(s.H + d * m.hs + d, D); return; }
  for (int a = 0; a < 4; a++) { int da = s.ten_cdof[4 * t + a]; if (da == 255) continue;
    for (int b = 0; b < 4; b++) { int db = s.ten_cdof[4 * t + b]; if (db == 255 || db > da) continue; atomicAdd(s.H + da * m.hs + db, D * s.tenJ[4 * t + a] * s.tenJ[4 * t + b]); } }   // (lower triangle: what the factorisation reads)
}
// the same in the per-tree block layout, lower triangle only (Hessian with the pattern of M)
__device__ __forceinline__ void srow_hess_tree(RgLds& s, int desc, float D) {
  int t = (desc >> 6) & 31;
  if (t == 31) { int d = desc & 63, blk = s.cblk[d]; atomicAdd(s.H + (blk & 0xFFFF) + d - ((blk >> 16) & 255), D); return; }
  for (int a = 0; a < 4; a++) { int da = s.ten_cdof[4 * t + a]; if (da == 255) continue;
    int blk = s.cblk[da];
    for (int b = 0; b < 4; b++) { int db = s.ten_cdof[4 * t + b]; if (db == 255 || db > da) continue; atomicAdd(s.H + (blk & 0xFFFF) + db - ((blk >> 16) & 255), D * s.tenJ[4 * t + a] * s.tenJ[4 * t + b]); } }
}
__device__ __forceinline__ int npyr(int dim) { return dim == 1 ? 1 : 2 * (dim - 1); }
__device__ __forceinline__ int nbasis(int dim) { return dim >= 4 ? 4 : (dim == 1 ? 1 : 3); }

__device__ __forceinline__ void rg_make_constraint(RgM m, RgLds& s, const float* P) {
  // contact basis Jacobians on the merged dof chains, packed into a shared pool (rows x nnz per contact)
  int ncon = s.ncon;
  // (bodies and the union of their dof masks come from the pair record: one load latency instead of pair -> geom -> body -> mask)
  PFOR(c, ncon) {
    const float* R = m.pair_rec + RG_PAIRREC * s.c_pair[c];
    s.c_nnz[c] = __popc(__builtin_bit_cast(uint32_t, R[23])) + __popc(__builtin_bit_cast(uint32_t, R[24]));
  }
  SYNC();
  {
    int c = LANE, off = 0, sz = 0;
    if (c < ncon) { for (int q = 0; q < c; q++) off += nbasis(s.c_dim[q]) * s.c_nnz[q]; sz = nbasis(s.c_dim[c]) * s.c_nnz[c]; s.c_off[c] = off; }
    int firstbad = wave_min_i((c < ncon && off + sz > RG_CPOOL) ? c : 0x7fffffff);
    if (firstbad < ncon) { ncon = firstbad; if (LANE == 0) { s.ncon = ncon; s.status |= RG_STATUS_CON_FULL; } }
  }
  SYNC();
  for (int w = LANE; w < ncon * RG_W; w += RG_WAVE) {
    int c = w / RG_W, sl = w - c * RG_W, p = s.c_pair[c], nnz = s.c_nnz[c];
    if (sl >= nnz) continue;
    const float* R = m.pair_rec + RG_PAIRREC * p;
    int bb = __builtin_bit_cast(int, R[19]), b1 = bb & 255, b2 = bb >> 8;
    uint32_t lo = __builtin_bit_cast(uint32_t, R[23]), hi = __builtin_bit_cast(uint32_t, R[24]);
    int k = sl; uint32_t bits = lo; int base = 0;   // sl-th set bit of (hi:lo)
    if (k >= __popc(lo)) { k -= __popc(lo); bits = hi; base = 32; }
    for (int q = 0; q < k; q++) bits &= bits - 1;
    int d = base + __builtin_ctz(bits);
    v3 pos = ld3(s.c_pos + 3 * c);
    v3 jp = mk3(0, 0, 0), jr = mk3(0, 0, 0);
    if (in_chain(m, b2, d)) { jp = jp + jac_col(s, d, pos - ld3(s.org + 3 * s.b2org[b2])); jr = jr + ld3(s.cdof + 6 * d); }
    if (in_chain(m, b1, d)) { jp = jp - jac_col(s, d, pos - ld3(s.org + 3 * s.b2org[b1])); jr = jr - ld3(s.cdof + 6 * d); }
    float f[9]; st3(f, ld3(s.c_normal + 3 * c)); make_frame(f);   // contact frame: normal, two tangents (mju_makeFrame)
    s.c_idx[c * RG_W + sl] = (unsigned char)m.d2c[d];
    float* Bc = s.c_pool + s.c_off[c]; int nb = nbasis(s.c_dim[c]);
    Bc[sl] = dot(ld3(f), jp);
    if (nb >= 3) { Bc[nnz + sl] = dot(ld3(f + 3), jp); Bc[2 * nnz + sl] = dot(ld3(f + 6), jp); }
    if (nb >= 4) Bc[3 * nnz + sl] = dot(ld3(f), jr);
  }
  SYNC();
  // contact parameters: D (shared by the pyramid), friction coefficients, reference accelerations
  PFOR(c, ncon) {
    int p = s.c_pair[c], dim = s.c_dim[c];
    const float* prm = m.pair_prm + 12 * p;
    const int hdr = __builtin_bit_cast(int, m.pair_rec[RG_PAIRREC * p]), bb = __builtin_bit_cast(int, m.pair_rec[RG_PAIRREC * p + 19]), b1 = bb & 255, b2 = bb >> 8;
    float tran = P[RG_PRM_BODY_INVWEIGHT0 + 2 * b1] + P[RG_PRM_BODY_INVWEIGHT0 + 2 * b2];
    float includemargin = prm[0] - prm[1], dist = s.c_dist[c];
    // solref / solimp of the pair from its geoms' (the env's own rows: GeomSolrefRandomizer / GeomSolimpRandomizer), mixed by solmix;
    // friction: element-wise max of its geoms' (engine_collision_driver: mj_contactParam)
    const int pg1 = hdr & 255, pg2 = (hdr >> 8) & 255;
    const float mix = m.pair_mix[p];
    const float* r1 = P + RG_PRM_GEOM_SOLREF + 2 * pg1; const float* r2 = P + RG_PRM_GEOM_SOLREF + 2 * pg2;
    const float* i1 = P + RG_PRM_GEOM_SOLIMP + 5 * pg1; const float* i2 = P + RG_PRM_GEOM_SOLIMP + 5 * pg2;
    float solref[2], solimp[5];
    if (r1[0] > 0 && r2[0] > 0) { solref[0] = mix * r1[0] + (1.f - mix) * r2[0]; solref[1] = mix * r1[1] + (1.f - mix) * r2[1]; }
    else { solref[0] = fminf(r1[0], r2[0]); solref[1] = fminf(r1[1], r2[1]); }
    for (int q = 0; q < 5; q++) solimp[q] = mix * i1[q] + (1.f - mix) * i2[q];
    float imp = impedance(solimp, dist, includemargin), K, B;
    kb(P[RG_PRM_TIMESTEP], solref, solimp, K, B);
    const float fr_slide = fmaxf(P[RG_PRM_GEOM_FRICTION + 3 * pg1], P[RG_PRM_GEOM_FRICTION + 3 * pg2]), fr_spin = fmaxf(P[RG_PRM_GEOM_FRICTION + 3 * pg1 + 1], P[RG_PRM_GEOM_FRICTION + 3 * pg2 + 1]);
    float mu0 = fr_slide;  // friction[0]
    // first pyramid row: diagApprox = tran + mu0^2 * tran ; all rows get R = 2 mu^2 R_first, mu = friction[0]/sqrt(impratio)
    float R;
    if (dim == 1) R = fmaxf(1e-15f, (1 - imp) * tran * rg_rcp(imp));
    else { float Rf = fmaxf(1e-15f, (1 - imp) * (tran + mu0 * mu0 * tran) * rg_rcp(imp)); float mu = mu0 * rg_rsqrt(m.impratio); R = 2 * mu * mu * Rf; }
    s.c_D[c] = rg_rcp(R);
    // friction coefficient of tangent direction k (k = 0,1 sliding; 2 spin)
    s.c_mu[2 * c] = fr_slide; s.c_mu[2 * c + 1] = fr_spin;
    // aref of pyramid row q = -K imp (dist - margin) - B (J_q qvel): the velocity term is evaluated by the row's lane at the start of the solve
    s.c_aref0[c] = -K * imp * (dist - includemargin); s.c_kb[c] = B;
  }
  SYNC();
}

// Residuals of the static rows (friction loss, limits) live in the registers of the lane that owns the row (row
// LANE + 64 k): only that lane ever touches them, so they need no LDS (1.2 kB per env that the occupancy wants back).
#define RG_RSLOTS ((RG_MAXSROW + RG_WAVE - 1) / RG_WAVE)
struct RowRegs { int desc[RG_RSLOTS]; float D[RG_RSLOTS], aref[RG_RSLOTS], floss[RG_RSLOTS], jar[RG_RSLOTS], jv[RG_RSLOTS]; int quad[RG_RSLOTS]; float pjar[RG_PSLOTS], pjv[RG_PSLOTS], paref[RG_PSLOTS]; };   // (row forces are functions of jar: recomputed where needed, not kept)
// force of a friction-loss / limit row from its residual (mj_constraintUpdate's three cases)
__device__ __forceinline__ float srow_force(float D, float f, float x) {
  if (!(D > 0)) return 0.f;
  if (f > 0) { float R = rg_rcp(D); return x <= -R * f ? f : (x >= R * f ? -f : -D * x); }
  return x < 0 ? -D * x : 0.f;
}   // p*: pyramid row LANE + 64 k (= 6 c + q)
// friction-loss and limit rows (mj_makeConstraint's first two blocks + mj_makeImpedance for them): impedance, regulariser,
// reference acceleration — straight into the registers of the lanes that own the rows (RowRegs), at the start of the solve
__device__ __forceinline__ void rg_static_rows(RgM m, const RgLds& s, const float* P, RowRegs& R) {
  int ns = nsrow(m);
#pragma unroll
  for (int k = 0; k < RG_RSLOTS; k++) {
    int r = LANE + RG_WAVE * k;
    if (r >= ns) { R.D[k] = 0.f; R.aref[k] = 0.f; R.floss[k] = 0.f; R.desc[k] = 0; continue; }
    // everything static about the row in one 64-byte record (rg_api.hip: srow_rec)
    const rgf4* S = (const rgf4*)m.srow_rec + 4 * r;
    const rgf4 sa = S[0], sb = S[1], sc = S[2], sd = S[3];
    const int w0 = __builtin_bit_cast(int, sa.x), kind = w0 >> 16, desc = w0 & 0xFFFF, src = __builtin_bit_cast(int, sa.y), ofl = __builtin_bit_cast(int, sa.w);
    const bool fric = kind < 2;
    const float diag = P[__builtin_bit_cast(int, sa.z)];
    const float floss = kind == 0 ? P[ofl] : sb.x;
    const float solref[2] = {sb.w, sc.x}, solimp[5] = {sc.y, sc.z, sc.w, sd.x, sd.y};
    float pos = 0, margin = 0; bool active = true;
    if (!fric) {
      const float x = kind == 2 ? s.qpos[src] : s.tenlen[src], bound = P[__builtin_bit_cast(int, sb.y)];
      pos = (desc >> 11) & 1 ? bound - x : x - bound;   // lower limit: J = +1, upper: J = -1
      margin = kind == 2 ? P[__builtin_bit_cast(int, sb.z)] : sb.z; active = pos < margin;   // (joint margins: the env's own, RG_PRM_JNT_MARGIN)
    }
    R.desc[k] = desc;
    float D = 0, aref = 0;
    if (active) {
      float imp = impedance(solimp, pos, margin), K, B;
      float R = fmaxf(1e-15f, (1 - imp) * diag * rg_rcp(imp));
      kb(P[RG_PRM_TIMESTEP], solref, solimp, K, B);
      if (fric) K = 0;
      float vel = srow_dot<true>(s, desc, s.qvel);
      D = rg_rcp(R); aref = -B * vel - K * imp * (pos - margin);
    }
    R.D[k] = D; R.aref[k] = aref; R.floss[k] = fric ? floss : 0.f;   // D == 0 marks an inactive slot
  }
}
// jar = J x - aref (or jv = J x) for every active row; x lives in the compact dof space
template <bool SROWS = true> __device__ __forceinline__ void rg_J_mul(RgM m, RgLds& s, RowRegs& R, const float* x, bool to_jv) {
  int ns = nsrow(m), ncon = s.ncon;
#pragma unroll
  for (int k = 0; k < (SROWS ? RG_RSLOTS : 0); k++) {
    int r = LANE + RG_WAVE * k;
    if (r < ns && R.D[k] > 0) { float v = srow_dot<false>(s, R.desc[k], x); if (to_jv) R.jv[k] = v; else R.jar[k] = v - R.aref[k]; }
  }
  // basis products of the contacts: a 16-lane group per contact (four per trip), lane (c, sl) multiplies x[dof of slot sl] into the slot's
  // <= 4 basis entries, a DPP row sum adds the slots up.  (Before: a lane per (contact, basis vector) walking the <= 14 slots one after the
  // other, two dependent LDS reads each.)
  for (int c0 = 0; c0 < ncon; c0 += 4) {
    const int c = c0 + (LANE >> 4), sl = LANE & 15, cc = c < ncon ? c : c0;
    const int nnz = s.c_nnz[cc], nb = nbasis(s.c_dim[cc]);
    const bool on = c < ncon && sl < nnz;
    const int slc = on ? sl : 0;
    const float* Bc = s.c_pool + s.c_off[cc] + slc;
    const float xv = on ? x[s.c_idx[cc * RG_W + slc]] : 0.f;
    float p0 = Bc[0] * xv, p1 = nb > 1 ? Bc[nnz] * xv : 0.f, p2 = nb > 2 ? Bc[2 * nnz] * xv : 0.f, p3 = nb > 3 ? Bc[3 * nnz] * xv : 0.f;
    p0 = grp_sum16(p0); p1 = grp_sum16(p1); p2 = grp_sum16(p2); p3 = grp_sum16(p3);
    if (c < ncon && sl < 4) s.c_bdot[4 * c + sl] = sl == 0 ? p0 : (sl == 1 ? p1 : (sl == 2 ? p2 : p3));
  }
  SYNC();
#pragma unroll
  for (int kk = 0; kk < RG_PSLOTS; kk++) {
    int w = LANE + RG_WAVE * kk;
    if (w >= ncon * 6) continue;
    int c = w / 6, q = w - 6 * c, dim = s.c_dim[c];
    if (q >= npyr(dim)) continue;
    float v;
    if (dim == 1) v = s.c_bdot[4 * c];
    else { int k = q >> 1; float mu = s.c_mu[2 * c + (k >> 1)]; v = s.c_bdot[4 * c] + ((q & 1) ? -mu : mu) * s.c_bdot[4 * c + k + 1]; }
    if (to_jv) R.pjv[kk] = v; else R.pjar[kk] = v - R.paref[kk];
  }
  SYNC();
}
// forces / quadratic flags from jar; returns the wave-summed constraint cost
// `changed`: whether any row's quadratic flag differs from what the arrays held before (the Hessian
// M + J' D J depends on the state only through these flags)
// Flag words (RowRegs::quad, RgLds::p_quad): bit 0 = the row is in its quadratic zone NOW, bit 1 = it was when the factor held in s.H was built
// (rg_solve sets it at every factorisation).  `nset`: the number of rows whose two bits differ = the rank of H(now) - H(factor).
__device__ __forceinline__ float rg_constraint_update(RgM m, RgLds& s, RowRegs& R, bool& changed, int& nset) {
  int ns = nsrow(m), ncon = s.ncon; float cost = 0; bool chg = false; nset = 0;
#ifdef RG_FINE_PROF
  float nchg_f = 0.f;
#endif
#pragma unroll
  for (int k = 0; k < RG_RSLOTS; k++) {
    int r = LANE + RG_WAVE * k;
    bool diff = false;
    if (r < ns) {
      float D = R.D[k]; const int oldw = R.quad[k], old = oldw & 1; int q = 0;
      if (D > 0) {
        float x = R.jar[k], f = R.floss[k];
        if (f > 0) {
          float R = rg_rcp(D);
          if (x <= -R * f) cost += f * (-0.5f * R * f - x);
          else if (x >= R * f) cost += f * (-0.5f * R * f + x);
          else { q = 1; cost += 0.5f * D * x * x; }
        } else if (x < 0) { q = 1; cost += 0.5f * D * x * x; }
      }
      R.quad[k] = q | (oldw & 2); chg |= q != old; diff = q != (oldw >> 1);
#ifdef RG_FINE_PROF
      nchg_f += q != old ? 1.f : 0.f;
#endif
    }
    nset += __popcll(__ballot(diff));
  }
#pragma unroll
  for (int kk = 0; kk < RG_PSLOTS; kk++) {
    int w = LANE + RG_WAVE * kk;
    bool diff = false;
    if (w < ncon * 6) {
      const int c = w / 6, k = w - 6 * c, oldw = s.p_quad[w], old = oldw & 1; int q = 0;
      if (k < npyr(s.c_dim[c])) {
        float x = R.pjar[kk], D = s.c_D[c];
        if (x < 0) { q = 1; cost += 0.5f * D * x * x; }
      }
      s.p_quad[w] = (unsigned char)(q | (oldw & 2)); chg |= q != old; diff = q != (oldw >> 1);
#ifdef RG_FINE_PROF
      nchg_f += q != old ? 1.f : 0.f;
#endif
    }
    nset += __popcll(__ballot(diff));
  }
  changed = __ballot(chg) != 0;
#ifdef RG_FINE_PROF
  s.prof[47] = 0.f;   // (analysis build: the number of rows that changed their zone, for the refactorisation statistics in rg_solve)
  SYNC();
  { float n = wave_sum(nchg_f); if (LANE == 0) s.prof[47] = n; }
#endif
  SYNC();
  return wave_sum(cost);
}
// dst = J^T force in the compact dof space (dst zeroed here)
__device__ __forceinline__ void rg_JT_force(RgM m, RgLds& s, const RowRegs& R, float* dst) {
  int ns = nsrow(m), ncon = s.ncon;
  PFOR(d, m.nvc) dst[d] = 0;
  // Round 6: the pyramid rows' forces are staged in LDS -- rows 0..3 of contact c in c_bdot[4 c ..] (free here: J x has been consumed), rows 4 and 5
  // in c_aref0[c] / c_kb[c] (dead since the rows took their reference accelerations at the start of the solve) -- and a 16-lane group per
  // contact then forms the contact's four basis forces itself and scatters slot sl's column: one LDS phase less than summing the basis forces
  // with atomics first, and no serial walk.
  static_assert(RG_W <= 16, "one 16-lane group per contact");
#pragma unroll
  for (int kk = 0; kk < RG_PSLOTS; kk++) {
    int w = LANE + RG_WAVE * kk;
    if (w < ncon * 6) {
      const int c = w / 6, q = w - 6 * c;
      const float f = (s.p_quad[w] & 1) ? -s.c_D[c] * R.pjar[kk] : 0.f;   // (a pyramid row carries force exactly when it is quadratic: x < 0)
      float* o = q < 4 ? s.c_bdot + 4 * c + q : (q == 4 ? s.c_aref0 + c : s.c_kb + c);
      *o = f;
    }
  }
  SYNC();
#pragma unroll
  for (int k = 0; k < RG_RSLOTS; k++) { int r = LANE + RG_WAVE * k; float f = r < ns ? srow_force(R.D[k], R.floss[k], R.jar[k]) : 0.f; if (f != 0) srow_scatter(s, R.desc[k], f, dst); }
  for (int c0 = 0; c0 < ncon; c0 += 4) {
    const int c = c0 + (LANE >> 4), sl = LANE & 15, cc = c < ncon ? c : c0;
    const int nnz = s.c_nnz[cc], nb = nbasis(s.c_dim[cc]);
    const float mu0 = s.c_mu[2 * cc], mu1 = s.c_mu[2 * cc + 1];
    const float* fq = s.c_bdot + 4 * cc;
    const float f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3], f4 = s.c_aref0[cc], f5 = s.c_kb[cc];
    if (c < ncon && sl < nnz) {
      const float* Bc = s.c_pool + s.c_off[cc] + sl;
      float v = Bc[0] * (((((f0 + f1) + f2) + f3) + f4) + f5);
      if (nb >= 3) v += Bc[nnz] * (mu0 * (f0 - f1)) + Bc[2 * nnz] * (mu0 * (f2 - f3));
      if (nb >= 4) v += Bc[3 * nnz] * (mu1 * (f4 - f5));
      atomicAdd(dst + s.c_idx[cc * RG_W + sl], v);
    }
  }
  SYNC();
}
// y = M x in the compact dof space from the tree-sparse entries (mj_mulM): a lane takes entries LANE, LANE + 64, ...
// (their compact row / column, prefetched once per solve: RgMEnt) and adds M_ij x_j to y_i and M_ij x_i to y_j with
// LDS atomics (issued by one wave in lane order: deterministic)
#define RG_MSLOTS ((RG_MAXNM + RG_WAVE - 1) / RG_WAVE)
struct RgMEnt { int w[RG_MSLOTS]; };   // ci | cj << 8 | valid << 16
__device__ __forceinline__ void rg_M_ent_load(RgM m, RgMEnt& E) {
#pragma unroll
  for (int k = 0; k < RG_MSLOTS; k++) { int e = LANE + RG_WAVE * k; E.w[k] = e < m.nM ? m.M_ent[2 * e + 1] : 0; }
}
__device__ __forceinline__ void rg_M_mul(RgM m, RgLds& s, const RgMEnt& E, const float* x, float* y) {
  PFOR(i, m.nvc) y[i] = 0;
  SYNC();
#pragma unroll
  for (int k = 0; k < RG_MSLOTS; k++) {
    int w = E.w[k];
    if (w >> 16) {
      int ci = w & 255, cj = (w >> 8) & 255; float mv = s.Msp[LANE + RG_WAVE * k];
      atomicAdd(y + ci, mv * x[cj]);
      if (ci != cj) atomicAdd(y + cj, mv * x[ci]);
    }
  }
  SYNC();
}
// Dense Cholesky of the n x n matrix in s.H (row stride hs, lower triangle, in place), left-looking, lane i
// owns row i, FOUR columns per step: one pass over the lane's own row chunks serves four dot products
// (pivot rows are broadcast reads), the 4x4 diagonal block is factored redundantly in every lane from ten
// v_readlane values, and each lane solves its four new entries against it.  n/4 LDS round trips, not n.
__device__ __forceinline__ float dot4(rgf4 a, rgf4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// RHS: row n of the work matrix holds a right-hand side g (columns n.. zero); lane n carries it through the same steps as the
// matrix rows, which is the forward substitution L y = g: y is left in row n (rg_chol_solve_bwd finishes the solve)
template <bool RHS> __device__ __forceinline__ void rg_chol(RgM m, RgLds& s) {
  int n = m.nvc, hs4 = m.hs >> 2, i = LANE;
  rgf4* H4 = (rgf4*)s.H;
  bool bad = false;
  for (int j0 = 0; j0 < n; j0 += 4) {
    int jb = j0 >> 2, nb = n - j0;  // nb >= 4: full block
    int r1 = j0 + 1 < n ? j0 + 1 : n - 1, r2 = j0 + 2 < n ? j0 + 2 : n - 1, r3 = j0 + 3 < n ? j0 + 3 : n - 1;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    bool mine = i >= j0 && (RHS ? i <= n : i < n);
    if (mine) {
      rgf4 own = H4[i * hs4 + jb];
      a0 = own.x; a1 = own.y; a2 = own.z; a3 = own.w;
      const rgf4 *ri = H4 + i * hs4, *p0 = H4 + j0 * hs4, *p1 = H4 + r1 * hs4, *p2 = H4 + r2 * hs4, *p3 = H4 + r3 * hs4;
#pragma unroll 2
      for (int c = 0; c < jb; c++) { rgf4 a = ri[c]; a0 -= dot4(a, p0[c]); a1 -= dot4(a, p1[c]); a2 -= dot4(a, p2[c]); a3 -= dot4(a, p3[c]); }
    }
    // diagonal block (lower triangle) from the four pivot lanes
    float t00 = lane_bcast(a0, j0);
    float t10 = lane_bcast(a0, j0 + 1), t11 = lane_bcast(a1, j0 + 1);
    float t20 = lane_bcast(a0, j0 + 2), t21 = lane_bcast(a1, j0 + 2), t22 = lane_bcast(a2, j0 + 2);
    float t30 = lane_bcast(a0, j0 + 3), t31 = lane_bcast(a1, j0 + 3), t32 = lane_bcast(a2, j0 + 3), t33 = lane_bcast(a3, j0 + 3);
    bool v1 = nb > 1, v2 = nb > 2, v3 = nb > 3;
    if (!(t00 > 1e-30f)) { bad = true; t00 = 1e-30f; }
    float i00 = rg_rsqrt(t00), l00 = t00 * i00;
    float l10 = v1 ? t10 * i00 : 0.f, d11 = v1 ? t11 - l10 * l10 : 1.f;
    if (!(d11 > 1e-30f)) { bad = true; d11 = 1e-30f; }
    float r11 = rg_rsqrt(d11), l11 = d11 * r11, i11 = v1 ? r11 : 0.f;
    float l20 = v2 ? t20 * i00 : 0.f, l21 = v2 ? (t21 - l20 * l10) * i11 : 0.f, d22 = v2 ? t22 - l20 * l20 - l21 * l21 : 1.f;
    if (!(d22 > 1e-30f)) { bad = true; d22 = 1e-30f; }
    float r22 = rg_rsqrt(d22), l22 = d22 * r22, i22 = v2 ? r22 : 0.f;
    float l30 = v3 ? t30 * i00 : 0.f, l31 = v3 ? (t31 - l30 * l10) * i11 : 0.f, l32 = v3 ? (t32 - l30 * l20 - l31 * l21) * i22 : 0.f;
    float d33 = v3 ? t33 - l30 * l30 - l31 * l31 - l32 * l32 : 1.f;
    if (!(d33 > 1e-30f)) { bad = true; d33 = 1e-30f; }
    float r33 = rg_rsqrt(d33), l33 = d33 * r33, i33 = v3 ? r33 : 0.f;
    if (mine) {
      int r = i - j0;
      float x0 = a0 * i00, x1 = (a1 - x0 * l10) * i11, x2 = (a2 - x0 * l20 - x1 * l21) * i22, x3 = (a3 - x0 * l30 - x1 * l31 - x2 * l32) * i33;
      if (RHS && i == n) r = 4;   // (the right-hand-side row is never a pivot row)
      if (r == 0) { x0 = l00; x1 = 0; x2 = 0; x3 = 0; }
      else if (r == 1) { x1 = l11; x2 = 0; x3 = 0; }
      else if (r == 2) { x2 = l22; x3 = 0; }
      else if (r == 3) x3 = l33;
      rgf4 o; o.x = x0; o.y = x1; o.z = x2; o.w = x3;
      H4[i * hs4 + jb] = o;
    }
    if (i == 0) { s.dinv[j0] = i00; if (v1) s.dinv[j0 + 1] = i11; if (v2) s.dinv[j0 + 2] = i22; if (v3) s.dinv[j0 + 3] = i33; }
    SYNC();
  }
  if (bad && i == 0) s.status |= RG_STATUS_BAD_FACTOR;
}
// x <- H^-1 x with the factor above, four pivots per step; lane i owns x[i].  The operands of step k+1 (pivot
// rows, reciprocal pivots, the lane's own chunk) do not depend on step k's result, so they are fetched one
// step ahead and the LDS latency hides behind the substitution arithmetic.
struct TriOps { rgf4 q1, q2, q3, own; float i00, i11, i22, i33; float c0, c1, c2, c3; };
__device__ __forceinline__ void tri_fetch(const RgLds& s, int n, int hs, int j0, int i, bool backward, TriOps& t) {
  const float* H = s.H; const rgf4* H4 = (const rgf4*)s.H;
  int hs4 = hs >> 2, jb = j0 >> 2, nb = n - j0;
  int r1 = j0 + 1 < n ? j0 + 1 : n - 1, r2 = j0 + 2 < n ? j0 + 2 : n - 1, r3 = j0 + 3 < n ? j0 + 3 : n - 1;
  t.q1 = H4[r1 * hs4 + jb]; t.q2 = H4[r2 * hs4 + jb]; t.q3 = H4[r3 * hs4 + jb];
  t.i00 = s.dinv[j0]; t.i11 = s.dinv[r1]; t.i22 = s.dinv[r2]; t.i33 = s.dinv[r3];
  if (nb <= 1) t.i11 = 0.f;
  if (nb <= 2) t.i22 = 0.f;
  if (nb <= 3) t.i33 = 0.f;
  int ic = i < n ? i : n - 1;
  if (!backward) t.own = H4[ic * hs4 + jb];
  else { t.c0 = H[j0 * hs + ic]; t.c1 = H[r1 * hs + ic]; t.c2 = H[r2 * hs + ic]; t.c3 = H[r3 * hs + ic]; }
}
__device__ __forceinline__ void tri_fwd(const TriOps& t, int n, int j0, int i, float& xi) {
  float x0 = lane_bcast(xi, j0) * t.i00;
  float x1 = (lane_bcast(xi, j0 + 1) - t.q1.x * x0) * t.i11;
  float x2 = (lane_bcast(xi, j0 + 2) - t.q2.x * x0 - t.q2.y * x1) * t.i22;
  float x3 = (lane_bcast(xi, j0 + 3) - t.q3.x * x0 - t.q3.y * x1 - t.q3.z * x2) * t.i33;
  int r = i - j0;
  if (r >= 4 && i < n) xi -= t.own.x * x0 + t.own.y * x1 + t.own.z * x2 + t.own.w * x3;
  else if (r == 0) xi = x0; else if (r == 1) xi = x1; else if (r == 2) xi = x2; else if (r == 3) xi = x3;
}
__device__ __forceinline__ void tri_bwd(const TriOps& t, int n, int j0, int i, float& xi) {
  int nb = n - j0;
  float l10 = nb > 1 ? t.q1.x : 0.f, l20 = nb > 2 ? t.q2.x : 0.f, l21 = nb > 2 ? t.q2.y : 0.f, l30 = nb > 3 ? t.q3.x : 0.f, l31 = nb > 3 ? t.q3.y : 0.f, l32 = nb > 3 ? t.q3.z : 0.f;
  float x3 = lane_bcast(xi, j0 + 3) * t.i33;
  float x2 = (lane_bcast(xi, j0 + 2) - l32 * x3) * t.i22;
  float x1 = (lane_bcast(xi, j0 + 1) - l21 * x2 - l31 * x3) * t.i11;
  float x0 = (lane_bcast(xi, j0) - l10 * x1 - l20 * x2 - l30 * x3) * t.i00;
  int r = i - j0;
  if (i < j0) xi -= t.c0 * x0 + (nb > 1 ? t.c1 * x1 : 0.f) + (nb > 2 ? t.c2 * x2 : 0.f) + (nb > 3 ? t.c3 * x3 : 0.f);
  else if (r == 0) xi = x0; else if (r == 1) xi = x1; else if (r == 2) xi = x2; else if (r == 3) xi = x3;
}
__device__ __forceinline__ void rg_chol_solve(RgM m, RgLds& s, float* x) {
  int n = m.nvc, hs = m.hs, i = LANE;
  float xi = i < n ? x[i] : 0.f;
  TriOps A, B;   // two operand sets used alternately: the fetch of one overlaps the arithmetic on the other, no copies
  tri_fetch(s, n, hs, 0, i, false, A);
  for (int j0 = 0; j0 < n; j0 += 8) {  // forward: L y = b
    if (j0 + 4 < n) tri_fetch(s, n, hs, j0 + 4, i, false, B);
    tri_fwd(A, n, j0, i, xi);
    if (j0 + 4 >= n) break;
    if (j0 + 8 < n) tri_fetch(s, n, hs, j0 + 8, i, false, A);
    tri_fwd(B, n, j0 + 4, i, xi);
  }
  int jl = ((n - 1) >> 2) << 2;
  tri_fetch(s, n, hs, jl, i, true, A);
  for (int j0 = jl; j0 >= 0; j0 -= 8) {  // backward: L' x = y
    if (j0 >= 4) tri_fetch(s, n, hs, j0 - 4, i, true, B);
    tri_bwd(A, n, j0, i, xi);
    if (j0 < 4) break;
    if (j0 >= 8) tri_fetch(s, n, hs, j0 - 8, i, true, A);
    tri_bwd(B, n, j0 - 4, i, xi);
  }
  if (i < n) x[i] = xi;
  SYNC();
}
// the second half alone: y = L^-1 g was produced by rg_chol<true> in row n of the work matrix
__device__ __forceinline__ void rg_chol_solve_bwd(RgM m, RgLds& s, float* x) {
  int n = m.nvc, hs = m.hs, i = LANE;
  float xi = i < n ? s.H[n * hs + i] : 0.f;
  TriOps A, B;
  int jl = ((n - 1) >> 2) << 2;
  tri_fetch(s, n, hs, jl, i, true, A);
  for (int j0 = jl; j0 >= 0; j0 -= 8) {  // backward: L' x = y
    if (j0 >= 4) tri_fetch(s, n, hs, j0 - 4, i, true, B);
    tri_bwd(A, n, j0, i, xi);
    if (j0 < 4) break;
    if (j0 >= 8) tri_fetch(s, n, hs, j0 - 8, i, true, A);
    tri_bwd(B, n, j0 - 4, i, xi);
  }
  if (i < n) x[i] = xi;
  SYNC();
}
// ---- The dense Newton step in REGISTERS (round 6).  The left-looking factorisation above makes n/4 LDS round trips with a serial 4x4
// block in each (25 k cycles for n = 30), and the substitutions another 2 n/4.  Here lane i <= n loads row i of the work matrix (row n =
// the right-hand side g) into 32 registers and the wave runs the RIGHT-looking elimination on them: column j's pivot and multipliers
// travel by v_readlane (wave-uniform, no LDS), every lane updates its own row.  The lanes the matrix leaves idle carry the IDENTITY as
// n more rows (lane n + 1 + m starts as e_m): what the elimination does to a row b is b <- b inv(L'), so those lanes end up holding the
// rows of inv(L') (= the columns of inv(L)) while lane n ends up holding y = inv(L) g -- and the solution is one dot product per lane,
// x_m = (row m of inv(L')) . y: no backward substitution at all.  W = inv(L') is left in the work matrix (row m, upper triangular) for
// the iterations that reuse the factor (rg_cholinv_apply: x = W (W' g), two passes of broadcast products, no LDS writes).
// Needs 2 n + 1 <= 64 lanes; launch flag bit 6 keeps the path above.
// The size is a template parameter (every loop static, no branch inside the elimination; the wave's code is straight-line): instantiated for
// the Newton spaces of the hand models (dactyl/locked 30, dactyl/reach 24); other sizes keep the path above.
__device__ __forceinline__ bool rg_cholreg_ok(int n) { return n == 30 || n == 24; }
template <int N> __device__ __forceinline__ void rg_chol_inv_solve_n(RgM m, RgLds& s, float* x) {
  static_assert(2 * N + 1 <= RG_WAVE && N <= RG_MAXNVC, "matrix rows + right-hand side + identity rows: one lane each");
  constexpr int NC = (N + 3) / 4;
  const int hs4 = m.hs >> 2, i = LANE;
  int zr = i - (N + 1);                // identity row carried by this lane (lanes > N)
#ifndef RG_EMUL
  asm volatile("" : "+v"(zr));         // (opaque: otherwise the 32 identity-row constants are hoisted out of the Newton loop and live in scratch)
#endif
  const bool isrow = i <= N;
  float a[4 * NC];
  {
    const rgf4* src = (const rgf4*)s.H + (isrow ? i : 0) * hs4;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const rgf4 v = src[c];
      a[4 * c + 0] = isrow ? v.x : (zr == 4 * c + 0 ? 1.f : 0.f);
      a[4 * c + 1] = isrow ? v.y : (zr == 4 * c + 1 ? 1.f : 0.f);
      a[4 * c + 2] = isrow ? v.z : (zr == 4 * c + 2 ? 1.f : 0.f);
      a[4 * c + 3] = isrow ? v.w : (zr == 4 * c + 3 ? 1.f : 0.f);
    }
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < N; j++) {
    float p = lane_bcast(a[j], j);
    if (!(p > 1e-30f)) bad = true;
    const float inv = rg_rsqrt(fmaxf(p, 1e-30f));
    a[j] *= inv;                       // lanes > j: L[i][j]; lane j: the pivot's root; lanes < j: never read
    const float naj = -a[j];
#pragma unroll
    for (int k = j + 1; k < N; k += 4) {   // multipliers four at a time: the readlane -> fma latency of one hides behind the others
      const float l0 = lane_bcast(a[j], k), l1 = k + 1 < N ? lane_bcast(a[j], k + 1) : 0.f, l2 = k + 2 < N ? lane_bcast(a[j], k + 2) : 0.f, l3 = k + 3 < N ? lane_bcast(a[j], k + 3) : 0.f;
      a[k] = __builtin_fmaf(naj, l0, a[k]);
      if (k + 1 < N) a[k + 1] = __builtin_fmaf(naj, l1, a[k + 1]);
      if (k + 2 < N) a[k + 2] = __builtin_fmaf(naj, l2, a[k + 2]);
      if (k + 3 < N) a[k + 3] = __builtin_fmaf(naj, l3, a[k + 3]);
    }
  }
  // x_m = sum_k W[m][k] y_k in lane N + 1 + m
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
  for (int k = 0; k < N; k += 4) {
    acc0 = __builtin_fmaf(a[k], lane_bcast(a[k], N), acc0);
    if (k + 1 < N) acc1 = __builtin_fmaf(a[k + 1], lane_bcast(a[k + 1], N), acc1);
    if (k + 2 < N) acc2 = __builtin_fmaf(a[k + 2], lane_bcast(a[k + 2], N), acc2);
    if (k + 3 < N) acc3 = __builtin_fmaf(a[k + 3], lane_bcast(a[k + 3], N), acc3);
  }
  if (zr >= 0 && zr < N) {
    x[zr] = (acc0 + acc1) + (acc2 + acc3);
    rgf4* dst = (rgf4*)s.H + zr * hs4;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      rgf4 o; o.x = a[4 * c]; o.y = 4 * c + 1 < N ? a[4 * c + 1] : 0.f; o.z = 4 * c + 2 < N ? a[4 * c + 2] : 0.f; o.w = 4 * c + 3 < N ? a[4 * c + 3] : 0.f;
      dst[c] = o;
    }
  }
  if (bad && i == 0) s.status |= RG_STATUS_BAD_FACTOR;
  SYNC();
}
// ---- The same step on the MATRIX PIPE (round 6, an OPTION: -DRG_CHOL_MFMA; measured and not the default, see the end of this comment).  The elimination above is ~1230 VALU
// instructions per factorisation in a kernel whose waves share a VALU that is ~45 % busy: alone it takes 5.3 k cycles, between two other waves
// 15 k.  Here the same right-looking elimination is two 32 x 32 f32 accumulator tiles and one rank-2 v_mfma_f32_32x32x2_f32 per tile and PAIR of
// columns (~30 VALU instructions per pair; the matrix pipe is otherwise idle in this kernel):
//   S  = [H g] in symmetric storage (column N = the right-hand side; its row is never a pivot),
//   T' = the transpose of the N rows that start as the identity (T'[c][i] = row i, column c): it ends up as inv(L), and its row N as -x.
// A pivot row j of either tile is ONE register across 32 lanes (the half l / 32 = (j / 4) % 2), so step (j, j + 1) is: pivots and the
// multiplier by v_readlane, w = row / root on the VALU (row j + 1 takes step j's correction there too), the two k-slots of the operands
// assembled with one v_permlane32_swap each, rows <= j + 1 masked out of the A operand (they are final), S -= w w', T' -= w z'.
// By symmetry column j of S is row j of S, and the transposed storage makes "column j of the identity rows" a row as well: no transposes.
// Measured (profiles/r06_ab_newton.txt): alone 5.5 k cycles per factorisation against 5.3 k for the VALU elimination (tools/ubench), but inside the kernel the two
// 16-register accumulator tuples cost more in spills around the solve than the ~800 VALU instructions they save: 1.46 M against 1.62 M env-steps/s.
template <int N> __device__ __forceinline__ void rg_chol_mfma_n(RgM m, RgLds& s, float* x) {
  static_assert(N % 2 == 0 && N <= 30, "pairs of columns; column N = right-hand side, row / column 31 stay zero");
  const int hs = m.hs, hs4 = hs >> 2, l = LANE, col = l & 31, hi = l >> 5;
  const rgf4* H4 = (const rgf4*)s.H;
  rgacc S, T;
  {
    const int cc = col < N + 1 ? col : N;   // (column 31: clamped here, zeroed below)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int i0 = 8 * g + 4 * hi;
      const rgf4 rowf = H4[cc * hs4 + 2 * g + hi];   // H[c][i0 .. i0 + 3]: the stored (lower) triangle where i <= c
      const float v[4] = {rowf.x, rowf.y, rowf.z, rowf.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = i0 + q, ic = i < N + 1 ? i : N;
        const float colf = s.H[ic * hs + cc];       // H[i][c]: stored where c <= i
        float val = i < cc ? v[q] : colf;
        if (i > N || col > N) val = 0.f;
        S[4 * g + q] = val;
        T[4 * g + q] = (i == col && col < N) ? 1.f : 0.f;
      }
    }
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    constexpr int dummy = 0; (void)dummy;
    const int rj = 4 * (j >> 3) + (j & 3), hj = (j >> 2) & 1, lj = 32 * hj + j;
    const float s0 = S[rj], s1 = S[rj + 1], t0 = T[rj], t1 = T[rj + 1];
    const float p = lane_bcast(s0, lj);
    if (!(p > 1e-30f)) bad = true;
    const float inv = rg_rsqrt(fmaxf(p, 1e-30f));
    const float w0 = s0 * inv, z0 = t0 * inv;
    const float mlt = lane_bcast(w0, lj + 1);
    const float s1c = __builtin_fmaf(-mlt, w0, s1), t1c = __builtin_fmaf(-mlt, z0, t1);
    const float p2 = lane_bcast(s1c, lj + 1);
    if (!(p2 > 1e-30f)) bad = true;
    const float inv2 = rg_rsqrt(fmaxf(p2, 1e-30f));
    const float w1 = s1c * inv2, z1 = t1c * inv2;
    const bool mine = hi == hj;
    T[rj] = mine ? z0 : t0; T[rj + 1] = mine ? z1 : t1;
    float WW, ZZ;
    if (hj == 0) { WW = rg_halves<0>(w0, w1); ZZ = rg_halves<0>(z0, z1); } else { WW = rg_halves<1>(w0, w1); ZZ = rg_halves<1>(z0, z1); }
    const float WWn = col > j + 1 ? -WW : 0.f;   // rows <= j + 1 are final (row j + 1 took its correction on the VALU)
    rg_mfma32(WWn, WW, S);
    rg_mfma32(WWn, ZZ, T);
  }
  // x_i = -T'[N][i]
  constexpr int rN = 4 * (N >> 3) + (N & 3), hN = (N >> 2) & 1;
  if (hi == hN && col < N) x[col] = -T[rN];
  // W[m][k] = T'[k][m] for the iterations that reuse the factor (rg_cholinv_apply)
  if (col < N) {
    rgf4* dst = (rgf4*)s.H + col * hs4;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int k0 = 8 * g + 4 * hi;
      if (k0 < ((N + 3) & ~3)) {
        rgf4 o; o.x = k0 < N ? T[4 * g] : 0.f; o.y = k0 + 1 < N ? T[4 * g + 1] : 0.f; o.z = k0 + 2 < N ? T[4 * g + 2] : 0.f; o.w = k0 + 3 < N ? T[4 * g + 3] : 0.f;
        dst[2 * g + hi] = o;
      }
    }
  }
  if (bad && l == 0) s.status |= RG_STATUS_BAD_FACTOR;
  SYNC();
}
// x <- inv(H) x from W = inv(L') left by rg_chol_inv_solve: y = W' x (lane k: column k of W), then x = W y (lane m: row m of W)
template <int N> __device__ __forceinline__ void rg_cholinv_apply_n(RgM m, RgLds& s, float* x) {
  constexpr int NC = (N + 3) / 4;
  const int hs = m.hs, hs4 = hs >> 2, i = LANE, ic = i < N ? i : 0;
  const float g = i < N ? x[i] : 0.f;
  float y0 = 0.f, y1 = 0.f;
  const float* col = s.H + ic;
#pragma unroll
  for (int mm = 0; mm < N; mm += 2) {
    y0 = __builtin_fmaf(col[mm * hs], lane_bcast(g, mm), y0);
    if (mm + 1 < N) y1 = __builtin_fmaf(col[(mm + 1) * hs], lane_bcast(g, mm + 1), y1);
  }
  const float y = i < N ? y0 + y1 : 0.f;
  const rgf4* row = (const rgf4*)s.H + ic * hs4;
  float xa = 0.f, xb = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const rgf4 w = row[c];
    xa = __builtin_fmaf(w.x, lane_bcast(y, 4 * c), xa);
    if (4 * c + 1 < N) xb = __builtin_fmaf(w.y, lane_bcast(y, 4 * c + 1), xb);
    if (4 * c + 2 < N) xa = __builtin_fmaf(w.z, lane_bcast(y, 4 * c + 2), xa);
    if (4 * c + 3 < N) xb = __builtin_fmaf(w.w, lane_bcast(y, 4 * c + 3), xb);
  }
  if (i < N) x[i] = xa + xb;
  SYNC();
}
// ---- A few rows changed zone since the factorisation (rank-k change of the Hessian, k <= RG_WOODBURY): x = inv(H + U C U') g by the Woodbury identity on
// the inverse factor W that is still in the work matrix, instead of assembling and factorising H again (mj_solNewton itself updates its factor row by row,
// engine_solver.c HessianIncremental).  U's columns are the changed rows' Jacobians j_a (sparse), C = diag(+D_a: the row entered its quadratic zone,
// -D_a: it left), P = W' U (lane = dof, one register per changed row):
//     x = W (y - P t),   y = W' g,   (inv(C) + P' P) t = P' y.
// Cost: the two passes of rg_cholinv_apply, ~14 LDS reads per changed row, k (k + 3) / 2 wave sums and a k x k elimination on wave-uniform numbers --
// ~4 k cycles against ~26 k for assembly + factorisation.  Returns false (x untouched) when the small system is numerically singular: the caller
// then factorises.  Measured on the bench stream: half of the refactorisations follow a change of <= 2 rows, 70 % of <= 4 (tools/stage_profile.py, RG_FINE_PROF).
#ifndef RG_WOODBURY
#define RG_WOODBURY 4
#endif
__device__ __forceinline__ int lane_bcast_i(int v, int src) { return __builtin_bit_cast(int, lane_bcast(__builtin_bit_cast(float, v), src)); }
template <int N> __device__ __forceinline__ bool rg_cholinv_woodbury_n(RgM m, RgLds& s, const RowRegs& R, float* x) {
  constexpr int NC = (N + 3) / 4, K = RG_WOODBURY;
#ifdef RG_FINE_PROF
  const long long tw_entry = rg_clock();
#endif
  const int hs = m.hs, hs4 = hs >> 2, i = LANE, ic = i < N ? i : 0, ns = nsrow(m), ncon = s.ncon;
  const float g = i < N ? x[i] : 0.f;
  const float* col = s.H + ic;     // column `ic` of W: W[mm][ic] = col[mm * hs]
  float p[K], cinv[K];
#pragma unroll
  for (int a = 0; a < K; a++) { p[a] = 0.f; cinv[a] = 1.f; }
  int na = 0;
  // P = W' U, one changed row after the other (wave-uniform loops over ballots)
#pragma unroll
  for (int k = 0; k < RG_RSLOTS; k++) {
    const int r = LANE + RG_WAVE * k, qw = R.quad[k];
    unsigned long long msk = __ballot(r < ns && R.D[k] > 0 && (qw & 1) != (qw >> 1));
    while (msk) {
      const int L = __builtin_ctzll(msk); msk &= msk - 1;
      const int desc = lane_bcast_i(R.desc[k], L), qq = lane_bcast_i(qw, L); const float D = lane_bcast(R.D[k], L);
      const int t = (desc >> 6) & 31; float v;
      if (t == 31) v = col[(desc & 63) * hs];
      else { v = 0.f; for (int e = 0; e < 4; e++) { const int d = s.ten_cdof[4 * t + e]; if (d != 255) v += s.tenJ[4 * t + e] * col[d * hs]; } }
      if ((desc >> 11) & 1) v = -v;
      const float ci = (qq & 1) ? rg_rcp(D) : -rg_rcp(D);
#pragma unroll
      for (int a = 0; a < K; a++) if (a == na) { p[a] = i < N ? v : 0.f; cinv[a] = ci; }
      na++;
    }
  }
#pragma unroll
  for (int kk = 0; kk < RG_PSLOTS; kk++) {
    const int w = LANE + RG_WAVE * kk; const int qw = w < ncon * 6 ? s.p_quad[w] : 0;
    unsigned long long msk = __ballot((qw & 1) != ((qw >> 1) & 1));
    while (msk) {
      const int L = __builtin_ctzll(msk); msk &= msk - 1;
      const int wr = L + RG_WAVE * kk, c = wr / 6, q = wr - 6 * c, qq = lane_bcast_i(qw, L);
      const int nnz = s.c_nnz[c], dim = s.c_dim[c];
      const float* Bc = s.c_pool + s.c_off[c];
      // lane sl < nnz holds the row's entry for slot sl and the slot's dof; the sum over the slots reads them back lane by lane
      float jv = 0.f; int jd = 0;
      if (i < nnz) {
        jv = Bc[i];
        if (dim > 1) { const int kb = q >> 1; const float mu = s.c_mu[2 * c + (kb >> 1)]; jv += ((q & 1) ? -mu : mu) * Bc[(kb + 1) * nnz + i]; }
        jd = s.c_idx[c * RG_W + i];
      }
      float v = 0.f;   // (all RG_W slots, the unused ones with a zero entry on dof 0: independent LDS reads, all in flight)
#pragma unroll
      for (int sl = 0; sl < RG_W; sl++) v = __builtin_fmaf(lane_bcast(jv, sl), col[lane_bcast_i(jd, sl) * hs], v);
      const float D = s.c_D[c];
      const float ci = (qq & 1) ? rg_rcp(D) : -rg_rcp(D);
#pragma unroll
      for (int a = 0; a < K; a++) if (a == na) { p[a] = i < N ? v : 0.f; cinv[a] = ci; }
      na++;
    }
  }
#ifdef RG_FINE_PROF
  long long tw0 = tw_entry, tw1;
#define WPROF(k) do { tw1 = rg_clock(); if (LANE == 0) s.prof[k] += (float)(tw1 - tw0); tw0 = tw1; } while (0)
  WPROF(50);   // (50: since function entry -- the rows' P columns; the caller's stamp covers the whole call)
#else
#define WPROF(k) do { } while (0)
#endif
  // y = W' g
  float y0 = 0.f, y1 = 0.f;
#pragma unroll
  for (int mm = 0; mm < N; mm += 2) {
    y0 = __builtin_fmaf(col[mm * hs], lane_bcast(g, mm), y0);
    if (mm + 1 < N) y1 = __builtin_fmaf(col[(mm + 1) * hs], lane_bcast(g, mm + 1), y1);
  }
  float y = i < N ? y0 + y1 : 0.f;
  WPROF(51);
  // S = inv(C) + P' P (symmetric, unused slots: identity), r = P' y
  float S[K][K], rr[K];
#pragma unroll
  for (int a = 0; a < K; a++) {
    rr[a] = a < na ? wave_sum(p[a] * y) : 0.f;
#pragma unroll
    for (int b = a; b < K; b++) { float v = (a < na && b < na) ? wave_sum(p[a] * p[b]) : 0.f; if (a == b) v += cinv[a]; S[a][b] = v; S[b][a] = v; }
  }
  // elimination with the diagonal as pivots (S is symmetric, not definite: a row that left makes a negative pivot), scale-relative singularity test
  bool ok = true;
#pragma unroll
  for (int a = 0; a < K; a++) {
    const float piv = S[a][a];
    if (!(fabsf(piv) > 1e-4f * fabsf(cinv[a]))) ok = false;   // (|pivot| / |1 / D| = 1 / (1 + D j' inv(H_rest) j): below this the difference has lost four digits)
    const float ip = rg_rcp(piv);
#pragma unroll
    for (int b = a + 1; b < K; b++) {
      const float f = S[b][a] * ip;
#pragma unroll
      for (int c2 = a + 1; c2 < K; c2++) S[b][c2] -= f * S[a][c2];
      rr[b] -= f * rr[a];
    }
  }
  float t[K];
#pragma unroll
  for (int a = K - 1; a >= 0; a--) { float v = rr[a];
#pragma unroll
    for (int b = a + 1; b < K; b++) v -= S[a][b] * t[b];
    t[a] = v * rg_rcp(S[a][a]); }
  WPROF(52);
  if (!ok) return false;
#pragma unroll
  for (int a = 0; a < K; a++) y -= p[a] * t[a];
  // x = W y
  const rgf4* row = (const rgf4*)s.H + ic * hs4;
  float xa = 0.f, xb = 0.f;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const rgf4 w = row[c];
    xa = __builtin_fmaf(w.x, lane_bcast(y, 4 * c), xa);
    if (4 * c + 1 < N) xb = __builtin_fmaf(w.y, lane_bcast(y, 4 * c + 1), xb);
    if (4 * c + 2 < N) xa = __builtin_fmaf(w.z, lane_bcast(y, 4 * c + 2), xa);
    if (4 * c + 3 < N) xb = __builtin_fmaf(w.w, lane_bcast(y, 4 * c + 3), xb);
  }
  if (i < N) x[i] = xa + xb;
  SYNC();
  WPROF(53);
  return true;
}
#undef WPROF
__device__ __forceinline__ bool rg_cholinv_woodbury(RgM m, RgLds& s, const RowRegs& R, float* x) { return m.nvc == 30 ? rg_cholinv_woodbury_n<30>(m, s, R, x) : rg_cholinv_woodbury_n<24>(m, s, R, x); }
#ifdef RG_CHOL_MFMA
__device__ __forceinline__ void rg_chol_inv_solve(RgM m, RgLds& s, float* x) { if (m.nvc == 30) rg_chol_mfma_n<30>(m, s, x); else rg_chol_mfma_n<24>(m, s, x); }
#else
__device__ __forceinline__ void rg_chol_inv_solve(RgM m, RgLds& s, float* x) { if (m.nvc == 30) rg_chol_inv_solve_n<30>(m, s, x); else rg_chol_inv_solve_n<24>(m, s, x); }
#endif
__device__ __forceinline__ void rg_cholinv_apply(RgM m, RgLds& s, float* x) { if (m.nvc == 30) rg_cholinv_apply_n<30>(m, s, x); else rg_cholinv_apply_n<24>(m, s, x); }
// Tree-sparse factorisation A = L' D L in the per-tree block storage (work copy in s.H) and its substitutions
// (mj_factorM / mj_solveM).  Only (dof, ancestor) entries exist and dofs of equal depth are independent, so the
// factorisation is one lane-parallel pass per depth (deepest first), every lane taking one (k, i, j) update
// A[i][j] -= A[k][i] A[k][j] / A[k][k] and adding it with an LDS atomic (several dofs share ancestors; one wave
// issues them in lane order: deterministic).  The substitutions are the same passes over the (k, i) pairs.  A lane
// loads all of its descriptors up front (one load latency); the passes themselves only touch LDS.
struct LtdlDesc { int t0[RG_LTDL_TRI_ROUNDS], t1[RG_LTDL_TRI_ROUNDS], pr[RG_LTDL_PAIR_ROUNDS], ntr, npr; };
__device__ __forceinline__ void rg_ltdl_load(const int* tri, const int* pair, int ntr, int npr, LtdlDesc& L) {
  L.ntr = ntr; L.npr = npr;
#pragma unroll
  for (int r = 0; r < RG_LTDL_TRI_ROUNDS; r++) { bool on = r < ntr; L.t0[r] = on ? tri[2 * (r * RG_WAVE + LANE)] : 0; L.t1[r] = on ? tri[2 * (r * RG_WAVE + LANE) + 1] : 0; }
#pragma unroll
  for (int r = 0; r < RG_LTDL_PAIR_ROUNDS; r++) L.pr[r] = r < npr ? pair[r * RG_WAVE + LANE] : -1;
}
// s.H (already holding A, lower triangle) <- its factor: D on the diagonal, L below it
__device__ __forceinline__ void rg_ltdl_factor(RgLds& s, const LtdlDesc& L) {
#pragma unroll
  for (int r = 0; r < RG_LTDL_TRI_ROUNDS; r++) {   // (no break / continue: the loops must unroll fully, or the descriptor arrays end up in scratch)
    if (r < L.ntr) {
      if (L.t1[r] < 0) {
        float v = s.H[L.t0[r] & 1023] * s.H[(L.t0[r] >> 10) & 1023] * rg_rcp(s.H[(L.t0[r] >> 20) & 1023]);
        atomicAdd(s.H + (L.t1[r] & 1023), -v);
      }
      SYNC();
    }
  }
#pragma unroll
  for (int r = 0; r < RG_LTDL_PAIR_ROUNDS; r++)
    if (r < L.npr && L.pr[r] != -1) { int a = (L.pr[r] >> 12) & 1023; s.H[a] *= rg_rcp(fmaxf(s.H[(L.pr[r] >> 22) & 1023], 1e-30f)); }
  SYNC();
}
// x <- A^-1 x with the factor in s.H; lane i owns x[i], akk = address of its pivot (on: lane has a dof)
__device__ __forceinline__ void rg_ltdl_solve(RgLds& s, const LtdlDesc& L, float* x, int i, bool on, int akk) {
  float dk = on ? s.H[akk] : 1.f;
  bool bad = on && !(dk > 1e-30f);
  float idk = rg_rcp(fmaxf(dk, 1e-30f));
#pragma unroll
  for (int r = 0; r < RG_LTDL_PAIR_ROUNDS; r++) {   // x <- inv(L') x : x[i] -= L[k][i] x[k], deepest first
    if (r < L.npr) {
      if (L.pr[r] != -1) atomicAdd(x + ((L.pr[r] >> 6) & 63), -s.H[(L.pr[r] >> 12) & 1023] * x[L.pr[r] & 63]);
      SYNC();
    }
  }
  if (on) x[i] *= idk;                              // x <- inv(D) x
  SYNC();
#pragma unroll
  for (int q = 0; q < RG_LTDL_PAIR_ROUNDS; q++) {   // x <- inv(L) x : x[k] -= L[k][i] x[i], shallowest first
    const int r = RG_LTDL_PAIR_ROUNDS - 1 - q;
    if (r < L.npr) {
      if (L.pr[r] != -1) atomicAdd(x + (L.pr[r] & 63), -s.H[(L.pr[r] >> 12) & 1023] * x[(L.pr[r] >> 6) & 63]);
      SYNC();
    }
  }
  if (bad) s.status |= RG_STATUS_BAD_FACTOR;
}
// M + scale*diag(extra) over ALL dofs, then x <- solve: qacc_smooth = M^-1 qfrc_smooth and the implicit-damping Euler solve
__device__ __forceinline__ void rg_ltdl_factor_solve(RgM m, RgLds& s, const float* extra_diag, float scale, float* x) {
  LtdlDesc L;
  rg_ltdl_load(m.ltdl_tri, m.ltdl_pair, m.n_tri_rounds, m.n_pair_rounds, L);
  rg_M_to_blocks(m, s);
  int d = LANE; bool on = d < m.nv;
  int blk = on ? m.dof_blk[d] : 0, akk = (blk & 0xFFFF) + d - ((blk >> 16) & 255);
  if (on && extra_diag) s.H[akk] += scale * extra_diag[d];
  SYNC();
  rg_ltdl_factor(s, L);
  rg_ltdl_solve(s, L, x, d, on, akk);
}

struct LsPt { float cost, grad, hess; };
// The rows a lane owns (static slots LANE, LANE+64; pyramid rows LANE+64k) do not change during a line search:
// their (D, floss, jar, jv) are read from LDS once and every trial step length is evaluated from registers.
struct LsRows { float rD[2], rf[2], rjar[2], rjv[2], pD[RG_PSLOTS], pjar[RG_PSLOTS], pjv[RG_PSLOTS]; };   // (copies of register-resident values: no storage of their own after inlining)
__device__ __forceinline__ void rg_ls_load(RgM m, const RgLds& s, const RowRegs& R, LsRows& L) {
  int ns = nsrow(m), ncon = s.ncon;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    int r = LANE + RG_WAVE * k; bool on = r < ns;
    float D = on ? R.D[k] : 0.f;
    L.rD[k] = D > 0 ? D : 0.f; L.rf[k] = on ? R.floss[k] : 0.f;
    L.rjar[k] = on ? R.jar[k] : 0.f; L.rjv[k] = on ? R.jv[k] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < RG_PSLOTS; k++) {
    int w = LANE + RG_WAVE * k; bool on = w < ncon * 6;
    int cc = on ? w / 6 : 0; on = on && (w - 6 * cc) < npyr(s.c_dim[cc]);
    L.pD[k] = on ? s.c_D[cc] : 0.f; L.pjar[k] = on ? R.pjar[k] : 0.f; L.pjv[k] = on ? R.pjv[k] : 0.f;
  }
}
__device__ __forceinline__ LsPt rg_ls_eval(const LsRows& L, float alpha, float q0, float q1, float q2) {
  float c = 0, g = 0, h = 0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    float D = L.rD[k];
    if (!(D > 0)) continue;
    float jv = L.rjv[k], x = L.rjar[k] + alpha * jv, f = L.rf[k];
    if (f > 0) {
      float R = rg_rcp(D);
      if (x <= -R * f) { c += f * (-0.5f * R * f - x); g += -f * jv; }
      else if (x >= R * f) { c += f * (-0.5f * R * f + x); g += f * jv; }
      else { c += 0.5f * D * x * x; g += D * x * jv; h += D * jv * jv; }
    } else if (x < 0) { c += 0.5f * D * x * x; g += D * x * jv; h += D * jv * jv; }
  }
#pragma unroll
  for (int k = 0; k < RG_PSLOTS; k++) {
    float D = L.pD[k], jv = L.pjv[k], x = L.pjar[k] + alpha * jv;
    if (D > 0 && x < 0) { c += 0.5f * D * x * x; g += D * x * jv; h += D * jv * jv; }
  }
  LsPt p;
  p.cost = wave_sum(c) + alpha * alpha * q2 + alpha * q1 + q0;
  p.grad = wave_sum(g) + 2 * alpha * q2 + q1;
  p.hess = wave_sum(h) + 2 * q2;
  return p;
}

// Newton solver on the primal problem (see oracle ro_solve), in the compact space of constrained dofs
// (trees no constraint row can touch keep qacc = qacc_smooth).  Result: s.qacc, s.qfrc_con (full space).
template <bool SENSORS> __device__ __forceinline__ int rg_solve(RgM m, RgLds& s, const float* P, int& nefc_out, int flags, float warm /* qacc_warmstart[LANE] */) {
  long long t0 = rg_clock(), t1;
#define PROFS(k) do { if (flags & 2) { t1 = rg_clock(); if (LANE == 0) s.prof[k] += (float)(t1 - t0); t0 = t1; } } while (0)
#ifdef RG_FINE_PROF
  long long tf0 = rg_clock(), tf1;
#define PROFF(k) do { if (flags & 2) { tf1 = rg_clock(); if (LANE == 0) s.prof[k] += (float)(tf1 - tf0); tf0 = tf1; } } while (0)
#else
#define PROFF(k) do { } while (0)
#endif
  int nv = m.nv, nvc = m.nvc, hs = m.hs, ns = nsrow(m), ncon = s.ncon;
  // count active rows (diagnostic only)
  RgMEnt ME; rg_M_ent_load(m, ME);
  RowRegs RR;
  rg_static_rows(m, s, P, RR);
  { float cnt = 0;
#pragma unroll
    for (int k = 0; k < RG_RSLOTS; k++) cnt += RR.D[k] > 0 ? 1.f : 0.f;
    PFOR(c, ncon) cnt += (float)npyr(s.c_dim[c]); nefc_out = (int)(wave_sum(cnt) + 0.5f); }
  float scale = 1.0f / (m.meaninertia * (nv > 1 ? nv : 1));
#ifndef RG_TOL_FLOOR
#define RG_TOL_FLOOR 3e-7f   /* fp32 cannot resolve (scaled) cost improvements below this: 1e-7 ran 0.33 more iterations per substep with
                                identical re-synchronised errors vs the fp64 oracle at its 1e-8 (profiles/r02_ab.txt, tools/parity_quick.py) */
#endif
  float tol = fmaxf(m.tolerance, RG_TOL_FLOOR);
  { const int d = LANE < nvc ? s.c2d[LANE] : 0; const float w = __shfl(warm, d);
    if (LANE < nvc) { s.as[LANE] = s.qacc_smooth[d]; s.fs[LANE] = s.qfrc_smooth[d]; s.a[LANE] = w; } }
  SYNC();
  // When every contact couples the dofs of ONE chain only (a body against a static geom, or against one of its own
  // ancestors), J'DJ has the pattern of M and the Hessian is factorised tree-sparsely like M instead of densely.
#ifdef RG_NO_TREE_NEWTON
  bool tree = false;
#else
  bool tree = m.tree_newton_ok != 0;
#endif
  if (tree) {
    bool cross = false;
    PFOR(c, ncon) {
      int gg = m.pair_gg[s.c_pair[c]], b1 = m.geom_bodyid[gg & 255], b2 = m.geom_bodyid[gg >> 8];
      uint32_t a0 = m.body_dofmask[2 * b1], a1 = m.body_dofmask[2 * b1 + 1], c0 = m.body_dofmask[2 * b2], c1 = m.body_dofmask[2 * b2 + 1];
      cross = cross || (((a0 & ~c0) | (a1 & ~c1)) != 0 && ((c0 & ~a0) | (c1 & ~a1)) != 0);
    }
    tree = __ballot(cross) == 0;
  }
#pragma unroll
  for (int k = 0; k < RG_RSLOTS; k++) { RR.jar[k] = RR.jv[k] = 0.f; RR.quad[k] = 0; }
#pragma unroll
  for (int k = 0; k < RG_PSLOTS; k++) RR.pjar[k] = RR.pjv[k] = 0.f;
  // reference accelerations of the pyramid rows (mj_referenceConstraint): aref = aref0(contact) - kb(contact) * (J_row qvel)
  PFOR(i, nvc) s.search[i] = s.qvel[s.c2d[i]];
  SYNC();
  rg_J_mul<false>(m, s, RR, s.search, true);
#pragma unroll
  for (int k = 0; k < RG_PSLOTS; k++) {
    int w = LANE + RG_WAVE * k, cc = w / 6;
    RR.paref[k] = w < ncon * 6 ? s.c_aref0[cc] - s.c_kb[cc] * RR.pjv[k] : 0.f; RR.pjv[k] = 0.f;
  }
  // (the tree-pattern descriptors are fetched where they are used — 32 registers that would otherwise stay live, or be
  //  spilled, through the whole dense path as well)
  int cblk_own = LANE < nvc ? s.cblk[LANE] : 0, akk_own = (cblk_own & 0xFFFF) + LANE - ((cblk_own >> 16) & 255);
  // warm start: the better of qacc_smooth and qacc_warmstart (evaluated last, so Ma / jar are left valid for it)
  float cost_pick[2];
  for (int pass = 0; pass < 2; pass++) {
    const float* a = pass == 0 ? s.as : s.a;
    float g = 0;
    if (pass == 1) rg_M_mul(m, s, ME, a, s.Ma);    // (at a = qacc_smooth the Gauss term is exactly zero: no M a needed to price it)
    rg_J_mul(m, s, RR, a, false);
    if (pass == 1) { PFOR(i, nvc) g += 0.5f * (s.Ma[i] - s.fs[i]) * (a[i] - s.as[i]); g = wave_sum(g); }
    bool chg; int nsd; cost_pick[pass] = g + rg_constraint_update(m, s, RR, chg, nsd);
  }
  if (!(cost_pick[1] < cost_pick[0])) {
    PFOR(i, nvc) s.a[i] = s.as[i];
    SYNC();
    rg_M_mul(m, s, ME, s.a, s.Ma);
    rg_J_mul(m, s, RR, s.a, false);
  }
  // Invariant at the top of every iteration: Ma = M a and jar = J a - aref (both linear in a, so they are
  // advanced by alpha * (M s, J s) after the line search instead of being recomputed).
  // The factor of H is kept while the set of quadratic rows stays the same (H depends on nothing else).
  PROFF(24);
  float cost = 0, oldcost = 0; int iters = 0; bool have_factor = false;
  bool fresh_rhs = false; const bool rhs_row = (nvc + 1) * hs <= RG_HWORDS && !(flags & 64);   // (room for the right-hand-side row under the work matrix; flag bit 6: test hook, the separate forward substitution)
  const bool regchol = rhs_row && rg_cholreg_ok(nvc);   // the dense step in registers (rg_chol_inv_solve); flag bit 6 therefore also selects the LDS factorisation
  for (int iter = 0;; iter++) {
    float gauss = 0; PFOR(i, nvc) gauss += 0.5f * (s.Ma[i] - s.fs[i]) * (s.a[i] - s.as[i]);
    gauss = wave_sum(gauss);
    bool flags_changed; int nset; float cc = rg_constraint_update(m, s, RR, flags_changed, nset);
    oldcost = cost; cost = gauss + cc;
#ifdef RG_FINE_PROF
    if ((flags & 2) && iter > 0 && flags_changed && LANE == 0) { const float k = s.prof[47]; s.prof[36] += 1.f; s.prof[37] += k; s.prof[38] += k <= 2.f ? 1.f : 0.f; s.prof[39] += k <= 4.f ? 1.f : 0.f; s.prof[40] += k <= 8.f ? 1.f : 0.f; }
    if ((flags & 2) && iter > 0 && !flags_changed && LANE == 0) s.prof[41] += 1.f;
#endif
    PROFF(25);
    rg_JT_force(m, s, RR, s.jtf);
    PROFF(26);
    float gn = 0; PFOR(i, nvc) { float gi = s.Ma[i] - s.fs[i] - s.jtf[i]; s.search[i] = -gi; gn += gi * gi; }
    gn = sqrtf(wave_sum(gn)) * scale;
#ifdef RG_EMUL_TRACE
    if (LANE == 0) printf("  newton it %d cost %.9e gn %.3e improvement %.3e\n", iter, cost, gn, scale * (oldcost - cost));
#endif
    if (iter > 0 && scale * (oldcost - cost) < tol) break;
    if (gn < tol || iter >= m.iterations) break;
    iters = iter + 1;
    PROFS(12); PROFF(27);
    // Factorise again?  Tree-sparse / LDS paths: whenever a row changed zone since the last iteration.  Register path: the inverse factor in the work
    // matrix stays valid while the rows that differ from ITS zones (nset) are few -- those iterations correct the solve (rg_cholinv_woodbury).
    const bool invfac = regchol && !tree;   // (the work matrix holds the inverse factor W)
    bool refactor = !have_factor || (invfac ? nset > RG_WOODBURY : flags_changed), corrected = false;
#ifdef RG_WOODBURY_CHECK   /* emulation-harness self check (tests/test_kernel_emul.py::test_woodbury_correction_matches_refactorisation_emul): every corrected solve is also done by refactorisation and the two are compared */
    float dbg_g = LANE < nvc ? s.search[LANE] : 0.f, dbg_xw = 0.f; bool dbg_chk = false;
#endif
#ifdef RG_FINE_PROF
    const long long twb0 = rg_clock();
#endif
    if (!refactor && invfac && nset > 0) { corrected = rg_cholinv_woodbury(m, s, RR, s.search); if (!corrected) refactor = true; }
#ifdef RG_FINE_PROF
    if ((flags & 2) && LANE == 0 && !(!have_factor) && invfac && nset > 0 && nset <= RG_WOODBURY) { s.prof[45] += 1.f; s.prof[46] += (float)(rg_clock() - twb0); if (!corrected) s.prof[49] += 1.f; }
    if ((flags & 2) && LANE == 0 && refactor) s.prof[48] += 1.f;
#endif
#ifdef RG_WOODBURY_CHECK
    if (corrected) { dbg_xw = LANE < nvc ? s.search[LANE] : 0.f; SYNC(); if (LANE < nvc) s.search[LANE] = dbg_g; SYNC(); corrected = false; refactor = true; dbg_chk = true; }
#endif
    if (refactor) {
    have_factor = true;
    if (tree) {
      // H <- M in the block layout (the factor only uses the lower triangle), then + J' D J on the same addresses
      rg_M_to_blocks(m, s);
#pragma unroll
      for (int k = 0; k < RG_RSLOTS; k++) { int r = LANE + RG_WAVE * k; if (r < ns && RR.D[k] > 0 && (RR.quad[k] & 1)) srow_hess_tree(s, RR.desc[k], RR.D[k]); }
    } else {
    // H = M + J' D J over the quadratic rows (LDS atomics from one wave: in-order, deterministic)
    {  // H <- M: zero the compact nvc x hs matrix, then scatter the tree-sparse entries (lower triangle: what the factorisation reads)
      rgf4 z; z.x = z.y = z.z = z.w = 0.f; rgf4* H4 = (rgf4*)s.H;
      for (int w = LANE; w < nvc * (hs >> 2); w += RG_WAVE) H4[w] = z;
      SYNC();
#pragma unroll
      for (int k = 0; k < RG_MSLOTS; k++) {
        int w = ME.w[k];
        if (w >> 16) { int ci = w & 255, cj = (w >> 8) & 255; s.H[(ci > cj ? ci : cj) * hs + (ci > cj ? cj : ci)] = s.Msp[LANE + RG_WAVE * k]; }
      }
    }
    SYNC();
#pragma unroll
    for (int k = 0; k < RG_RSLOTS; k++) { int r = LANE + RG_WAVE * k; if (r < ns && RR.D[k] > 0 && (RR.quad[k] & 1)) srow_hess(m, s, RR.desc[k], RR.D[k]); }
    }
    // per contact, C = P' D_act P in the basis (normal, t1, t2, spin) has only its first row/column and its diagonal
    // non-zero: cn, ck[3], cd[3].  Lane c computes them for contact c; the block loop below reads them lane to lane.
    static_assert(RG_MAXCON <= RG_WAVE, "one lane per contact");
    PROFF(28);
    // Round 6: a lane per (contact, block row).  Four contacts per trip, one 16-lane group each (a row of a contact has <= RG_W = 14 non-zeros);
    // lane (c, a) forms the contact's coefficients itself (one round of independent LDS reads, no staging), folds them with its own row's basis
    // entries -- v(a, b) = n_b w0 + sum_k t_kb w_k -- and walks b = 0..a, two entries in flight.  (Before: one contact after the other, a
    // lane per (a, b) pair with the coefficients staged through LDS: 3.7 k cycles per contact, all of it LDS latency.)
    static_assert(RG_W <= 16, "one 16-lane group per contact");
    for (int c0 = 0; c0 < ncon; c0 += 4) {
      const int c = c0 + (LANE >> 4), a = LANE & 15, cc = c < ncon ? c : c0;
      const int dim = s.c_dim[cc], nnz = s.c_nnz[cc], nb = nbasis(dim);
      const float D = s.c_D[cc], mu0 = s.c_mu[2 * cc], mu1 = s.c_mu[2 * cc + 1];
      const unsigned char* pq = s.p_quad + 6 * cc;
      const int q0 = pq[0] & 1, q1 = pq[1] & 1, q2 = pq[2] & 1, q3 = pq[3] & 1, q4 = pq[4] & 1, q5 = pq[5] & 1;
      float cn, ck0 = 0.f, ck1 = 0.f, ck2 = 0.f, cd0 = 0.f, cd1 = 0.f, cd2 = 0.f;
      if (dim == 1) cn = q0 ? D : 0.f;
      else {
        cn = D * (float)(q0 + q1); ck0 = D * mu0 * (float)(q0 - q1); cd0 = D * mu0 * mu0 * (float)(q0 + q1);
        if (dim > 2) { cn += D * (float)(q2 + q3); ck1 = D * mu0 * (float)(q2 - q3); cd1 = D * mu0 * mu0 * (float)(q2 + q3); }
        if (dim > 3) { cn += D * (float)(q4 + q5); ck2 = D * mu1 * (float)(q4 - q5); cd2 = D * mu1 * mu1 * (float)(q4 + q5); }
      }
      if (c < ncon && a < nnz && cn != 0.f) {
        const float* Bc = s.c_pool + s.c_off[cc];
        const unsigned char* ix = s.c_idx + cc * RG_W;
        const float na = Bc[a], ta0 = nb > 1 ? Bc[nnz + a] : 0.f, ta1 = nb > 2 ? Bc[2 * nnz + a] : 0.f, ta2 = nb > 3 ? Bc[3 * nnz + a] : 0.f;
        const int ia = ix[a];
        const float w0 = cn * na + ck0 * ta0 + ck1 * ta1 + ck2 * ta2, w1 = ck0 * na + cd0 * ta0, w2 = ck1 * na + cd1 * ta1, w3 = ck2 * na + cd2 * ta2;
        for (int b = 0; b <= a; b += 2) {
          const bool two = b + 1 <= a; const int b1 = two ? b + 1 : b;
          const float n0 = Bc[b], n1 = Bc[b1];
          const float s0 = nb > 1 ? Bc[nnz + b] : 0.f, s1 = nb > 1 ? Bc[nnz + b1] : 0.f;
          const float u0 = nb > 2 ? Bc[2 * nnz + b] : 0.f, u1 = nb > 2 ? Bc[2 * nnz + b1] : 0.f;
          const float r0 = nb > 3 ? Bc[3 * nnz + b] : 0.f, r1 = nb > 3 ? Bc[3 * nnz + b1] : 0.f;
          const int i0 = ix[b], i1 = ix[b1];
          const float v0 = n0 * w0 + s0 * w1 + u0 * w2 + r0 * w3, v1 = n1 * w0 + s1 * w1 + u1 * w2 + r1 * w3;
          const int h0 = ia > i0 ? ia : i0, l0 = ia > i0 ? i0 : ia, h1 = ia > i1 ? ia : i1, l1 = ia > i1 ? i1 : ia;
          if (!tree) { atomicAdd(s.H + h0 * hs + l0, v0); if (two) atomicAdd(s.H + h1 * hs + l1, v1); }
          else {
            const int k0 = s.cblk[h0], k1 = s.cblk[h1];
            atomicAdd(s.H + (k0 & 0xFFFF) + l0 - ((k0 >> 16) & 255), v0);
            if (two) atomicAdd(s.H + (k1 & 0xFFFF) + l1 - ((k1 >> 16) & 255), v1);
          }
        }
      }
    }
    // the zones this factor is built with (bit 1 of the flag words)
#pragma unroll
    for (int k = 0; k < RG_RSLOTS; k++) RR.quad[k] = (RR.quad[k] & 1) * 3;
#pragma unroll
    for (int kk = 0; kk < RG_PSLOTS; kk++) { const int w = LANE + RG_WAVE * kk; if (w < ncon * 6) s.p_quad[w] = (unsigned char)((s.p_quad[w] & 1) * 3); }
    // dense path: the gradient rides through the factorisation as one more row (row nvc), so the forward substitution costs nothing
    if (!tree && rhs_row) { if (LANE < hs) s.H[nvc * hs + LANE] = LANE < nvc ? s.search[LANE] : 0.f; fresh_rhs = true; }
    SYNC();
    PROFS(13); PROFF(29);
    if (tree) { LtdlDesc LT; rg_ltdl_load(m.ltdl_tri_c, m.ltdl_pair_c, m.n_tri_rounds_c, m.n_pair_rounds_c, LT); rg_ltdl_factor(s, LT); }
    else if (regchol) rg_chol_inv_solve(m, s, s.search);   // factorisation AND solution (the right-hand side rode along)
    else if (rhs_row) rg_chol<true>(m, s); else rg_chol<false>(m, s);
    }
    PROFS(14);
    if (tree) { LtdlDesc LT; rg_ltdl_load((const int*)0, m.ltdl_pair_c, 0, m.n_pair_rounds_c, LT); rg_ltdl_solve(s, LT, s.search, LANE, LANE < nvc, akk_own); }
    else if (regchol) { if (!fresh_rhs && !corrected) rg_cholinv_apply(m, s, s.search); }
    else if (fresh_rhs) rg_chol_solve_bwd(m, s, s.search); else rg_chol_solve(m, s, s.search);
    fresh_rhs = false;
#ifdef RG_WOODBURY_CHECK
    if (dbg_chk) { float xf = LANE < nvc ? s.search[LANE] : 0.f; float e = wave_max(fabsf(xf - dbg_xw)), n = wave_max(fabsf(xf)); if (LANE == 0) { const float rel = e / (n + 1e-30f); if (rel > rg_wchk_max) rg_wchk_max = rel; rg_wchk_n++; rg_wchk_rows += nset; } }
#endif
    PROFS(15); PROFF(30);
    // exact line search along `search`
    rg_M_mul(m, s, ME, s.search, s.Mv);
    PROFF(31);
    rg_J_mul(m, s, RR, s.search, true);
    PROFF(32);
    float q1 = 0, q2 = 0, sn = 0;
    PFOR(i, nvc) { q1 += s.search[i] * (s.Ma[i] - s.fs[i]); q2 += 0.5f * s.search[i] * s.Mv[i]; sn += s.search[i] * s.search[i]; }
    q1 = wave_sum(q1); q2 = wave_sum(q2); sn = sqrtf(wave_sum(sn));
    if (sn < 1e-15f) break;
    float gtol = tol * 0.01f * sn / scale;
    LsRows rows; rg_ls_load(m, s, RR, rows);
    LsPt p0 = rg_ls_eval(rows, 0.f, gauss, q1, q2);
    float alpha = 0;
    if (p0.grad < 0 && p0.hess > 0) {
      float lo = 0, hi = -1, glo = p0.grad, hlo = p0.hess, ghi = 0, hhi = 1;
      alpha = -p0.grad * rg_rcp(p0.hess);
      for (int it = 0; it < 12; it++) {
        LsPt p = rg_ls_eval(rows, alpha, gauss, q1, q2);
        if (fabsf(p.grad) < gtol) break;
        if (p.grad < 0) { lo = alpha; glo = p.grad; hlo = p.hess; } else { hi = alpha; ghi = p.grad; hhi = p.hess; }
        float cand = lo - glo * rg_rcp(hlo);
        if (hi >= 0 && !(cand > lo && cand < hi)) { cand = hi - ghi * rg_rcp(hhi); if (!(cand > lo && cand < hi)) cand = 0.5f * (lo + hi); }
        if (cand == alpha) break;
        alpha = cand;
      }
    }
#ifdef RG_EMUL_TRACE
    if (LANE == 0) printf("     alpha %.6e p0.grad %.3e p0.hess %.3e gtol %.3e\n", alpha, p0.grad, p0.hess, gtol);
#endif
    PROFF(33);
    if (alpha == 0) break;
    PFOR(i, nvc) { s.a[i] += alpha * s.search[i]; s.Ma[i] += alpha * s.Mv[i]; }
#pragma unroll
    for (int k = 0; k < RG_RSLOTS; k++) RR.jar[k] += alpha * RR.jv[k];
#pragma unroll
    for (int k = 0; k < RG_PSLOTS; k++) RR.pjar[k] += alpha * RR.pjv[k];
    SYNC();
    PROFS(10); PROFF(34);
  }
  PROFF(35);
  // Every exit of the loop leaves s.jtf = J' f(a) for the final a (the gradient evaluation at the top of the pass that
  // broke out, or the pass whose step length came out zero), so the forces at the solution are already there; only the
  // expansion to the full dof space remains.  (RG_SOLVE_RECOMPUTE: evaluate them once more from J a - aref computed from
  // scratch instead of the incrementally advanced residuals: identical to ~1e-7 relative, and 3.5 % of the step.)
#ifdef RG_SOLVE_RECOMPUTE
  rg_J_mul(m, s, RR, s.a, false);
  { bool chg; int nsd; rg_constraint_update(m, s, RR, chg, nsd); }
  rg_JT_force(m, s, RR, s.jtf);
#endif
  if (SENSORS) {   // sensor pass (its own instantiation: the hot path's solver carries none of this): normal force of every contact = sum of its pyramid edge forces (mju_decodePyramid), f = -D jar where jar < 0
    PFOR(c, ncon) s.c_bdot[c] = 0.f;
    SYNC();
#pragma unroll
    for (int k = 0; k < RG_PSLOTS; k++) {
      int w = LANE + RG_WAVE * k;
      if (w < ncon * 6) { int cc = w / 6; if ((w - 6 * cc) < npyr(s.c_dim[cc]) && RR.pjar[k] < 0) atomicAdd(s.c_bdot + cc, -s.c_D[cc] * RR.pjar[k]); }
    }
    SYNC();
  }
  SYNC();   // (qacc / qfrc_con overlay as, fs, search, Mv)
  PFOR(d, nv) { int i = m.d2c[d]; s.qacc[d] = i >= 0 ? s.a[i] : s.qacc_smooth[d]; s.qfrc_con[d] = i >= 0 ? s.jtf[i] : 0.f; }
  SYNC();
  return iters;
}

// ------------------------------------------------------------------------------------------------- integration
__device__ __forceinline__ void rg_euler(RgM m, RgLds& s, const float* P, float& warm) {
  float h = P[RG_PRM_TIMESTEP];
  PFOR(i, m.nv) s.tmpv[i] = s.qfrc_smooth[i] + s.qfrc_con[i];
  SYNC();
  rg_ltdl_factor_solve(m, s, P + RG_PRM_DOF_DAMPING, h, s.tmpv);
  if (LANE < m.nv) { s.qvel[LANE] += h * s.tmpv[LANE]; warm = s.qacc[LANE]; }
  SYNC();
  PFOR(j, m.njnt) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j], t = m.jnt_type[j];
    if (t == RG_JNT_FREE) { for (int k = 0; k < 3; k++) s.qpos[qa + k] += h * s.qvel[da + k]; qa += 3; da += 3; }
    if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
      v3 w = ld3(s.qvel + da); float ang = norm(w) * h;
      if (ang > 0) { q4 q = qnormalize(qmul(ldq(s.qpos + qa), axisangle(normalized(w), ang))); stq(s.qpos + qa, q); }
    } else s.qpos[qa] += h * s.qvel[da];
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------------- the env-step kernel

// stage dump in three parts, each taken while the arrays it reads are alive
__device__ __forceinline__ void rg_dump_kin(RgM m, RgLds& s, float* dbg) {
  PFOR(i, m.nbody * 3) dbg[RG_DBG_XPOS + i] = s.xpos[i];
  PFOR(i, m.nbody * 4) dbg[RG_DBG_XQUAT + i] = s.xquat[i];
  PFOR(i, m.nsite * 3) dbg[RG_DBG_SITE + i] = s.spos[i];
  PFOR(i, m.ntendon) dbg[RG_DBG_TENLEN + i] = s.tenlen[i];
  PFOR(i, m.ntendon * 4) dbg[RG_DBG_TENJ + i] = s.tenJ[i];
  SYNC();
}
__device__ __forceinline__ void rg_dump_pos(RgM m, RgLds& s, float* dbg) {
  for (int w = LANE; w < m.nv * m.nv; w += RG_WAVE) dbg[RG_DBG_M + w] = 0.f;
  SYNC();
  PFOR(e, m.nM) { int i = m.M_i[e], j = m.M_j[e]; dbg[RG_DBG_M + i * m.nv + j] = s.Msp[e]; dbg[RG_DBG_M + j * m.nv + i] = s.Msp[e]; }
  PFOR(i, m.nv) { dbg[RG_DBG_BIAS + i] = s.qfrc_bias[i]; dbg[RG_DBG_PASSIVE + i] = s.qfrc_passive[i]; dbg[RG_DBG_ACTFRC + i] = s.qfrc_act[i]; }
  if (LANE == 0) { dbg[RG_DBG_NCON] = (float)s.ncon; dbg[RG_DBG_NCON + 3] = (float)s.ncand; }
  PFOR(c, (s.ncon < RG_DBG_MAXCON ? s.ncon : RG_DBG_MAXCON)) { float* o = dbg + RG_DBG_CON + 8 * c; o[0] = s.c_dist[c]; o[1] = s.c_pos[3 * c]; o[2] = s.c_pos[3 * c + 1]; o[3] = s.c_pos[3 * c + 2]; o[4] = s.c_normal[3 * c]; o[5] = s.c_normal[3 * c + 1]; o[6] = s.c_normal[3 * c + 2]; o[7] = (float)s.c_pair[c]; }
  SYNC();
}
__device__ __forceinline__ void rg_dump_slv(RgM m, RgLds& s, float* dbg, int nefc, int iters) {
  PFOR(i, m.nv) { dbg[RG_DBG_QACCS + i] = s.qacc_smooth[i]; dbg[RG_DBG_QACC + i] = s.qacc[i]; }
  if (LANE == 0) { dbg[RG_DBG_NCON + 1] = (float)nefc; dbg[RG_DBG_NCON + 2] = (float)iters; }
}

// ------------------------------------------------------------------------------------------------- stage calls (definitions of RgLaunch / RgCtx / RG_STAGE: top of the file)
// ------------------------------------------------------------------------------------------------- touch sensors
// mj_sensorAcc, mjSENS_TOUCH (oracle: ro_sensor): the sum of the normal forces of the contacts on the site's body whose contact
// point sees the site's volume along the contact normal.  The geometric half runs while the position stage is alive
// (c_touch masks), the force half after the solve.
__device__ __forceinline__ bool ray_hits_sphere(v3 c, float r, v3 p, v3 v) {
  v3 w = p - c; float vv = dot(v, v);
  float t = vv > 0 ? -dot(w, v) / vv : 0.f; t = t < 0 ? 0.f : t;
  v3 q = w + v * t;
  return dot(q, q) <= r * r;
}
__device__ __forceinline__ bool ray_hits_site(int type, v3 size, v3 p, v3 v) {   // p, v in the site frame
  if (type == RG_GEOM_SPHERE) return ray_hits_sphere(mk3(0, 0, 0), size.x, p, v);
  if (type == RG_GEOM_CAPSULE || type == RG_GEOM_CYLINDER) {
    float r = size.x, h = size.y;
    float a = v.x * v.x + v.y * v.y, b = p.x * v.x + p.y * v.y, cc = p.x * p.x + p.y * p.y - r * r;
    bool in = false; float t0 = 0, t1 = 3.0e38f;
    if (a < 1e-30f) in = !(cc > 0);
    else { float det = b * b - a * cc; if (det >= 0) { float sq = sqrtf(det); t0 = (-b - sq) / a; t1 = (-b + sq) / a; in = t1 >= 0; t0 = t0 < 0 ? 0.f : t0; } }
    if (in) {
      float z0 = p.z + t0 * v.z, z1 = t1 > 1e37f ? (v.z > 0 ? 3.0e38f : (v.z < 0 ? -3.0e38f : p.z)) : p.z + t1 * v.z;
      float lo = fminf(z0, z1), hi = fmaxf(z0, z1);
      if (lo <= h && hi >= -h) return true;
    }
    if (type == RG_GEOM_CYLINDER) return false;
    return ray_hits_sphere(mk3(0, 0, h), r, p, v) || ray_hits_sphere(mk3(0, 0, -h), r, p, v);
  }
  if (type == RG_GEOM_ELLIPSOID) return ray_hits_sphere(mk3(0, 0, 0), 1.f, mk3(p.x / size.x, p.y / size.y, p.z / size.z), mk3(v.x / size.x, v.y / size.y, v.z / size.z));
  if (type == RG_GEOM_BOX) {
    float t0 = 0, t1 = 3.0e38f; float pp[3] = {p.x, p.y, p.z}, vv[3] = {v.x, v.y, v.z}, ss[3] = {size.x, size.y, size.z};
    for (int k = 0; k < 3; k++) {
      if (fabsf(vv[k]) < 1e-30f) { if (fabsf(pp[k]) > ss[k]) return false; continue; }
      float a = (-ss[k] - pp[k]) / vv[k], b = (ss[k] - pp[k]) / vv[k];
      t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
    }
    return t0 <= t1;
  }
  return false;
}
#if RG_SENSORS
__device__ __forceinline__ void rg_touch_geom(RgM m, RgLds& s, const float* P) {
  PFOR(c, s.ncon) {
    const float* R = m.pair_rec + RG_PAIRREC * s.c_pair[c];
    int hdr = __builtin_bit_cast(int, R[0]), bb = __builtin_bit_cast(int, R[19]), b1 = bb & 255, b2 = bb >> 8;
    v3 pos = ld3(s.c_pos + 3 * c), n = ld3(s.c_normal + 3 * c);
    int mask = 0;
    for (int k = 0; k < m.nsensor; k++) {
      int site = m.sensor_site[k], body = m.site_bodyid[site];
      if (body != b1 && body != b2) continue;
      // the body frames of the position stage are gone by now (their LDS was reused); the frame of the contact's own geom on
      // that body is still there: body = geom frame o inverse(geom's local frame)
      int g = body == b1 ? (hdr & 255) : ((hdr >> 8) & 255);
      q4 gl = ldq(m.geom_quat + 4 * g); gl.x = -gl.x; gl.y = -gl.y; gl.z = -gl.z;
      q4 bq = qmul(ldq(s.gquat + 4 * g), gl);
      v3 bp = ld3(s.gpos + 3 * g) - qrot(bq, ld3(m.geom_pos + 3 * g));
      q4 q = qmul(bq, ldq(m.site_quat + 4 * site));
      v3 lp = qrotT(q, pos - (bp + qrot(bq, ld3(P + RG_PRM_SITE_POS + 3 * site)))), lv = qrotT(q, body == b2 ? n * -1.0f : n);
      if (ray_hits_site(m.site_type[site], ld3(m.site_size + 3 * site), lp, lv)) mask |= 1 << k;
    }
    s.c_touch[c] = (unsigned char)mask;
  }
  SYNC();
}
// after the solve: s.c_bdot[c] holds the normal force of contact c (rg_solve, sensor pass; the array is free by then)
__device__ __forceinline__ void rg_touch_write(RgM m, RgLds& s, float* xd) {
  PFOR(k, m.nsensor) {
    float f = 0;
    for (int c = 0; c < s.ncon; c++) { float nf = s.c_bdot[c]; if (((s.c_touch[c] >> k) & 1) && nf > 0) f += nf; }
    xd[RG_XD_SENSOR + k] = f;
  }
  SYNC();
}
#endif

RG_STAGE void st_kinematics(RgCtx c) { RgM m = RG_M(c); RgLRef L = RG_L(c); rg_kinematics(m, RG_S(), rg_prm(m, L)); }
RG_STAGE void st_com_pos(RgCtx c) { RgM m = RG_M(c); rg_com_pos(m, RG_S(), rg_prm(m, RG_L(c))); }
RG_STAGE void st_tendon(RgCtx c) { rg_tendon(RG_M(c), RG_S()); }
RG_STAGE void st_crb(RgCtx c) { RgM m = RG_M(c); RgLRef L = RG_L(c); rg_crb(m, RG_S(), rg_prm(m, L), L.x.subtree_adr, L.x.subtree); }
RG_STAGE void st_velocity(RgCtx c) { RgM m = RG_M(c); RgLRef L = RG_L(c); rg_velocity(m, RG_S(), rg_prm(m, L), L.x.dof_velmask, L.x.subtree_adr, L.x.subtree); }
RG_STAGE_BIG void st_collision(RgCtx c) {
  RgM m = RG_M(c); RgLRef L = RG_L(c); RgLds& s = RG_S();
  int e = rg_env(L), flags = L.flags;
  rg_collision(c, m, s, rg_prm(m, L), (flags & 2) ? s.prof : (float*)0, L.bt.sepdir ? (rgf4*)L.bt.sepdir + (size_t)e * m.npair : (rgf4*)0,
               (L.bt.pairlb && !(flags & 4)) ? L.bt.pairlb + (size_t)e * m.npair : (float*)0, !(flags & 8));
}
RG_STAGE void st_make_constraint(RgCtx c) { RgM m = RG_M(c); rg_make_constraint(m, RG_S(), rg_prm(m, RG_L(c))); }
RG_STAGE void st_pid(RgCtx c, float* pid) { RgM m = RG_M(c); rg_pid(m, RG_S(), rg_prm(m, RG_L(c)), pid); }
RG_STAGE void st_smooth(RgCtx c) { RgM m = RG_M(c); rg_smooth(m, RG_S(), rg_prm(m, RG_L(c))); }
RG_STAGE void st_factor_smooth(RgCtx c) { RgLds& s = RG_S(); rg_ltdl_factor_solve(RG_M(c), s, (const float*)0, 0.f, s.qacc_smooth); }
// The Newton stage runs at raised wave priority (s_setprio): it is the kernel's longest dependent instruction chain (the register elimination alone takes 5.3 k cycles
// for a wave by itself and 15 k between two others), while the waves it shares its SIMD with are mostly in stages with independent work in flight -- the arbiter
// then serves the chain first.  Measured +0.4 ... +0.6 % on the bench line (profiles/r06_ab_newton.txt); level 3, or raising the collision stage / the tree sweeps as
// well, measured no better.  -DRG_NO_PRIO builds without it.
#if !defined(RG_EMUL) && !defined(RG_NO_PRIO)
#define RG_SOLVE_PRIO(level) __builtin_amdgcn_s_setprio(level)
#else
#define RG_SOLVE_PRIO(level) do { } while (0)
#endif
RG_STAGE_BIG int st_solve(RgCtx c, float warm) { RgM m = RG_M(c); int nefc = 0; RG_SOLVE_PRIO(2); int it = rg_solve<false>(m, RG_S(), rg_prm(m, RG_L(c)), nefc, RG_L(c).flags, warm); RG_SOLVE_PRIO(0); return it | (nefc << 8); }
#if RG_SENSORS
RG_STAGE_BIG int st_solve_sensors(RgCtx c, float warm) { RgM m = RG_M(c); int nefc = 0; RG_SOLVE_PRIO(2); int it = rg_solve<true>(m, RG_S(), rg_prm(m, RG_L(c)), nefc, RG_L(c).flags, warm); RG_SOLVE_PRIO(0); return it | (nefc << 8); }
#endif
#if RG_SENSORS
RG_STAGE void st_touch_geom(RgCtx c) { RgM m = RG_M(c); rg_touch_geom(m, RG_S(), rg_prm(m, RG_L(c))); }
RG_STAGE void st_touch_write(RgCtx c) { RgM m = RG_M(c); RgLRef L = RG_L(c); rg_touch_write(m, RG_S(), L.bt.xdata + (size_t)rg_env(L) * RG_XDATA); }
#endif
RG_STAGE float st_euler(RgCtx c, float warm) { RgM m = RG_M(c); rg_euler(m, RG_S(), rg_prm(m, RG_L(c)), warm); return warm; }
RG_STAGE void st_build_row_desc(RgCtx c) { rg_build_row_desc(RG_M(c), RG_S()); }
RG_STAGE void st_dump(RgCtx c, int which, int nefc, int iters) {
  RgM m = RG_M(c); RgLRef L = RG_L(c); RgLds& s = RG_S();
  float* dbg = L.bt.dbg + (size_t)rg_env(L) * RG_DBG_SIZE;
  if (which == 0) rg_dump_kin(m, s, dbg); else if (which == 1) rg_dump_pos(m, s, dbg); else rg_dump_slv(m, s, dbg, nefc, iters);
}

// kernel arguments: the model descriptor's device address at kernarg offset 0, the launch descriptor (8-byte
// aligned) at offset 8 of the kernarg segment
#ifdef RG_EMUL
#define RG_MAKE_CTX() RgCtx c{mp, &launch}
#else
#define RG_MAKE_CTX() RgCtx c{mp, (const void*)((const RG_AS4 char*)__builtin_amdgcn_kernarg_segment_ptr() + 8)}
#endif
// (RG_WAVES_PER_SIMD 3: <= 168 VGPRs, so that the LDS-permitted 9 envs per CU are also register-permitted; measured
//  +3 % over the 256-VGPR build at 8 per CU in spite of ~80 spilled registers, profiles/r02_ab.txt)
#ifndef RG_WAVES_PER_SIMD
#define RG_WAVES_PER_SIMD 3
#endif
#if !RG_ITEMS
__global__ void __launch_bounds__(RG_WAVE, RG_WAVES_PER_SIMD) rg_step_kernel(const RgModelDev* mp, RgLaunch launch) {
  RG_MAKE_CTX();
  RgM m = RG_M(c);
  RgLRef L = RG_L(c);
  RgLds& s = RG_S();
  const int nsubsteps = L.nsubsteps, nforward_ticks = L.nforward_ticks, flags = L.flags;
  if ((int)blockIdx.x >= L.bt.B) return;
  int e = rg_env(L);
  if (L.bt.active && !L.bt.active[e]) return;
  // flags bit 8: `active` is the redo array of a rollout launch, whose entries say where to resume: entry - 1 = the first substep
  // that is still to do (the substep-granular rollout kernel hands an env over in the middle of its env.step)
  const int sub0 = ((flags & 256) && L.bt.active) ? L.bt.active[e] - 1 : 0;
  long long tk0 = rg_clock();
  // ---- load the env's state row
  PFOR(i, m.nq) s.qpos[i] = L.bt.qpos[(size_t)e * m.nq + i];
  float warmr = 0.f, pidr[3] = {0.f, 0.f, 0.f};
  if (LANE < m.nv) { s.qvel[LANE] = L.bt.qvel[(size_t)e * m.nv + LANE]; warmr = L.bt.qacc_warmstart[(size_t)e * m.nv + LANE]; }
  if (LANE < m.nu) for (int k = 0; k < 3; k++) pidr[k] = L.bt.pid[(size_t)e * 3 * m.nu + 3 * LANE + k];
  const float* P = rg_prm(m, L);
  { float nz = 0; PFOR(i, 6 * m.nbody) nz += P[RG_PRM_XFRC + i] != 0.f ? 1.f : 0.f; nz = wave_sum(nz); if (LANE == 0) s.has_xfrc = nz > 0; }
  const unsigned status0 = L.bt.status[e];
  if (LANE == 0) s.status = status0;
  if ((L.flags & 2) && LANE < RG_NPROF) s.prof[LANE] = 0;
  st_build_row_desc(c);
  // ---- action -> ctrl (robot_interface.py:247-278 with the hand's position->control matrix)
  if (L.bt.preticks && sub0 == 0) {
    // reset_goal's two state-less forwards (robot_env.py:893-909 -> _observe_sync), booked by rg_env_post_step: same state,
    // same stored ctrl as when they were owed, so running them here is the same arithmetic
    const int pre = L.bt.preticks[e];
    if (pre > 0) {
      PFOR(u, m.nu) s.ctrl[u] = L.bt.ctrl[(size_t)e * m.nu + u];
      SYNC();
      st_kinematics(c); st_com_pos(c); st_tendon(c);
      for (int k = 0; k < pre; k++) st_pid(c, pidr);
      // (cleared in the write-back section: an env that is handed to the large configuration through `redo` returns
      //  without writing its PID row, so the ticks must still be owed when that launch picks it up)
    }
  }
  // envs on `hold` (scripted reset recipe) and envs whose action row holds a non-finite entry keep their stored ctrl row
  bool use_action = L.bt.action && !(L.bt.hold && L.bt.hold[e]) && sub0 == 0;   // (a resumed env.step: the action went into the stored ctrl row at its first substep)
  if (use_action) {
    float nf = 0; PFOR(u, m.nu) nf += (fabsf(L.bt.action[(size_t)e * m.nu + u]) <= 3.0e38f) ? 0.f : 1.f;
    if (wave_sum(nf) > 0) { use_action = false; if (LANE == 0) s.status |= RG_STATUS_BAD_ACTION; }
  }
  if (use_action) {
    PFOR(u, m.nu) {
      float lo = P[RG_PRM_ACT_CTRLRANGE + 2 * u], hi = P[RG_PRM_ACT_CTRLRANGE + 2 * u + 1], centre;
      if (L.env.relative_action) { centre = 0; for (int j = 0; j < L.env.n_hand_jnt; j++) centre += L.env.pos_to_ctrl[u * L.env.n_hand_jnt + j] * s.qpos[L.env.hand_qposadr + j]; }
      else centre = 0.5f * (hi + lo);
      float a = clampf(L.bt.action[(size_t)e * m.nu + u], -1.f, 1.f);
      s.ctrl[u] = clampf(centre + a * 0.5f * (hi - lo), lo, hi);
    }
  } else { PFOR(u, m.nu) s.ctrl[u] = L.bt.ctrl[(size_t)e * m.nu + u]; }
  SYNC();
  float st_ncon = 0, st_nefc = 0, st_iter = 0;
  bool bad = false;
  for (int sub = sub0; sub < nsubsteps; sub++) {
    // mj_checkPos / mj_checkVel
    float bd = 0; PFOR(i, m.nq) bd += (fabsf(s.qpos[i]) < 1e10f) ? 0.f : 1.f; PFOR(i, m.nv) bd += (fabsf(s.qvel[i]) < 1e10f) ? 0.f : 1.f;
    if (wave_sum(bd) > 0) { bad = true; break; }
    long long t0 = rg_clock(), t1;
#define PROF(k) do { if (flags & 2) { t1 = rg_clock(); if (LANE == 0) s.prof[k] += (float)(t1 - t0); t0 = t1; } } while (0)
#ifdef RG_EMUL_POISON
    { unsigned int* u = (unsigned int*)s.H; int nw = (int)((sizeof(RgLds) - ((char*)s.H - (char*)&s)) / 4); for (int w = LANE; w < nw; w += RG_WAVE) u[w] = 0x7fc00000u; SYNC(); }
#endif
    st_kinematics(c); PROF(0);
    st_com_pos(c); PROF(1);
    st_tendon(c); PROF(2);
    if (sub == 0 && (flags & 1) && L.bt.dbg) st_dump(c, 0, 0, 0);
    st_crb(c); PROF(3);
    st_velocity(c); PROF(7);
    st_collision(c); PROF(6);
    if ((flags & 2) && LANE == 0 && (float)s.ncon > s.prof[23]) s.prof[23] = (float)s.ncon;
    st_make_constraint(c); PROF(8);
    if (L.bt.xdata && sub == nsubsteps - 1) {   // data.ncon / data.contact[i].{geom1, geom2, dist} of the last mj_step
      float* xd = L.bt.xdata + (size_t)e * RG_XDATA;
      const int nc = s.ncon < RG_DBG_MAXCON ? s.ncon : RG_DBG_MAXCON;
      if (LANE == 0) xd[RG_XD_NCON] = (float)s.ncon;
      PFOR(ci, nc) { int gg = m.pair_gg[s.c_pair[ci]]; float* o = xd + RG_XD_CONTACT + 3 * ci; o[0] = (float)(gg & 255); o[1] = (float)(gg >> 8); o[2] = s.c_dist[ci]; }
    }
    if (L.bt.redo && ((s.status & ~status0 & (RG_STATUS_CON_FULL | RG_STATUS_CAND_FULL)) || ((RG_L(c).flags & 512) && s.ncon > 5))) {   // (flags bit 9: test hook, "more than 5 contacts do not fit")
      // more contacts / candidates than this configuration holds: leave the env exactly as it was (state rows are
      // written at the end; the distance-bound cache was advanced by the substeps done so far, so it is voided) and
      // hand it to the large configuration
      if (L.bt.pairlb) { float* lb = L.bt.pairlb + (size_t)e * m.npair; PFOR(i, m.npair) lb[i] = 0.f; }
      if (LANE == 0) L.bt.redo[e] = 1;
      return;
    }
    st_pid(c, pidr);
    st_smooth(c); PROF(9);
    if (sub == 0 && (flags & 1) && L.bt.dbg) st_dump(c, 1, 0, 0);
    // ---- the position-stage scratch is dead from here on; the solver scratch takes its place
#ifdef RG_EMUL_POISON
    { unsigned int* u = (unsigned int*)s.H; int nw = (int)((sizeof(RgLds) - ((char*)s.H - (char*)&s)) / 4); for (int w = LANE; w < nw; w += RG_WAVE) u[w] = 0x7fc00000u; SYNC(); }
#endif
    st_factor_smooth(c); PROF(4);
    int packed = st_solve(c, warmr), iters = packed & 255, nefc = packed >> 8; t0 = rg_clock();
    st_ncon += s.ncon; st_nefc += nefc; st_iter += iters;
    if (sub == 0 && (flags & 1) && L.bt.dbg) st_dump(c, 2, nefc, iters);
    bd = 0; PFOR(i, m.nv) bd += (fabsf(s.qacc[i]) < 1e10f) ? 0.f : 1.f;
    if (wave_sum(bd) > 0) { bad = true; break; }
    warmr = st_euler(c, warmr); PROF(11);
  }
  if (bad && LANE == 0) s.status |= RG_STATUS_BAD_STATE;
  // ---- state-less forward() calls of the reference (simulation_interface.py:185, robot_env.py:677,
  //      observation/mujoco.py:22-27): only their PID-controller side effect touches the state
  const int nticks = L.bt.nticks ? L.bt.nticks[e] : nforward_ticks;
  if (nticks > 0 || L.bt.obs || L.bt.xdata) {
    st_kinematics(c);
    if (L.bt.xdata) {   // body / site frames of the final state, before the later stages reuse their LDS
      float* xd = L.bt.xdata + (size_t)e * RG_XDATA;
      PFOR(i, 3 * m.nbody) xd[RG_XD_XPOS + i] = s.xpos[i];
      PFOR(i, 4 * m.nbody) xd[RG_XD_XQUAT + i] = s.xquat[i];
      PFOR(i, 3 * m.nsite) xd[RG_XD_SITE_XPOS + i] = s.spos[i];
      if (nsubsteps == 0 && LANE == 0) xd[RG_XD_NCON] = 0.f;
    }
    st_com_pos(c); st_tendon(c);
#if RG_SENSORS
    const bool sensors = (flags & 32) && L.bt.xdata && m.nsensor > 0 && nticks > 0;
#else
    const bool sensors = false;
#endif
    for (int k = 0; k < nticks - (sensors ? 1 : 0); k++) st_pid(c, pidr);
#if RG_SENSORS
    if (sensors) {
      // data.sensordata: the LAST state-less forward of the reference in full — contacts and their forces at the final state
      // (mj_sensorAcc reads efc_force); its actuation stage is the last controller tick
      st_crb(c); st_velocity(c); st_collision(c); st_make_constraint(c);
      if (L.bt.redo && ((s.status & ~status0 & (RG_STATUS_CON_FULL | RG_STATUS_CAND_FULL)) || ((RG_L(c).flags & 512) && s.ncon > 5))) {   // (flags bit 9: test hook, "more than 5 contacts do not fit")
        if (L.bt.pairlb) { float* lb = L.bt.pairlb + (size_t)e * m.npair; PFOR(i, m.npair) lb[i] = 0.f; }
        if (LANE == 0) L.bt.redo[e] = 1;
        return;
      }
      st_touch_geom(c);
      st_pid(c, pidr); st_smooth(c);
      st_factor_smooth(c);
      st_solve_sensors(c, warmr);
      st_touch_write(c);
    }
#endif
    if (L.bt.xdata) PFOR(u, m.nu) L.bt.xdata[(size_t)e * RG_XDATA + RG_XD_ACT_FORCE + u] = s.actfrc[u];
  }
  // ---- write back
  PFOR(i, m.nq) L.bt.qpos[(size_t)e * m.nq + i] = s.qpos[i];
  if (LANE < m.nv) { L.bt.qvel[(size_t)e * m.nv + LANE] = s.qvel[LANE]; L.bt.qacc_warmstart[(size_t)e * m.nv + LANE] = warmr; }
  if (LANE < m.nu) for (int k = 0; k < 3; k++) L.bt.pid[(size_t)e * 3 * m.nu + 3 * LANE + k] = pidr[k];
  PFOR(u, m.nu) L.bt.ctrl[(size_t)e * m.nu + u] = s.ctrl[u];
  if (LANE == 0) {
    if (L.bt.preticks) L.bt.preticks[e] = 0;
    L.bt.status[e] = s.status; L.bt.time[e] += nsubsteps * P[RG_PRM_TIMESTEP];
#if defined(RG_CLOCK_REALTIME) && RG_CLOCK_REALTIME == 2
    L.bt.time[e] = (float)(tk0 & 0xFFFFFF);   // analysis build only: when the env's wave started (low 24 bits of the 100 MHz counter) instead of the simulation time
#endif
    if (L.bt.cost) L.bt.cost[e] = (float)(rg_clock() - tk0);
    if (L.bt.stats) { float* st = L.bt.stats + 4 * (size_t)e; st[0] += st_ncon; st[1] += st_nefc; st[2] += st_iter; st[3] += nsubsteps - sub0; }
  }
  if ((flags & 2) && L.bt.dbg && LANE < RG_NPROF) L.bt.dbg[(size_t)e * RG_DBG_SIZE + RG_DBG_CON + LANE] = s.prof[LANE];  // stage cycle counters (overlays the contact dump)
  // ---- observation row (robot_env.py:714-743; keys/order: DESIGN.md "observation layout")
  if (L.bt.obs) {
    int od = 3 + 4 + m.nq + m.nv + L.env.n_hand_jnt + 15;
    float* o = L.bt.obs + (size_t)e * od;
    // cube_pos = the three slide-joint coordinates, cube_quat sign-normalised to w >= 0
    // (envs/dactyl/observation/cube.py:8-29)
    PFOR(i, 3) o[i] = s.qpos[L.env.cube_pos_qposadr + i];
    { float sg = s.qpos[L.env.cube_quat_qposadr] < 0 ? -1.f : 1.f; PFOR(i, 4) o[3 + i] = sg * s.qpos[L.env.cube_quat_qposadr + i]; }
    PFOR(i, m.nq) o[7 + i] = (i >= L.env.target_qposadr && i < L.env.target_qposadr + L.env.target_nq) ? 0.f : s.qpos[i];
    PFOR(i, m.nv) o[7 + m.nq + i] = (i >= L.env.target_dofadr && i < L.env.target_dofadr + L.env.target_nv) ? 0.f : s.qvel[i];
    PFOR(i, L.env.n_hand_jnt) o[7 + m.nq + m.nv + i] = s.qpos[L.env.hand_qposadr + i];
    // fingertips relative to the three reference sites (hand_forward_kinematics.py:39-50)
    PFOR(i, 5) {
      v3 r0 = ld3(s.spos + 3 * L.env.ref_site[0]), r1 = ld3(s.spos + 3 * L.env.ref_site[1]), r2 = ld3(s.spos + 3 * L.env.ref_site[2]);
      v3 a = normalized(r0 - r1), c = normalized(r2 - r1), b = cross(a, c);
      v3 t = ld3(s.spos + 3 * L.env.tip_site[i]) - r1;
      float* ot = o + 7 + m.nq + m.nv + L.env.n_hand_jnt + 3 * i;
      ot[0] = dot(t, a); ot[1] = dot(t, b); ot[2] = dot(t, c);
    }
    // goal distance: 2 acos(|w|) of q_goal * conj(q_cube)  (locked_parallel.py:54-76, rotation.py:271-286)
    if (L.bt.goal_quat && L.bt.goal_dist && LANE == 0) {
      q4 g = ldq(L.bt.goal_quat + 4 * (size_t)e), c = ldq(s.qpos + L.env.cube_quat_qposadr);
      c.x = -c.x; c.y = -c.y; c.z = -c.z;
      q4 dq = qmul(g, c);
      L.bt.goal_dist[e] = 2.0f * acosf(clampf(fabsf(dq.w), -1.f, 1.f));
    }
  }
}

#endif   // !RG_ITEMS

#if RG_ITEMS
// The env.step of the rollout configuration as B x nsubsteps WORK ITEMS instead of B jobs (VERDICT r02 item 6 i).
// With one workgroup per env.step a launch of 8192 envs on 3072 wave slots is 2.67 jobs per slot whose lengths vary
// 3.5 x: the last third of the launch is ramp-down (slots 81 % utilised, profiles/r02_residency.txt).  Here a launch is a
// grid of PERSISTENT workgroups (one per wave slot) that draw (env, substep) items from a queue in substep-major order --
// substep 0 of every env (longest-expected env first), then substep 1 of every env, ... -- so all envs advance together
// and the slots stay full until the last row: the tail is one mj_step long instead of one env.step.
//   * item (e, s) needs item (e, s - 1): it waits on the env's progress word.  Its predecessor was drawn earlier from the
//     same queue by a workgroup that is therefore running (or done): no order of dispatch can deadlock, and with ~1000 other
//     items between the two the wait is almost never entered.
//   * the env's state travels through its HBM rows (qpos, qvel, ctrl, PID state, warm start: 760 B) and its collision caches;
//     nothing else crosses a substep boundary in the one-workgroup kernel either (registers: PID state, warm start).
//   * one queue per XCD (the L2s of different XCDs are not coherent): queue x holds the envs at positions k = x (mod nqueues)
//     of the dispatch order and is served only by workgroups whose XCC id is x, so an env never leaves its XCD during a
//     launch, that XCD's L2 is the point of coherence, and the hand-off needs no cache-wide fence: plain stores, a vmcnt
//     drain, then the progress word; the consumer polls it and reads the rows with L1-bypassing loads (RG_ROW_LD).
//   * an env that exceeds the rollout capacities in substep s raises redo[e] = s + 1: its later items skip, and the large
//     configuration resumes it at substep s (rg_step_kernel, flags bit 8).
// Results are bit-identical to rg_step_kernel (same stages on the same bytes; tested).
__device__ __forceinline__ void rg_item_publish(int* prog, int e, int value) {
  rg_drain_stores();   // this wave's row stores have reached L2 (gfx9: stores count on vmcnt; the vector L1 is write-through)
  SYNC();
  if (LANE == 0) rg_st_sc1(prog + e, value);
}
__global__ void __launch_bounds__(RG_WAVE, RG_WAVES_PER_SIMD) rg_step_items_kernel(const RgModelDev* mp, RgLaunch launch) {
  RG_MAKE_CTX();
  RgM m = RG_M(c);
  RgLRef L = RG_L(c);
  RgLds& s = RG_S();
  const int nsub = L.nsubsteps, B = L.bt.B, nforward_ticks = L.nforward_ticks, nq_ = L.nqueues;
  int* sched = L.bt.sched;
  int* prog = sched + RG_SCHED_PROG;
  const int xq = rg_xcc_id() % nq_;
  const int nx = (B - xq + nq_ - 1) / nq_;   // envs at positions k = nq_ * j + xq < B
  const int nitems = nx * nsub;
  st_build_row_desc(c);   // (static index tables in the persistent part of the LDS image: once per workgroup, not per item)
  int xfrc_default = -1;  // batches without per-env parameter rows: "any external wrench?" is a property of the model's default row
  if (!L.bt.envprm) { const float* P0 = m.prm_default; float nz = 0; PFOR(i, 6 * m.nbody) nz += P0[RG_PRM_XFRC + i] != 0.f ? 1.f : 0.f; xfrc_default = wave_sum(nz) > 0 ? 1 : 0; }
  const bool desert = (L.flags & 1024) && xq == 0;   // TEST HOOK (flags bit 10): queue 0 is left unserved, as if its XCD had received no workgroup
  for (;;) {
    SYNC();   // (the previous item's LDS image is dead only when every lane is here)
    if (desert) break;
    int t = 0;
    if (LANE == 0) t = rg_ticket(sched + 16 * xq);
    t = rg_first(t);
    if (t >= nitems) break;
    const int sub = t / nx, k = nq_ * (t - sub * nx) + xq;
    const int e = L.bt.order ? rg_first(L.bt.order[k]) : k;
    if (L.bt.active && !L.bt.active[e]) continue;   // (every item of an inactive env skips: nothing to publish)
    if (LANE == 0) s.cur_env = e;
    int pv = 0;
    if (sub > 0) {
      if (LANE == 0) { while (((pv = rg_ld_sc1(prog + e)) & 0xFFFF) < sub) rg_pause(); }
      pv = rg_first(pv);
    }
    SYNC();
    const bool last = sub == nsub - 1;
    const bool handed_over = sub > 0 && L.bt.redo && rg_first(rg_ld_sc1(L.bt.redo + e)) != 0;
    if (handed_over) { rg_item_publish(prog, e, (sub + 1) | (pv & RG_SCHED_BAD)); continue; }
    const long long tk0 = rg_clock();
    // ---- the env's state row
    PFOR(i, m.nq) s.qpos[i] = RG_ROW_LD(L.bt.qpos + (size_t)e * m.nq + i);
    float warmr = 0.f, pidr[3] = {0.f, 0.f, 0.f};
    if (LANE < m.nv) { s.qvel[LANE] = RG_ROW_LD(L.bt.qvel + (size_t)e * m.nv + LANE); warmr = RG_ROW_LD(L.bt.qacc_warmstart + (size_t)e * m.nv + LANE); }
    if (LANE < m.nu) for (int q = 0; q < 3; q++) pidr[q] = RG_ROW_LD(L.bt.pid + (size_t)e * 3 * m.nu + 3 * LANE + q);
    const float* P = rg_prm(m, L);
    if (xfrc_default >= 0) { if (LANE == 0) s.has_xfrc = xfrc_default; }
    else { float nz = 0; PFOR(i, 6 * m.nbody) nz += P[RG_PRM_XFRC + i] != 0.f ? 1.f : 0.f; nz = wave_sum(nz); if (LANE == 0) s.has_xfrc = nz > 0; }
    const unsigned status0 = RG_ROW_LD(L.bt.status + e);
    if (LANE == 0) s.status = status0;
    bool cleared_preticks = false;
    if (sub == 0) {
      if (L.bt.preticks) {   // reset_goal's two state-less forwards, owed from the previous step (see rg_step_kernel)
        const int pre = L.bt.preticks[e];
        if (pre > 0) {
          PFOR(u, m.nu) s.ctrl[u] = L.bt.ctrl[(size_t)e * m.nu + u];
          SYNC();
          st_kinematics(c); st_com_pos(c); st_tendon(c);
          for (int q = 0; q < pre; q++) st_pid(c, pidr);
        }
        cleared_preticks = true;
      }
      bool use_action = L.bt.action && !(L.bt.hold && L.bt.hold[e]);
      if (use_action) {
        float nf = 0; PFOR(u, m.nu) nf += (fabsf(L.bt.action[(size_t)e * m.nu + u]) <= 3.0e38f) ? 0.f : 1.f;
        if (wave_sum(nf) > 0) { use_action = false; if (LANE == 0) s.status |= RG_STATUS_BAD_ACTION; }
      }
      if (use_action) {
        PFOR(u, m.nu) {
          float lo = P[RG_PRM_ACT_CTRLRANGE + 2 * u], hi = P[RG_PRM_ACT_CTRLRANGE + 2 * u + 1], centre;
          if (L.env.relative_action) { centre = 0; for (int j = 0; j < L.env.n_hand_jnt; j++) centre += L.env.pos_to_ctrl[u * L.env.n_hand_jnt + j] * s.qpos[L.env.hand_qposadr + j]; }
          else centre = 0.5f * (hi + lo);
          float a = clampf(L.bt.action[(size_t)e * m.nu + u], -1.f, 1.f);
          s.ctrl[u] = clampf(centre + a * 0.5f * (hi - lo), lo, hi);
        }
      } else { PFOR(u, m.nu) s.ctrl[u] = L.bt.ctrl[(size_t)e * m.nu + u]; }
    } else { PFOR(u, m.nu) s.ctrl[u] = RG_ROW_LD(L.bt.ctrl + (size_t)e * m.nu + u); }
    SYNC();
    // ---- one mj_step (skipped once a substep of this env.step has failed its checks: rg_step_kernel breaks out of its loop)
    bool bad = (pv & RG_SCHED_BAD) != 0, newly_bad = false;
    float st_ncon = 0, st_nefc = 0, st_iter = 0;
    if (!bad) {
      float bd = 0; PFOR(i, m.nq) bd += (fabsf(s.qpos[i]) < 1e10f) ? 0.f : 1.f; PFOR(i, m.nv) bd += (fabsf(s.qvel[i]) < 1e10f) ? 0.f : 1.f;
      if (wave_sum(bd) > 0) bad = newly_bad = true;
    }
    if (!bad) {
      st_kinematics(c); st_com_pos(c); st_tendon(c);
      st_crb(c); st_velocity(c);
      st_collision(c);
      st_make_constraint(c);
      if (L.bt.xdata && last) {   // data.ncon / data.contact[i].{geom1, geom2, dist} of the last mj_step
        float* xd = L.bt.xdata + (size_t)e * RG_XDATA;
        const int nc = s.ncon < RG_DBG_MAXCON ? s.ncon : RG_DBG_MAXCON;
        if (LANE == 0) xd[RG_XD_NCON] = (float)s.ncon;
        PFOR(ci, nc) { int gg = m.pair_gg[s.c_pair[ci]]; float* o = xd + RG_XD_CONTACT + 3 * ci; o[0] = (float)(gg & 255); o[1] = (float)(gg >> 8); o[2] = s.c_dist[ci]; }
      }
      if (L.bt.redo && ((s.status & ~status0 & (RG_STATUS_CON_FULL | RG_STATUS_CAND_FULL)) || ((RG_L(c).flags & 512) && s.ncon > 5))) {   // (flags bit 9: test hook, "more than 5 contacts do not fit")
        // more contacts / candidates than this configuration holds: the env's rows stay as this item found them (in substep 0:
        // as the launch found them, owed ticks and action included) and the large configuration takes over at this substep
        if (L.bt.pairlb) { float* lb = L.bt.pairlb + (size_t)e * m.npair; PFOR(i, m.npair) lb[i] = 0.f; }
        if (LANE == 0) L.bt.redo[e] = sub + 1;
        rg_item_publish(prog, e, sub + 1);
        continue;
      }
      st_pid(c, pidr);
      st_smooth(c);
      st_factor_smooth(c);
      const int packed = st_solve(c, warmr), iters = packed & 255, nefc = packed >> 8;
      st_ncon = (float)s.ncon; st_nefc = (float)nefc; st_iter = (float)iters;
      float bd = 0; PFOR(i, m.nv) bd += (fabsf(s.qacc[i]) < 1e10f) ? 0.f : 1.f;
      if (wave_sum(bd) > 0) bad = newly_bad = true;
      else warmr = st_euler(c, warmr);
    }
    if (newly_bad && LANE == 0) s.status |= RG_STATUS_BAD_STATE;
    // ---- last item: the state-less forward() calls of the reference and the readout (see rg_step_kernel)
    if (last) {
      const int nticks = L.bt.nticks ? L.bt.nticks[e] : nforward_ticks;
      if (nticks > 0 || L.bt.obs || L.bt.xdata) {
        st_kinematics(c);
        if (L.bt.xdata) {
          float* xd = L.bt.xdata + (size_t)e * RG_XDATA;
          PFOR(i, 3 * m.nbody) xd[RG_XD_XPOS + i] = s.xpos[i];
          PFOR(i, 4 * m.nbody) xd[RG_XD_XQUAT + i] = s.xquat[i];
          PFOR(i, 3 * m.nsite) xd[RG_XD_SITE_XPOS + i] = s.spos[i];
        }
        st_com_pos(c); st_tendon(c);
        for (int q = 0; q < nticks; q++) st_pid(c, pidr);
        if (L.bt.xdata) PFOR(u, m.nu) L.bt.xdata[(size_t)e * RG_XDATA + RG_XD_ACT_FORCE + u] = s.actfrc[u];
      }
    }
    // ---- write back
    PFOR(i, m.nq) RG_ROW_ST(L.bt.qpos + (size_t)e * m.nq + i, s.qpos[i]);
    if (LANE < m.nv) { RG_ROW_ST(L.bt.qvel + (size_t)e * m.nv + LANE, s.qvel[LANE]); RG_ROW_ST(L.bt.qacc_warmstart + (size_t)e * m.nv + LANE, warmr); }
    if (LANE < m.nu) for (int q = 0; q < 3; q++) RG_ROW_ST(L.bt.pid + (size_t)e * 3 * m.nu + 3 * LANE + q, pidr[q]);
    if (sub == 0) PFOR(u, m.nu) RG_ROW_ST(L.bt.ctrl + (size_t)e * m.nu + u, s.ctrl[u]);
    if (LANE == 0) {
      if (cleared_preticks) L.bt.preticks[e] = 0;
      if (s.status != status0) RG_ROW_ST(L.bt.status + e, s.status);
      if (last) L.bt.time[e] += nsub * P[RG_PRM_TIMESTEP];
      if (L.bt.cost) { const float cy = (float)(rg_clock() - tk0); RG_ROW_ST(L.bt.cost + e, sub == 0 ? cy : RG_ROW_LD(L.bt.cost + e) + cy); }
      if (L.bt.stats) { float* st = L.bt.stats + 4 * (size_t)e; RG_ROW_ST(st, RG_ROW_LD(st) + st_ncon); RG_ROW_ST(st + 1, RG_ROW_LD(st + 1) + st_nefc); RG_ROW_ST(st + 2, RG_ROW_LD(st + 2) + st_iter); RG_ROW_ST(st + 3, RG_ROW_LD(st + 3) + 1.f); }
    }
    // ---- observation row of the final state (robot_env.py:714-743; see rg_step_kernel)
    if (last && L.bt.obs) {
      int od = 3 + 4 + m.nq + m.nv + L.env.n_hand_jnt + 15;
      float* o = L.bt.obs + (size_t)e * od;
      PFOR(i, 3) o[i] = s.qpos[L.env.cube_pos_qposadr + i];
      { float sg = s.qpos[L.env.cube_quat_qposadr] < 0 ? -1.f : 1.f; PFOR(i, 4) o[3 + i] = sg * s.qpos[L.env.cube_quat_qposadr + i]; }
      PFOR(i, m.nq) o[7 + i] = (i >= L.env.target_qposadr && i < L.env.target_qposadr + L.env.target_nq) ? 0.f : s.qpos[i];
      PFOR(i, m.nv) o[7 + m.nq + i] = (i >= L.env.target_dofadr && i < L.env.target_dofadr + L.env.target_nv) ? 0.f : s.qvel[i];
      PFOR(i, L.env.n_hand_jnt) o[7 + m.nq + m.nv + i] = s.qpos[L.env.hand_qposadr + i];
      PFOR(i, 5) {
        v3 r0 = ld3(s.spos + 3 * L.env.ref_site[0]), r1 = ld3(s.spos + 3 * L.env.ref_site[1]), r2 = ld3(s.spos + 3 * L.env.ref_site[2]);
        v3 a = normalized(r0 - r1), cc = normalized(r2 - r1), b = cross(a, cc);
        v3 tt = ld3(s.spos + 3 * L.env.tip_site[i]) - r1;
        float* ot = o + 7 + m.nq + m.nv + L.env.n_hand_jnt + 3 * i;
        ot[0] = dot(tt, a); ot[1] = dot(tt, b); ot[2] = dot(tt, cc);
      }
      if (L.bt.goal_quat && L.bt.goal_dist && LANE == 0) {
        q4 g = ldq(L.bt.goal_quat + 4 * (size_t)e), cq = ldq(s.qpos + L.env.cube_quat_qposadr);
        cq.x = -cq.x; cq.y = -cq.y; cq.z = -cq.z;
        q4 dq = qmul(g, cq);
        L.bt.goal_dist[e] = 2.0f * acosf(clampf(fabsf(dq.w), -1.f, 1.f));
      }
    }
    rg_item_publish(prog, e, (sub + 1) | (bad ? RG_SCHED_BAD : 0));
  }
  // ---- the last workgroup to leave checks that every queue was drained (a queue whose XCD received no workgroup would be left
  //      standing: the host probes the dispatch once before it enables this kernel, this is the run-time net under that)
  // Round 4: the net is a FALLBACK, not a flag on env 0.  Every env whose progress word is short of nsub (its remaining items were never drawn)
  // gets redo[e] = progress + 1, i.e. it is handed to the large-configuration launch that follows every rollout launch and resumes an env.step at
  // the substep named there (flags bit 8) — the mid-step hand-over path — and RG_STATUS_SCHED on ITS OWN status word (informational: the env WAS
  // stepped).  All other workgroups have left, so nobody else touches these rows; the kernel boundary makes them visible to the next launch.
  int g = 0;
  if (LANE == 0) g = rg_ticket(sched + RG_SCHED_FIN);
  g = rg_first(g);
  if (g == (int)gridDim.x - 1) {
    int ok = 1;
    if (LANE == 0) for (int x = 0; x < nq_; x++) { const int n = (B - x + nq_ - 1) / nq_ * nsub; if (rg_ld_sc1(sched + 16 * x) < n) ok = 0; }
    ok = rg_first(ok);
    if (!ok) {
      PFOR(e, B) {
        if (L.bt.active && !L.bt.active[e]) continue;
        const int p = rg_ld_sc1(prog + e) & 0xFFFF;
        if (p < nsub) { if (L.bt.redo) L.bt.redo[e] = p + 1; L.bt.status[e] |= RG_STATUS_SCHED; }
      }
    }
  }
}
#endif   // RG_ITEMS

#if !RG_ITEMS
// Collision unit-test hook: kinematics of each env's stored qpos, then one MPR query between two geoms.
// out[e][8] = hit, depth, dir3, pos3 (world)
__global__ void __launch_bounds__(RG_WAVE) rg_mpr_pair_kernel(const RgModelDev* mp, RgLaunch launch, int g1, int g2, float margin, float* out) {
  RG_MAKE_CTX();
  RgM m = RG_M(c);
  RgLRef L = RG_L(c);
  RgLds& s = RG_S();
  int e = blockIdx.x;
  PFOR(i, m.nq) s.qpos[i] = L.bt.qpos[(size_t)e * m.nq + i];
  SYNC();
  st_kinematics(c);
  MprGeom A, B;
  v3 p1 = ld3(s.gpos + 3 * g1), p2 = ld3(s.gpos + 3 * g2);
  A.type = m.geom_type[g1]; A.quat = s.gquat + 4 * g1; A.size = ld3(m.geom_size + 3 * g1); A.margin = 0.5f * margin; A.pos = mk3(0, 0, 0);
  B.type = m.geom_type[g2]; B.quat = s.gquat + 4 * g2; B.size = ld3(m.geom_size + 3 * g2); B.margin = 0.5f * margin; B.pos = p2 - p1;
  A.mesh = B.mesh = -1; A.vertadr = B.vertadr = 0; A.nvert = B.nvert = 0;
  if (A.type == RG_GEOM_MESH) { A.mesh = m.geom_dataid[g1]; A.vertadr = m.mesh_vertadr[A.mesh]; A.nvert = m.mesh_vertnum[A.mesh]; }
  if (B.type == RG_GEOM_MESH) { B.mesh = m.geom_dataid[g2]; B.vertadr = m.mesh_vertadr[B.mesh]; B.nvert = m.mesh_vertnum[B.mesh]; }
  MprEnv E = rg_mpr_env(m, (float*)0, true);   // the hook exercises the cell-list supports
  E.plane_depth = true;
  float depth = 0; v3 dir = mk3(0, 0, 0), pos = mk3(0, 0, 0);
  v3 sep; bool hit = rg_mpr<64>(E, A, B, m.mpr_iterations, m.mpr_tolerance, depth, dir, pos, sep, true);
  if (LANE == 0) {
    float* o = out + 8 * (size_t)e;
    o[0] = hit ? 1.f : 0.f; o[1] = depth; o[2] = dir.x; o[3] = dir.y; o[4] = dir.z; o[5] = pos.x + p1.x; o[6] = pos.y + p1.y; o[7] = pos.z + p1.z;
  }
}

#if RG_SETCONST
// mj_setConst on the device (reference: mujoco_simulation.set_constants() in every _reset, cube_env.py:346-349,
// simulation_interface.py:199-201): for every env of the mask, recompute the quantities MuJoCo derives from the model at
// qpos0 and the kernel reads through the env's parameter row -- dof_invweight0 (diagonal of inv(M), averaged over the
// components of a ball / free joint), body_invweight0 (mean diagonal of J inv(M) J' for the translational and the rotational
// Jacobian of the body's com) and tendon_invweight0 (J inv(M) J') -- from the row's CURRENT body_mass / body_inertia /
// dof_armature (and site_pos, for the tendon paths).  They scale the regulariser R of every constraint row
// (rg_make_constraint), so a mass / inertia randomisation without this pass steps with the wrong softness.
// One wave per env: the position stages of the step kernel at qpos0, the tree-sparse L'DL of M, then one substitution per
// needed row of inv(M) J' (nv + 6 per moving body + 1 per tendon: ~230 for the hand models, about one mj_step of work).
// Scratch that has to survive the factorisation (which overlays the position-stage arrays) sits behind the RgLds image.
struct RgSetconstX { float cdof[6 * RG_MAXNV], off[3 * RG_MAXBODY], x[RG_MAXNV], y[RG_MAXNV], dg[RG_MAXNV]; };
static inline size_t rg_lds_setconst_bytes() { return sizeof(RgLds) + sizeof(RgSetconstX); }
__global__ void __launch_bounds__(RG_WAVE) rg_setconst_kernel(const RgModelDev* mp, RgLaunch launch, float* envprm) {
  RG_MAKE_CTX();
  RgM m = RG_M(c);
  RgLRef L = RG_L(c);
  RgLds& s = RG_S();
  RgSetconstX& X = *(RgSetconstX*)((char*)&s + sizeof(RgLds));
  const int e = blockIdx.x;
  if (e >= L.bt.B) return;
  if (L.bt.active && !L.bt.active[e]) return;
  PFOR(i, m.nq) s.qpos[i] = m.qpos0[i];
  if (LANE == 0) { s.has_xfrc = 0; s.status = 0; }
  st_build_row_desc(c);
  SYNC();
  st_kinematics(c); st_com_pos(c); st_tendon(c);
  PFOR(i, 6 * m.nv) X.cdof[i] = s.cdof[i];
  for (int b = 1 + LANE; b < m.nbody; b += RG_WAVE) st3(X.off + 3 * b, rg_xipos(m, s, b) - ld3(s.org + 3 * s.b2org[b]));
  SYNC();
  st_crb(c);
  LtdlDesc D;
  rg_ltdl_load(m.ltdl_tri, m.ltdl_pair, m.n_tri_rounds, m.n_pair_rounds, D);
  rg_M_to_blocks(m, s);
  const int d = LANE; const bool on = d < m.nv;
  const int blk = on ? m.dof_blk[d] : 0, akk = (blk & 0xFFFF) + d - ((blk >> 16) & 255);
  rg_ltdl_factor(s, D);
  float* P = envprm + (size_t)e * RG_NPRM;
  // ---- dof_invweight0
  for (int i = 0; i < m.nv; i++) {
    if (on) X.x[d] = d == i ? 1.f : 0.f;
    SYNC();
    rg_ltdl_solve(s, D, X.x, d, on, akk);
    if (LANE == 0) X.dg[i] = X.x[i];
    SYNC();
  }
  PFOR(j, m.njnt) {
    const int da = m.jnt_dofadr[j], t = m.jnt_type[j];
    if (t == RG_JNT_FREE || t == RG_JNT_BALL) {
      for (int h = 0; h < (t == RG_JNT_FREE ? 2 : 1); h++) {
        const float a = (X.dg[da + 3 * h] + X.dg[da + 3 * h + 1] + X.dg[da + 3 * h + 2]) * (1.f / 3.f);
        for (int k = 0; k < 3; k++) P[RG_PRM_DOF_INVWEIGHT0 + da + 3 * h + k] = a;
      }
    } else P[RG_PRM_DOF_INVWEIGHT0 + da] = X.dg[da];
  }
  // ---- body_invweight0
  for (int b = 1; b < m.nbody; b++) {
    if ((m.body_depth[b] & 255) == 0) { if (LANE == 0) { P[RG_PRM_BODY_INVWEIGHT0 + 2 * b] = 0.f; P[RG_PRM_BODY_INVWEIGHT0 + 2 * b + 1] = 0.f; } continue; }
    float tr[2] = {0.f, 0.f};
    const v3 off = ld3(X.off + 3 * b);
    for (int k = 0; k < 6; k++) {
      float v = 0.f;
      if (on && in_chain(m, b, d)) {
        const v3 ang = ld3(X.cdof + 6 * d), jc = k < 3 ? ld3(X.cdof + 6 * d + 3) + cross(ang, off) : ang;
        const int kk = k < 3 ? k : k - 3;
        v = kk == 0 ? jc.x : (kk == 1 ? jc.y : jc.z);
      }
      if (on) { X.x[d] = v; X.y[d] = v; }
      SYNC();
      rg_ltdl_solve(s, D, X.x, d, on, akk);
      tr[k < 3 ? 0 : 1] += wave_sum(on ? X.x[d] * X.y[d] : 0.f);
      SYNC();
    }
    if (LANE == 0) { P[RG_PRM_BODY_INVWEIGHT0 + 2 * b] = fmaxf(1e-15f, tr[0] * (1.f / 3.f)); P[RG_PRM_BODY_INVWEIGHT0 + 2 * b + 1] = fmaxf(1e-15f, tr[1] * (1.f / 3.f)); }
  }
  // ---- tendon_invweight0
  for (int t = 0; t < m.ntendon; t++) {
    float v = 0.f;
    if (on) for (int q = 0; q < 4; q++) if (m.ten_dofs[4 * t + q] == d) v += s.tenJ[4 * t + q];
    if (on) { X.x[d] = v; X.y[d] = v; }
    SYNC();
    rg_ltdl_solve(s, D, X.x, d, on, akk);
    const float w = wave_sum(on ? X.x[d] * X.y[d] : 0.f);
    SYNC();
    if (LANE == 0) P[RG_PRM_TENDON_INVWEIGHT0 + t] = fmaxf(1e-15f, w);
  }
  if (LANE == 0 && s.status) L.bt.status[e] |= s.status;
}
#endif

// masked row copy (rg_batch_copy_rows): one workgroup per env
__global__ void rg_copy_rows_kernel(float* dst, const float* src, const int* mask, int n, int col0, int ncols, float* pairlb, int npair) {
  int e = blockIdx.x;
  if (!mask[e]) return;
  for (int i = LANE; i < ncols; i += RG_WAVE) dst[(size_t)e * n + col0 + i] = src[(size_t)e * ncols + i];
  if (pairlb) for (int i = LANE; i < npair; i += RG_WAVE) pairlb[(size_t)e * npair + i] = 0.f;
}
#endif   // !RG_ITEMS (hooks and helper kernels exist once, in the classic configurations)
}  // namespace RG_NS
#undef RG_MAXPYR
#undef RG_SENSORS
#undef RG_SETCONST
#undef RG_ITEMS
#undef RG_PSLOTS
#undef RG_RSLOTS
#undef RG_MSLOTS
#undef PROF
#undef PROFS
