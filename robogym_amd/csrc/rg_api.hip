// rg_api.hip — host side of librgstep: the C ABI declared in include/rgstep.h.
// Parses the RGMODEL1 blob, keeps fp32/int32 copies of every model table in HBM, owns the
// per-batch state rows and launches the env-step kernel (one 64-lane workgroup per env).
#include "../../include/rgstep.h"
// two kernel configurations of the same source (rg_kernel.h, "Everything below depends on ..."): LDS per env is what
// bounds the envs in flight per CU, so the hot path runs with capacities sized for rollouts and the reset recipe
// (2-3 x more contacts) with larger ones
typedef rg_post_args RgPostArgs;
typedef rb_post_args RbPostArgs;
typedef ra_post_args RaPostArgs;
typedef ra_recipe_args RaRecipeArgs;
#define RG_NS rgs
#define RG_MAXCON 24
#define RG_CPOOL 768
#define RG_MAXCAND 128
#define RG_MAXCAND2 64
#include "rg_kernel.h"
#undef RG_NS
#undef RG_MAXCON
#undef RG_CPOOL
#undef RG_MAXCAND
#undef RG_MAXCAND2
#define RG_NS rgl
#define RG_SETCONST 1
#define RG_MAXCON 64
#define RG_CPOOL 2048
#define RG_MAXCAND 256
#define RG_MAXCAND2 128
#include "rg_kernel.h"
#undef RG_NS
#undef RG_MAXCON
#undef RG_CPOOL
#undef RG_MAXCAND
#undef RG_MAXCAND2
// the large capacities once more, with data.sensordata evaluation compiled in (launches with flags bit 5)
#define RG_NS rgx
#define RG_MAXCON 64
#define RG_CPOOL 2048
#define RG_MAXCAND 256
#define RG_MAXCAND2 128
#define RG_SENSORS 1
#include "rg_kernel.h"
#undef RG_NS
#undef RG_MAXCON
#undef RG_CPOOL
#undef RG_MAXCAND
#undef RG_MAXCAND2

// the rollout capacities once more as the substep-granular configuration (rg_step_items_kernel: persistent workgroups drawing
// (env, substep) work items; rows written by a previous item are read past the L1)
#define RG_NS rgi
#define RG_ITEMS 1
#define RG_MAXCON 24
#define RG_CPOOL 768
#define RG_MAXCAND 128
#define RG_MAXCAND2 64
#include "rg_kernel.h"
#undef RG_NS
#undef RG_MAXCON
#undef RG_CPOOL
#undef RG_MAXCAND
#undef RG_MAXCAND2

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <sys/wait.h>
#include <spawn.h>
#include <signal.h>
#include <time.h>
#include <mutex>
extern char** environ;

#include <algorithm>
#include <string>
#include <vector>

#ifdef RG_EMUL
// CPU emulation harness (tests/emul): "device memory" is host memory
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 1; }
static hipError_t hipFree(void* p) { free(p); return 0; }
enum { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static hipError_t hipSetDevice(int) { return 0; }
static hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static hipError_t hipDeviceSynchronize() { return 0; }
static const char* hipGetErrorString(hipError_t) { return "emul"; }
#endif

#include "rg_env_kernel.h"
#define RB_NS rgb            /* large configuration: 4 waves per env (dactyl/full_perpendicular) */
#define RB_T RB_T_LARGE
#define RB_MAXGROUP RB_MAXGROUP_LARGE
#define RB_MAXNV RB_MAXNV_LARGE
#define RB_MAXNQ RB_MAXNQ_LARGE
#define RB_WG_PER_CU 4
#include "rb_kernel.h"
#undef RB_NS
#undef RB_T
#undef RB_MAXGROUP
#undef RB_MAXNV
#undef RB_MAXNQ
#undef RB_WG_PER_CU
// The one-wave configurations hold a contact's Jacobian rows at RB_CONW_ONEWAVE dofs (the rearrange worlds need 13: a free object against the arm + gripper chain),
// the large one at 24 (finger chain + wrist against the cube's chain): 384 instead of 576 bytes of Jacobian per contact, 4 instead of 6 16-byte loads per row.
#ifndef RB_CONW_ONEWAVE
#define RB_CONW_ONEWAVE 16
#endif
#define RB_CONW_LARGE RB_CONW
#undef RB_CONW
#define RB_CONW RB_CONW_ONEWAVE
#define RB_MAXNU_ONEWAVE 8   /* the rearrange worlds have 7 and 1 actuators; a model with more runs on the large configuration */
#undef RB_MAXNU
#define RB_MAXNU RB_MAXNU_ONEWAVE
#define RB_NS rgbs           /* small configuration: one wave per env, 16 envs per CU (the rearrange worlds) */
#define RB_T RB_T_SMALL
#define RB_MAXGROUP RB_MAXGROUP_SMALL
#define RB_MAXNV RB_MAXNV_SMALL
#define RB_MAXNQ RB_MAXNQ_SMALL
#ifndef RB_SMALL_WAVES
#define RB_SMALL_WAVES 4     /* waves per SIMD the small configuration's registers are budgeted for (4: 128 VGPRs, 16 envs per CU) */
#endif
#define RB_WG_PER_CU RB_SMALL_WAVES
#include "rb_kernel.h"
#undef RB_NS
#undef RB_T
#undef RB_MAXGROUP
#undef RB_MAXNV
#undef RB_MAXNQ
#undef RB_WG_PER_CU
#define RB_NS rgbm           /* medium configuration: one wave per env, 56 dofs (rearrange/ycb with 8 objects), 10 envs per CU */
#define RB_T RB_T_MEDIUM
#define RB_MAXGROUP RB_MAXGROUP_MEDIUM
#define RB_MAXNV RB_MAXNV_MEDIUM
#define RB_MAXNQ RB_MAXNQ_MEDIUM
#define RB_WG_PER_CU 3
#include "rb_kernel.h"
#undef RB_NS
#undef RB_T
#undef RB_MAXGROUP
#undef RB_MAXNV
#undef RB_MAXNQ
#undef RB_WG_PER_CU
#undef RB_MAXNU
#undef RB_CONW
#define RB_CONW 24           /* (= rb_types.h; the env kernels and the host code below see the large configuration's width unless they ask per model) */
#include "rb_env_kernel.h"
#include "ra_env_kernel.h"
#define RG_WAVES_PER_SIMD_HOST 3   /* = RG_WAVES_PER_SIMD of rg_kernel.h (its default) */
struct rg_batch;
extern "C" { static void rg_items_probe(rg_batch* b); }

static thread_local std::string g_err;
#ifdef RG_EMUL
struct DeviceGuard { explicit DeviceGuard(int) {} };
#else
// every entry point runs on the batch's device and leaves the caller's current device as it found it
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#endif
static int fail(const std::string& msg) { g_err = msg; return -1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct blob_entry { char name[40]; uint32_t dtype, count; uint64_t offset; };

struct rg_model {
  int device = 0;
  RgModelDev dev;
  const RgModelDev* dev_copy = nullptr;   // the same descriptor in device memory (kernels read it through the constant address space)
  RgAux aux;
  std::vector<void*> allocs;
  std::vector<float> qpos0, prm_default;
  std::vector<std::string> blob_keys;
  int ok = 0;
};
struct rg_batch {
  const rg_model* model;
  RgBatchDev dev;
  RgEnvDev env;
  int device;
  int has_env = 0;
  int items_slots = 0;   // substep-granular dispatch: persistent workgroups per launch (wave slots of the device), 0 = not available
  int items_queues = 1;  // work queues (XCDs)
  std::vector<void*> allocs;
};

namespace {
struct Blob {
  const char* base; size_t nbytes;
  // directory entries are validated against the blob size here, so the readers below cannot run past the buffer
  const blob_entry* find(const char* name) const {
    uint32_t n = *(const uint32_t*)(base + 8);
    if (16 + (uint64_t)n * sizeof(blob_entry) > nbytes) return nullptr;
    const blob_entry* e = (const blob_entry*)(base + 16);
    for (uint32_t i = 0; i < n; i++) if (strncmp(e[i].name, name, 40) == 0) {
      uint64_t esz = e[i].dtype == 0 ? 8 : 4;
      if (e[i].dtype > 2 || e[i].offset > nbytes || (uint64_t)e[i].count * esz > nbytes - e[i].offset) return nullptr;
      return &e[i];
    }
    return nullptr;
  }
};

template <class T> bool upload(rg_model* m, const std::vector<T>& host, const T** dst) {
  void* p = nullptr;
  if (hipMalloc(&p, (host.size() ? host.size() : 1) * sizeof(T)) != hipSuccess) return false;
  m->allocs.push_back(p);   // owned by the model from here on (freed by rg_model_free also on a failed create)
  if (host.size() && hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
  *dst = (const T*)p;
  return true;
}
// the arrays a model create asked the blob for, in order (rg_model_blob_keys / rb_model_blob_keys: the schema of that model kind, stated by the reader itself)
thread_local std::vector<std::string>* g_key_log = nullptr;
void log_key(const char* name) { if (g_key_log && std::find(g_key_log->begin(), g_key_log->end(), name) == g_key_log->end()) g_key_log->push_back(name); }
bool get_f(const Blob& b, const char* name, std::vector<float>& out, std::string& err) {
  log_key(name);
  const blob_entry* e = b.find(name);
  if (!e) { err = std::string("model blob lacks '") + name + "' (or its directory entry is out of bounds)"; return false; }
  out.resize(e->count);
  if (e->dtype == 0) { const double* p = (const double*)(b.base + e->offset); for (uint32_t i = 0; i < e->count; i++) out[i] = (float)p[i]; }
  else if (e->dtype == 2) { const float* p = (const float*)(b.base + e->offset); for (uint32_t i = 0; i < e->count; i++) out[i] = p[i]; }
  else { const int* p = (const int*)(b.base + e->offset); for (uint32_t i = 0; i < e->count; i++) out[i] = (float)p[i]; }
  return true;
}
bool get_i(const Blob& b, const char* name, std::vector<int>& out, std::string& err) {
  log_key(name);
  const blob_entry* e = b.find(name);
  if (!e) { err = std::string("model blob lacks '") + name + "' (or its directory entry is out of bounds)"; return false; }
  if (e->dtype != 1) { err = std::string("'") + name + "' is not int32"; return false; }
  out.assign((const int*)(b.base + e->offset), (const int*)(b.base + e->offset) + e->count);
  return true;
}
}  // namespace

extern "C" {

const char* rg_last_error(void) { return g_err.c_str(); }
int rg_debug_size(void) { return RG_DBG_SIZE; }
int rg_lds_bytes(void) { return (int)rgs::rg_lds_launch_bytes(false); }
int rg_lds_bytes_cfg(int config) { return (int)(config == RG_CFG_LARGE ? rgl::rg_lds_launch_bytes(false) : rgs::rg_lds_launch_bytes(false)); }

rg_model* rg_model_create(const void* blob, size_t nbytes, char* err, int errlen) {
  auto bail = [&](const std::string& msg, rg_model* m) -> rg_model* {
    g_key_log = nullptr;
    g_err = msg;
    if (err && errlen > 0) { strncpy(err, msg.c_str(), errlen - 1); err[errlen - 1] = 0; }
    if (m) rg_model_free(m);
    return nullptr;
  };
  if (!blob || nbytes < 16 || memcmp(blob, "RGMODEL1", 8) != 0) return bail("not an RGMODEL1 blob", nullptr);
  Blob B{(const char*)blob, nbytes};
  rg_model* m = new rg_model();
  g_key_log = &m->blob_keys;
#ifndef RG_EMUL
  if (hipGetDevice(&m->device) != hipSuccess) return bail("hipGetDevice failed", m);
#endif
  RgModelDev& d = m->dev;
  memset(&d, 0, sizeof d);
  std::string e;
  std::vector<int> iv; std::vector<float> fv;
#define GI(name) if (!get_i(B, name, iv, e)) return bail(e, m)
#define GF(name) if (!get_f(B, name, fv, e)) return bail(e, m)
#define UPI(field, name) do { GI(name); if (!upload<int>(m, iv, &d.field)) return bail("hipMalloc failed", m); } while (0)
#define UPF(field, name) do { GF(name); if (!upload<float>(m, fv, &d.field)) return bail("hipMalloc failed", m); } while (0)
  GI("dims");
  d.nq = iv[0]; d.nv = iv[1]; d.nu = iv[2]; d.nbody = iv[3]; d.njnt = iv[4]; d.ngeom = iv[5]; d.nsite = iv[6]; d.ntendon = iv[7]; d.nwrap = iv[8]; d.nmesh = iv[9];
  GI("k_dims");
  d.nlevel = iv[0]; d.ndoflevel = iv[1]; d.nM = iv[2]; d.npair = iv[3]; d.nstatic = iv[4];
  if (iv[5] > RG_W) return bail("constraint row wider than RG_W", m);
  if (d.nq > RG_MAXNQ || d.nv > RG_MAXNV || d.nbody > RG_MAXBODY || d.njnt > RG_MAXJNT || d.ngeom > RG_MAXGEOM || d.nsite > RG_MAXSITE ||
      d.ntendon > RG_MAXTEN || d.nu > RG_MAXU)
    return bail("model exceeds the compiled kernel capacities (rg_types.h)", m);
  GI("k_blk_dims");
  d.nvc = iv[0]; d.hs = iv[1]; d.blkwords = iv[2]; d.ntree = iv[3]; d.maxtree = iv[4];
  if (d.nvc > RG_MAXNVC || d.nvc * d.hs > RG_HWORDS || d.blkwords > RG_HWORDS || d.npair >= 32768)
    return bail("model exceeds the compiled solver capacities (rg_types.h)", m);
  UPI(dof_blk, "k_dof_blk"); UPI(dof_blk2, "k_dof_blk2"); UPI(c_blk, "k_c_blk"); UPI(d2c, "k_d2c"); UPI(c2d, "k_c2d");
  GI("opt_int"); d.iterations = iv[0]; d.cone = iv[1]; d.mpr_iterations = iv[3];
  if (d.cone != 0) return bail("elliptic friction cones are not implemented in this kernel configuration", m);
  GF("opt_timestep"); d.timestep = fv[0];
  GF("opt_gravity"); d.gravity[0] = fv[0]; d.gravity[1] = fv[1]; d.gravity[2] = fv[2];
  GF("opt_tolerance"); d.tolerance = fv[0];
  GF("opt_impratio"); d.impratio = fv[0];
  GF("opt_mpr_tolerance"); d.mpr_tolerance = fv[0];
  GF("stat_meaninertia"); d.meaninertia = fv[0];
  UPI(body_parentid, "body_parentid"); UPI(body_rootid, "body_rootid"); UPI(body_jntadr, "body_jntadr"); UPI(body_jntnum, "body_jntnum");
  UPI(body_dofadr, "body_dofadr"); UPI(body_dofnum, "body_dofnum"); UPI(body_lastdof, "k_body_lastdof");
  UPF(body_pos, "body_pos"); UPF(body_quat, "body_quat"); UPF(body_ipos, "body_ipos"); UPF(body_iquat, "body_iquat");
  UPF(body_mass, "body_mass"); UPF(body_inertia, "body_inertia"); UPF(body_invweight0, "body_invweight0");
  UPI(subtree_mask, "k_subtree_mask");
  GI("k_ltdl_tri"); d.n_tri_rounds = (int)(iv.size() / 128); if (!upload<int>(m, iv, &d.ltdl_tri)) return bail("hipMalloc failed", m);
  GI("k_ltdl_pair"); d.n_pair_rounds = (int)(iv.size() / 64); if (!upload<int>(m, iv, &d.ltdl_pair)) return bail("hipMalloc failed", m);
  GI("k_ltdl_tri_c"); d.n_tri_rounds_c = (int)(iv.size() / 128); if (!upload<int>(m, iv, &d.ltdl_tri_c)) return bail("hipMalloc failed", m);
  GI("k_ltdl_pair_c"); d.n_pair_rounds_c = (int)(iv.size() / 64); if (!upload<int>(m, iv, &d.ltdl_pair_c)) return bail("hipMalloc failed", m);
  GI("k_tree_newton_ok"); d.tree_newton_ok = iv.empty() ? 0 : iv[0];
  if (d.n_tri_rounds_c > RG_LTDL_TRI_ROUNDS || d.n_pair_rounds_c > RG_LTDL_PAIR_ROUNDS) d.tree_newton_ok = 0;
  if (d.n_tri_rounds > RG_LTDL_TRI_ROUNDS || d.n_pair_rounds > RG_LTDL_PAIR_ROUNDS) return bail("dof tree too large for the L'DL descriptor caps", m);
  UPF(static_xpos, "k_static_xpos"); UPF(static_xquat, "k_static_xquat");
  UPI(root_origin_body, "k_root_origin_body"); UPI(body_orgslot, "k_body_orgslot"); UPF(root_origin_const, "k_root_origin_const");
  { GI("k_body_dofmask"); std::vector<uint32_t> u(iv.begin(), iv.end()); for (size_t i = 0; i < iv.size(); i++) u[i] = (uint32_t)iv[i]; if (!upload<uint32_t>(m, u, &d.body_dofmask)) return bail("hipMalloc failed", m); }
  UPI(jnt_type, "jnt_type"); UPI(jnt_qposadr, "jnt_qposadr"); UPI(jnt_dofadr, "jnt_dofadr"); UPI(jnt_bodyid, "jnt_bodyid");
  UPF(jnt_pos, "jnt_pos"); UPF(jnt_axis, "jnt_axis"); UPF(jnt_stiffness, "jnt_stiffness"); UPF(jnt_range, "jnt_range");
  UPF(jnt_margin, "jnt_margin"); UPF(jnt_solref, "jnt_solref"); UPF(jnt_solimp, "jnt_solimp");
  UPI(dof_bodyid, "dof_bodyid"); UPI(dof_jntid, "dof_jntid"); UPI(dof_parentid, "dof_parentid");
  UPF(dof_armature, "dof_armature"); UPF(dof_damping, "dof_damping"); UPF(dof_frictionloss, "dof_frictionloss");
  UPF(dof_solref, "dof_solref"); UPF(dof_solimp, "dof_solimp"); UPF(dof_invweight0, "dof_invweight0");
  UPF(qpos0, "qpos0"); m->qpos0 = fv; UPF(qpos_spring, "qpos_spring");
  {  // kinematics records (host side: one per level slot, copied per body below): w0 body | parent<<8 | njnt<<16 | type(first joint)<<20, w1 first joint | its qposadr<<16,
     // w2-4 body_pos, w5-8 body_quat, w9-11 body_ipos, w12-14 jnt_pos, w15-17 jnt_axis, w18 qpos0 of the joint
    std::vector<int> lb, par, jn, ja, jt, jq; std::vector<float> bp, bq, bi, jp, jx, q0;
    if (!get_i(B, "k_lvl_body", lb, e) || !get_i(B, "body_parentid", par, e) || !get_i(B, "body_jntnum", jn, e) || !get_i(B, "body_jntadr", ja, e) ||
        !get_i(B, "jnt_type", jt, e) || !get_i(B, "jnt_qposadr", jq, e) || !get_f(B, "body_pos", bp, e) || !get_f(B, "body_quat", bq, e) ||
        !get_f(B, "body_ipos", bi, e) || !get_f(B, "jnt_pos", jp, e) || !get_f(B, "jnt_axis", jx, e) || !get_f(B, "qpos0", q0, e)) return bail(e, m);
    std::vector<float> rec((lb.size() ? lb.size() : 1) * RG_KINREC, 0.f);
    for (size_t k = 0; k < lb.size(); k++) {
      int b = lb[k], j = jn[b] > 0 ? ja[b] : 0, t = jn[b] > 0 ? jt[j] : 0, qa = jn[b] > 0 ? jq[j] : 0;
      float* r = rec.data() + k * RG_KINREC;
      int w0 = b | (par[b] << 8) | (jn[b] << 16) | (t << 20), w1 = j | (qa << 16);
      memcpy(r, &w0, 4); memcpy(r + 1, &w1, 4);
      for (int c = 0; c < 3; c++) { r[2 + c] = bp[3 * b + c]; r[9 + c] = bi[3 * b + c]; r[12 + c] = jn[b] > 0 ? jp[3 * j + c] : 0.f; r[15 + c] = jn[b] > 0 ? jx[3 * j + c] : 0.f; }
      for (int c = 0; c < 4; c++) r[5 + c] = bq[4 * b + c];
      r[18] = jn[b] > 0 ? q0[qa] : 0.f;
    }
    // per moving body: its own record once more (indexed by body), and its chain of moving ancestors (root first, the body
    // last; one byte each) — rg_kinematics composes the bodies' relative frames down that chain, one lane per body
    {
      static_assert(RG_MAXDEPTH <= 8 && RG_MAXBODY <= 256, "body chain: 8 bytes per body");
      std::vector<int> slot(d.nbody, -1), depth(d.nbody, 0), chainw((size_t)(d.nbody ? d.nbody : 1) * 2, 0);
      for (size_t k = 0; k < lb.size(); k++) if (lb[k] >= 0 && lb[k] < d.nbody) slot[lb[k]] = (int)k;
      std::vector<float> brec((size_t)(d.nbody ? d.nbody : 1) * RG_KINREC, 0.f);
      for (int b0 = 0; b0 < d.nbody; b0++) {
        if (slot[b0] < 0) continue;
        memcpy(brec.data() + (size_t)b0 * RG_KINREC, rec.data() + (size_t)slot[b0] * RG_KINREC, RG_KINREC * 4);
        int chain[RG_MAXDEPTH + 1], n = 0, a0 = b0;
        for (; a0 > 0 && slot[a0] >= 0; a0 = par[a0]) { if (n >= RG_MAXDEPTH) return bail("kinematic chain deeper than RG_MAXDEPTH", m); chain[n++] = a0; }
        unsigned long long cw = 0; int absmask = 0;
        for (int i = 0; i < n; i++) {
          int a = chain[n - 1 - i];
          cw |= (unsigned long long)(a & 255) << (8 * i);
          for (int jj = 0; jj < jn[a]; jj++) if (jt[ja[a] + jj] == RG_JNT_FREE) absmask |= 1 << i;
        }
        depth[b0] = n | (absmask << 8) | ((a0 & 255) << 16);
        chainw[2 * b0] = (int)(unsigned)(cw & 0xFFFFFFFFull); chainw[2 * b0 + 1] = (int)(unsigned)(cw >> 32);
      }
      if (!upload<float>(m, brec, &d.body_rec) || !upload<int>(m, depth, &d.body_depth) || !upload<int>(m, chainw, &d.body_chain)) return bail("hipMalloc failed", m);
    }
    {  // com-frame origin slots (the kernel's `org`): one record per slot instead of body -> root -> origin body -> ipos
      std::vector<int> os, rid, rob; std::vector<float> roc;
      if (!get_i(B, "k_body_orgslot", os, e) || !get_i(B, "body_rootid", rid, e) || !get_i(B, "k_root_origin_body", rob, e) || !get_f(B, "k_root_origin_const", roc, e)) return bail(e, m);
      std::vector<float> orec(16, 0.f);
      for (int sl = 0; sl < 4; sl++) { int unused = -2; memcpy(&orec[4 * sl], &unused, 4); }
      for (int b0 = 0; b0 < d.nbody; b0++) {
        int sl = os[b0], r = rid[b0], ob = rob[r];
        if (sl < 0 || sl > 3) return bail("com-frame slot out of range", m);
        memcpy(&orec[4 * sl], &ob, 4);
        for (int c = 0; c < 3; c++) orec[4 * sl + 1 + c] = ob >= 0 ? bi[3 * ob + c] : roc[3 * r + c];
      }
      if (!upload<float>(m, orec, &d.org_rec)) return bail("hipMalloc failed", m);
    }
  }
  UPI(lvl_dof, "k_lvl_dof"); UPI(lvl_dof_adr, "k_lvl_dof_adr"); UPI(M_i, "k_M_i"); UPI(M_j, "k_M_j"); { std::vector<int> mi0, mj0, db0; if (!get_i(B, "k_M_i", mi0, e) || !get_i(B, "k_M_j", mj0, e) || !get_i(B, "dof_bodyid", db0, e)) return bail(e, m); std::vector<int> w(mi0.size() ? mi0.size() : 1, 0); for (size_t q = 0; q < mi0.size(); q++) w[q] = mi0[q] | (mj0[q] << 8) | (db0[mi0[q]] << 16); if (!upload<int>(m, w, &d.M_ijb)) return bail("hipMalloc failed", m); }   /* M entry: dof i | dof j << 8 | body of i << 16 */ UPI(M_lvl_adr, "k_M_lvl_adr");
  UPI(desc_adr, "k_desc_adr"); UPI(desc, "k_desc");
  {  // per tree-sparse entry of M: its two words in the per-tree block layout and its place in the compact Newton space
    std::vector<int> mi, mj, blk, d2c;
    if (!get_i(B, "k_M_i", mi, e) || !get_i(B, "k_M_j", mj, e) || !get_i(B, "k_dof_blk", blk, e) || !get_i(B, "k_d2c", d2c, e)) return bail(e, m);
    if ((int)mi.size() > RG_MAXNM || mi.size() != mj.size()) return bail("inertia matrix has more non-zeros than RG_MAXNM", m);
    std::vector<int> ent(2 * (mi.size() ? mi.size() : 1), 0);
    for (size_t k = 0; k < mi.size(); k++) {
      int i = mi[k], j = mj[k], s0 = (blk[i] >> 16) & 255;
      int aij = (blk[i] & 0xFFFF) + (j - s0), aji = (blk[j] & 0xFFFF) + (i - s0);
      if (aij >= RG_HWORDS || aji >= RG_HWORDS || aij < 0 || aji < 0) return bail("inertia block layout exceeds RG_HWORDS", m);
      ent[2 * k] = aij | (aji << 16);
      ent[2 * k + 1] = (d2c[i] >= 0 && d2c[j] >= 0) ? (d2c[i] | (d2c[j] << 8) | (1 << 16)) : 0;
    }
    if (!upload<int>(m, ent, &d.M_ent)) return bail("hipMalloc failed", m);
  }
  UPI(geom_type, "geom_type"); UPI(geom_bodyid, "geom_bodyid"); UPI(geom_dataid, "geom_dataid"); UPI(body_geomadr, "body_geomadr"); UPI(body_geomnum, "body_geomnum");
  UPF(geom_size, "geom_size"); UPF(geom_rbound, "geom_rbound"); UPF(geom_pos, "geom_pos"); UPF(geom_quat, "geom_quat"); UPF(geom_aabb, "k_geom_aabb");
  UPI(site_bodyid, "site_bodyid"); UPF(site_pos, "site_pos");
  UPI(site_type, "site_type"); UPF(site_size, "site_size"); UPF(site_quat, "site_quat");
  { std::vector<int> st, so, sty;
    if (!get_i(B, "sensor_type", st, e) || !get_i(B, "sensor_objid", so, e) || !get_i(B, "site_type", sty, e)) return bail(e, m);
    if (st.size() > RG_MAXSENSOR) return bail("more than RG_MAXSENSOR sensors", m);
    for (size_t k = 0; k < st.size(); k++) {
      if (st[k] != 0) return bail("only touch sensors are implemented", m);
      int t = sty[so[k]];
      if (t != RG_GEOM_SPHERE && t != RG_GEOM_CAPSULE && t != RG_GEOM_ELLIPSOID && t != RG_GEOM_CYLINDER && t != RG_GEOM_BOX) return bail("touch sensor on a site of unsupported shape", m);
    }
    d.nsensor = (int)st.size();
    if (so.empty()) so.push_back(0);
    if (!upload<int>(m, so, &d.sensor_site)) return bail("hipMalloc failed", m); }
  UPI(mesh_vertadr, "mesh_vertadr"); UPI(mesh_vertnum, "mesh_vertnum");
  { GF("mesh_vert"); std::vector<float> v4(fv.size() / 3 * 4, 0.f);  // 16-byte vertex records: one dwordx4 load per vertex
    for (size_t i = 0; i < fv.size() / 3; i++) { v4[4 * i] = fv[3 * i]; v4[4 * i + 1] = fv[3 * i + 1]; v4[4 * i + 2] = fv[3 * i + 2]; }
    if (!upload<float>(m, v4, &d.mesh_vert)) return bail("hipMalloc failed", m);
    // support-cell records: the cell lists' vertices copied out so that one cell is one contiguous run
    std::vector<int> vadr, cadr, vidx;
    if (!get_i(B, "mesh_vertadr", vadr, e) || !get_i(B, "k_mesh_cell_adr", cadr, e) || !get_i(B, "k_mesh_cell_vidx", vidx, e)) return bail(e, m);
    if (cadr.size() != (size_t)d.nmesh * RG_NCELL) return bail("k_mesh_cell_adr has the wrong size (RG_CELLN mismatch?)", m);
    std::vector<float> blk((size_t)d.nmesh * RG_NCELL * 16, 0.f), ovf(4, 0.f);
    for (int mi = 0; mi < d.nmesh; mi++) for (int c = 0; c < RG_NCELL; c++) {
      size_t ce = (size_t)mi * RG_NCELL + c;
      int e = cadr[ce], start = e >> 8, cnt = e & 255;
      if (cnt < 1 || (size_t)(start + cnt) > vidx.size()) return bail("bad support cell", m);
      cadr[ce] = (int)((ovf.size() / 4) << 8) | cnt;
      for (int k = 0; k < (cnt > 4 ? cnt : 4); k++) {
        int vi = vidx[start + (k < cnt ? k : cnt - 1)]; size_t src = (size_t)(vadr[mi] + vi);
        if (vi < 0 || src * 3 + 2 >= fv.size()) return bail("bad support cell vertex", m);
        float o[4] = {fv[3 * src], fv[3 * src + 1], fv[3 * src + 2], 0.f}; memcpy(o + 3, &vi, 4);
        if (k < 4) memcpy(blk.data() + ce * 16 + 4 * k, o, 16); else ovf.insert(ovf.end(), o, o + 4);
      }
    }
    if (!upload<int>(m, cadr, &d.mesh_cell_adr) || !upload<float>(m, blk, &d.mesh_cell_blk) || !upload<float>(m, ovf, &d.mesh_cell_ovf)) return bail("hipMalloc failed", m); }
  UPI(pair_geom, "k_pair_geom"); UPF(pair_prm, "k_pair_prm"); UPF(pair_mix, "k_pair_mix");
  {  // pair records: w0 g1 | g2<<8 | condim<<16 | type1<<20 | type2<<24, w1 margin, w2/w3 mesh ids (-1: none),
     // w4-6 size1, w7 nvert1, w8-10 size2, w11 nvert2, w12/w13 first vertex of the meshes, w14/w15 bounding radii,
     // w16-18 / w20-22 box half extents of the geoms (geom frame), w19 body1 | body2 << 8, w23 / w24 union of the two bodies' dof masks
    std::vector<int> pg, gt, gd, va, vn, gb, bdm; std::vector<float> pp, gs, gr, ga;
    if (!get_i(B, "geom_bodyid", gb, e) || !get_i(B, "k_body_dofmask", bdm, e)) return bail(e, m);
    if (!get_i(B, "k_pair_geom", pg, e) || !get_i(B, "geom_type", gt, e) || !get_i(B, "geom_dataid", gd, e) || !get_i(B, "mesh_vertadr", va, e) || !get_i(B, "mesh_vertnum", vn, e) ||
        !get_f(B, "k_pair_prm", pp, e) || !get_f(B, "geom_size", gs, e) || !get_f(B, "geom_rbound", gr, e) || !get_f(B, "k_geom_aabb", ga, e)) return bail(e, m);
    std::vector<int> gsc;   // optional: geoms whose size follows the env's RG_PRM_GEOM_SCALE (boxes only: the bounds scale exactly)
    if (B.find("k_geom_scaled") && !get_i(B, "k_geom_scaled", gsc, e)) return bail(e, m);
    for (size_t g = 0; g < gsc.size(); g++) if (gsc[g] && (g >= gt.size() || gt[g] != RG_GEOM_BOX)) return bail("k_geom_scaled flags a geom that is not a box", m);
    size_t np = pg.size() / 3;
    std::vector<int> gg(np ? np : 1, 0); std::vector<float> rec((np ? np : 1) * RG_PAIRREC, 0.f);
    for (size_t p = 0; p < np; p++) {
      int g[2] = {pg[3 * p], pg[3 * p + 1]};
      if (g[0] < 0 || g[1] < 0 || g[0] > 255 || g[1] > 255 || g[0] >= d.ngeom || g[1] >= d.ngeom) return bail("pair geom id out of range", m);
      gg[p] = g[0] | (g[1] << 8);
      float* r = rec.data() + p * RG_PAIRREC;
      int hdr = g[0] | (g[1] << 8) | (pg[3 * p + 2] << 16) | (gt[g[0]] << 20) | (gt[g[1]] << 24);
      if ((size_t)g[0] < gsc.size() && gsc[g[0]]) hdr |= RG_PAIR_SCALED1;
      if ((size_t)g[1] < gsc.size() && gsc[g[1]]) hdr |= RG_PAIR_SCALED2;
      memcpy(r, &hdr, 4); r[1] = pp[12 * p];
      for (int k = 0; k < 2; k++) {
        int id = gt[g[k]] == RG_GEOM_MESH ? gd[g[k]] : -1, nvert = id >= 0 ? vn[id] : 0, vadr0 = id >= 0 ? va[id] : 0;
        memcpy(r + 2 + k, &id, 4);
        for (int c = 0; c < 3; c++) { r[4 + 4 * k + c] = gs[3 * g[k] + c]; r[16 + 4 * k + c] = ga[3 * g[k] + c]; }
        memcpy(r + 7 + 4 * k, &nvert, 4); memcpy(r + 12 + k, &vadr0, 4);
        r[14 + k] = gr[g[k]];
      }
      int b1 = gb[g[0]], b2 = gb[g[1]], bb = b1 | (b2 << 8), lo = bdm[2 * b1] | bdm[2 * b2], hi = bdm[2 * b1 + 1] | bdm[2 * b2 + 1];
      if (b1 > 255 || b2 > 255) return bail("pair body id out of range", m);
      memcpy(r + 19, &bb, 4); memcpy(r + 23, &lo, 4); memcpy(r + 24, &hi, 4);
    }
    if (!upload<int>(m, gg, &d.pair_gg) || !upload<float>(m, rec, &d.pair_rec)) return bail("hipMalloc failed", m);
  }
  UPI(tendon_adr, "tendon_adr"); UPI(tendon_num, "tendon_num"); UPI(wrap_type, "wrap_type"); UPI(wrap_objid, "wrap_objid"); UPI(ten_dofs, "k_ten_dofs");
  UPI(ten_path, "k_ten_path"); UPI(ten_path_adr, "k_ten_path_adr");
  UPF(wrap_prm, "wrap_prm"); UPF(tendon_range, "tendon_range"); UPF(tendon_margin, "tendon_margin"); UPF(tendon_stiffness, "tendon_stiffness");
  UPF(tendon_damping, "tendon_damping"); UPF(tendon_frictionloss, "tendon_frictionloss"); UPF(tendon_lengthspring, "tendon_lengthspring");
  UPF(tendon_solref_lim, "tendon_solref_lim"); UPF(tendon_solimp_lim, "tendon_solimp_lim"); UPF(tendon_solref_fri, "tendon_solref_fri");
  UPF(tendon_solimp_fri, "tendon_solimp_fri"); UPF(tendon_invweight0, "tendon_invweight0");
  UPI(dof_ten_adr, "k_dof_ten_adr"); UPI(dof_ten, "k_dof_ten"); UPI(dof_act_adr, "k_dof_act_adr"); UPI(dof_act, "k_dof_act");
  {  // per-dof record for the velocity / actuation stages (one load instead of dof -> joint -> stiffness / qposadr -> spring chains):
     // w0 body | tendon entries begin << 8 | end << 16, w1 actuator entries begin | end << 8, w2 joint stiffness (0 unless hinge / slide), w3 qposadr, w4 qpos_spring, w5-7 unused
    std::vector<int> db, dj, jt, jq, ta, aa; std::vector<float> js, qs;
    if (!get_i(B, "dof_bodyid", db, e) || !get_i(B, "dof_jntid", dj, e) || !get_i(B, "jnt_type", jt, e) || !get_i(B, "jnt_qposadr", jq, e) ||
        !get_i(B, "k_dof_ten_adr", ta, e) || !get_i(B, "k_dof_act_adr", aa, e) || !get_f(B, "jnt_stiffness", js, e) || !get_f(B, "qpos_spring", qs, e)) return bail(e, m);
    std::vector<float> drec((size_t)(d.nv ? d.nv : 1) * 8, 0.f);
    for (int i = 0; i < d.nv; i++) {
      int j = dj[i], t = jt[j], qa = jq[j];
      if (ta[i + 1] > 255 || aa[i + 1] > 255 || db[i] > 255) return bail("dof record: table index out of range", m);
      int w0 = db[i] | (ta[i] << 8) | (ta[i + 1] << 16), w1 = aa[i] | (aa[i + 1] << 8);
      float* r = drec.data() + 8 * (size_t)i;
      memcpy(r, &w0, 4); memcpy(r + 1, &w1, 4);
      r[2] = (t == RG_JNT_HINGE || t == RG_JNT_SLIDE) ? js[j] : 0.f;
      memcpy(r + 3, &qa, 4); r[4] = qs[qa];
    }
    if (!upload<float>(m, drec, &d.dof_rec)) return bail("hipMalloc failed", m);
  }
  UPI(actuator_trntype, "actuator_trntype"); UPI(actuator_trnid, "actuator_trnid"); UPI(actuator_ctrllimited, "actuator_ctrllimited");
  UPI(actuator_forcelimited, "actuator_forcelimited"); UPI(actuator_biastype, "actuator_biastype");
  UPF(actuator_gear, "actuator_gear"); UPF(actuator_ctrlrange, "actuator_ctrlrange"); UPF(actuator_forcerange, "actuator_forcerange");
  UPF(actuator_gainprm, "actuator_gainprm"); UPF(actuator_biasprm, "actuator_biasprm");
  GI("k_fric_dof"); d.nfric_dof = (int)iv.size(); if (!upload<int>(m, iv, &d.fric_dof)) return bail("hipMalloc failed", m);
  GI("k_fric_ten"); d.nfric_ten = (int)iv.size(); if (!upload<int>(m, iv, &d.fric_ten)) return bail("hipMalloc failed", m);
  GI("k_lim_jnt"); d.nlim_jnt = (int)iv.size(); if (!upload<int>(m, iv, &d.lim_jnt)) return bail("hipMalloc failed", m);
  GI("k_lim_ten"); d.nlim_ten = (int)iv.size(); if (!upload<int>(m, iv, &d.lim_ten)) return bail("hipMalloc failed", m);
  if (d.nfric_dof + d.nfric_ten + 2 * d.nlim_jnt + 2 * d.nlim_ten > RG_MAXSROW || d.nfric_dof + d.nfric_ten > RG_MAXFRIC) return bail("too many friction/limit rows for RG_MAXSROW / RG_MAXFRIC", m);
  {  // static constraint rows (friction loss of dofs / tendons, joint / tendon limits), one 16-word record per row in row order:
     // w0 Jacobian descriptor (compact dof | tendon << 6 | sign << 11) | kind << 16, w1 qposadr / tendon whose position the limit reads,
     // w2 parameter-row offset of the row's diagApprox, w3 of its friction loss (-1: none or constant), w4 constant friction loss,
     // w5 parameter-row offset of the limit, w6 margin, w7-8 solref, w9-13 solimp
    std::vector<int> fd, ft, lj, lt, d2c, jq, jd; std::vector<float> tfl, jm, tm, dsr, dsi, tsrf, tsif, jsr, jsi, tsrl, tsil;
    if (!get_i(B, "k_fric_dof", fd, e) || !get_i(B, "k_fric_ten", ft, e) || !get_i(B, "k_lim_jnt", lj, e) || !get_i(B, "k_lim_ten", lt, e) || !get_i(B, "k_d2c", d2c, e) ||
        !get_i(B, "jnt_qposadr", jq, e) || !get_i(B, "jnt_dofadr", jd, e) || !get_f(B, "tendon_frictionloss", tfl, e) || !get_f(B, "jnt_margin", jm, e) || !get_f(B, "tendon_margin", tm, e) ||
        !get_f(B, "dof_solref", dsr, e) || !get_f(B, "dof_solimp", dsi, e) || !get_f(B, "tendon_solref_fri", tsrf, e) || !get_f(B, "tendon_solimp_fri", tsif, e) ||
        !get_f(B, "jnt_solref", jsr, e) || !get_f(B, "jnt_solimp", jsi, e) || !get_f(B, "tendon_solref_lim", tsrl, e) || !get_f(B, "tendon_solimp_lim", tsil, e)) return bail(e, m);
    const int ns = d.nfric_dof + d.nfric_ten + 2 * d.nlim_jnt + 2 * d.nlim_ten;
    std::vector<float> srec((size_t)(ns ? ns : 1) * 16, 0.f);
    for (int r = 0; r < ns; r++) {
      float* o = srec.data() + 16 * (size_t)r;
      int rr = r, kind, dof = 0, ten = 31, neg = 0, src = 0, odiag, ofl = -1, olim = 0, omargin = -1; float cfl = 0.f, margin = 0.f; const float *sr, *si;
      if (rr < d.nfric_dof) { int q = fd[rr]; kind = 0; dof = d2c[q]; ofl = RG_PRM_DOF_FRICTIONLOSS + q; odiag = RG_PRM_DOF_INVWEIGHT0 + q; sr = &dsr[2 * q]; si = &dsi[5 * q]; }
      else if ((rr -= d.nfric_dof) < d.nfric_ten) { int t = ft[rr]; kind = 1; ten = t; cfl = tfl[t]; odiag = RG_PRM_TENDON_INVWEIGHT0 + t; sr = &tsrf[2 * t]; si = &tsif[5 * t]; }
      else if ((rr -= d.nfric_ten) < 2 * d.nlim_jnt) { int j = lj[rr >> 1]; kind = 2; dof = d2c[jd[j]]; neg = rr & 1; src = jq[j]; olim = RG_PRM_JNT_RANGE + 2 * j + (rr & 1); margin = 0.f; omargin = RG_PRM_JNT_MARGIN + j; (void)jm; odiag = RG_PRM_DOF_INVWEIGHT0 + jd[j]; sr = &jsr[2 * j]; si = &jsi[5 * j]; }
      else { rr -= 2 * d.nlim_jnt; int t = lt[rr >> 1]; kind = 3; ten = t; neg = rr & 1; src = t; olim = RG_PRM_TENDON_RANGE + 2 * t + (rr & 1); margin = tm[t]; odiag = RG_PRM_TENDON_INVWEIGHT0 + t; sr = &tsrl[2 * t]; si = &tsil[5 * t]; }
      int w0 = (dof & 63) | (ten << 6) | (neg << 11) | (kind << 16);
      memcpy(o, &w0, 4); memcpy(o + 1, &src, 4); memcpy(o + 2, &odiag, 4); memcpy(o + 3, &ofl, 4); o[4] = cfl; memcpy(o + 5, &olim, 4); o[6] = margin; if (omargin >= 0) memcpy(o + 6, &omargin, 4);   // (w6: joint limits carry the parameter-row offset of their margin)
      o[7] = sr[0]; o[8] = sr[1]; for (int c = 0; c < 5; c++) o[9 + c] = si[c];
    }
    if (!upload<float>(m, srec, &d.srow_rec)) return bail("hipMalloc failed", m);
  }
  { GI("k_pair_geom"); for (size_t i = 2; i < iv.size(); i += 3) if (iv[i] > 4) return bail("condim 6 (rolling friction) contacts are not implemented", m); }
  GI("k_subtree_adr"); if (!upload<int>(m, iv, &m->aux.subtree_adr)) return bail("hipMalloc failed", m);
  GI("k_subtree"); if (!upload<int>(m, iv, &m->aux.subtree)) return bail("hipMalloc failed", m);
  { GI("k_dof_velmask"); std::vector<uint32_t> u(iv.size()); for (size_t i = 0; i < iv.size(); i++) u[i] = (uint32_t)iv[i]; if (!upload<uint32_t>(m, u, &m->aux.dof_velmask)) return bail("hipMalloc failed", m); }
  {  // the model's own values in the per-env parameter layout (RG_PRM_*): the row every env reads unless the batch overrides it
    std::vector<float> prm(RG_NPRM, 0.f), v;
    auto put = [&](const char* name, int off, size_t cap) -> bool {
      if (!get_f(B, name, v, e)) return false;
      if (v.size() > cap) { e = std::string(name) + " exceeds the per-env parameter layout"; return false; }
      for (size_t i = 0; i < v.size(); i++) prm[off + i] = v[i];
      return true;
    };
    if (!put("opt_gravity", RG_PRM_GRAVITY, 3) || !put("opt_timestep", RG_PRM_TIMESTEP, 1) || !put("dof_damping", RG_PRM_DOF_DAMPING, RG_MAXNV) ||
        !put("dof_armature", RG_PRM_DOF_ARMATURE, RG_MAXNV) || !put("dof_frictionloss", RG_PRM_DOF_FRICTIONLOSS, RG_MAXNV) || !put("dof_invweight0", RG_PRM_DOF_INVWEIGHT0, RG_MAXNV) ||
        !put("body_mass", RG_PRM_BODY_MASS, RG_MAXBODY) || !put("body_inertia", RG_PRM_BODY_INERTIA, 3 * RG_MAXBODY) || !put("body_invweight0", RG_PRM_BODY_INVWEIGHT0, 2 * RG_MAXBODY) ||
        !put("jnt_range", RG_PRM_JNT_RANGE, 2 * RG_MAXJNT) || !put("tendon_range", RG_PRM_TENDON_RANGE, 2 * RG_MAXTEN) || !put("tendon_invweight0", RG_PRM_TENDON_INVWEIGHT0, RG_MAXTEN) ||
        !put("actuator_gainprm", RG_PRM_ACT_GAINPRM, 10 * RG_MAXU) || !put("actuator_ctrlrange", RG_PRM_ACT_CTRLRANGE, 2 * RG_MAXU) ||
        !put("actuator_forcerange", RG_PRM_ACT_FORCERANGE, 2 * RG_MAXU) || !put("geom_friction", RG_PRM_GEOM_FRICTION, 3 * RG_MAXGEOM) ||
        !put("site_pos", RG_PRM_SITE_POS, 3 * RG_MAXSITE) || !put("jnt_margin", RG_PRM_JNT_MARGIN, RG_MAXJNT) || !put("geom_solref", RG_PRM_GEOM_SOLREF, 2 * RG_MAXGEOM) ||
        !put("geom_solimp", RG_PRM_GEOM_SOLIMP, 5 * RG_MAXGEOM)) return bail(e, m);
    prm[RG_PRM_GEOM_SCALE] = 1.f;
    m->prm_default = prm;
    if (!upload<float>(m, prm, &d.prm_default)) return bail("hipMalloc failed", m);
  }
  { std::vector<RgModelDev> one(1, d); if (!upload<RgModelDev>(m, one, &m->dev_copy)) return bail("hipMalloc failed", m); }
  m->ok = 1;
  g_key_log = nullptr;
  return m;
}

rg_model* rg_model_create_on(const void* blob, size_t nbytes, int device, char* err, int errlen) {
  DeviceGuard g(device);
  return rg_model_create(blob, nbytes, err, errlen);
}
void rg_model_free(rg_model* m) {
  if (!m) return;
  { DeviceGuard g(m->device); for (void* p : m->allocs) hipFree(p); }
  delete m;
}
int rg_model_npair(const rg_model* m) { return m ? m->dev.npair : -1; }
int rg_model_dims(const rg_model* m, int* out) {
  if (!m) return fail("null model");
  out[0] = m->dev.nq; out[1] = m->dev.nv; out[2] = m->dev.nu; out[3] = m->dev.nbody; out[4] = m->dev.nsite;
  return 0;
}

static void* balloc(rg_batch* b, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr;
  hipMemset(p, 0, n ? n : 1);
  b->allocs.push_back(p);
  return p;
}

rg_batch* rg_batch_create(const rg_model* m, int B, int device) {
  if (!m || !m->ok || B <= 0) { fail("bad arguments to rg_batch_create"); return nullptr; }
  if (device != m->device) { fail("rg_batch_create: the model tables live on another device (create the model with rg_model_create_on)"); return nullptr; }
  DeviceGuard g(device);
  rg_batch* b = new rg_batch();
  b->model = m; b->device = device;
  memset(&b->dev, 0, sizeof b->dev); memset(&b->env, 0, sizeof b->env);
  const RgModelDev& d = m->dev;
  RgBatchDev& s = b->dev;
  s.B = B;
  s.qpos = (float*)balloc(b, (size_t)B * d.nq * 4); s.qvel = (float*)balloc(b, (size_t)B * d.nv * 4);
  s.ctrl = (float*)balloc(b, (size_t)B * d.nu * 4); s.pid = (float*)balloc(b, (size_t)B * 3 * d.nu * 4);
  s.qacc_warmstart = (float*)balloc(b, (size_t)B * d.nv * 4); s.time = (float*)balloc(b, (size_t)B * 4);
  s.status = (uint32_t*)balloc(b, (size_t)B * 4); s.stats = (float*)balloc(b, (size_t)B * 16);
  s.dbg = (float*)balloc(b, (size_t)B * RG_DBG_SIZE * 4);
  s.cost = (float*)balloc(b, (size_t)B * 4);
  s.sepdir = (float*)balloc(b, (size_t)B * (d.npair > 0 ? d.npair : 1) * 16);
  s.pairlb = (float*)balloc(b, (size_t)B * (d.npair > 0 ? d.npair : 1) * 4);
  s.sched = (int*)balloc(b, ((size_t)RG_SCHED_PROG + B) * 4);
  if (!s.sched) { fail("hipMalloc failed"); rg_batch_free(b); return nullptr; }
  rg_items_probe(b);
  if (!s.qpos || !s.qvel || !s.ctrl || !s.pid || !s.qacc_warmstart || !s.time || !s.status || !s.stats || !s.dbg || !s.sepdir || !s.pairlb || !s.cost) { fail("hipMalloc failed"); rg_batch_free(b); return nullptr; }
  if (rg_batch_reset(b) != 0) { rg_batch_free(b); return nullptr; }
  return b;
}
void rg_batch_free(rg_batch* b) {
  if (!b) return;
  { DeviceGuard g(b->device); for (void* p : b->allocs) hipFree(p); }
  delete b;
}
int rg_batch_reset(rg_batch* b) {
  if (!b) return fail("null batch");
  DeviceGuard g(b->device);
  const RgModelDev& d = b->model->dev;
  RgBatchDev& s = b->dev;
  HIPCHK(hipDeviceSynchronize());   // slow-path call: ordered after everything queued on any stream (step launches run on the caller's streams)
  std::vector<float> q((size_t)s.B * d.nq);
  for (int e = 0; e < s.B; e++) memcpy(q.data() + (size_t)e * d.nq, b->model->qpos0.data(), d.nq * 4);
  HIPCHK(hipMemcpy(s.qpos, q.data(), q.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(s.qvel, 0, (size_t)s.B * d.nv * 4)); HIPCHK(hipMemset(s.ctrl, 0, (size_t)s.B * d.nu * 4));
  HIPCHK(hipMemset(s.pid, 0, (size_t)s.B * 3 * d.nu * 4)); HIPCHK(hipMemset(s.qacc_warmstart, 0, (size_t)s.B * d.nv * 4));
  HIPCHK(hipMemset(s.time, 0, (size_t)s.B * 4)); HIPCHK(hipMemset(s.status, 0, (size_t)s.B * 4)); HIPCHK(hipMemset(s.stats, 0, (size_t)s.B * 16));
  HIPCHK(hipMemset(s.pairlb, 0, (size_t)s.B * (d.npair > 0 ? d.npair : 1) * 4));
  return 0;
}

int rg_batch_enable_env_params(rg_batch* b) {
  if (!b) return fail("null batch");
  if (b->dev.envprm) return 0;
  DeviceGuard g(b->device);
  const size_t n = (size_t)b->dev.B * RG_NPRM;
  float* p = (float*)balloc(b, n * 4);
  if (!p) return fail("hipMalloc failed");
  std::vector<float> rows(n);
  for (int e = 0; e < b->dev.B; e++) memcpy(rows.data() + (size_t)e * RG_NPRM, b->model->prm_default.data(), RG_NPRM * 4);
  HIPCHK(hipMemcpy(p, rows.data(), n * 4, hipMemcpyHostToDevice));
  b->dev.envprm = p;
  return 0;
}
int rg_xdata_layout(int* out, int n) {
  const int lay[] = {RG_XDATA, RG_XD_XPOS, RG_XD_XQUAT, RG_XD_SITE_XPOS, RG_XD_ACT_FORCE, RG_XD_NCON, RG_XD_CONTACT, RG_DBG_MAXCON, RG_XD_SENSOR, RG_MAXSENSOR};
  const int k = (int)(sizeof lay / sizeof lay[0]);
  for (int i = 0; i < k && i < n; i++) out[i] = lay[i];
  return k;
}
int rg_prm_layout(int* out, int n) {
  const int lay[] = {RG_NPRM, RG_PRM_GRAVITY, RG_PRM_TIMESTEP, RG_PRM_DOF_DAMPING, RG_PRM_DOF_ARMATURE, RG_PRM_DOF_FRICTIONLOSS, RG_PRM_DOF_INVWEIGHT0, RG_PRM_BODY_MASS,
                     RG_PRM_BODY_INERTIA, RG_PRM_BODY_INVWEIGHT0, RG_PRM_JNT_RANGE, RG_PRM_TENDON_RANGE, RG_PRM_TENDON_INVWEIGHT0, RG_PRM_ACT_GAINPRM, RG_PRM_ACT_CTRLRANGE,
                     RG_PRM_ACT_FORCERANGE, RG_PRM_GEOM_FRICTION, RG_PRM_XFRC, RG_PRM_SITE_POS, RG_PRM_GEOM_SCALE, RG_PRM_JNT_MARGIN, RG_PRM_GEOM_SOLREF, RG_PRM_GEOM_SOLIMP};
  const int k = (int)(sizeof lay / sizeof lay[0]);
  for (int i = 0; i < k && i < n; i++) out[i] = lay[i];
  return k;
}
int rg_batch_set_env(rg_batch* b, const int* ints, int nints, const float* p2c, float thr) {
  if (!b || !ints || nints < 20) return fail("rg_batch_set_env: need 20 ints");
  DeviceGuard g(b->device);
  RgEnvDev& e = b->env;
  e.hand_qposadr = ints[0]; e.n_hand_jnt = ints[1]; e.cube_pos_qposadr = ints[2]; e.cube_quat_qposadr = ints[3];
  e.target_qposadr = ints[4]; e.target_nq = ints[5]; e.target_dofadr = ints[6]; e.target_nv = ints[7]; e.cube_body = ints[8];
  for (int i = 0; i < 3; i++) e.ref_site[i] = ints[9 + i];
  for (int i = 0; i < 5; i++) e.tip_site[i] = ints[12 + i];
  e.relative_action = ints[17];
  e.success_threshold = thr;
  size_t n = (size_t)b->model->dev.nu * e.n_hand_jnt;
  float* dp = (float*)balloc(b, n * 4);
  if (!dp) return fail("hipMalloc failed");
  HIPCHK(hipMemcpy(dp, p2c, n * 4, hipMemcpyHostToDevice));
  e.pos_to_ctrl = dp;
  b->has_env = 1;
  return 0;
}
int rg_obs_dim(const rg_batch* b) {
  if (!b || !b->has_env) return -1;
  return 7 + b->model->dev.nq + b->model->dev.nv + b->env.n_hand_jnt + 15;
}

static void* field_ptr(rg_batch* b, int field, size_t* n) {
  const RgModelDev& d = b->model->dev;
  RgBatchDev& s = b->dev;
  switch (field) {
    case RG_F_QPOS: *n = (size_t)d.nq; return s.qpos;
    case RG_F_QVEL: *n = (size_t)d.nv; return s.qvel;
    case RG_F_CTRL: *n = (size_t)d.nu; return s.ctrl;
    case RG_F_PID: *n = (size_t)3 * d.nu; return s.pid;
    case RG_F_WARMSTART: *n = (size_t)d.nv; return s.qacc_warmstart;
    case RG_F_TIME: *n = 1; return s.time;
    case RG_F_STATUS: *n = 1; return s.status;
    case RG_F_STATS: *n = 4; return s.stats;
    case RG_F_DEBUG: *n = RG_DBG_SIZE; return s.dbg;
    case RG_F_COST: *n = 1; return s.cost;
    case RG_F_PAIRLB: *n = (size_t)(d.npair > 0 ? d.npair : 1); return s.pairlb;
    case RG_F_ENVPRM: *n = RG_NPRM; return (void*)s.envprm;   // (null until rg_batch_enable_env_params)
    default: return nullptr;
  }
}
void* rg_batch_field_ptr(rg_batch* b, int field, int* row_words) {
  if (!b) { fail("null batch"); return nullptr; }
  size_t n = 0; void* p = field_ptr(b, field, &n);
  if (!p) { fail("rg_batch_field_ptr: unknown field"); return nullptr; }
  if (row_words) *row_words = (int)n;
  return p;
}
int rg_batch_copy(rg_batch* b, int field, void* ptr, int to_batch, int ptr_is_device) {
  if (!b || !ptr) return fail("rg_batch_copy: null argument");
  DeviceGuard g(b->device);
  const RgModelDev& d = b->model->dev;
  RgBatchDev& s = b->dev;
  size_t n = 0; void* p = field_ptr(b, field, &n);
  if (!p) return fail("rg_batch_copy: unknown field");
  size_t bytes = n * 4 * (size_t)s.B;
  HIPCHK(hipDeviceSynchronize());   // slow-path call: ordered after everything queued on any stream (the blocking copies below then complete before it returns)
  if (to_batch) {
    HIPCHK(hipMemcpy(p, ptr, bytes, ptr_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    // positions changed behind the kernel's back: the cached pair distance bounds are void
    if (field == RG_F_QPOS) HIPCHK(hipMemset(s.pairlb, 0, (size_t)s.B * (d.npair > 0 ? d.npair : 1) * 4));
  }
  else HIPCHK(hipMemcpy(ptr, p, bytes, ptr_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
  return 0;
}

#ifdef RG_EMUL
struct EmulCopyArgs { float* dst; const float* src; const int* mask; int n, col0, ncols; float* lb; int npair; };
static void emul_copy_entry(void* a) { EmulCopyArgs* p = (EmulCopyArgs*)a; rgs::rg_copy_rows_kernel(p->dst, p->src, p->mask, p->n, p->col0, p->ncols, p->lb, p->npair); }
#endif
int rg_batch_copy_rows(rg_batch* b, int field, const void* src_dev, const int* mask_dev, int col0, int ncols, void* stream) {
  if (!b || !src_dev || !mask_dev) return fail("rg_batch_copy_rows: null argument");
  DeviceGuard g(b->device);
  const RgModelDev& d = b->model->dev;
  RgBatchDev& s = b->dev;
  void* p; int n;
  switch (field) {
    case RG_F_QPOS: p = s.qpos; n = d.nq; break;
    case RG_F_QVEL: p = s.qvel; n = d.nv; break;
    case RG_F_CTRL: p = s.ctrl; n = d.nu; break;
    case RG_F_PID: p = s.pid; n = 3 * d.nu; break;
    case RG_F_WARMSTART: p = s.qacc_warmstart; n = d.nv; break;
    case RG_F_TIME: p = s.time; n = 1; break;
    case RG_F_STATUS: p = s.status; n = 1; break;
    default: return fail("rg_batch_copy_rows: field cannot be written row-wise");
  }
  if (ncols <= 0) { col0 = 0; ncols = n; }
  if (col0 < 0 || col0 + ncols > n) return fail("rg_batch_copy_rows: column range outside the row");
  float* lb = field == RG_F_QPOS ? s.pairlb : nullptr;
#ifdef RG_EMUL
  EmulCopyArgs args{(float*)p, (const float*)src_dev, mask_dev, n, col0, ncols, lb, d.npair};
  emul_launch(s.B, 16, emul_copy_entry, &args);
#else
  hipLaunchKernelGGL(rgs::rg_copy_rows_kernel, dim3(s.B), dim3(RG_WAVE), 0, (hipStream_t)stream, (float*)p, (const float*)src_dev, mask_dev, n, col0, ncols, lb, d.npair);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

// ---- substep-granular dispatch: is it available on this device?  One queue per XCD, served only by workgroups of that XCD:
// every XCD must receive workgroups of a slots-sized grid.  Probed once per batch with a tiny kernel that counts workgroups
// per XCC id; if any XCD stays empty (or the ids are not 0..7) the batch keeps the one-workgroup-per-env kernel.
#ifndef RG_EMUL
__global__ void rg_xcc_probe_kernel(int* counts) { if (threadIdx.x == 0) atomicAdd(counts + (rg_xcc_id() & 15), 1); }
#endif
static void rg_items_probe(rg_batch* b) {
#ifdef RG_EMUL
  b->items_slots = 2; b->items_queues = 1;   // (workgroups run one after the other: the first drains the queue, the second finds it empty)
#else
  b->items_slots = 0; b->items_queues = 1;
  // the probe's answer is a property of the device: measured once per device and process (it used to allocate, launch and synchronise in every
  // rg_batch_create), and a failed probe launch is reported through rg_last_error instead of being swallowed -- the mode then stays off, which is always correct
  // (the cache is shared by every thread that creates batches -- one per device is a normal host layout -- so it sits behind a mutex, and a probe whose
  //  launch FAILED is cached too, as "off", instead of allocating and launching again in every rg_batch_create)
  static int cached_slots[64], cached_queues[64]; static bool cached[64];
  static std::mutex probe_mutex;
  std::lock_guard<std::mutex> probe_lock(probe_mutex);
  const bool cacheable = b->device >= 0 && b->device < 64;
  if (cacheable && cached[b->device]) {
    b->items_slots = cached_slots[b->device]; b->items_queues = cached_queues[b->device];
    if (const char* ov = getenv("RG_ITEMS_SLOTS")) { const int v = atoi(ov); if (v > 0 && b->items_slots > 0) b->items_slots = v; }
    return;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, b->device) != hipSuccess) return;
  const int cus = prop.multiProcessorCount;
  // wave slots: LDS granules (1280 B) per CU / per workgroup, capped by the 3 waves per SIMD the kernel is compiled for
  const int per_cu_lds = (160 * 1024) / (int)(((rgi::rg_lds_launch_bytes(false) + 1279) / 1280) * 1280);
  const int per_cu = per_cu_lds < 4 * RG_WAVES_PER_SIMD_HOST ? per_cu_lds : 4 * RG_WAVES_PER_SIMD_HOST;
  int* counts = nullptr;
  if (hipMalloc((void**)&counts, 16 * 4) != hipSuccess) return;
  (void)hipMemset(counts, 0, 16 * 4);
  const int grid = cus * per_cu;
  hipLaunchKernelGGL(rg_xcc_probe_kernel, dim3(grid), dim3(RG_WAVE), 0, 0, counts);
  const hipError_t launch_err = hipGetLastError();
  int h[16];
  if (launch_err != hipSuccess) {
    (void)fail(std::string("rg_items_probe: probe launch failed (substep-granular dispatch stays off): ") + hipGetErrorString(launch_err));
    if (cacheable) { cached_slots[b->device] = 0; cached_queues[b->device] = 1; cached[b->device] = true; }
  } else if (hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
    int nq = 0; bool ok = true;
    for (int x = 0; x < 16; x++) if (h[x] > 0) nq = x + 1;
    for (int x = 0; x < nq; x++) if (h[x] <= 0) ok = false;
    if (ok && nq >= 1 && nq <= 8) { b->items_slots = grid; b->items_queues = nq; }
    if (cacheable) { cached_slots[b->device] = b->items_slots; cached_queues[b->device] = b->items_queues; cached[b->device] = true; }
    if (const char* ov = getenv("RG_ITEMS_SLOTS")) { const int v = atoi(ov); if (v > 0 && b->items_slots > 0) b->items_slots = v; }   // (experiments: persistent workgroups per launch)
  }
  (void)hipFree(counts);
#endif
}
int rg_batch_items_info(const rg_batch* b, int* slots, int* queues) {
  if (!b) return fail("null batch");
  if (slots) *slots = b->items_slots;
  if (queues) *queues = b->items_queues;
  return 0;
}

#ifdef RG_EMUL
struct EmulArgs { const RgModelDev* m; RgLaunch launch; };
static void emul_entry_items(void* a);
static void emul_entry(void* a) { EmulArgs* p = (EmulArgs*)a; rgs::rg_step_kernel(p->m, p->launch); }
static void emul_entry_large(void* a) { EmulArgs* p = (EmulArgs*)a; rgl::rg_step_kernel(p->m, p->launch); }
static void emul_entry_sensors(void* a) { EmulArgs* p = (EmulArgs*)a; rgx::rg_step_kernel(p->m, p->launch); }
static void emul_entry_items(void* a) { EmulArgs* p = (EmulArgs*)a; rgi::rg_step_items_kernel(p->m, p->launch); }
#endif

int rg_batch_step_ex(rg_batch* b, const rg_step_args* a) {
  if (!b || !a) return fail("null argument");
  if ((a->action_dev || a->obs_dev) && !b->has_env) return fail("rg_batch_set_env must be called before stepping with actions/observations");
  if (a->nsubsteps < 0 || a->nforward_ticks < 0) return fail("rg_batch_step: negative step counts");
  DeviceGuard g(b->device);
  RgBatchDev bt = b->dev;
  bt.action = a->action_dev; bt.goal_quat = a->goal_quat_dev; bt.obs = a->obs_dev; bt.goal_dist = a->goal_dist_dev; bt.active = a->active_dev;
  bt.hold = a->hold_dev; bt.nticks = a->nticks_dev; bt.order = a->order_dev;
  // flags bit 5 (sensordata): the sensor-evaluating instantiation, which has the large capacities (no redo hand-off needed)
  const bool sens = (a->flags & 32) != 0 && a->xdata_dev && b->model->dev.nsensor > 0;
  const bool large = a->config == RG_CFG_LARGE || sens, prof = (a->flags & 2) != 0;
  if (a->config != RG_CFG_LARGE && a->config != RG_CFG_ROLLOUT) return fail("rg_batch_step: unknown kernel configuration");
  bt.redo = large ? nullptr : a->redo_dev;
  bt.preticks = a->preticks_dev;
  bt.xdata = a->xdata_dev;
  // flags bit 7: substep-granular dispatch (rollout configuration, >= 2 substeps, no debug / profiling / sensor pass): see rg_step_items_kernel
  const bool items = (a->flags & 128) && !large && !(a->flags & (1 | 2 | 256)) && a->nsubsteps >= 2 && b->items_slots > 0;
  RgLaunch launch{b->model->aux, b->env, bt, a->nsubsteps, a->nforward_ticks, a->flags, items ? b->items_queues : 1};
  if (items) {
    const long long nitems = (long long)bt.B * a->nsubsteps;
    const int grid = (int)(nitems < b->items_slots ? nitems : b->items_slots);
    const size_t ldsi = rgi::rg_lds_launch_bytes(false);
#ifdef RG_EMUL
    memset(bt.sched, 0, ((size_t)RG_SCHED_PROG + bt.B) * 4);
    EmulArgs args{b->model->dev_copy, launch};
    emul_launch(grid, ldsi, emul_entry_items, &args);
#else
    HIPCHK(hipMemsetAsync(bt.sched, 0, ((size_t)RG_SCHED_PROG + bt.B) * 4, (hipStream_t)a->stream));
    hipLaunchKernelGGL(rgi::rg_step_items_kernel, dim3(grid), dim3(RG_WAVE), ldsi, (hipStream_t)a->stream, b->model->dev_copy, launch);
    HIPCHK(hipGetLastError());
#endif
    return 0;
  }
  const size_t lds = sens ? rgx::rg_lds_launch_bytes(prof) : (large ? rgl::rg_lds_launch_bytes(prof) : rgs::rg_lds_launch_bytes(prof));
#ifdef RG_EMUL
  EmulArgs args{b->model->dev_copy, launch};
  emul_launch(bt.B, lds, sens ? emul_entry_sensors : (large ? emul_entry_large : emul_entry), &args);
#else
  if (sens) hipLaunchKernelGGL(rgx::rg_step_kernel, dim3(bt.B), dim3(RG_WAVE), lds, (hipStream_t)a->stream, b->model->dev_copy, launch);
  else if (large) hipLaunchKernelGGL(rgl::rg_step_kernel, dim3(bt.B), dim3(RG_WAVE), lds, (hipStream_t)a->stream, b->model->dev_copy, launch);
  else hipLaunchKernelGGL(rgs::rg_step_kernel, dim3(bt.B), dim3(RG_WAVE), lds, (hipStream_t)a->stream, b->model->dev_copy, launch);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
int rg_batch_step(rg_batch* b, const float* action_dev, const float* goal_quat_dev, float* obs_dev, float* goal_dist_dev,
                  const int* active_dev, int nsubsteps, int nforward_ticks, int flags, void* stream) {
  rg_step_args a;
  memset(&a, 0, sizeof a);
  a.action_dev = action_dev; a.goal_quat_dev = goal_quat_dev; a.obs_dev = obs_dev; a.goal_dist_dev = goal_dist_dev; a.active_dev = active_dev;
  a.nsubsteps = nsubsteps; a.nforward_ticks = nforward_ticks; a.flags = flags; a.stream = stream;
  return rg_batch_step_ex(b, &a);
}

#ifdef RG_EMUL
struct EmulMprArgs { const RgModelDev* m; RgLaunch launch; int g1, g2; float margin; float* out; };
static void emul_mpr_entry(void* a) { EmulMprArgs* p = (EmulMprArgs*)a; rgs::rg_mpr_pair_kernel(p->m, p->launch, p->g1, p->g2, p->margin, p->out); }
#endif
int rg_batch_mpr_pair(rg_batch* b, int g1, int g2, float margin, float* out_dev, void* stream) {
  if (!b || !out_dev) return fail("null argument");
  const RgModelDev& d = b->model->dev;
  if (g1 < 0 || g2 < 0 || g1 >= d.ngeom || g2 >= d.ngeom) return fail("geom id out of range");
  DeviceGuard g(b->device);
  RgLaunch launch{b->model->aux, b->env, b->dev, 0, 0, 0, 1};
#ifdef RG_EMUL
  EmulMprArgs args{b->model->dev_copy, launch, g1, g2, margin, out_dev};
  emul_launch(b->dev.B, sizeof(rgs::RgLds), emul_mpr_entry, &args);
#else
  hipLaunchKernelGGL(rgs::rg_mpr_pair_kernel, dim3(b->dev.B), dim3(RG_WAVE), sizeof(rgs::RgLds), (hipStream_t)stream, b->model->dev_copy, launch, g1, g2, margin, out_dev);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

#ifdef RG_EMUL
struct EmulPostArgs { RgBatchDev bt; RgPostArgs a; int nq, nv, nu, npair; };
static void emul_post_entry(void* p_) { EmulPostArgs* p = (EmulPostArgs*)p_; rg_post_step_kernel(p->bt, p->a, p->nq, p->nv, p->nu, p->npair); }
#endif
int rg_post_args_size(void) { return (int)sizeof(rg_post_args); }
int rg_env_post_step(rg_batch* b, const rg_post_args* args, void* stream) {
  if (!b || !args) return fail("null argument");
  if (!b->has_env) return fail("rg_batch_set_env must be called first");
  const rg_post_args& a = *args;
  if (!a.goal_dist || !a.obs || !a.t || !a.phase || !a.tries || !a.steps || !a.steps_since_last_goal || !a.successes_so_far || !a.goals_so_far || !a.consecutive ||
      !a.prev_dist || !a.prev_valid || !a.is_successful || !a.goal_quat || !a.qpos_goal || !a.preticks || !a.reward || !a.done || !a.goal_reset || !a.trial_success ||
      !a.sub_goal_ok || !a.env_crash || !a.resetting || !a.episode_started || !a.info_ssl || !a.nticks_next || !a.reset_mask || !a.live_mask || !a.goal_dist_before || !a.parallel_quats || !a.qpos0 || !a.zero_ctrl ||
      !a.ctrl_lo || !a.ctrl_hi) return fail("rg_env_post_step: a required array is NULL");
  const RgModelDev& d = b->model->dev;
  if (d.nu > 20) return fail("rg_env_post_step: nu exceeds RG_POST_NDRAW's action slots");
  DeviceGuard g(b->device);
#ifdef RG_EMUL
  EmulPostArgs ea{b->dev, a, d.nq, d.nv, d.nu, d.npair};
  emul_launch(b->dev.B, 256, emul_post_entry, &ea);
#else
  hipLaunchKernelGGL(rg_post_step_kernel, dim3(b->dev.B), dim3(RG_WAVE), 0, (hipStream_t)stream, b->dev, a, d.nq, d.nv, d.nu, d.npair);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

#ifdef RG_EMUL
struct EmulSetconstArgs { const RgModelDev* m; RgLaunch launch; float* envprm; };
static void emul_setconst_entry(void* a) { EmulSetconstArgs* p = (EmulSetconstArgs*)a; rgl::rg_setconst_kernel(p->m, p->launch, p->envprm); }
#endif
int rg_batch_set_constants(rg_batch* b, const int* mask_dev, void* stream) {
  if (!b) return fail("null batch");
  if (!b->dev.envprm) return fail("rg_batch_set_constants: the batch has no per-env parameter rows (rg_batch_enable_env_params); the model's own constants are already consistent");
  DeviceGuard g(b->device);
  RgBatchDev bt = b->dev;
  bt.active = mask_dev; bt.order = nullptr;
  RgLaunch launch{b->model->aux, b->env, bt, 0, 0, 0, 1};
  const size_t lds = rgl::rg_lds_setconst_bytes();
#ifdef RG_EMUL
  EmulSetconstArgs args{b->model->dev_copy, launch, (float*)b->dev.envprm};
  emul_launch(bt.B, lds, emul_setconst_entry, &args);
#else
  hipLaunchKernelGGL(rgl::rg_setconst_kernel, dim3(bt.B), dim3(RG_WAVE), lds, (hipStream_t)stream, b->model->dev_copy, launch, (float*)b->dev.envprm);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

// =====================================================================================================================
// Large-model path (rb_kernel.h): models beyond the compile-time layout of the Shadow-hand kernel.  Same blob, same
// ownership and error conventions; entry points rb_* (include/rgstep.h).
// =====================================================================================================================
struct rb_model {
  int device = 0;
  RbModelDev dev;
  const RbModelDev* dev_copy = nullptr;
  std::vector<void*> allocs;
  std::vector<float> qpos0, mocap0, eq_data0;
  std::vector<int> eq_active0;
  int config = 0;   // 0: large configuration of rb_kernel.h, 1: small, 2: medium
  std::vector<std::string> blob_keys;
  std::vector<float> prm_default;   // the model's own values in the layout of an env's parameter block (RB_P_* order)
  int prm_len[RB_NPRMF] = {0};      // words of each field
  int nbatches = 0;                 // batches created from this model (the per-env parameter switch must precede them)
};
struct rb_batch {
  const rb_model* model;
  RbBatchDev dev;
  RbEnvDev env;
  int device;
  int has_env = 0;
  std::vector<void*> allocs;
};
static bool rb_upload_bytes(rb_model* m, const void* host, size_t bytes, const void** dst) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return false;
  m->allocs.push_back(p);
  if (bytes && hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
  *dst = p;
  return true;
}
void rb_model_free(rb_model* m) {
  if (!m) return;
  { DeviceGuard g(m->device); for (void* p : m->allocs) hipFree(p); }
  delete m;
}
// the per-env scratch row: stage arrays at the model's capacities, then (models switched to per-env parameters) the env's parameter block
static void rb_layout(rb_model* m) {
  RbModelDev& d = m->dev;
  int o = 0;
  auto take = [&](int which, int words) { d.off[which] = o; o += (words + 3) & ~3; };
  take(RB_O_XPOS, 3 * d.nbody); take(RB_O_XQUAT, 4 * d.nbody); take(RB_O_XIPOS, 3 * d.nbody); take(RB_O_XIQUAT, 4 * d.nbody);
  take(RB_O_XANCHOR, 3 * d.njnt); take(RB_O_XAXIS, 3 * d.njnt); take(RB_O_GPOS, 3 * d.ngeom); take(RB_O_GQUAT, 4 * d.ngeom); take(RB_O_SPOS, 3 * d.nsite);
  take(RB_O_ROOTCOM, 3 * d.nbody); take(RB_O_CINERT, 10 * d.nbody); take(RB_O_CRB, 10 * d.nbody); take(RB_O_CDOF, 6 * d.nv); take(RB_O_CDOFDOT, 6 * d.nv);
  take(RB_O_CVEL, 6 * d.nbody); take(RB_O_CACC, 6 * d.nbody); take(RB_O_CFRC, 6 * d.nbody);
  take(RB_O_TENLEN, d.ntendon); take(RB_O_TENJ, RB_TENW * d.ntendon); take(RB_O_TENVEL, d.ntendon); take(RB_O_MSP, d.nM);
  const int cw = m->config ? RB_CONW_ONEWAVE : RB_CONW;      // dofs per contact row the model's configuration is compiled for
  take(RB_O_CAND, d.maxcand); take(RB_O_CON, RB_CONREC * d.maxcon); take(RB_O_CONJ, 6 * cw * d.maxcon); take(RB_O_CONIDX, cw * d.maxcon);
  take(RB_O_ROW, RB_ROWREC * d.maxrow); take(RB_O_DOFCON_ADR, d.nv + 1); take(RB_O_DOFCON, cw * d.maxcon); take(RB_O_CONF, RB_NW * d.maxcon); take(RB_O_DBG, 8 + 5 * d.nv + 16); take(RB_O_CFRCEXT, 6 * d.nbody); take(RB_O_CONLOC, cw * d.maxcon);
  d.prm_words = 0;
  for (int k = 0; k < RB_NPRMF; k++) d.prm_off[k] = 0;
  d.off[RB_O_PRM] = o;
  if (d.prm_on) {
    int w = 0;
    for (int k = 0; k < RB_NPRMF; k++) { d.prm_off[k] = o + w; w += (m->prm_len[k] + 3) & ~3; }
    d.prm_words = w; o += w;
  }
  d.scratch_words = o;
}
rb_model* rb_model_create(const void* blob, size_t nbytes, char* err, int errlen) {
  auto bail = [&](const std::string& msg, rb_model* m) -> rb_model* {
    g_key_log = nullptr;
    g_err = msg;
    if (err && errlen > 0) { strncpy(err, msg.c_str(), errlen - 1); err[errlen - 1] = 0; }
    if (m) rb_model_free(m);
    return nullptr;
  };
  if (!blob || nbytes < 16 || memcmp(blob, "RGMODEL1", 8) != 0) return bail("not an RGMODEL1 blob", nullptr);
  Blob B{(const char*)blob, nbytes};
  rb_model* m = new rb_model();
  g_key_log = &m->blob_keys;
#ifndef RG_EMUL
  if (hipGetDevice(&m->device) != hipSuccess) return bail("hipGetDevice failed", m);
#endif
  RbModelDev& d = m->dev;
  memset(&d, 0, sizeof d);
  std::string e;
  std::vector<int> iv; std::vector<float> fv;
  if (!get_i(B, "dims", iv, e)) return bail(e, m);
  d.nq = iv[0]; d.nv = iv[1]; d.nu = iv[2]; d.nbody = iv[3]; d.njnt = iv[4]; d.ngeom = iv[5]; d.nsite = iv[6]; d.ntendon = iv[7]; d.nmesh = iv[9];
  if (!get_i(B, "b_dims", iv, e)) return bail(e + " (derive_big_tables was not run on the model)", m);
  d.nlevel = iv[0]; d.nM = iv[1]; d.npair = iv[2]; d.ngroup = iv[3]; d.gmax = iv[4]; d.nroot = iv[5]; d.conw = iv[6];
  if (d.gmax > RB_MAXGROUP_LARGE) return bail("a constraint-coupled group of trees has more dofs than RB_MAXGROUP", m);
  if (d.nv > RB_MAXNV_LARGE || d.nq > RB_MAXNQ_LARGE || d.nu > 32 || d.conw > RB_CONW) return bail("model exceeds the LDS vector capacities of rb_kernel.h", m);
  {
    // the small configuration (one wave per env) when the model fits it: LDS vectors, the dense block, and the star trees' branch lists
    // (one thread per branch, 28 words of the block's storage per branch + 27)
    std::vector<int> td, te;
    if (!get_i(B, "b_tree_desc", td, e) || !get_i(B, "b_tree_brn_end", te, e)) return bail(e, m);
    int tmax = 0;
    for (size_t t = 0; t < te.size() && 4 * t + 3 < td.size(); t++) tmax = std::max(tmax, te[t] - td[4 * t + 3]);
    const char* force = getenv("RB_CONFIG");
    auto fits = [&](int maxgroup, int maxnv, int maxnq, int threads) {
      return d.gmax <= maxgroup && d.nv <= maxnv && d.nq <= maxnq && tmax <= threads && 28 * tmax + 27 <= maxgroup * (maxgroup + 1) / 2 + 8 && d.conw <= RB_CONW_ONEWAVE && d.nu <= RB_MAXNU_ONEWAVE;
    };
    m->config = 0;
    if (!(force && !strcmp(force, "large"))) {
      if (fits(RB_MAXGROUP_SMALL, RB_MAXNV_SMALL, RB_MAXNQ_SMALL, RB_T_SMALL) && !(force && !strcmp(force, "medium"))) m->config = 1;
      else if (fits(RB_MAXGROUP_MEDIUM, RB_MAXNV_MEDIUM, RB_MAXNQ_MEDIUM, RB_T_MEDIUM)) m->config = 2;
    }
  }
#define X(n) if (!get_i(B, #n, iv, e)) return bail(e, m); if (!rb_upload_bytes(m, iv.data(), iv.size() * 4, (const void**)&d.n)) return bail("hipMalloc failed", m);
  RB_INT_ARRAYS(X)
#undef X
#define X(n) if (!get_f(B, #n, fv, e)) return bail(e, m); if (!rb_upload_bytes(m, fv.data(), fv.size() * 4, (const void**)&d.n)) return bail("hipMalloc failed", m);
  RB_FLT_ARRAYS(X)
#undef X
  if (!get_i(B, "b_fric_dof", iv, e)) return bail(e, m); d.nfric_dof = (int)iv.size();
  if (!get_i(B, "b_fric_ten", iv, e)) return bail(e, m); d.nfric_ten = (int)iv.size();
  if (!get_i(B, "b_lim_jnt", iv, e)) return bail(e, m); d.nlim_jnt = (int)iv.size();
  if (!get_i(B, "b_lim_ten", iv, e)) return bail(e, m); d.nlim_ten = (int)iv.size();
  if (!get_i(B, "actuator_biastype", iv, e)) return bail(e, m);
  for (int v : iv) if (v != 2) return bail("rb_model_create: only mujoco-py PID actuators (biastype user) are implemented", m);
  if (!get_i(B, "opt_int", iv, e)) return bail(e, m);
  d.iterations = iv[0]; d.ls_iterations = iv[2]; d.mpr_iterations = iv[3];
  d.cone = iv[1];   // 0 pyramidal, 1 elliptic (ur16e/base.xml:3)
  if (!get_i(B, "eq_type", iv, e)) return bail(e, m); d.neq = (int)iv.size();
  if (!get_i(B, "nmocap", iv, e)) return bail(e, m); d.nmocap = iv[0];
  if (d.nmocap > 2) return bail("rb_model_create: at most two mocap bodies", m);
  if (!get_i(B, "sensor_type", iv, e)) return bail(e, m); d.nsensor = (int)iv.size();
  { std::vector<int> sa, sd; if (!get_i(B, "sensor_adr", sa, e) || !get_i(B, "sensor_dim", sd, e)) return bail(e, m); d.nsensordata = d.nsensor ? sa.back() + sd.back() : 0; }
  for (int k = 0; k < d.nsensor; k++) if (iv[k] != 0 && iv[k] != 4 && iv[k] != 5 && iv[k] != 8) return bail("rb_model_create: sensor type not implemented", m);
  if (!get_i(B, "size_int", iv, e)) return bail(e, m);
  d.maxrow = iv[0] > 0 ? iv[0] : 2000; d.maxcon = iv[1] > 0 ? iv[1] : 200;   // njmax / nconmax of the model
  // RB_SCRATCH_MAXCON / RB_SCRATCH_MAXROW: capacities of the per-env scratch row below the model's nconmax / njmax (the row's arrays are laid out at capacity:
  // rearrange's 500 / 2000 spread the ~40 kB an mj_step touches over 660 kB per env).  Exceeding them raises RG_STATUS_CON_FULL / ROW_FULL as the model's own do.
  if (const char* ov = getenv("RB_SCRATCH_MAXCON")) { const int v = atoi(ov); if (v > 0 && v < d.maxcon) d.maxcon = v; }
  if (const char* ov = getenv("RB_SCRATCH_MAXROW")) { const int v = atoi(ov); if (v > 0 && v < d.maxrow) d.maxrow = v; }
  d.maxcand = 4 * d.maxcon + 256;
  if (!get_f(B, "opt_timestep", fv, e)) return bail(e, m); d.timestep = fv[0];
  if (!get_f(B, "opt_gravity", fv, e)) return bail(e, m); for (int k = 0; k < 3; k++) d.gravity[k] = fv[k];
  if (!get_f(B, "opt_tolerance", fv, e)) return bail(e, m); d.tolerance = fv[0];
  if (!get_f(B, "opt_impratio", fv, e)) return bail(e, m); d.impratio = fv[0];
  if (!get_f(B, "opt_mpr_tolerance", fv, e)) return bail(e, m); d.mpr_tolerance = fv[0];
  if (!get_f(B, "stat_meaninertia", fv, e)) return bail(e, m); d.meaninertia = fv[0];
  if (!get_f(B, "qpos0", m->qpos0, e)) return bail(e, m);
  if (!get_f(B, "eq_data", m->eq_data0, e) || !get_i(B, "eq_active", m->eq_active0, e)) return bail(e, m);
  {
    std::vector<int> mid; std::vector<float> bp, bq;
    if (!get_i(B, "body_mocapid", mid, e) || !get_f(B, "body_pos", bp, e) || !get_f(B, "body_quat", bq, e)) return bail(e, m);
    m->mocap0.assign(7 * d.nmocap, 0.f);
    for (int bb = 0; bb < d.nbody; bb++) if (mid[bb] >= 0 && mid[bb] < d.nmocap) { for (int k = 0; k < 3; k++) m->mocap0[7 * mid[bb] + k] = bp[3 * bb + k]; for (int k = 0; k < 4; k++) m->mocap0[7 * mid[bb] + 3 + k] = bq[4 * bb + k]; }
  }
  {  // the model's own values of the per-env parameter fields, in block order (rb_types.h RB_P_*)
    static const char* fields[RB_NPRMF] = {"opt_gravity", "dof_damping", "dof_armature", "dof_frictionloss", "dof_invweight0", "jnt_stiffness", "jnt_margin", "jnt_range",
                                           "body_pos", "body_mass", "body_inertia", "body_invweight0", "actuator_gainprm", "actuator_forcerange", "actuator_ctrlrange",
                                           "geom_pos", "geom_margin", "geom_gap", "geom_friction", "geom_solref", "geom_solimp", "tendon_range", "tendon_invweight0"};
    std::vector<std::vector<float>> vals(RB_NPRMF);
    for (int k = 0; k < RB_NPRMF; k++) { if (!get_f(B, fields[k], vals[k], e)) return bail(e, m); m->prm_len[k] = (int)vals[k].size(); }
    m->prm_default.clear();
    for (int k = 0; k < RB_NPRMF; k++) { m->prm_default.insert(m->prm_default.end(), vals[k].begin(), vals[k].end()); m->prm_default.resize((m->prm_default.size() + 3) & ~(size_t)3, 0.f); }
  }
  rb_layout(m);
  // ---- LDS residency of stage arrays (rb_types.h lds_off): RB_LDS_PLACE = a preset or a comma-separated list of array names.  Only arrays whose length does not
  // depend on the number of contacts / rows of the mj_step can be placed (their capacity is the model's own size).
  {
    static const char* names[RB_NOFF] = {"xpos", "xquat", "xipos", "xiquat", "xanchor", "xaxis", "gpos", "gquat", "spos", "rootcom", "cinert", "crb", "cdof", "cdofdot", "cvel", "cacc", "cfrc",
                                         "tenlen", "tenj", "tenvel", "msp", "cand", "con", "conj", "conidx", "row", "dofcon_adr", "dofcon", "conf", "dbg", "cfrcext", "conloc", "prm"};
    const int cwl = m->config ? RB_CONW_ONEWAVE : RB_CONW;
    const int len[RB_NOFF] = {3 * d.nbody, 4 * d.nbody, 3 * d.nbody, 4 * d.nbody, 3 * d.njnt, 3 * d.njnt, 3 * d.ngeom, 4 * d.ngeom, 3 * d.nsite, 3 * d.nbody, 10 * d.nbody, 10 * d.nbody, 6 * d.nv, 6 * d.nv,
                              6 * d.nbody, 6 * d.nbody, 6 * d.nbody, d.ntendon, RB_TENW * d.ntendon, d.ntendon, d.nM, d.maxcand, RB_CONREC * d.maxcon, 6 * cwl * d.maxcon, cwl * d.maxcon, RB_ROWREC * d.maxrow, d.nv + 1, cwl * d.maxcon, RB_NW * d.maxcon, 0, 6 * d.nbody, cwl * d.maxcon, 0};
    for (int k = 0; k < RB_NOFF; k++) { d.lds_off[k] = -1; d.lds_len[k] = 0; }
    d.lds_words = 0;
    const char* place = getenv("RB_LDS_PLACE");
    std::string want = place ? place : "";
    if (want == "frames") want = "xpos,xquat,xipos,xiquat,xanchor,xaxis,gpos,gquat,spos,rootcom";
    else if (want == "kin") want = "xpos,xquat,xipos,xiquat,xanchor,xaxis,gpos,gquat,spos,rootcom,cinert,crb,cdof,cdofdot,cvel,cacc,cfrc,tenlen,tenj,tenvel,msp,dofcon_adr,cfrcext";
    else if (want == "dyn") want = "cinert,crb,cdof,cdofdot,cvel,cacc,cfrc,msp";
#ifndef RB_LDS_ARENA
    if (!want.empty()) return bail("RB_LDS_PLACE: this library was built without -DRB_LDS_ARENA (rb_kernel.h)", m);
#endif
    if (!want.empty() && m->config != 0) {   // (the one-wave configurations; the large configuration's LDS is spoken for by its 96-dof block)
      int lo = 0;
      size_t at = 0;
      while (at <= want.size()) {
        const size_t e2 = want.find(',', at);
        const std::string nm = want.substr(at, e2 == std::string::npos ? std::string::npos : e2 - at);
        at = e2 == std::string::npos ? want.size() + 1 : e2 + 1;
        if (nm.empty()) continue;
        int k = -1;
        for (int q = 0; q < RB_NOFF; q++) if (nm == names[q]) k = q;
        if (k < 0 || (len[k] == 0 && k != RB_O_TENLEN && k != RB_O_TENJ && k != RB_O_TENVEL)) return bail("RB_LDS_PLACE: unknown or unplaceable stage array name", m);
        if (d.lds_off[k] >= 0 || len[k] == 0) continue;
        d.lds_off[k] = lo; d.lds_len[k] = len[k]; lo += (len[k] + 3) & ~3;
      }
      d.lds_words = lo;
      const size_t base = m->config == 1 ? ((sizeof(rgbs::RbLds) + 15) & ~(size_t)15) : ((sizeof(rgbm::RbLds) + 15) & ~(size_t)15);
      if (base + 4 * (size_t)lo > 64 * 1024) return bail("RB_LDS_PLACE: the arena exceeds 64 kB of LDS per workgroup", m);
    }
  }
  void* p = nullptr;
  if (hipMalloc(&p, sizeof(RbModelDev)) != hipSuccess) return bail("hipMalloc failed", m);
  m->allocs.push_back(p);
  if (hipMemcpy(p, &d, sizeof(RbModelDev), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy failed", m);
  m->dev_copy = (const RbModelDev*)p;
  g_key_log = nullptr;
  return m;
}
// ---- the model blob as an interface of its own (SURVEY 8b lists `rg_compile_mjcf`: compilation stays host-side, the blob is what crosses the ABI)
static int join_keys(const std::vector<std::string>& keys, char* out, int outlen) {
  std::string j;
  for (size_t i = 0; i < keys.size(); i++) { if (i) j += ","; j += keys[i]; }
  if (out && outlen > 0) { strncpy(out, j.c_str(), outlen - 1); out[outlen - 1] = 0; }
  return (int)j.size() + 1;
}
// ---- MJCF across the boundary (SURVEY 8b `rg_compile_mjcf`; the reference's seam: MujocoXML.build -> mujoco_py.load_model_from_xml(xml_string), mujoco_xml.py:249-260).
// The compiler is the package's Python module; the library runs it as a HELPER PROCESS (robogym_amd/mujoco/compile_cli.py: document -> RGMODEL1 blob file) and
// creates the model from the blob it wrote, so that a host in any language hands over an XML string as the reference hands one to MuJoCo.
static std::string lib_root() {
  if (const char* ov = getenv("RGSTEP_PYTHONPATH")) return ov;
  Dl_info info;
  if (!dladdr((const void*)&rg_last_error, &info) || !info.dli_fname) return ".";
  std::string p = info.dli_fname;                       // <root>/robogym_amd/csrc/librgstep.so (or <root>/tests/emul/librgstep_emul.so): three levels up
  for (int k = 0; k < 3; k++) { const size_t at = p.find_last_of('/'); if (at == std::string::npos) return "."; p.erase(at); }
  return p.empty() ? "/" : p;
}
static bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n <= 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}
static bool compile_mjcf_to_blob(const char* xml, const char* meshdir, int kind, std::vector<char>& blob, std::string& err) {
  if (!xml || !*xml) { err = "rg_compile_mjcf: empty MJCF document"; return false; }
  if (kind != 0 && kind != 1) { err = "rg_compile_mjcf: kind must be 0 (rg_model) or 1 (rb_model)"; return false; }
  const char* tmp = getenv("TMPDIR");
  std::string dirs = std::string(tmp && *tmp ? tmp : "/tmp") + "/rgstep_mjcf_XXXXXX";
  std::vector<char> dirbuf(dirs.begin(), dirs.end()); dirbuf.push_back(0);
  char* dir = dirbuf.data();
  if (!mkdtemp(dir)) { err = "rg_compile_mjcf: mkdtemp failed under " + dirs; return false; }
  const std::string d = dir, fx = d + "/model.xml", fb = d + "/model.blob", fe = d + "/stderr.txt";
  auto cleanup = [&]() { unlink(fx.c_str()); unlink(fb.c_str()); unlink(fe.c_str()); rmdir(dir); };
  { FILE* f = fopen(fx.c_str(), "wb"); if (!f || fwrite(xml, 1, strlen(xml), f) != strlen(xml)) { if (f) fclose(f); cleanup(); err = "rg_compile_mjcf: cannot write the document to " + fx; return false; } fclose(f); }
  const char* py = getenv("RGSTEP_PYTHON");
  const std::string python = py && *py ? py : "python3", root = lib_root();
  // The helper is started with posix_spawnp from argv / envp / file actions that are complete BEFORE the call (the process holds HIP runtime threads,
  // so nothing that allocates or takes a lock may run between fork and exec); stdout / stderr go to a file of the scratch directory, stdin to /dev/null.
  // (Descriptors the runtime opened without O_CLOEXEC -- the GPU device nodes -- are inherited by the helper, which never touches them and exits.)
  std::vector<std::string> envs;
  bool have_pp = false;
  for (char** e = environ; e && *e; e++) {
    if (strncmp(*e, "PYTHONPATH=", 11) == 0) { envs.push_back(std::string("PYTHONPATH=") + root + ((*e)[11] ? std::string(":") + (*e + 11) : std::string())); have_pp = true; }
    else envs.push_back(*e);
  }
  if (!have_pp) envs.push_back("PYTHONPATH=" + root);
  std::vector<char*> envp; for (auto& e : envs) envp.push_back((char*)e.c_str()); envp.push_back(nullptr);
  std::vector<std::string> args = {python, "-m", "robogym_amd.mujoco.compile_cli", "--kind", kind == 0 ? "rg" : "rb", "--xml", fx, "--out", fb};
  if (meshdir && *meshdir) { args.push_back("--meshdir"); args.push_back(meshdir); }
  std::vector<char*> av; for (auto& a : args) av.push_back((char*)a.c_str()); av.push_back(nullptr);
  posix_spawn_file_actions_t fa; posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
  posix_spawn_file_actions_addopen(&fa, 1, fe.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  posix_spawn_file_actions_adddup2(&fa, 1, 2);
  pid_t pid = -1;
  const int rc = posix_spawnp(&pid, python.c_str(), &fa, nullptr, av.data(), envp.data());
  posix_spawn_file_actions_destroy(&fa);
  if (rc != 0) { cleanup(); err = "rg_compile_mjcf: the MJCF compiler (" + python + ") could not be started: " + strerror(rc); return false; }
  // bounded wait: RGSTEP_COMPILE_TIMEOUT seconds (default 600; the largest shipped document compiles in under a minute), then the helper is killed
  double limit = 600.0;
  if (const char* t = getenv("RGSTEP_COMPILE_TIMEOUT")) { const double v = atof(t); if (v > 0) limit = v; }
  int status = 0; bool timed_out = false;
  { struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
      const pid_t w = waitpid(pid, &status, WNOHANG);
      if (w == pid) break;
      if (w < 0 && errno != EINTR) { status = 0x7f00; break; }
      struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) { kill(pid, SIGKILL); while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {} timed_out = true; break; }
      struct timespec nap = {0, 20 * 1000 * 1000}; nanosleep(&nap, nullptr);
    } }
  if (timed_out) { cleanup(); err = "rg_compile_mjcf: the MJCF compiler did not finish within RGSTEP_COMPILE_TIMEOUT = " + std::to_string((int)limit) + " s and was killed"; return false; }
  const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0 && read_file(fb, blob) && blob.size() >= 16;
  if (!ok) {
    std::vector<char> msg; read_file(fe, msg);
    std::string tail(msg.begin(), msg.end());
    while (!tail.empty() && (tail.back() == '\n' || tail.back() == ' ')) tail.pop_back();
    if (tail.size() > 400) tail = tail.substr(tail.size() - 400);
    err = "rg_compile_mjcf: the MJCF compiler (" + python + " -m robogym_amd.mujoco.compile_cli, PYTHONPATH " + root + ") " +
          (WIFEXITED(status) && WEXITSTATUS(status) == 127 ? "could not be started" : "failed") + (tail.empty() ? "" : ": " + tail);
  }
  cleanup();
  return ok;
}
static void put_err(const std::string& e, char* err, int errlen) { g_err = e; if (err && errlen > 0) { strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; } }
int rg_compile_mjcf_blob(const char* xml, const char* meshdir, int kind, void** blob_out, size_t* nbytes_out, char* err, int errlen) {
  if (!blob_out || !nbytes_out) { put_err("rg_compile_mjcf_blob: null output pointer", err, errlen); return -1; }
  std::vector<char> blob; std::string e;
  if (!compile_mjcf_to_blob(xml, meshdir, kind, blob, e)) { put_err(e, err, errlen); return -1; }
  void* p = malloc(blob.size());
  if (!p) { put_err("rg_compile_mjcf_blob: out of memory", err, errlen); return -1; }
  memcpy(p, blob.data(), blob.size());
  *blob_out = p; *nbytes_out = blob.size();
  return 0;
}
void rg_blob_free(void* blob) { free(blob); }
rg_model* rg_compile_mjcf(const char* xml, const char* meshdir, char* err, int errlen) {
  std::vector<char> blob; std::string e;
  if (!compile_mjcf_to_blob(xml, meshdir, 0, blob, e)) { put_err(e, err, errlen); return nullptr; }
  return rg_model_create(blob.data(), blob.size(), err, errlen);
}
rb_model* rb_compile_mjcf(const char* xml, const char* meshdir, char* err, int errlen) {
  std::vector<char> blob; std::string e;
  if (!compile_mjcf_to_blob(xml, meshdir, 1, blob, e)) { put_err(e, err, errlen); return nullptr; }
  return rb_model_create(blob.data(), blob.size(), err, errlen);
}
int rg_model_blob_keys(const rg_model* m, char* out, int outlen) { return m ? join_keys(m->blob_keys, out, outlen) : fail("null model"); }
int rb_model_blob_keys(const rb_model* m, char* out, int outlen) { return m ? join_keys(m->blob_keys, out, outlen) : fail("null model"); }
int rg_blob_entry(const void* blob, size_t nbytes, int index, char* name40, int* dtype, unsigned* count) {
  if (!blob || nbytes < 16 || memcmp(blob, "RGMODEL1", 8) != 0) return fail("not an RGMODEL1 blob");
  const uint32_t n = *(const uint32_t*)((const char*)blob + 8);
  if (16 + (uint64_t)n * sizeof(blob_entry) > nbytes) return fail("blob directory out of bounds");
  if (index < 0) return (int)n;
  if ((uint32_t)index >= n) return fail("blob entry index out of range");
  const blob_entry& e = ((const blob_entry*)((const char*)blob + 16))[index];
  const uint64_t esz = e.dtype == 0 ? 8 : 4;
  if (e.dtype > 2 || e.offset > nbytes || (uint64_t)e.count * esz > nbytes - e.offset) return fail("blob entry out of bounds");
  if (name40) { memcpy(name40, e.name, 40); }
  if (dtype) *dtype = (int)e.dtype;
  if (count) *count = e.count;
  return (int)n;
}
int rb_model_info(const rb_model* m, int* out, int n) {
  if (!m) return fail("null model");
  const RbModelDev& d = m->dev;
  const int v[] = {d.nq, d.nv, d.nu, d.nbody, d.njnt, d.ngeom, d.nsite, d.ntendon, d.nM, d.npair, d.ngroup, d.gmax, d.maxcon, d.maxrow, d.scratch_words, RB_CONREC, RB_ROWREC, m->config ? RB_CONW_ONEWAVE : RB_CONW, RB_TENW,
                   (m->config == 1 ? (int)sizeof(rgbs::RbLds) : m->config == 2 ? (int)sizeof(rgbm::RbLds) : (int)sizeof(rgb::RbLds)) + (d.lds_words ? 4 * d.lds_words + 16 : 0), m->config == 1 ? RB_T_SMALL : m->config == 2 ? RB_T_MEDIUM : RB_T_LARGE};
  const int k = (int)(sizeof v / sizeof v[0]);
  for (int i = 0; i < k && i < n; i++) out[i] = v[i];
  return k;
}
// Per-env model parameters on this stepper (base.py:1008-1092's randomizers write these fields of `sim.model`): switches the MODEL so that every
// batch created from it afterwards carries a parameter block per env (rb_types.h RB_P_*, initialised with the model's values) which the kernel reads instead of
// the model's arrays.  Must precede rb_batch_create for this model.
int rb_model_enable_env_params(rb_model* m) {
  if (!m) return fail("null model");
  if (m->dev.prm_on) return 0;
  if (m->nbatches > 0) return fail("rb_model_enable_env_params: call it before the model's first rb_batch_create");
  DeviceGuard g(m->device);
  m->dev.prm_on = 1;
  rb_layout(m);
  HIPCHK(hipMemcpy((void*)m->dev_copy, &m->dev, sizeof(RbModelDev), hipMemcpyHostToDevice));
  return 0;
}
// out[0] = 1 if the model carries per-env blocks, out[1] = words of a block, then per field (RB_P_* order) its word offset in the scratch row and its length
int rb_prm_layout(const rb_model* m, int* out, int n) {
  if (!m || !out) return fail("rb_prm_layout: null argument");
  const int k = 2 + 2 * RB_NPRMF;
  std::vector<int> v(k);
  v[0] = m->dev.prm_on; v[1] = m->dev.prm_words;
  for (int f = 0; f < RB_NPRMF; f++) { v[2 + 2 * f] = m->dev.prm_off[f]; v[3 + 2 * f] = m->prm_len[f]; }
  for (int i = 0; i < k && i < n; i++) out[i] = v[i];
  return k;
}
int rb_scratch_offset(const rb_model* m, int which) { return (m && which >= 0 && which < RB_NOFF) ? m->dev.off[which] : -1; }
static void* rb_balloc(rb_batch* b, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr;
  hipMemset(p, 0, n ? n : 1);
  b->allocs.push_back(p);
  return p;
}
void rb_batch_free(rb_batch* b) {
  if (!b) return;
  { DeviceGuard g(b->device); for (void* p : b->allocs) hipFree(p); }
  delete b;
}
int rb_batch_reset(rb_batch* b) {
  if (!b) return fail("null batch");
  DeviceGuard g(b->device);
  const RbModelDev& d = b->model->dev; RbBatchDev& s = b->dev;
  HIPCHK(hipDeviceSynchronize());
  std::vector<float> q((size_t)s.B * d.nq);
  for (int e = 0; e < s.B; e++) memcpy(q.data() + (size_t)e * d.nq, b->model->qpos0.data(), d.nq * 4);
  HIPCHK(hipMemcpy(s.qpos, q.data(), q.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(s.qvel, 0, (size_t)s.B * d.nv * 4)); HIPCHK(hipMemset(s.ctrl, 0, (size_t)s.B * d.nu * 4));
  HIPCHK(hipMemset(s.pid, 0, (size_t)s.B * 3 * d.nu * 4)); HIPCHK(hipMemset(s.qacc_warmstart, 0, (size_t)s.B * d.nv * 4));
  HIPCHK(hipMemset(s.time, 0, (size_t)s.B * 4)); HIPCHK(hipMemset(s.status, 0, (size_t)s.B * 4)); HIPCHK(hipMemset(s.stats, 0, (size_t)s.B * 16));
  if (d.nmocap) {   // mj_resetData: mocap pose <- the mocap bodies' model pose
    std::vector<float> mc((size_t)s.B * 7 * d.nmocap);
    for (int e = 0; e < s.B; e++) memcpy(mc.data() + (size_t)e * 7 * d.nmocap, b->model->mocap0.data(), 7 * d.nmocap * 4);
    HIPCHK(hipMemcpy(s.mocap, mc.data(), mc.size() * 4, hipMemcpyHostToDevice));
  }
  if (d.nsensordata) HIPCHK(hipMemset(s.sensordata, 0, (size_t)s.B * d.nsensordata * 4));
  return 0;
}
rb_batch* rb_batch_create(const rb_model* m, int B) {
  if (!m || B <= 0) { fail("bad arguments to rb_batch_create"); return nullptr; }
  DeviceGuard g(m->device);
  rb_batch* b = new rb_batch();
  b->model = m; b->device = m->device;
  memset(&b->dev, 0, sizeof b->dev); memset(&b->env, 0, sizeof b->env);
  const RbModelDev& d = m->dev; RbBatchDev& s = b->dev;
  s.B = B;
  s.qpos = (float*)rb_balloc(b, (size_t)B * d.nq * 4); s.qvel = (float*)rb_balloc(b, (size_t)B * d.nv * 4);
  s.ctrl = (float*)rb_balloc(b, (size_t)B * d.nu * 4); s.pid = (float*)rb_balloc(b, (size_t)B * 3 * d.nu * 4);
  s.qacc_warmstart = (float*)rb_balloc(b, (size_t)B * d.nv * 4); s.time = (float*)rb_balloc(b, (size_t)B * 4);
  s.status = (uint32_t*)rb_balloc(b, (size_t)B * 4); s.stats = (float*)rb_balloc(b, (size_t)B * 16);
  s.scratch = (float*)rb_balloc(b, (size_t)B * d.scratch_words * 4);
  s.mocap = (float*)rb_balloc(b, (size_t)B * 7 * d.nmocap * 4); s.eq_data = (float*)rb_balloc(b, (size_t)B * 7 * d.neq * 4);
  s.eq_active = (int*)rb_balloc(b, (size_t)B * d.neq * 4); s.sensordata = (float*)rb_balloc(b, (size_t)B * d.nsensordata * 4);
  if (!s.mocap || !s.eq_data || !s.eq_active || !s.sensordata) { fail("hipMalloc failed"); rb_batch_free(b); return nullptr; }
  {  // equality data and flags are MODEL fields in MuJoCo: they start at the model's values and survive MjSim.reset (the envs write them)
    std::vector<float> ed((size_t)B * 7 * d.neq); std::vector<int> ea((size_t)B * d.neq);
    for (int e = 0; e < B; e++) { for (int k = 0; k < 7 * d.neq; k++) ed[(size_t)e * 7 * d.neq + k] = m->eq_data0[k]; for (int k = 0; k < d.neq; k++) ea[(size_t)e * d.neq + k] = m->eq_active0[k]; }
    if (d.neq && (hipMemcpy(s.eq_data, ed.data(), ed.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(s.eq_active, ea.data(), ea.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) { fail("hipMemcpy failed"); rb_batch_free(b); return nullptr; }
  }
  if (!s.qpos || !s.qvel || !s.ctrl || !s.pid || !s.qacc_warmstart || !s.time || !s.status || !s.stats || !s.scratch) { fail("hipMalloc failed"); rb_batch_free(b); return nullptr; }
  if (d.prm_on) {   // every env's parameter block starts as the model's own values
    std::vector<float> rows((size_t)B * d.prm_words);
    for (int e = 0; e < B; e++) memcpy(rows.data() + (size_t)e * d.prm_words, m->prm_default.data(), (size_t)d.prm_words * 4);
    bool ok = true;   // (one strided copy per env block: set-up time only)
    for (int e = 0; e < B && ok; e++)
      ok = hipMemcpy(s.scratch + (size_t)e * d.scratch_words + d.off[RB_O_PRM], rows.data() + (size_t)e * d.prm_words, (size_t)d.prm_words * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { fail("hipMemcpy failed"); rb_batch_free(b); return nullptr; }
  }
  const_cast<rb_model*>(m)->nbatches++;
  if (rb_batch_reset(b) != 0) { rb_batch_free(b); return nullptr; }
  return b;
}
int rb_batch_set_env(rb_batch* b, int hand_qposadr, int n_hand_jnt, int relative_action, const float* p2c) {
  if (!b || !p2c) return fail("rb_batch_set_env: null argument");
  DeviceGuard g(b->device);
  const size_t n = (size_t)b->model->dev.nu * n_hand_jnt;
  float* dp = (float*)rb_balloc(b, n * 4);
  if (!dp) return fail("hipMalloc failed");
  HIPCHK(hipMemcpy(dp, p2c, n * 4, hipMemcpyHostToDevice));
  b->env.hand_qposadr = hand_qposadr; b->env.n_hand_jnt = n_hand_jnt; b->env.relative_action = relative_action; b->env.pos_to_ctrl = dp;
  b->has_env = 1;
  return 0;
}
int rb_batch_set_action_limits(rb_batch* b, float max_position_change, unsigned ctrl_centre_mask) {
  if (!b) return fail("rb_batch_set_action_limits: null batch");
  if (!b->has_env) return fail("rb_batch_set_action_limits: rb_batch_set_env first");
  if (!(max_position_change >= 0.f)) return fail("rb_batch_set_action_limits: max_position_change must be >= 0 (0 = not capped)");
  if (b->model->dev.nu < 32 && (ctrl_centre_mask >> b->model->dev.nu) != 0) return fail("rb_batch_set_action_limits: ctrl_centre_mask names an actuator the model does not have");
  b->env.max_position_change = max_position_change; b->env.ctrl_centre_mask = ctrl_centre_mask;
  return 0;
}
void* rb_batch_field_ptr(rb_batch* b, int field, int* row_words) {
  if (!b) { fail("null batch"); return nullptr; }
  const RbModelDev& d = b->model->dev; RbBatchDev& s = b->dev;
  void* p = nullptr; int n = 0;
  switch (field) {
    case RG_F_QPOS: p = s.qpos; n = d.nq; break;
    case RG_F_QVEL: p = s.qvel; n = d.nv; break;
    case RG_F_CTRL: p = s.ctrl; n = d.nu; break;
    case RG_F_PID: p = s.pid; n = 3 * d.nu; break;
    case RG_F_WARMSTART: p = s.qacc_warmstart; n = d.nv; break;
    case RG_F_TIME: p = s.time; n = 1; break;
    case RG_F_STATUS: p = s.status; n = 1; break;
    case RG_F_STATS: p = s.stats; n = 4; break;
    case RG_F_DEBUG: p = s.scratch; n = d.scratch_words; break;   // the whole scratch row (stage arrays, rb_scratch_offset)
    case RB_F_MOCAP: p = s.mocap; n = 7 * d.nmocap; break;
    case RB_F_EQ_DATA: p = s.eq_data; n = 7 * d.neq; break;
    case RB_F_EQ_ACTIVE: p = s.eq_active; n = d.neq; break;
    case RB_F_SENSORDATA: p = s.sensordata; n = d.nsensordata; break;
    default: fail("rb_batch_field_ptr: unknown field"); return nullptr;
  }
  if (row_words) *row_words = n;
  return p;
}
#ifdef RG_EMUL
struct EmulRbArgs { const RbModelDev* m; RbLaunch launch; };
static void emul_rb_entry(void* a) { EmulRbArgs* p = (EmulRbArgs*)a; rgb::rb_step_kernel(p->m, p->launch); }
static void emul_rbs_entry(void* a) { EmulRbArgs* p = (EmulRbArgs*)a; rgbs::rb_step_kernel(p->m, p->launch); }
static void emul_rbm_entry(void* a) { EmulRbArgs* p = (EmulRbArgs*)a; rgbm::rb_step_kernel(p->m, p->launch); }
#endif
int rb_batch_step_ex(rb_batch* b, const float* action_dev, const int* active_dev, const int* hold_dev, const int* nticks_dev, int nsubsteps, int nforward_ticks, int flags, void* stream);
int rb_tcp_args_size(void) { return (int)sizeof(rb_tcp_args); }
static thread_local const RbTcpHook* g_tcp_hook = nullptr;
// ---- several batches in ONE launch (rb_multi_begin / rb_multi_launch): between the two calls the rb_batch_step* entry points of this thread validate and record
// their launch instead of issuing it
struct RbCollected { const rb_batch* b; RbLaunch launch; };
static thread_local std::vector<RbCollected>* g_multi = nullptr;
#ifdef RG_EMUL
static void emul_rbs_multi_entry(void* a) { rgbs::rb_step_multi_kernel(*(RbMultiLaunch*)a); }
static void emul_rbm_multi_entry(void* a) { rgbm::rb_step_multi_kernel(*(RbMultiLaunch*)a); }
#endif
static int rb_issue(const rb_batch* b, const RbLaunch& launch, void* stream);
int rb_multi_begin(void) {
  if (g_multi) return fail("rb_multi_begin: already collecting");
  g_multi = new std::vector<RbCollected>();
  return 0;
}
int rb_multi_launch(void* stream) {
  if (!g_multi) return fail("rb_multi_launch without rb_multi_begin");
  std::vector<RbCollected> v; v.swap(*g_multi);
  delete g_multi; g_multi = nullptr;
  if (v.empty()) return 0;
  // (a batch recorded twice would put two workgroups on the same rows of one launch)
  for (size_t a = 0; a < v.size(); a++) for (size_t c = a + 1; c < v.size(); c++) if (v[a].b == v[c].b) return fail("rb_multi_launch: the same batch was recorded twice between rb_multi_begin and rb_multi_launch");
  bool same = v.size() <= RB_MAXMULTI;
  for (const RbCollected& c : v)
    same = same && c.b->dev.B == v[0].b->dev.B && c.b->device == v[0].b->device && c.b->model->config == v[0].b->model->config && (c.b->model->config == 1 || c.b->model->config == 2) &&
           c.b->model->dev.lds_words == v[0].b->model->dev.lds_words;
  if (!same || v.size() == 1) {      // different sizes / configurations (or a single batch): one launch each, in order
    for (const RbCollected& c : v) { const int rc = rb_issue(c.b, c.launch, stream); if (rc) return rc; }
    return 0;
  }
  DeviceGuard g(v[0].b->device);
  RbMultiLaunch ml; memset(&ml, 0, sizeof ml);
  ml.n = (int)v.size(); ml.group_size = v[0].b->dev.B;
  for (size_t k = 0; k < v.size(); k++) { ml.m[k] = v[k].b->model->dev_copy; ml.L[k] = v[k].launch; }
  const int total = ml.n * ml.group_size, cfg = v[0].b->model->config;
#ifdef RG_EMUL
  const size_t arena = 4 * (size_t)v[0].b->model->dev.lds_words + 16;
  if (cfg == 1) emul_launch_n(total, RB_T_SMALL, sizeof(rgbs::RbLds) + arena, emul_rbs_multi_entry, &ml);
  else emul_launch_n(total, RB_T_MEDIUM, sizeof(rgbm::RbLds) + arena, emul_rbm_multi_entry, &ml);
#else
  const size_t arena = v[0].b->model->dev.lds_words ? 4 * (size_t)v[0].b->model->dev.lds_words + 16 : 0;
  if (cfg == 1) hipLaunchKernelGGL(rgbs::rb_step_multi_kernel, dim3(total), dim3(RB_T_SMALL), sizeof(rgbs::RbLds) + arena, (hipStream_t)stream, ml);
  else hipLaunchKernelGGL(rgbm::rb_step_multi_kernel, dim3(total), dim3(RB_T_MEDIUM), sizeof(rgbm::RbLds) + arena, (hipStream_t)stream, ml);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}   // set by rb_batch_step_tcp around its launch
// JointControlledTcpArm.set_position_control as ONE launch of the solver simulation (RbTcpHook, rb_types.h)
int rb_batch_step_tcp(rb_batch* solver, rb_batch* main_batch, const float* action_dev, const int* active_dev, const rb_tcp_args* a, int nsubsteps, int flags, void* stream) {
  if (a && a->self_world) main_batch = solver;      // tcp_solver_mode mocap: one world
  if (!solver || !main_batch || !a || (!action_dev && !a->action_index)) return fail("rb_batch_step_tcp: null argument");
  if ((a->hold != nullptr) != (a->scripted != nullptr)) return fail("rb_batch_step_tcp: hold and scripted go together");
  if (a->action_index && (!a->bins || a->nbins < 1)) return fail("rb_batch_step_tcp: action_index needs bins");
  if ((a->ema_value != nullptr) != (a->ema_t != nullptr)) return fail("rb_batch_step_tcp: ema_value and ema_t go together");
  if (solver->dev.B != main_batch->dev.B || solver->device != main_batch->device) return fail("rb_batch_step_tcp: the two batches must have the same size and device");
  const RbModelDev& ds = solver->model->dev; const RbModelDev& dm = main_batch->model->dev;
  if (ds.nmocap != 1) return fail("rb_batch_step_tcp: the solver model needs exactly one mocap body");
  if (!a->self_world && dm.nu < 6) return fail("rb_batch_step_tcp: the main model's first six actuators must be the arm's (JointControlledArm.set_position_control writes ctrl[:6])");
  if (a->self_world && a->reset_controller_error) return fail("rb_batch_step_tcp: self_world has no second arm to synchronise (reset_controller_error must be 0)");
  if (!a->self_world && (a->nforward_ticks != 0 || a->nticks)) return fail("rb_batch_step_tcp: nforward_ticks / nticks belong to self_world (the solver's own world takes no state-less forwards)");
  if (a->nforward_ticks < 0) return fail("rb_batch_step_tcp: negative nforward_ticks");
  for (int k = 0; k < 6; k++) if (a->arm_qposadr[k] < 0 || a->arm_qposadr[k] >= ds.nq || a->main_arm_qposadr[k] < 0 || a->main_arm_qposadr[k] >= dm.nq) return fail("rb_batch_step_tcp: joint address out of range");
  if (a->main_gripper_actuator < 0 || a->main_gripper_actuator >= dm.nu || a->tcp_body <= 0 || a->tcp_body >= ds.nbody || a->wrist_joint < 0 || a->wrist_joint >= ds.njnt) return fail("rb_batch_step_tcp: id out of range");
  RbTcpHook h; memset(&h, 0, sizeof h);
  h.enabled = 1; h.sync = a->reset_controller_error; h.action = action_dev; h.main_qpos = main_batch->dev.qpos; h.main_ctrl = main_batch->dev.ctrl;
  h.main_nq = dm.nq; h.main_nu = dm.nu;
  for (int k = 0; k < 6; k++) { h.main_arm_q[k] = a->main_arm_qposadr[k]; h.arm_q[k] = a->arm_qposadr[k]; }
  h.main_grip_act = a->main_gripper_actuator; h.tcp_body = a->tcp_body; h.wrist_jnt = a->wrist_joint;
  h.max_position_change = a->max_position_change; h.speed[0] = a->speed_roll; h.speed[1] = a->speed_pitch; h.drift_threshold = a->joint_drift_threshold;
  h.grip_lo = a->gripper_ctrl_lo; h.grip_hi = a->gripper_ctrl_hi;
  h.action_index = a->action_index; h.bins = a->bins; h.nbins = a->nbins; h.ema_alpha = a->ema_alpha; h.ema_value = a->ema_value; h.ema_t = a->ema_t; h.action_out = a->action_out; h.hold = a->hold; h.scripted = a->scripted;
  h.wrist_only = a->wrist_only != 0; h.self_world = a->self_world != 0; h.skip = a->skip;
  g_tcp_hook = &h;
  const int rc = rb_batch_step_ex(solver, nullptr, active_dev, nullptr, a->self_world ? a->nticks : nullptr, nsubsteps, a->self_world ? a->nforward_ticks : 0, flags, stream);
  g_tcp_hook = nullptr;
  return rc;
}
int rb_batch_step(rb_batch* b, const float* action_dev, const int* active_dev, int nsubsteps, int nforward_ticks, int flags, void* stream) {
  return rb_batch_step_ex(b, action_dev, active_dev, nullptr, nullptr, nsubsteps, nforward_ticks, flags, stream);
}
int rb_batch_step_ex(rb_batch* b, const float* action_dev, const int* active_dev, const int* hold_dev, const int* nticks_dev, int nsubsteps, int nforward_ticks, int flags, void* stream) {
  if (!b) return fail("null batch");
  if (action_dev && !b->has_env) return fail("rb_batch_set_env must be called before stepping with actions");
  if (nsubsteps < 0 || nforward_ticks < 0) return fail("rb_batch_step: negative step counts");
  DeviceGuard g(b->device);
  RbBatchDev bt = b->dev;
  bt.action = action_dev; bt.active = active_dev; bt.hold = hold_dev; bt.nticks = nticks_dev;
  RbLaunch launch{b->env, bt, nsubsteps, nforward_ticks, flags, RbTcpHook{}};
  if (g_tcp_hook) { launch.tcp = *g_tcp_hook; }
  if (g_multi) { g_multi->push_back(RbCollected{b, launch}); return 0; }
  return rb_issue(b, launch, stream);
}
static int rb_issue(const rb_batch* b, const RbLaunch& launch, void* stream) {
  const RbBatchDev& bt = launch.bt;
  DeviceGuard g(b->device);
#ifdef RG_EMUL
  EmulRbArgs args{b->model->dev_copy, launch};
  const size_t arena = 4 * (size_t)b->model->dev.lds_words + 16;
  if (b->model->config == 1) emul_launch_n(bt.B, RB_T_SMALL, sizeof(rgbs::RbLds) + arena, emul_rbs_entry, &args);
  else if (b->model->config == 2) emul_launch_n(bt.B, RB_T_MEDIUM, sizeof(rgbm::RbLds) + arena, emul_rbm_entry, &args);
  else emul_launch_n(bt.B, RB_T_LARGE, sizeof(rgb::RbLds), emul_rb_entry, &args);
#else
  const size_t arena = b->model->dev.lds_words ? 4 * (size_t)b->model->dev.lds_words + 16 : 0;   // (+16: the arena starts at the next 16-byte boundary behind RbLds)
  if (b->model->config == 1) hipLaunchKernelGGL(rgbs::rb_step_kernel, dim3(bt.B), dim3(RB_T_SMALL), sizeof(rgbs::RbLds) + arena, (hipStream_t)stream, b->model->dev_copy, launch);
  else if (b->model->config == 2) hipLaunchKernelGGL(rgbm::rb_step_kernel, dim3(bt.B), dim3(RB_T_MEDIUM), sizeof(rgbm::RbLds) + arena, (hipStream_t)stream, b->model->dev_copy, launch);
  else hipLaunchKernelGGL(rgb::rb_step_kernel, dim3(bt.B), dim3(RB_T_LARGE), sizeof(rgb::RbLds), (hipStream_t)stream, b->model->dev_copy, launch);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

#ifdef RG_EMUL
struct EmulRbPostArgs { const RbModelDev* m; RbBatchDev bt; RbPostArgs a; };
static void emul_rb_post_entry(void* p_) { EmulRbPostArgs* p = (EmulRbPostArgs*)p_; rgb::rb_post_step_kernel(p->m, p->bt, p->a); }
struct EmulRbCubeArgs { RbBatchDev bt; int nq, col; const int* tab; const float* ops; int nops; };
static void emul_rb_cube_entry(void* p_) { EmulRbCubeArgs* p = (EmulRbCubeArgs*)p_; rgb::rb_cube_ops_kernel(p->bt, p->nq, p->col, p->tab, p->ops, p->nops); }
#endif
int rb_post_args_size(void) { return (int)sizeof(rb_post_args); }
int rb_env_post_step(rb_batch* b, const rb_post_args* args, void* stream) {
  if (g_multi) return fail("rb_env_post_step: not recordable -- called between rb_multi_begin and rb_multi_launch it would run AHEAD of the recorded physics launches; issue it after rb_multi_launch");
  if (!b || !args) return fail("null argument");
  const rb_post_args& a = *args;
  const RbModelDev& d = b->model->dev;
  if (!a.obs || !a.t || !a.steps || !a.steps_since_last_goal || !a.successes_so_far || !a.goals_so_far || !a.consecutive || !a.prev_dist || !a.prev_valid ||
      !a.is_successful || !a.goal || !a.reward || !a.goal_dist || !a.done || !a.goal_reset || !a.trial_success || !a.sub_goal_ok || !a.env_crash || !a.info_ssl ||
      !a.cube_tab || !a.face_up_quats) return fail("rb_env_post_step: a required array is NULL");
  if (a.obs_dim != 13 + a.n_hand + 15 + 13) return fail("rb_env_post_step: obs_dim does not match the row layout");
  if (a.pipelined) {
    if (!a.phase || !a.tries || !a.nticks_next || !a.hold_next || !a.resetting || !a.episode_started || !a.qpos0 || !a.ctrl_lo || !a.ctrl_hi)
      return fail("rb_env_post_step: pipelined resets need phase, tries, nticks_next, hold_next, resetting, episode_started, qpos0, ctrl_lo, ctrl_hi");
    if (a.num_scramble_steps < 0 || a.num_scramble_steps > 50 || d.nu > 20) return fail("rb_env_post_step: num_scramble_steps / nu exceed RB_RESET_NDRAW's slots");
    if (a.reset_initial_steps < 1 || a.n_random_initial_steps < 1 || a.max_pose_resets < 1) return fail("rb_env_post_step: recipe lengths must be positive");
  }
  if (d.nu > 64) return fail("rb_env_post_step: nu exceeds the workgroup");
  for (int k = 0; k < 6; k++) if (a.face_geom[k] < 0 || a.face_geom[k] >= d.ngeom) return fail("rb_env_post_step: face geom id out of range");
  for (int k = 0; k < 5; k++) if (a.tip_site[k] < 0 || a.tip_site[k] >= d.nsite) return fail("rb_env_post_step: site id out of range");
  for (int k = 0; k < 3; k++) if (a.ref_site[k] < 0 || a.ref_site[k] >= d.nsite) return fail("rb_env_post_step: site id out of range");
  if (a.center_site < 0 || a.center_site >= d.nsite) return fail("rb_env_post_step: site id out of range");
  if (a.cube_block_col < 0 || a.cube_block_col + 66 > d.nq || a.target_block_col < 0 || a.target_block_col + 66 > d.nq || a.cube_quat_col < 0 || a.cube_quat_col + 4 > d.nq ||
      a.cube_pos_col < 0 || a.cube_pos_col + 3 > d.nq || a.hand_col < 0 || a.hand_col + a.n_hand > d.nq) return fail("rb_env_post_step: qpos column out of range");
  DeviceGuard g(b->device);
#ifdef RG_EMUL
  EmulRbPostArgs ea{b->model->dev_copy, b->dev, a};
  emul_launch(b->dev.B, 1024, emul_rb_post_entry, &ea);
#else
  hipLaunchKernelGGL(rgb::rb_post_step_kernel, dim3(b->dev.B), dim3(64), 0, (hipStream_t)stream, b->model->dev_copy, b->dev, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
#ifdef RG_EMUL
struct EmulRaPostArgs { const RbModelDev* m; RbBatchDev bt; RaPostArgs a; };
static void emul_ra_post_entry(void* p_) { EmulRaPostArgs* p = (EmulRaPostArgs*)p_; rgb::ra_post_step_kernel(p->m, p->bt, p->a); }
#endif
int ra_post_args_size(void) { return (int)sizeof(ra_post_args); }
int ra_env_post_step(rb_batch* b, rb_batch* solver, const ra_post_args* args, void* stream) {
  if (g_multi) return fail("ra_env_post_step: not recordable -- called between rb_multi_begin and rb_multi_launch it would run AHEAD of the recorded physics launches; issue it after rb_multi_launch");
  if (!b || !args) return fail("null argument");
  ra_post_args a = *args;
  const RbModelDev& d = b->model->dev;
  if (!a.obs || !a.t || !a.steps || !a.steps_since_last_goal || !a.successes_so_far || !a.consecutive || !a.prev_nsucc || !a.prev_valid || !a.goal || !a.goal_rot || !a.qpos_goal ||
      !a.static_obs || !a.reward || !a.goal_dist || !a.done || !a.goal_reset || !a.trial_success || !a.sub_goal_ok || !a.env_crash || !a.objects_off_table || !a.info_ssl)
    return fail("ra_env_post_step: a required array is NULL");
  if (a.num_objects < 1 || a.num_objects > RA_MAXOBJ) return fail("ra_env_post_step: num_objects out of range");
  if (a.obs_dim != 36 * a.num_objects + 23 + 2 * d.nq) return fail("ra_env_post_step: obs_dim does not match the row layout");
  if (d.nsensordata < 1 || a.force_adr < 0 || a.force_adr + 3 > d.nsensordata || a.torque_adr < 0 || a.torque_adr + 3 > d.nsensordata) return fail("ra_env_post_step: sensor address out of range");
  for (int k = 0; k < a.num_objects; k++) if (a.obj_body[k] <= 0 || a.obj_body[k] >= d.nbody) return fail("ra_env_post_step: object body id out of range");
  if (a.tcp_body <= 0 || a.tcp_body >= d.nbody || a.grip_act < 0 || a.grip_act >= d.nu || a.grip_qposadr < 0 || a.grip_qposadr >= d.nq || a.grip_dofadr < 0 || a.grip_dofadr >= d.nv)
    return fail("ra_env_post_step: robot id out of range");
  for (int k = 0; k < 6; k++) if (a.arm_qposadr[k] < 0 || a.arm_qposadr[k] >= d.nq) return fail("ra_env_post_step: arm joint address out of range");
  for (int k = 0; k < 2; k++) if (a.finger_geom[k] < 0 || a.finger_geom[k] >= d.ngeom) return fail("ra_env_post_step: finger pad geom id out of range");
  if (a.table_plane_geom < 0 || a.table_plane_geom >= d.ngeom) return fail("ra_env_post_step: table plane geom id out of range");
  if (d.ngeom < 64 && (a.gripper_geom_mask >> d.ngeom) != 0) return fail("ra_env_post_step: gripper_geom_mask names a geom the model does not have (bit g = geom g, g < 64)");
  if (solver) {
    if (solver->dev.B != b->dev.B || solver->device != b->device) return fail("ra_env_post_step: the two batches must have the same size and device");
    const RbModelDev& ds = solver->model->dev;
    if (a.solver_grip_qposadr < 0 || a.solver_grip_qposadr >= ds.nq || a.solver_grip_act < 0 || a.solver_grip_act >= ds.nu) return fail("ra_env_post_step: solver gripper id out of range");
    a.solver_qpos = solver->dev.qpos; a.solver_ctrl = solver->dev.ctrl; a.solver_nq = ds.nq; a.solver_nu = ds.nu;
  } else { a.solver_qpos = nullptr; a.solver_ctrl = nullptr; }
  DeviceGuard g(b->device);
#ifdef RG_EMUL
  EmulRaPostArgs ea{b->model->dev_copy, b->dev, a};
  emul_launch(b->dev.B, 1024, emul_ra_post_entry, &ea);
#else
  hipLaunchKernelGGL(rgb::ra_post_step_kernel, dim3(b->dev.B), dim3(64), 0, (hipStream_t)stream, b->model->dev_copy, b->dev, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
#ifdef RG_EMUL
struct EmulRaRecipeArgs { const RbModelDev* m; RbBatchDev bt; const RbModelDev* ms; RbBatchDev sb; RaRecipeArgs a; };
static void emul_ra_recipe_entry(void* p_) { EmulRaRecipeArgs* p = (EmulRaRecipeArgs*)p_; rgb::ra_recipe_kernel(p->m, p->bt, p->ms, p->sb, p->a); }
#endif
int ra_recipe_args_size(void) { return (int)sizeof(ra_recipe_args); }
int ra_env_recipe_step(rb_batch* b, rb_batch* solver, const ra_recipe_args* args, void* stream) {
  if (g_multi) return fail("ra_env_recipe_step: not recordable -- called between rb_multi_begin and rb_multi_launch it would run AHEAD of the recorded physics launches; issue it after rb_multi_launch");
  if (!b || !args) return fail("null argument");
  const ra_recipe_args& a = *args;
  const RbModelDev& d = b->model->dev;
  if (!a.stage || !a.left || !a.yaw || !a.done || !a.goal_reset || !a.hold || !a.hold_ctrl || !a.solver_active || !a.nticks || !a.scripted || !a.frozen || !a.resetting ||
      !a.episode_started || !a.reobserve || !a.ended || !a.stabilised || !a.placement_failed || !a.t || !a.steps || !a.steps_since_last_goal || !a.successes_so_far ||
      !a.consecutive || !a.prev_valid || !a.ema_t || !a.ema_value || !a.action_ema || !a.goal || !a.goal_rot || !a.qpos_goal || !a.static_obs)
    return fail("ra_env_recipe_step: a required array is NULL");
  if (a.num_objects < 1 || a.num_objects > RA_MAXOBJ) return fail("ra_env_recipe_step: num_objects out of range");
  if (a.action_dim < 1 || a.action_dim > 16) return fail("ra_env_recipe_step: action_dim out of range");
  if (d.nu < 6) return fail("ra_env_recipe_step: the model has fewer than the arm's six actuators");
  for (int k = 0; k < a.num_objects; k++) {
    if (a.obj_qposadr[k] < 0 || a.obj_qposadr[k] + 7 > d.nq) return fail("ra_env_recipe_step: object joint address out of range");
    if (!(a.obj_half[k][0] > 0.f) || !(a.obj_half[k][1] > 0.f)) return fail("ra_env_recipe_step: an object's bounding box is empty");
  }
  for (int k = 0; k < 6; k++) if (a.arm_qposadr[k] < 0 || a.arm_qposadr[k] >= d.nq) return fail("ra_env_recipe_step: arm joint address out of range");
  if (!(a.area_size[0] > 0.f) || !(a.area_size[1] > 0.f)) return fail("ra_env_recipe_step: empty placement area");
  if (a.stabilize_steps < 0 || a.n_random_initial_steps < 0 || a.settle_steps < 0) return fail("ra_env_recipe_step: negative step count");
  RbBatchDev sb; memset(&sb, 0, sizeof sb);
  const RbModelDev* ms = nullptr;
  if (solver) {
    if (solver->dev.B != b->dev.B || solver->device != b->device) return fail("ra_env_recipe_step: the two batches must have the same size and device");
    for (int k = 0; k < 6; k++) if (a.solver_arm_qposadr[k] < 0 || a.solver_arm_qposadr[k] >= solver->model->dev.nq) return fail("ra_env_recipe_step: solver arm joint address out of range");
    sb = solver->dev; ms = solver->model->dev_copy;
  }
  DeviceGuard g(b->device);
#ifdef RG_EMUL
  EmulRaRecipeArgs ea{b->model->dev_copy, b->dev, ms, sb, a};
  emul_launch(b->dev.B, 1024, emul_ra_recipe_entry, &ea);
#else
  hipLaunchKernelGGL(rgb::ra_recipe_kernel, dim3(b->dev.B), dim3(64), 0, (hipStream_t)stream, b->model->dev_copy, b->dev, ms, sb, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}
int rb_cube_ops(rb_batch* b, int block_col, const int* cube_tab_dev, const float* ops_dev, int nops, const int* active_dev, void* stream) {
  if (g_multi) return fail("rb_cube_ops: not recordable -- called between rb_multi_begin and rb_multi_launch it would run AHEAD of the recorded physics launches; issue it after rb_multi_launch");
  if (!b || !cube_tab_dev || !ops_dev || nops < 0) return fail("rb_cube_ops: bad argument");
  const RbModelDev& d = b->model->dev;
  if (block_col < 0 || block_col + 66 > d.nq) return fail("rb_cube_ops: block column out of range");
  DeviceGuard g(b->device);
  RbBatchDev bt = b->dev; bt.active = active_dev;
#ifdef RG_EMUL
  EmulRbCubeArgs ea{bt, d.nq, block_col, cube_tab_dev, ops_dev, nops};
  emul_launch(bt.B, 1024, emul_rb_cube_entry, &ea);
#else
  hipLaunchKernelGGL(rgb::rb_cube_ops_kernel, dim3(bt.B), dim3(64), 0, (hipStream_t)stream, bt, d.nq, block_col, cube_tab_dev, ops_dev, nops);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

int rg_sync(void* stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

}  // extern "C"

#if defined(RG_WOODBURY_CHECK) && defined(RG_EMUL)
// test harness only (tests/test_kernel_emul.py): the statistics of the Woodbury self check since the last call, then reset
extern "C" void rg_emul_woodbury_check(float* max_rel, int* solves, int* rows) { *max_rel = rg_wchk_max; *solves = rg_wchk_n; *rows = rg_wchk_rows; rg_wchk_max = 0.f; rg_wchk_n = 0; rg_wchk_rows = 0; }
#endif
