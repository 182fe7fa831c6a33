/*
 * rg_oracle.c — CPU restatement (double precision, scalar, one env) of the arithmetic on
 * robogym's env.step hot path: MuJoCo-2.0 mj_step / mj_forward as driven by
 * SimulationInterface.step (/root/reference/robogym/mujoco/simulation_interface.py:176-189)
 * plus mujoco-py's PID actuator callbacks (cymj.set_pid_control, simulation_interface.py:86-88).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path (robogym_amd/) never does.
 *
 * PARITY UNPINNED vs MuJoCo: the algorithm lives in third-party, un-vendored
 * dependencies of the reference (setup.py:14,16: mujoco-py==2.0.2.13 -> MuJoCo 2.0 binary,
 * mjpid.pyx) that are absent from /root/reference and from this machine, and the
 * reference ships no golden trajectories.  Each stage below restates
 * the published MuJoCo computation pipeline ("Computation" chapter of the MuJoCo
 * documentation; engine_* function names given per stage) and is pinned only by
 * (i) the in-tree pure-numpy pieces of the reference that do run (hand forward
 * kinematics, rotation utilities, hand control tables: tests/golden/), and
 * (ii) physical invariants (energy, momentum, M symmetry/CRB-vs-Jacobian agreement,
 * constraint KKT conditions) checked in tests/.
 */
#ifdef RO_F32
/* single-precision build (make librg_oracle_f32.so: -DRO_F32 -fsingle-precision-constant): every stored value and every
 * operation in float (tgmath dispatches sqrt/sin/... to their f suffix forms).  It exists to answer one question: how
 * far do two runs of THIS algorithm drift apart through precision alone (tests/tools/precision_report.py)? */
#include <tgmath.h>
#undef I
typedef float real;
#else
#include <math.h>
typedef double real;
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#define MINVAL 1e-15
#define MAXCON 200   /* the largest nconmax / njmax the reference sets on this path: cube_env.py:239-242 (full cube) */
#define MAXEFC 2000
#define MAXCONPAIR 8
#define PI 3.14159265358979323846

enum { JNT_FREE = 0, JNT_BALL, JNT_SLIDE, JNT_HINGE };
enum { GEOM_PLANE = 0, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH };
enum { WRAP_JOINT = 1, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, WRAP_CYLINDER };
enum { TRN_JOINT = 0, TRN_TENDON = 3 };
enum { EFC_EQUALITY = 0, EFC_FRICTION_DOF = 1, EFC_FRICTION_TENDON, EFC_LIMIT_JOINT, EFC_LIMIT_TENDON, EFC_CONTACT_PYRAMIDAL, EFC_CONTACT_ELLIPTIC = 7 };
enum { EQ_WELD = 1, EQ_JOINT = 2 };                               /* mjtEq */
enum { SENS_TOUCH = 0, SENS_FORCE = 4, SENS_TORQUE = 5, SENS_JOINTPOS = 8 };   /* mjtSensor (MuJoCo 2.0) */

typedef struct {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nwrap, nmesh, nmeshvert, nexclude, nsensor;
  int neq, nmocap, nsensordata;   /* rearrange models (ur16e/base.xml:52-54, gripper_actuators.xml:3); 0 for the dactyl models */
  const int *eq_type, *eq_obj1id, *eq_obj2id, *eq_active0, *body_mocapid, *sensor_adr, *sensor_dim;
  const real *eq_solref, *eq_solimp, *eq_data0;
  real timestep, gravity[3], tolerance, impratio, ls_tolerance, mpr_tolerance, meaninertia;
  int iterations, cone, ls_iterations, mpr_iterations;
  int nconmax, njmax;
  const int *body_parentid, *body_rootid, *body_weldid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum,
      *body_geomadr, *body_geomnum;
  const real *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_subtreemass,
      *body_invweight0;
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const real *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  const int *dof_bodyid, *dof_jntid, *dof_parentid;
  const real *dof_armature, *dof_damping, *dof_frictionloss, *dof_solref, *dof_solimp, *dof_invweight0;
  const real *qpos0, *qpos_spring;
  const int *geom_type, *geom_bodyid, *geom_dataid, *geom_contype, *geom_conaffinity, *geom_condim;
  const real *geom_size, *geom_rbound, *geom_pos, *geom_quat, *geom_friction, *geom_margin, *geom_gap, *geom_solmix,
      *geom_solref, *geom_solimp;
  const int *site_bodyid, *site_type, *sensor_type, *sensor_objid;
  const real *site_pos, *site_quat, *site_size;
  const int *mesh_vertadr, *mesh_vertnum;
  const float *mesh_vert;
  const int *exclude_signature;
  const int *tendon_adr, *tendon_num, *tendon_limited;
  const real *tendon_range, *tendon_margin, *tendon_stiffness, *tendon_damping, *tendon_frictionloss,
      *tendon_lengthspring, *tendon_solref_lim, *tendon_solimp_lim, *tendon_solref_fri, *tendon_solimp_fri,
      *tendon_invweight0;
  const int *wrap_type, *wrap_objid;
  const real *wrap_prm;
  const int *actuator_trntype, *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited, *actuator_gaintype,
      *actuator_biastype;
  const real *actuator_gear, *actuator_ctrlrange, *actuator_forcerange, *actuator_gainprm, *actuator_biasprm,
      *actuator_user;
  void* blob;
  void* conv[128]; int nconv; /* RO_F32: the blob's float64 arrays converted */
} ro_model;

typedef struct {
  real dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2, efc_address;
} ro_contact;

typedef struct ro_data_s {
  /* state */
  real *qpos, *qvel, *ctrl, *pid, *qacc_warmstart, time;
  real *mocap_pos, *mocap_quat;   /* [nmocap][3], [nmocap][4]: mjData.mocap_pos / mocap_quat */
  real* eq_data; int* eq_active;  /* [neq][7], [neq]: run-time copies (gym's reset_mocap_welds writes eq_data; the envs toggle eq_active) */
  int ne;                         /* equality rows (they come first, as in mj_makeConstraint) */
  real *cacc, *cfrc_int, *cfrc_ext;   /* [nbody][6]: mj_rnePostConstraint, for the force / torque sensors */
  real* xfrc_applied; /* [nbody][6] force, torque in world coordinates, applied at the body's com (mjData.xfrc_applied) */
  /* position stage */
  real *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  real *subtree_com, *cinert, *cdof, *crb;
  real *ten_length, *ten_J, *actuator_length, *actuator_moment;
  real *qM, *qL; /* dense nv*nv inertia and its Cholesky factor (lower) */
  int ncon, nefc, nf, nl;
  ro_contact contact[MAXCON];
  real *efc_J, *efc_pos, *efc_margin, *efc_frictionloss, *efc_diagApprox, *efc_R, *efc_D, *efc_KBIP, *efc_vel,
      *efc_aref, *efc_force;
  int *efc_type, *efc_id;
  /* velocity / force stage */
  real *ten_velocity, *actuator_velocity, *cvel, *cdof_dot, *qfrc_passive, *qfrc_bias, *actuator_force,
      *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
  real* sensordata; /* [nsensor] */
  /* diagnostics */
  int solver_iter, warn_contact_full, warn_efc_full, warn_bad;
  long stat_ncon, stat_nefc, stat_iter, stat_steps, stat_mpr_calls, stat_mpr_iter;
} ro_data;

/* ------------------------------------------------------------------------------------------ small math */
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void copy3(real* r, const real* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void zero3(real* r) { r[0] = r[1] = r[2] = 0; }
static inline void sub3(real* r, const real* a, const real* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void add3(real* r, const real* a, const real* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void addscl3(real* r, const real* a, real s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static inline void scl3(real* r, const real* a, real s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static inline real norm3(const real* a) { return sqrt(dot3(a, a)); }
static inline real normalize3(real* a) {
  real n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = a[2] = 0; return n; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
/* r = M v, M row-major 3x3 */
static inline void mulmat3(real* r, const real* M, const real* v) {
  real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2],
         z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmatT3(real* r, const real* M, const real* v) {
  real x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2], y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2],
         z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulquat(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void normalize4(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat2mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void axisangle2quat(real* q, const real* axis, real angle) {
  real s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static inline real clampd(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------------------------------ model blob */
typedef struct { char name[40]; uint32_t dtype, count; uint64_t offset; } blob_entry;

static const void* blob_find(const void* blob, const char* name, uint32_t* count, int required) {
  const char* b = (const char*)blob;
  uint32_t n = *(const uint32_t*)(b + 8);
  const blob_entry* e = (const blob_entry*)(b + 16);
  for (uint32_t i = 0; i < n; i++)
    if (strncmp(e[i].name, name, 40) == 0) {
      if (count) *count = e[i].count;
      return b + e[i].offset;
    }
  if (required) { fprintf(stderr, "rg_oracle: model blob lacks '%s'\n", name); abort(); }
  return NULL;
}
#ifdef RO_F32
static const real* blob_f64_as_real(ro_model* m, const char* name) {
  uint32_t n = 0;
  const double* src = (const double*)blob_find(m->blob, name, &n, 1);
  real* dst = (real*)malloc((n ? n : 1) * sizeof(real));
  for (uint32_t i = 0; i < n; i++) dst[i] = (real)src[i];
  if (m->nconv < 128) m->conv[m->nconv++] = dst;
  return dst;
}
#define F64(field) m->field = blob_f64_as_real(m, #field)
#else
#define F64(field) m->field = (const real*)blob_find(m->blob, #field, NULL, 1)
#endif
#define I32(field) m->field = (const int*)blob_find(m->blob, #field, NULL, 1)

ro_model* ro_model_load(const void* blob_in, size_t nbytes) {
  if (nbytes < 16 || memcmp(blob_in, "RGMODEL1", 8) != 0) return NULL;
  ro_model* m = (ro_model*)calloc(1, sizeof(ro_model));
  m->blob = malloc(nbytes);
  memcpy(m->blob, blob_in, nbytes);
  const int* dims = (const int*)blob_find(m->blob, "dims", NULL, 1);
  m->nq = dims[0]; m->nv = dims[1]; m->nu = dims[2]; m->nbody = dims[3]; m->njnt = dims[4]; m->ngeom = dims[5];
  m->nsite = dims[6]; m->ntendon = dims[7]; m->nwrap = dims[8]; m->nmesh = dims[9]; m->nmeshvert = dims[10];
  m->nexclude = dims[11]; m->nsensor = dims[12];
  m->timestep = (real)*(const double*)blob_find(m->blob, "opt_timestep", NULL, 1);
  { const double* g = (const double*)blob_find(m->blob, "opt_gravity", NULL, 1); for (int k = 0; k < 3; k++) m->gravity[k] = (real)g[k]; }
  m->tolerance = (real)*(const double*)blob_find(m->blob, "opt_tolerance", NULL, 1);
  m->impratio = (real)*(const double*)blob_find(m->blob, "opt_impratio", NULL, 1);
  m->ls_tolerance = (real)*(const double*)blob_find(m->blob, "opt_ls_tolerance", NULL, 1);
  m->mpr_tolerance = (real)*(const double*)blob_find(m->blob, "opt_mpr_tolerance", NULL, 1);
  m->meaninertia = (real)*(const double*)blob_find(m->blob, "stat_meaninertia", NULL, 1);
  const int* oi = (const int*)blob_find(m->blob, "opt_int", NULL, 1);
  m->iterations = oi[0]; m->cone = oi[1]; m->ls_iterations = oi[2]; m->mpr_iterations = oi[3];
  const int* si = (const int*)blob_find(m->blob, "size_int", NULL, 1);
  m->njmax = si[0] > 0 && si[0] <= MAXEFC ? si[0] : MAXEFC;
  m->nconmax = si[1] > 0 && si[1] <= MAXCON ? si[1] : MAXCON;
  I32(body_parentid); I32(body_rootid); I32(body_weldid); I32(body_jntadr); I32(body_jntnum); I32(body_dofadr);
  I32(body_dofnum); I32(body_geomadr); I32(body_geomnum);
  F64(body_pos); F64(body_quat); F64(body_ipos); F64(body_iquat); F64(body_mass); F64(body_inertia);
  F64(body_subtreemass); F64(body_invweight0);
  I32(jnt_type); I32(jnt_qposadr); I32(jnt_dofadr); I32(jnt_bodyid); I32(jnt_limited);
  F64(jnt_pos); F64(jnt_axis); F64(jnt_stiffness); F64(jnt_range); F64(jnt_margin); F64(jnt_solref); F64(jnt_solimp);
  I32(dof_bodyid); I32(dof_jntid); I32(dof_parentid);
  F64(dof_armature); F64(dof_damping); F64(dof_frictionloss); F64(dof_solref); F64(dof_solimp); F64(dof_invweight0);
  F64(qpos0); F64(qpos_spring);
  I32(geom_type); I32(geom_bodyid); I32(geom_dataid); I32(geom_contype); I32(geom_conaffinity); I32(geom_condim);
  F64(geom_size); F64(geom_rbound); F64(geom_pos); F64(geom_quat); F64(geom_friction); F64(geom_margin); F64(geom_gap);
  F64(geom_solmix); F64(geom_solref); F64(geom_solimp);
  I32(site_bodyid); F64(site_pos); F64(site_quat); I32(site_type); F64(site_size); I32(sensor_type); I32(sensor_objid);
  I32(mesh_vertadr); I32(mesh_vertnum);
  m->mesh_vert = (const float*)blob_find(m->blob, "mesh_vert", NULL, 1);
  I32(exclude_signature);
  I32(tendon_adr); I32(tendon_num); I32(tendon_limited);
  F64(tendon_range); F64(tendon_margin); F64(tendon_stiffness); F64(tendon_damping); F64(tendon_frictionloss);
  F64(tendon_lengthspring); F64(tendon_solref_lim); F64(tendon_solimp_lim); F64(tendon_solref_fri);
  F64(tendon_solimp_fri); F64(tendon_invweight0);
  I32(wrap_type); I32(wrap_objid); F64(wrap_prm);
  I32(actuator_trntype); I32(actuator_trnid); I32(actuator_ctrllimited); I32(actuator_forcelimited);
  I32(actuator_gaintype); I32(actuator_biastype);
  F64(actuator_gear); F64(actuator_ctrlrange); F64(actuator_forcerange); F64(actuator_gainprm); F64(actuator_biasprm);
  F64(actuator_user);
  /* optional sections (absent from model files compiled before the rearrange features existed) */
  { uint32_t n = 0;
    m->eq_type = (const int*)blob_find(m->blob, "eq_type", &n, 0); m->neq = m->eq_type ? (int)n : 0;
    if (m->neq) { I32(eq_obj1id); I32(eq_obj2id); m->eq_active0 = (const int*)blob_find(m->blob, "eq_active", NULL, 1); F64(eq_solref); F64(eq_solimp);
#ifdef RO_F32
      m->eq_data0 = blob_f64_as_real(m, "eq_data");
#else
      m->eq_data0 = (const real*)blob_find(m->blob, "eq_data", NULL, 1);
#endif
    }
    m->body_mocapid = (const int*)blob_find(m->blob, "body_mocapid", NULL, 0);
    const int* nm = (const int*)blob_find(m->blob, "nmocap", NULL, 0); m->nmocap = (nm && m->body_mocapid) ? nm[0] : 0;
    m->sensor_adr = (const int*)blob_find(m->blob, "sensor_adr", NULL, 0); m->sensor_dim = (const int*)blob_find(m->blob, "sensor_dim", NULL, 0);
    m->nsensordata = m->nsensor;
    if (m->sensor_adr && m->sensor_dim && m->nsensor > 0) m->nsensordata = m->sensor_adr[m->nsensor - 1] + m->sensor_dim[m->nsensor - 1];
  }
  return m;
}
void ro_model_free(ro_model* m) { if (m) { for (int i = 0; i < m->nconv; i++) free(m->conv[i]); free(m->blob); free(m); } }

static real* dalloc(size_t n) { return (real*)calloc(n ? n : 1, sizeof(real)); }
/* the parts of mjData / mjModel that mj_resetData and a fresh model restore: mocap pose <- body pose, equality data and flags */
static void ro_reset_model_state(const ro_model* m, ro_data* d);

ro_data* ro_data_new(const ro_model* m) {
  ro_data* d = (ro_data*)calloc(1, sizeof(ro_data));
  int nv = m->nv, nb = m->nbody;
  d->qpos = dalloc(m->nq); d->qvel = dalloc(nv); d->ctrl = dalloc(m->nu); d->pid = dalloc(3 * m->nu);
  d->qacc_warmstart = dalloc(nv); d->xfrc_applied = dalloc(6 * m->nbody);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb); d->xipos = dalloc(3 * nb);
  d->ximat = dalloc(9 * nb); d->xanchor = dalloc(3 * m->njnt); d->xaxis = dalloc(3 * m->njnt);
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom);
  d->site_xpos = dalloc(3 * m->nsite); d->site_xmat = dalloc(9 * m->nsite);
  d->subtree_com = dalloc(3 * nb); d->cinert = dalloc(10 * nb); d->cdof = dalloc(6 * nv); d->crb = dalloc(10 * nb);
  d->ten_length = dalloc(m->ntendon); d->ten_J = dalloc((size_t)m->ntendon * nv);
  d->actuator_length = dalloc(m->nu); d->actuator_moment = dalloc((size_t)m->nu * nv);
  d->qM = dalloc((size_t)nv * nv); d->qL = dalloc((size_t)nv * nv);
  d->efc_J = dalloc((size_t)MAXEFC * nv); d->efc_pos = dalloc(MAXEFC); d->efc_margin = dalloc(MAXEFC);
  d->efc_frictionloss = dalloc(MAXEFC); d->efc_diagApprox = dalloc(MAXEFC); d->efc_R = dalloc(MAXEFC);
  d->efc_D = dalloc(MAXEFC); d->efc_KBIP = dalloc(4 * MAXEFC); d->efc_vel = dalloc(MAXEFC);
  d->efc_aref = dalloc(MAXEFC); d->efc_force = dalloc(MAXEFC);
  d->efc_type = (int*)calloc(MAXEFC, sizeof(int)); d->efc_id = (int*)calloc(MAXEFC, sizeof(int));
  d->ten_velocity = dalloc(m->ntendon); d->actuator_velocity = dalloc(m->nu); d->cvel = dalloc(6 * nb);
  d->cdof_dot = dalloc(6 * nv); d->qfrc_passive = dalloc(nv); d->qfrc_bias = dalloc(nv);
  d->actuator_force = dalloc(m->nu); d->qfrc_actuator = dalloc(nv); d->qfrc_smooth = dalloc(nv);
  d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qacc = dalloc(nv);
  d->sensordata = dalloc(m->nsensordata > 0 ? m->nsensordata : 1);
  d->mocap_pos = dalloc(3 * m->nmocap); d->mocap_quat = dalloc(4 * m->nmocap);
  d->eq_data = dalloc(7 * m->neq); d->eq_active = (int*)calloc(m->neq ? m->neq : 1, sizeof(int));
  d->cacc = dalloc(6 * nb); d->cfrc_int = dalloc(6 * nb); d->cfrc_ext = dalloc(6 * nb);
  memcpy(d->qpos, m->qpos0, m->nq * sizeof(real));
  ro_reset_model_state(m, d);
  return d;
}
void ro_data_free(ro_data* d) {
  if (!d) return;
  real** p[] = {&d->qpos, &d->qvel, &d->ctrl, &d->pid, &d->qacc_warmstart, &d->xfrc_applied, &d->xpos, &d->xquat, &d->xmat, &d->xipos,
                  &d->ximat, &d->xanchor, &d->xaxis, &d->geom_xpos, &d->geom_xmat, &d->site_xpos, &d->site_xmat,
                  &d->subtree_com, &d->cinert, &d->cdof, &d->crb, &d->ten_length, &d->ten_J, &d->actuator_length,
                  &d->actuator_moment, &d->qM, &d->qL, &d->efc_J, &d->efc_pos, &d->efc_margin, &d->efc_frictionloss,
                  &d->efc_diagApprox, &d->efc_R, &d->efc_D, &d->efc_KBIP, &d->efc_vel, &d->efc_aref, &d->efc_force,
                  &d->ten_velocity, &d->actuator_velocity, &d->cvel, &d->cdof_dot, &d->qfrc_passive, &d->qfrc_bias,
                  &d->actuator_force, &d->qfrc_actuator, &d->qfrc_smooth, &d->qacc_smooth, &d->qfrc_constraint, &d->qacc,
                  &d->mocap_pos, &d->mocap_quat, &d->eq_data, &d->cacc, &d->cfrc_int, &d->cfrc_ext, &d->sensordata};
  for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); i++) free(*p[i]);
  free(d->efc_type); free(d->efc_id); free(d->eq_active); free(d);
}
static void ro_reset_model_state(const ro_model* m, ro_data* d) {
  for (int b = 0; b < m->nbody && m->nmocap; b++) {
    int k = m->body_mocapid[b];
    if (k < 0) continue;
    copy3(d->mocap_pos + 3 * k, m->body_pos + 3 * b); memcpy(d->mocap_quat + 4 * k, m->body_quat + 4 * b, 4 * sizeof(real));
  }
  for (int e = 0; e < m->neq; e++) { d->eq_active[e] = m->eq_active0[e]; memcpy(d->eq_data + 7 * e, m->eq_data0 + 7 * e, 7 * sizeof(real)); }
}

/* mj_resetData: qpos <- qpos0, everything else zero (mujoco-py MjSim.reset) */
void ro_reset(const ro_model* m, ro_data* d) {
  memcpy(d->qpos, m->qpos0, m->nq * sizeof(real));
  memset(d->qvel, 0, m->nv * sizeof(real)); memset(d->ctrl, 0, m->nu * sizeof(real));
  memset(d->pid, 0, 3 * m->nu * sizeof(real)); memset(d->qacc_warmstart, 0, m->nv * sizeof(real));
  memset(d->xfrc_applied, 0, 6 * m->nbody * sizeof(real));
  d->time = 0; d->warn_bad = d->warn_contact_full = d->warn_efc_full = 0;
  /* mj_resetData restores the mocap pose; eq_data / eq_active are model fields and survive a reset */
  for (int b = 0; b < m->nbody && m->nmocap; b++) {
    int k = m->body_mocapid[b];
    if (k >= 0) { copy3(d->mocap_pos + 3 * k, m->body_pos + 3 * b); memcpy(d->mocap_quat + 4 * k, m->body_quat + 4 * b, 4 * sizeof(real)); }
  }
}

/* ------------------------------------------------------------------------------------------ kinematics
 * engine_core_smooth.c: mj_kinematics */
static void ro_kinematics(const ro_model* m, ro_data* d) {
  zero3(d->xpos); d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat); zero3(d->xipos); quat2mat(d->ximat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    real pos[3], quat[4], tmp[3], mat[9];
    mulmat3(tmp, d->xmat + 9 * p, m->body_pos + 3 * b);
    add3(pos, d->xpos + 3 * p, tmp);
    mulquat(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
    if (m->nmocap && m->body_mocapid[b] >= 0) {   /* mj_kinematics: a mocap body takes its pose from mjData.mocap_pos / mocap_quat (normalised) */
      copy3(pos, d->mocap_pos + 3 * m->body_mocapid[b]); memcpy(quat, d->mocap_quat + 4 * m->body_mocapid[b], 4 * sizeof(real));
    }
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, qa = m->jnt_qposadr[j], t = m->jnt_type[j];
      if (t == JNT_FREE) {
        copy3(pos, d->qpos + qa);
        memcpy(quat, d->qpos + qa + 3, 4 * sizeof(real)); normalize4(quat);
        copy3(d->xanchor + 3 * j, pos); d->xaxis[3 * j] = 0; d->xaxis[3 * j + 1] = 0; d->xaxis[3 * j + 2] = 1;
        continue;
      }
      quat2mat(mat, quat);
      mulmat3(tmp, mat, m->jnt_pos + 3 * j); add3(d->xanchor + 3 * j, pos, tmp);
      mulmat3(d->xaxis + 3 * j, mat, m->jnt_axis + 3 * j);
      if (t == JNT_SLIDE) {
        addscl3(pos, d->xaxis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
      } else {
        real ql[4], qn[4];
        if (t == JNT_BALL) { memcpy(ql, d->qpos + qa, 4 * sizeof(real)); normalize4(ql); }
        else axisangle2quat(ql, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
        mulquat(qn, quat, ql); memcpy(quat, qn, sizeof qn);
        /* keep the anchor fixed: xpos = xanchor - R_new * jnt_pos */
        quat2mat(mat, quat); mulmat3(tmp, mat, m->jnt_pos + 3 * j); sub3(pos, d->xanchor + 3 * j, tmp);
      }
    }
    normalize4(quat);
    copy3(d->xpos + 3 * b, pos); memcpy(d->xquat + 4 * b, quat, 4 * sizeof(real));
    quat2mat(d->xmat + 9 * b, quat);
    mulmat3(tmp, d->xmat + 9 * b, m->body_ipos + 3 * b); add3(d->xipos + 3 * b, pos, tmp);
    real iq[4]; mulquat(iq, quat, m->body_iquat + 4 * b); quat2mat(d->ximat + 9 * b, iq);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    real tmp[3], q[4];
    mulmat3(tmp, d->xmat + 9 * b, m->geom_pos + 3 * g); add3(d->geom_xpos + 3 * g, d->xpos + 3 * b, tmp);
    mulquat(q, d->xquat + 4 * b, m->geom_quat + 4 * g); quat2mat(d->geom_xmat + 9 * g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    real tmp[3], q[4];
    mulmat3(tmp, d->xmat + 9 * b, m->site_pos + 3 * s); add3(d->site_xpos + 3 * s, d->xpos + 3 * b, tmp);
    mulquat(q, d->xquat + 4 * b, m->site_quat + 4 * s); quat2mat(d->site_xmat + 9 * s, q);
  }
}

/* engine_core_smooth.c: mj_comPos — subtree COMs, body inertias and motion axes in the com-based frame */
static void ro_com_pos(const ro_model* m, ro_data* d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) scl3(d->subtree_com + 3 * b, d->xipos + 3 * b, m->body_mass[b]);
  for (int b = nb - 1; b > 0; b--) add3(d->subtree_com + 3 * m->body_parentid[b], d->subtree_com + 3 * m->body_parentid[b], d->subtree_com + 3 * b);
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < MINVAL) copy3(d->subtree_com + 3 * b, d->xipos + 3 * b);
    else scl3(d->subtree_com + 3 * b, d->subtree_com + 3 * b, 1.0 / m->body_subtreemass[b]);
  }
  memset(d->cinert, 0, 10 * sizeof(real));
  for (int b = 1; b < nb; b++) {
    real dif[3], *ci = d->cinert + 10 * b, tmp[9], I[9];
    const real *R = d->ximat + 9 * b, *in = m->body_inertia + 3 * b;
    real mass = m->body_mass[b];
    sub3(dif, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
    /* I = R diag(in) R^T */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[3 * i + j] = R[3 * i + j] * in[j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[3 * i + j] = tmp[3 * i] * R[3 * j] + tmp[3 * i + 1] * R[3 * j + 1] + tmp[3 * i + 2] * R[3 * j + 2];
    real d2 = dot3(dif, dif);
    ci[0] = I[0] + mass * (d2 - dif[0] * dif[0]); ci[1] = I[4] + mass * (d2 - dif[1] * dif[1]); ci[2] = I[8] + mass * (d2 - dif[2] * dif[2]);
    ci[3] = I[1] - mass * dif[0] * dif[1]; ci[4] = I[2] - mass * dif[0] * dif[2]; ci[5] = I[5] - mass * dif[1] * dif[2];
    ci[6] = mass * dif[0]; ci[7] = mass * dif[1]; ci[8] = mass * dif[2]; ci[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j], t = m->jnt_type[j];
    real off[3];
    sub3(off, d->subtree_com + 3 * m->body_rootid[b], d->xanchor + 3 * j);
    const real* R = d->xmat + 9 * b;
    if (t == JNT_FREE || t == JNT_BALL) {
      if (t == JNT_FREE) {
        for (int k = 0; k < 3; k++) { real* c = d->cdof + 6 * (da + k); memset(c, 0, 6 * sizeof(real)); c[3 + k] = 1; }
        da += 3;
      }
      for (int k = 0; k < 3; k++) {
        real ax[3] = {R[k], R[3 + k], R[6 + k]}, *c = d->cdof + 6 * (da + k);
        copy3(c, ax); cross3(c + 3, ax, off);
      }
    } else if (t == JNT_SLIDE) {
      real* c = d->cdof + 6 * da; zero3(c); copy3(c + 3, d->xaxis + 3 * j);
    } else {
      real* c = d->cdof + 6 * da; copy3(c, d->xaxis + 3 * j); cross3(c + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* y = cinert * v  (mju_mulInertVec) */
static void mul_inert_vec(real* r, const real* i, const real* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(real* r, const real* vel, const real* v) {
  real a[3], b[3];
  cross3(r, vel, v); cross3(a, vel, v + 3); cross3(b, vel + 3, v); add3(r + 3, a, b);
}
static void cross_force(real* r, const real* vel, const real* f) {
  real a[3], b[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); add3(r, a, b); cross3(r + 3, vel, f + 3);
}

/* translational/rotational Jacobian of a world point attached to `body` (mj_jac); jp, jr: 3 x nv or NULL */
static void ro_jac(const ro_model* m, const ro_data* d, real* jp, real* jr, const real* point, int body) {
  int nv = m->nv;
  if (jp) memset(jp, 0, 3 * nv * sizeof(real));
  if (jr) memset(jr, 0, 3 * nv * sizeof(real));
  if (body <= 0) return;
  real off[3];
  sub3(off, point, d->subtree_com + 3 * m->body_rootid[body]);
  /* last dof of the nearest body (self or ancestor) that has dofs */
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b <= 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    const real* c = d->cdof + 6 * i;
    if (jr) { jr[i] = c[0]; jr[nv + i] = c[1]; jr[2 * nv + i] = c[2]; }
    if (jp) {
      real t[3]; cross3(t, c, off);
      jp[i] = c[3] + t[0]; jp[nv + i] = c[4] + t[1]; jp[2 * nv + i] = c[5] + t[2];
    }
  }
}

/* ------------------------------------------------------------------------------------------ tendons
 * engine_util_misc.c: mju_wrap / wrap_circle / is_intersect; engine_core_smooth.c: mj_tendon */
static int is_intersect(const real* p1, const real* p2, const real* p3, const real* p4) {
  real det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1]);
  if (fabs(det) < MINVAL) return 0;
  real a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det;
  real b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det;
  return a >= 0 && a <= 1 && b >= 0 && b <= 1;
}
static real wrap_circle(real* pnt, const real* d, const real* sd, real rad) {
  real sq0 = d[0] * d[0] + d[1] * d[1], sq1 = d[2] * d[2] + d[3] * d[3], sqr = rad * rad;
  real dif[2] = {d[2] - d[0], d[3] - d[1]}, dd = dif[0] * dif[0] + dif[1] * dif[1];
  if (sq0 < sqr || sq1 < sqr || rad < MINVAL || dd < MINVAL) return -1;
  real a = clampd(-(dif[0] * d[0] + dif[1] * d[1]) / dd, 0, 1);
  real nr[2] = {a * dif[0] + d[0], a * dif[1] + d[1]};
  if (nr[0] * nr[0] + nr[1] * nr[1] > sqr && (!sd || sd[0] * nr[0] + sd[1] * nr[1] >= 0)) return -1;
  real sol[2][4], good[2];
  for (int i = 0; i < 2; i++) {
    real sgn = i == 0 ? 1 : -1, r0 = sqrt(sq0 - sqr), r1 = sqrt(sq1 - sqr);
    sol[i][0] = (d[0] * sqr + sgn * rad * d[1] * r0) / sq0; sol[i][1] = (d[1] * sqr - sgn * rad * d[0] * r0) / sq0;
    sol[i][2] = (d[2] * sqr - sgn * rad * d[3] * r1) / sq1; sol[i][3] = (d[3] * sqr + sgn * rad * d[2] * r1) / sq1;
    if (sd) {
      real mx = sol[i][0] + sol[i][2], my = sol[i][1] + sol[i][3], n = sqrt(mx * mx + my * my);
      if (n < MINVAL) n = MINVAL;
      good[i] = (mx * sd[0] + my * sd[1]) / n;
    } else {
      real tx = sol[i][0] - sol[i][2], ty = sol[i][1] - sol[i][3];
      good[i] = -(tx * tx + ty * ty);
    }
    if (is_intersect(d, sol[i], d + 2, sol[i] + 2)) good[i] = -10000;
  }
  int k = good[0] > good[1] ? 0 : 1;
  memcpy(pnt, sol[k], 4 * sizeof(real));
  if (is_intersect(d, pnt, d + 2, pnt + 2)) return -1;
  return rad * acos(clampd((pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqr, -1, 1));
}
/* returns wrap length (<0: straight) and the two tangent points wpnt[6] in world frame */
static real ro_wrap(real* wpnt, const real* x0, const real* x1, const real* gpos, const real* gmat,
                      real radius, int type, const real* side) {
  real p0[3], p1[3], t[3];
  sub3(t, x0, gpos); mulmatT3(p0, gmat, t); sub3(t, x1, gpos); mulmatT3(p1, gmat, t);
  if (norm3(p0) < MINVAL || norm3(p1) < MINVAL) return -1;
  real ax0[3], ax1[3];
  if (type == WRAP_SPHERE) {
    real nrm[3];
    copy3(ax0, p0); normalize3(ax0); cross3(nrm, p0, p1);
    if (norm3(nrm) < MINVAL) { real e[3] = {fabs(ax0[0]) < 0.9, fabs(ax0[0]) >= 0.9, 0}; cross3(nrm, ax0, e); }
    normalize3(nrm); cross3(ax1, nrm, ax0); normalize3(ax1);
  } else { ax0[0] = 1; ax0[1] = ax0[2] = 0; ax1[0] = ax1[2] = 0; ax1[1] = 1; }
  real dd[4] = {dot3(p0, ax0), dot3(p0, ax1), dot3(p1, ax0), dot3(p1, ax1)}, sd[2], *psd = NULL;
  if (side) {
    real s[3]; sub3(t, side, gpos); mulmatT3(s, gmat, t);
    sd[0] = dot3(s, ax0); sd[1] = dot3(s, ax1);
    real n = sqrt(sd[0] * sd[0] + sd[1] * sd[1]);
    if (n < radius) { fprintf(stderr, "rg_oracle: inside tendon wrap not supported\n"); abort(); }
    sd[0] *= radius / n; sd[1] *= radius / n; psd = sd;
  }
  real pnt[4], wlen = wrap_circle(pnt, dd, psd, radius);
  if (wlen < 0) return -1;
  real r0[3], r1[3];
  for (int k = 0; k < 3; k++) { r0[k] = ax0[k] * pnt[0] + ax1[k] * pnt[1]; r1[k] = ax0[k] * pnt[2] + ax1[k] * pnt[3]; }
  if (type == WRAP_CYLINDER) {
    real L0 = hypot(dd[0] - pnt[0], dd[1] - pnt[1]), L1 = hypot(dd[2] - pnt[2], dd[3] - pnt[3]);
    r0[2] = p0[2] + (p1[2] - p0[2]) * L0 / (L0 + wlen + L1);
    r1[2] = p0[2] + (p1[2] - p0[2]) * (L0 + wlen) / (L0 + wlen + L1);
    wlen = hypot(wlen, r1[2] - r0[2]);
  }
  mulmat3(t, gmat, r0); add3(wpnt, t, gpos); mulmat3(t, gmat, r1); add3(wpnt + 3, t, gpos);
  return wlen;
}

static void ro_tendon(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  real* ja = (real*)malloc(6 * nv * sizeof(real));
  real* jb = ja + 3 * nv;
  memset(d->ten_J, 0, (size_t)m->ntendon * nv * sizeof(real));
  for (int t = 0; t < m->ntendon; t++) {
    int adr = m->tendon_adr[t], num = m->tendon_num[t];
    real L = 0, *J = d->ten_J + (size_t)t * nv;
    if (m->wrap_type[adr] == WRAP_JOINT) {
      for (int w = adr; w < adr + num; w++) {
        int j = m->wrap_objid[w];
        L += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]];
        J[m->jnt_dofadr[j]] = m->wrap_prm[w];
      }
      d->ten_length[t] = L;
      continue;
    }
    real divisor = 1;
    int w = adr;
    while (w < adr + num - 1) {
      int t0 = m->wrap_type[w], t1 = m->wrap_type[w + 1];
      if (t0 == WRAP_PULLEY || t1 == WRAP_PULLEY) { if (t0 == WRAP_PULLEY) divisor = m->wrap_prm[w]; w++; continue; }
      real pnt[12]; int body[4], cnt;
      int s0 = m->wrap_objid[w];
      copy3(pnt, d->site_xpos + 3 * s0); body[0] = m->site_bodyid[s0];
      real wlen = -1;
      if (t1 == WRAP_SPHERE || t1 == WRAP_CYLINDER) {
        int g = m->wrap_objid[w + 1], s1 = m->wrap_objid[w + 2], sid = (int)lround(m->wrap_prm[w + 1]);
        wlen = ro_wrap(pnt + 3, pnt, d->site_xpos + 3 * s1, d->geom_xpos + 3 * g, d->geom_xmat + 9 * g,
                       m->geom_size[3 * g], t1, sid >= 0 ? d->site_xpos + 3 * sid : NULL);
        if (wlen < 0) { copy3(pnt + 3, d->site_xpos + 3 * s1); body[1] = m->site_bodyid[s1]; cnt = 2; }
        else { copy3(pnt + 9, d->site_xpos + 3 * s1); body[1] = body[2] = m->geom_bodyid[g]; body[3] = m->site_bodyid[s1]; cnt = 4; }
        w += 2;
      } else {
        int s1 = m->wrap_objid[w + 1];
        copy3(pnt + 3, d->site_xpos + 3 * s1); body[1] = m->site_bodyid[s1]; cnt = 2;
        w += 1;
      }
      if (wlen >= 0) L += wlen / divisor;
      for (int k = 0; k < cnt - 1; k++) {
        if (cnt == 4 && k == 1) continue; /* the arc lies on the wrapping geom */
        real dif[3]; sub3(dif, pnt + 3 * (k + 1), pnt + 3 * k);
        real dist = norm3(dif);
        L += dist / divisor;
        if (body[k] != body[k + 1] && dist > MINVAL) {
          scl3(dif, dif, 1 / dist);
          ro_jac(m, d, ja, NULL, pnt + 3 * k, body[k]); ro_jac(m, d, jb, NULL, pnt + 3 * (k + 1), body[k + 1]);
          for (int i = 0; i < nv; i++)
            J[i] += (dif[0] * (jb[i] - ja[i]) + dif[1] * (jb[nv + i] - ja[nv + i]) + dif[2] * (jb[2 * nv + i] - ja[2 * nv + i])) / divisor;
        }
      }
    }
    d->ten_length[t] = L;
  }
  free(ja);
}

/* engine_core_smooth.c: mj_transmission (joint and tendon transmissions, scalar gear) */
static void ro_transmission(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  memset(d->actuator_moment, 0, (size_t)m->nu * nv * sizeof(real));
  for (int i = 0; i < m->nu; i++) {
    real g = m->actuator_gear[i];
    int id = m->actuator_trnid[i];
    if (m->actuator_trntype[i] == TRN_JOINT) {
      d->actuator_length[i] = g * d->qpos[m->jnt_qposadr[id]];
      d->actuator_moment[(size_t)i * nv + m->jnt_dofadr[id]] = g;
    } else {
      d->actuator_length[i] = g * d->ten_length[id];
      for (int k = 0; k < nv; k++) d->actuator_moment[(size_t)i * nv + k] = g * d->ten_J[(size_t)id * nv + k];
    }
  }
}

/* engine_core_smooth.c: mj_crb + mj_factorM (dense Cholesky here; MuJoCo uses sparse L'DL — same solution) */
static int cholesky(real* L, const real* A, int n) {
  memcpy(L, A, (size_t)n * n * sizeof(real));
  for (int j = 0; j < n; j++) {
    real s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (s < MINVAL) return -1;
    s = sqrt(s); L[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      real t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / s;
    }
  }
  return 0;
}
static void chol_solve(const real* L, real* x, int n) { /* in place: x <- A^-1 x */
  for (int i = 0; i < n; i++) { real s = x[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { real s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}
static void ro_crb(const ro_model* m, ro_data* d) {
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crb, d->cinert, 10 * nb * sizeof(real));
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  memset(d->qM, 0, (size_t)nv * nv * sizeof(real));
  for (int i = 0; i < nv; i++) {
    real buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      const real* c = d->cdof + 6 * j;
      real v = c[0] * buf[0] + c[1] * buf[1] + c[2] * buf[2] + c[3] * buf[3] + c[4] * buf[4] + c[5] * buf[5];
      d->qM[i * nv + j] = d->qM[j * nv + i] = v;
    }
    d->qM[i * nv + i] += m->dof_armature[i];
  }
  if (cholesky(d->qL, d->qM, nv) != 0) d->warn_bad |= 1;
}

/* ------------------------------------------------------------------------------------------ collision
 * engine_collision_driver.c (filters), engine_collision_convex.c (mjc_Convex on libccd's MPR),
 * engine_collision_primitive.c (plane cases).  libccd: ccd/mpr.c ccdMPRPenetration. */
typedef struct { const ro_model* m; const ro_data* d; int geom; real margin; } ccd_obj;
typedef struct { real v[3], v1[3], v2[3]; } ccd_support;

static void geom_support(const ccd_obj* o, const real* dir, real* res) {
  const ro_model* m = o->m; int g = o->geom;
  const real* mat = o->d->geom_xmat + 9 * g; const real* pos = o->d->geom_xpos + 3 * g;
  real ld[3], lr[3] = {0, 0, 0};
  mulmatT3(ld, mat, dir);
  const real* sz = m->geom_size + 3 * g;
  switch (m->geom_type[g]) {
    case GEOM_SPHERE: scl3(lr, ld, sz[0]); break;
    case GEOM_CAPSULE: scl3(lr, ld, sz[0]); lr[2] += (ld[2] >= 0 ? 1 : -1) * sz[1]; break;
    case GEOM_ELLIPSOID: {
      real t[3] = {ld[0] * sz[0], ld[1] * sz[1], ld[2] * sz[2]}; real n = norm3(t); if (n < MINVAL) n = MINVAL;
      lr[0] = t[0] * sz[0] / n; lr[1] = t[1] * sz[1] / n; lr[2] = t[2] * sz[2] / n; break; }
    case GEOM_CYLINDER: {
      real n = sqrt(ld[0] * ld[0] + ld[1] * ld[1]);
      if (n > MINVAL) { lr[0] = ld[0] / n * sz[0]; lr[1] = ld[1] / n * sz[0]; }
      lr[2] = (ld[2] >= 0 ? 1 : -1) * sz[1]; break; }
    case GEOM_BOX: for (int k = 0; k < 3; k++) lr[k] = (ld[k] >= 0 ? 1 : -1) * sz[k]; break;
    case GEOM_MESH: {
      int id = m->geom_dataid[g], adr = m->mesh_vertadr[id], n = m->mesh_vertnum[id], best = 0;
      real bv = -1e300;
      for (int i = 0; i < n; i++) {
        const float* v = m->mesh_vert + 3 * (adr + i);
        real s = ld[0] * v[0] + ld[1] * v[1] + ld[2] * v[2];
        if (s > bv) { bv = s; best = i; }
      }
      const float* v = m->mesh_vert + 3 * (adr + best);
      lr[0] = v[0]; lr[1] = v[1]; lr[2] = v[2]; break; }
    default: break;
  }
  addscl3(lr, ld, o->margin);
  mulmat3(res, mat, lr); add3(res, res, pos);
}
static void mpr_support(const ccd_obj* o1, const ccd_obj* o2, const real* dir, ccd_support* s) {
  real nd[3] = {-dir[0], -dir[1], -dir[2]};
  geom_support(o1, dir, s->v1); geom_support(o2, nd, s->v2); sub3(s->v, s->v1, s->v2);
}
#define CCD_EPS 2.220446049250313e-16
static inline int ccd_zero(real x) { return fabs(x) < CCD_EPS; }
static inline int ccd_eq(real a, real b) {
  real ab = fabs(a - b);
  if (ab < CCD_EPS) return 1;
  real aa = fabs(a), bb = fabs(b);
  return ab < CCD_EPS * (bb > aa ? bb : aa);
}
static void portal_dir(const ccd_support* p, real* dir) {
  real a[3], b[3];
  sub3(a, p[2].v, p[1].v); sub3(b, p[3].v, p[1].v); cross3(dir, a, b); normalize3(dir);
}
static int portal_reach_tol(const ccd_support* p, const ccd_support* v4, const real* dir, real tol) {
  real dv1 = dot3(p[1].v, dir), dv2 = dot3(p[2].v, dir), dv3 = dot3(p[3].v, dir), dv4 = dot3(v4->v, dir);
  real d1 = dv4 - dv1, d2 = dv4 - dv2, d3 = dv4 - dv3;
  real mn = d1 < d2 ? d1 : d2; mn = mn < d3 ? mn : d3;
  return ccd_eq(mn, tol) || mn < tol;
}
static void expand_portal(ccd_support* p, const ccd_support* v4) {
  real v4v0[3]; cross3(v4v0, v4->v, p[0].v);
  if (dot3(p[1].v, v4v0) > 0) { if (dot3(p[2].v, v4v0) > 0) p[1] = *v4; else p[3] = *v4; }
  else { if (dot3(p[3].v, v4v0) > 0) p[2] = *v4; else p[1] = *v4; }
}
/* squared distance from the origin to triangle (a,b,c); closest point in `w` */
static real origin_tri_dist2(const real* a, const real* b, const real* c, real* w) {
  real ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(ab, b, a); sub3(ac, c, a);
  real d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { copy3(w, a); return dot3(w, w); }
  real bp[3] = {-b[0], -b[1], -b[2]}, d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { copy3(w, b); return dot3(w, w); }
  real vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { real v = d1 / (d1 - d3); copy3(w, a); addscl3(w, ab, v); return dot3(w, w); }
  real cp[3] = {-c[0], -c[1], -c[2]}, d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { copy3(w, c); return dot3(w, w); }
  real vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { real v = d2 / (d2 - d6); copy3(w, a); addscl3(w, ac, v); return dot3(w, w); }
  real va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    real v = (d4 - d3) / ((d4 - d3) + (d5 - d6)), bc[3]; sub3(bc, c, b); copy3(w, b); addscl3(w, bc, v); return dot3(w, w);
  }
  real den = 1.0 / (va + vb + vc), v = vb * den, u = vc * den;
  copy3(w, a); addscl3(w, ab, v); addscl3(w, ac, u);
  return dot3(w, w);
}
/* 1 (default): libccd verbatim, which is what MuJoCo 2.0's mjc_Convex runs.  0: the HIP kernel's documented
   deviation (portal plane), kept so that the size of the deviation can be measured (DESIGN.md "Deviations"). */
static int ro_mpr_libccd_tridist = 1;
void ro_set_mpr_libccd_tridist(int on) { ro_mpr_libccd_tridist = on; }
/* 1 (default): box-box pairs go through the dedicated multi-point routine (MuJoCo: mjc_BoxBox).  0: through the
   generic convex (MPR, one contact) path, the HIP kernel's documented deviation. */
static int ro_boxbox_multipoint = 1;
void ro_set_boxbox_multipoint(int on) { ro_boxbox_multipoint = on; }
static void find_pos(const ccd_support* p, real* pos) {
  real dir[3], b[4], t[3];
  portal_dir(p, dir);
  cross3(t, p[1].v, p[2].v); b[0] = dot3(t, p[3].v);
  cross3(t, p[3].v, p[2].v); b[1] = dot3(t, p[0].v);
  cross3(t, p[0].v, p[1].v); b[2] = dot3(t, p[3].v);
  cross3(t, p[2].v, p[1].v); b[3] = dot3(t, p[0].v);
  real sum = b[0] + b[1] + b[2] + b[3];
  if (ccd_zero(sum) || sum < 0) {
    b[0] = 0;
    cross3(t, p[2].v, p[3].v); b[1] = dot3(t, dir);
    cross3(t, p[3].v, p[1].v); b[2] = dot3(t, dir);
    cross3(t, p[1].v, p[2].v); b[3] = dot3(t, dir);
    sum = b[1] + b[2] + b[3];
  }
  real inv = 1.0 / sum, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++) { addscl3(p1, p[i].v1, b[i]); addscl3(p2, p[i].v2, b[i]); }
  for (int k = 0; k < 3; k++) pos[k] = 0.5 * inv * (p1[k] + p2[k]);
}
/* returns 0 and fills depth/dir/pos on penetration, -1 otherwise */
static int mpr_penetration(const ccd_obj* o1, const ccd_obj* o2, int max_iter, real tol, real* depth, real* dir_out,
                           real* pos, long* iters) {
  ccd_support p[4], v4;
  real dir[3], va[3], vb[3], dt;
  /* discoverPortal */
  copy3(p[0].v1, o1->d->geom_xpos + 3 * o1->geom); copy3(p[0].v2, o2->d->geom_xpos + 3 * o2->geom);
  sub3(p[0].v, p[0].v1, p[0].v2);
  if (ccd_zero(p[0].v[0]) && ccd_zero(p[0].v[1]) && ccd_zero(p[0].v[2])) p[0].v[0] += CCD_EPS * 10;
  scl3(dir, p[0].v, -1); normalize3(dir);
  mpr_support(o1, o2, dir, &p[1]);
  dt = dot3(p[1].v, dir);
  if (ccd_zero(dt) || dt < 0) return -1;
  cross3(dir, p[0].v, p[1].v);
  if (ccd_zero(dot3(dir, dir))) {
    if (ccd_zero(p[1].v[0]) && ccd_zero(p[1].v[1]) && ccd_zero(p[1].v[2])) {
      /* touching contact at v1 (findPenetrTouch) */
      *depth = 0; zero3(dir_out); for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
      return 0;
    }
    /* origin on the v0-v1 segment (findPenetrSegment) */
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
    copy3(dir_out, p[1].v); *depth = normalize3(dir_out);
    return 0;
  }
  normalize3(dir);
  mpr_support(o1, o2, dir, &p[2]);
  dt = dot3(p[2].v, dir);
  if (ccd_zero(dt) || dt < 0) return -1;
  sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v); cross3(dir, va, vb); normalize3(dir);
  if (dot3(dir, p[0].v) > 0) { ccd_support t = p[1]; p[1] = p[2]; p[2] = t; scl3(dir, dir, -1); }
  for (int guard = 0;; guard++) {
    if (guard > 100) return -1;
    mpr_support(o1, o2, dir, &p[3]);
    dt = dot3(p[3].v, dir);
    if (ccd_zero(dt) || dt < 0) return -1;
    int cont = 0;
    cross3(va, p[1].v, p[3].v); dt = dot3(va, p[0].v);
    if (dt < 0 && !ccd_zero(dt)) { p[2] = p[3]; cont = 1; }
    if (!cont) {
      cross3(va, p[3].v, p[2].v); dt = dot3(va, p[0].v);
      if (dt < 0 && !ccd_zero(dt)) { p[1] = p[3]; cont = 1; }
    }
    if (!cont) break;
    sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v); cross3(dir, va, vb); normalize3(dir);
  }
  /* refinePortal */
  for (int guard = 0;; guard++) {
    if (guard > 1000) return -1;
    portal_dir(p, dir);
    dt = dot3(dir, p[1].v);
    if (ccd_zero(dt) || dt > 0) break; /* portal encapsulates the origin */
    mpr_support(o1, o2, dir, &v4);
    dt = dot3(v4.v, dir);
    if (!(ccd_zero(dt) || dt > 0) || portal_reach_tol(p, &v4, dir, tol)) return -1;
    expand_portal(p, &v4);
  }
  /* findPenetr */
  for (int it = 0;; it++) {
    portal_dir(p, dir);
    mpr_support(o1, o2, dir, &v4);
    if (iters) (*iters)++;
    if (portal_reach_tol(p, &v4, dir, tol) || it > max_iter) {
      if (ro_mpr_libccd_tridist) {
        /* libccd verbatim: distance/direction to the closest point of the final portal TRIANGLE.  On flat
           (face-face) contacts the final triangle depends on rounding-level tie breaks among equal support
           points, and the answer jumps whenever the origin's projection leaves the triangle. */
        real w[3];
        *depth = sqrt(origin_tri_dist2(p[1].v, p[2].v, p[3].v, w));
        if (ccd_zero(*depth)) zero3(dir_out);
        else { copy3(dir_out, w); normalize3(dir_out); }
      } else {
        /* kernel variant: the portal PLANE (its normal and distance).  Identical to the above whenever the
           projection lies inside the triangle, independent of which triangle of the supporting plane
           the refinement ended on, and a tighter bound on the penetration depth.  (DESIGN.md "MPR".) */
        *depth = (dot3(p[1].v, dir) + dot3(p[2].v, dir) + dot3(p[3].v, dir)) / 3.0;
        if (*depth < 0) *depth = 0;
        copy3(dir_out, dir);
      }
      find_pos(p, pos);
      return 0;
    }
    expand_portal(p, &v4);
  }
}

static void make_frame(real* f) {
  normalize3(f);
  if (norm3(f + 3) < 0.5) { zero3(f + 3); if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1; }
  real t = dot3(f, f + 3); addscl3(f + 3, f, -t); normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

static int add_contact(const ro_model* m, ro_data* d, int g1, int g2, real dist, const real* pos, const real* normal,
                       real margin, real gap) {
  if (d->ncon >= m->nconmax) { d->warn_contact_full = 1; return 0; }
  ro_contact* c = &d->contact[d->ncon++];
  memset(c, 0, sizeof *c);
  c->dist = dist; copy3(c->pos, pos); copy3(c->frame, normal); make_frame(c->frame);
  c->includemargin = margin - gap; c->geom1 = g1; c->geom2 = g2;
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  real f[3];
  for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[3 * g1 + k], m->geom_friction[3 * g2 + k]);
  c->friction[0] = c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = c->friction[4] = f[2];
  real mix1 = m->geom_solmix[g1], mix2 = m->geom_solmix[g2], mix;
  if (mix1 >= MINVAL && mix2 >= MINVAL) mix = mix1 / (mix1 + mix2);
  else if (mix1 < MINVAL && mix2 < MINVAL) mix = 0.5;
  else mix = mix1 < MINVAL ? 0.0 : 1.0;
  const real *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
  if (r1[0] > 0 && r2[0] > 0) for (int k = 0; k < 2; k++) c->solref[k] = mix * r1[k] + (1 - mix) * r2[k];
  else for (int k = 0; k < 2; k++) c->solref[k] = fmin(r1[k], r2[k]);
  for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
  return 1;
}

/* engine_collision_box.c: mjc_BoxBox — multi-point box-box contacts (up to 8).  MuJoCo 2.0's routine is closed
 * source; what is restated here is the scheme its documentation describes and that every engine of that lineage
 * uses (separating-axis test over the 15 candidate axes; edge-edge: one contact at the closest points of the two
 * edges; face: the incident face clipped against the reference face, one contact per clipped vertex within the
 * margin), with MuJoCo's contact conventions: frame normal from geom1 to geom2, dist < 0 in penetration, position
 * midway between the two surfaces, contacts included while dist < margin.  PROVENANCE: recalled, unverified. */
static void box_axis(real* a, const real* R, int j) { a[0] = R[j]; a[1] = R[3 + j]; a[2] = R[6 + j]; }
static int clip_poly(real (*poly)[2], int n, int axis, real sign, real lim) {
  /* Sutherland-Hodgman against the half plane sign * x[axis] <= lim */
  real out[16][2]; int no = 0;
  for (int i = 0; i < n; i++) {
    const real *a = poly[i], *b = poly[(i + 1) % n];
    real da = sign * a[axis] - lim, db = sign * b[axis] - lim;
    if (da <= 0) { out[no][0] = a[0]; out[no][1] = a[1]; no++; }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      real t = da / (da - db);
      out[no][0] = a[0] + t * (b[0] - a[0]); out[no][1] = a[1] + t * (b[1] - a[1]); no++;
    }
    if (no >= 15) break;
  }
  for (int i = 0; i < no; i++) { poly[i][0] = out[i][0]; poly[i][1] = out[i][1]; }
  return no;
}
static void collide_box_box(const ro_model* m, ro_data* d, int g1, int g2, real margin, real gap) {
  const real *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2, *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  const real *A = m->geom_size + 3 * g1, *B = m->geom_size + 3 * g2;
  real t[3], ax1[3][3], ax2[3][3], R[3][3], Q[3][3], ta[3], tb[3];
  sub3(t, p2, p1);
  for (int j = 0; j < 3; j++) { box_axis(ax1[j], R1, j); box_axis(ax2[j], R2, j); }
  for (int i = 0; i < 3; i++) { ta[i] = dot3(t, ax1[i]); tb[i] = dot3(t, ax2[i]); for (int j = 0; j < 3; j++) { R[i][j] = dot3(ax1[i], ax2[j]); Q[i][j] = fabs(R[i][j]); } }
  real best = -1e300, n[3] = {0, 0, 0}; int code = 0;
  for (int i = 0; i < 3; i++) {   /* face normals of box 1 */
    real s = fabs(ta[i]) - (A[i] + B[0] * Q[i][0] + B[1] * Q[i][1] + B[2] * Q[i][2]);
    if (s > margin) return;
    if (s > best) { best = s; code = 1 + i; scl3(n, ax1[i], ta[i] < 0 ? -1 : 1); }
  }
  for (int j = 0; j < 3; j++) {   /* face normals of box 2 */
    real s = fabs(tb[j]) - (B[j] + A[0] * Q[0][j] + A[1] * Q[1][j] + A[2] * Q[2][j]);
    if (s > margin) return;
    if (s > best) { best = s; code = 4 + j; scl3(n, ax2[j], tb[j] < 0 ? -1 : 1); }
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {   /* edge x edge */
    real c[3]; cross3(c, ax1[i], ax2[j]);
    real l = norm3(c);
    if (l < 1e-8) continue;   /* parallel edges: covered by the face axes */
    scl3(c, c, 1.0 / l);
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    real ra = A[i1] * fabs(dot3(c, ax1[i1])) + A[i2] * fabs(dot3(c, ax1[i2]));
    real rb = B[j1] * fabs(dot3(c, ax2[j1])) + B[j2] * fabs(dot3(c, ax2[j2]));
    real tc = dot3(t, c), s = fabs(tc) - (ra + rb);
    if (s > margin) return;
    /* a face axis is preferred unless the edge axis is clearly better (the usual 5 % hysteresis) */
    if (s > best + 0.05 * fabs(best) + 1e-12 && s > best) { best = s; code = 7 + 3 * i + j; scl3(n, c, tc < 0 ? -1 : 1); }
  }
  if (code >= 7) {
    int i = (code - 7) / 3, j = (code - 7) % 3;
    real pa[3], pb[3];
    copy3(pa, p1); copy3(pb, p2);
    for (int k = 0; k < 3; k++) { if (k != i) addscl3(pa, ax1[k], (dot3(n, ax1[k]) > 0 ? 1 : -1) * A[k]); if (k != j) addscl3(pb, ax2[k], (dot3(n, ax2[k]) > 0 ? -1 : 1) * B[k]); }
    /* closest points of the lines pa + a ua, pb + b ub */
    real w[3]; sub3(w, pb, pa);
    real uaub = dot3(ax1[i], ax2[j]), q1 = dot3(ax1[i], w), q2 = -dot3(ax2[j], w), den = 1 - uaub * uaub;
    real alpha = den > 1e-12 ? (q1 + uaub * q2) / den : 0, beta = den > 1e-12 ? (uaub * q1 + q2) / den : 0;
    alpha = clampd(alpha, -A[i], A[i]); beta = clampd(beta, -B[j], B[j]);
    addscl3(pa, ax1[i], alpha); addscl3(pb, ax2[j], beta);
    real pos[3] = {0.5 * (pa[0] + pb[0]), 0.5 * (pa[1] + pb[1]), 0.5 * (pa[2] + pb[2])};
    add_contact(m, d, g1, g2, best, pos, n, margin, gap);
    return;
  }
  /* face contact: reference box owns the axis; nr = its outward normal towards the other box */
  int ref1 = code <= 3, ia = ref1 ? code - 1 : code - 4;
  const real (*rax)[3] = ref1 ? ax1 : ax2; const real (*iax)[3] = ref1 ? ax2 : ax1;
  const real *rp = ref1 ? p1 : p2, *ip = ref1 ? p2 : p1, *rs = ref1 ? A : B, *is = ref1 ? B : A;
  real nr[3]; scl3(nr, n, ref1 ? 1 : -1);
  int ib = 0; real bd = 1e300;   /* incident face: the one facing the reference face most directly */
  for (int k = 0; k < 3; k++) { real dk = dot3(nr, iax[k]); if (-fabs(dk) < bd) { bd = -fabs(dk); ib = k; } }
  real isgn = dot3(nr, iax[ib]) > 0 ? -1 : 1;
  real ic[3]; copy3(ic, ip); addscl3(ic, iax[ib], isgn * is[ib]);
  int u = (ia + 1) % 3, v = (ia + 2) % 3, iu = (ib + 1) % 3, iv = (ib + 2) % 3;
  real poly[16][2]; int np = 4;
  for (int k = 0; k < 4; k++) {
    real x[3]; copy3(x, ic);
    addscl3(x, iax[iu], ((k == 0 || k == 3) ? 1 : -1) * is[iu]); addscl3(x, iax[iv], (k < 2 ? 1 : -1) * is[iv]);
    real rel[3]; sub3(rel, x, rp);
    poly[k][0] = dot3(rel, rax[u]); poly[k][1] = dot3(rel, rax[v]);
  }
  np = clip_poly(poly, np, 0, 1, rs[u]); np = clip_poly(poly, np, 0, -1, rs[u]);
  np = clip_poly(poly, np, 1, 1, rs[v]); np = clip_poly(poly, np, 1, -1, rs[v]);
  /* lift the clipped vertices back onto the incident face plane: x = rp + a ru + b rv + h nr with (x - ic) . inorm = 0 */
  real inorm[3]; scl3(inorm, iax[ib], isgn);
  real dn = dot3(nr, inorm);
  int cnt = 0;
  for (int k = 0; k < np && cnt < 8; k++) {
    real base[3]; copy3(base, rp); addscl3(base, rax[u], poly[k][0]); addscl3(base, rax[v], poly[k][1]);
    real rel[3]; sub3(rel, ic, base);
    real h = fabs(dn) > 1e-12 ? dot3(rel, inorm) / dn : dot3(rel, nr);
    real dist = h - rs[ia];   /* signed distance of the incident-face point from the reference face */
    if (dist > margin) continue;
    int dup = 0;
    for (int q = 0; q < k; q++) if (fabs(poly[q][0] - poly[k][0]) + fabs(poly[q][1] - poly[k][1]) < 1e-12) dup = 1;
    if (dup) continue;
    real pos[3]; copy3(pos, base); addscl3(pos, nr, rs[ia] + 0.5 * dist);
    cnt += add_contact(m, d, g1, g2, dist, pos, n, margin, gap);
  }
}

static void collide_pair(const ro_model* m, ro_data* d, int g1, int g2) {
  if (m->geom_type[g1] > m->geom_type[g2]) { int t = g1; g1 = g2; g2 = t; }
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) return;
  real margin = fmax(m->geom_margin[g1], m->geom_margin[g2]), gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
  const real *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
  const real *R1 = d->geom_xmat + 9 * g1;
  if (t1 == GEOM_PLANE) {
    real n[3] = {R1[2], R1[5], R1[8]}, dif[3];
    sub3(dif, p2, p1);
    if (t2 == GEOM_PLANE) return;
    if (dot3(dif, n) > m->geom_rbound[g2] + margin) return;
    if (t2 == GEOM_BOX) { /* mjc_PlaneBox: corners below plane+margin, at most 4 */
      const real *R2 = d->geom_xmat + 9 * g2, *sz = m->geom_size + 3 * g2;
      int cnt = 0;
      for (int i = 0; i < 8 && cnt < 4; i++) {
        real lc[3] = {(i & 1 ? 1 : -1) * sz[0], (i & 2 ? 1 : -1) * sz[1], (i & 4 ? 1 : -1) * sz[2]}, c[3];
        mulmat3(c, R2, lc); add3(c, c, p2);
        real t[3]; sub3(t, c, p1);
        real dist = dot3(t, n);
        if (dist > margin) continue;
        real pos[3]; copy3(pos, c); addscl3(pos, n, -0.5 * dist);
        cnt += add_contact(m, d, g1, g2, dist, pos, n, margin, gap);
      }
    } else { /* mjc_PlaneConvex (deepest support point) / plane-sphere etc. through the support map */
      ccd_obj o = {m, d, g2, 0};
      real nd[3] = {-n[0], -n[1], -n[2]}, s[3], t[3];
      geom_support(&o, nd, s); sub3(t, s, p1);
      real dist = dot3(t, n);
      if (dist > margin) return;
      real pos[3]; copy3(pos, s); addscl3(pos, n, -0.5 * dist);
      add_contact(m, d, g1, g2, dist, pos, n, margin, gap);
    }
    return;
  }
  real dif[3]; sub3(dif, p2, p1);
  real bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
  if (dot3(dif, dif) > bound * bound) return;
  if (t1 == GEOM_BOX && t2 == GEOM_BOX && ro_boxbox_multipoint) { collide_box_box(m, d, g1, g2, margin, gap); return; }
  /* mjc_Convex: MPR on shapes inflated by margin/2 each; dist = margin - depth */
  ccd_obj o1 = {m, d, g1, 0.5 * margin}, o2 = {m, d, g2, 0.5 * margin};
  real depth, dir[3], pos[3];
  d->stat_mpr_calls++;
  if (mpr_penetration(&o1, &o2, m->mpr_iterations, m->mpr_tolerance, &depth, dir, pos, &d->stat_mpr_iter) != 0) return;
  if (norm3(dir) < 0.5) return; /* contact found but normal undefined */
  /* libccd reports the direction that separates obj2 from obj1 when applied to obj2's negative;
     MuJoCo's frame normal points from geom1 to geom2 */
  add_contact(m, d, g1, g2, margin - depth, pos, dir, margin, gap);
}

static int body_pair_excluded(const ro_model* m, int b1, int b2) {
  int sig = (b1 < b2 ? (b1 << 16) + b2 : (b2 << 16) + b1);
  for (int i = 0; i < m->nexclude; i++) if (m->exclude_signature[i] == sig) return 1;
  return 0;
}
static void ro_collision(const ro_model* m, ro_data* d) {
  d->ncon = 0;
  for (int b1 = 0; b1 < m->nbody; b1++) {
    if (!m->body_geomnum[b1]) continue;
    for (int b2 = b1 + 1; b2 < m->nbody; b2++) {
      if (!m->body_geomnum[b2]) continue;
      int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
      if (w1 == w2) continue;
      int pw1 = m->body_weldid[m->body_parentid[w1]], pw2 = m->body_weldid[m->body_parentid[w2]];
      if (w1 != 0 && w2 != 0 && (w1 == pw2 || w2 == pw1)) continue;
      if (body_pair_excluded(m, b1, b2)) continue;
      for (int i = 0; i < m->body_geomnum[b1]; i++)
        for (int j = 0; j < m->body_geomnum[b2]; j++) collide_pair(m, d, m->body_geomadr[b1] + i, m->body_geomadr[b2] + j);
    }
  }
}

/* ------------------------------------------------------------------------------------------ constraints
 * engine_core_constraint.c: mj_makeConstraint (friction, limit, contact rows), mj_makeImpedance */
static real get_impedance(const real* si, real pos, real margin) {
  real dmin = clampd(si[0], 1e-4, 0.9999), dmax = clampd(si[1], 1e-4, 0.9999), width = fmax(si[2], MINVAL);
  real mid = clampd(si[3], 1e-4, 0.9999), power = fmax(si[4], 1.0);
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  real x = fabs((pos - margin) / width);
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  real y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
static int add_row(const ro_model* m, ro_data* d, int type, int id, real pos, real margin, real floss, real diag) {
  if (d->nefc >= m->njmax) { d->warn_efc_full = 1; return -1; }
  int r = d->nefc++;
  memset(d->efc_J + (size_t)r * m->nv, 0, m->nv * sizeof(real));
  d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_margin[r] = margin;
  d->efc_frictionloss[r] = floss; d->efc_diagApprox[r] = diag;
  return r;
}
static void ro_make_constraint(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  d->nefc = d->nf = d->nl = d->ne = 0;
  /* equality constraints (mj_instantiateEquality), MuJoCo 2.0 semantics:
   *   weld  (ur16e/base.xml:52-54, the mocap weld of the TCP solver simulation): eq_data = pose of body2 in the frame of body1
   *         (3 position + 4 quaternion numbers); residual position = (xpos1 + R1 relpos) - xpos2, residual rotation = vector part
   *         of conj(q2) q1 relquat; Jacobian = J(body1, point 1) - J(body2, point 2), rotational rows through the half-quaternion
   *         product ("0.5 * neg(q1) * (jac0 - jac1) * q0 * relpose").
   *   joint (gripper_actuators.xml:3, the two-finger coupling): q1 - q1_0 - poly(q2 - q2_0), J = [1, -poly'].
   * Rows are always active (two-sided, quadratic); margin 0.  PROVENANCE: MuJoCo documentation ("Computation: Equality") and the
   * later open-sourced engine_core_constraint.c, recalled. */
  for (int e = 0; e < m->neq; e++) {
    if (!d->eq_active[e]) continue;
    const real* data = d->eq_data + 7 * e;
    if (m->eq_type[e] == EQ_WELD) {
      int b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e];
      real p1[3], cpos[6], q[4], qn[4], q2[4], tmp[3];
      mulmat3(tmp, d->xmat + 9 * b1, data); add3(p1, d->xpos + 3 * b1, tmp);
      const real* p2 = d->xpos + 3 * b2;
      sub3(cpos, p1, p2);
      mulquat(q, d->xquat + 4 * b1, data + 3);                                   /* q = q1 * relquat */
      qn[0] = d->xquat[4 * b2]; qn[1] = -d->xquat[4 * b2 + 1]; qn[2] = -d->xquat[4 * b2 + 2]; qn[3] = -d->xquat[4 * b2 + 3];
      mulquat(q2, qn, q);                                                        /* conj(q2) * q1 * relquat */
      cpos[3] = q2[1]; cpos[4] = q2[2]; cpos[5] = q2[3];
      real* jb = (real*)malloc(12 * nv * sizeof(real));
      real *jp1 = jb, *jr1 = jb + 3 * nv, *jp2 = jb + 6 * nv, *jr2 = jb + 9 * nv;
      ro_jac(m, d, jp1, jr1, p1, b1); ro_jac(m, d, jp2, jr2, p2, b2);
      real tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2], rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
      int rows[6];
      for (int k = 0; k < 6; k++) rows[k] = add_row(m, d, EFC_EQUALITY, e, cpos[k], 0, 0, k < 3 ? tran : rot);
      for (int i = 0; i < nv; i++) {
        for (int k = 0; k < 3; k++) if (rows[k] >= 0) d->efc_J[(size_t)rows[k] * nv + i] = jp1[k * nv + i] - jp2[k * nv + i];
        real ax[4] = {0, jr1[i] - jr2[i], jr1[nv + i] - jr2[nv + i], jr1[2 * nv + i] - jr2[2 * nv + i]}, t1[4], t2[4];
        mulquat(t1, qn, ax); mulquat(t2, t1, q);
        for (int k = 0; k < 3; k++) if (rows[3 + k] >= 0) d->efc_J[(size_t)rows[3 + k] * nv + i] = 0.5 * t2[1 + k];
      }
      free(jb);
    } else if (m->eq_type[e] == EQ_JOINT) {
      int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
      int q1a = m->jnt_qposadr[j1], d1 = m->jnt_dofadr[j1];
      real pos = d->qpos[q1a] - m->qpos0[q1a] - data[0], deriv = 0, diag = m->dof_invweight0[d1];
      if (j2 >= 0) {
        int q2a = m->jnt_qposadr[j2];
        real dif = d->qpos[q2a] - m->qpos0[q2a];
        pos -= data[1] * dif + data[2] * dif * dif + data[3] * dif * dif * dif + data[4] * dif * dif * dif * dif;
        deriv = data[1] + 2 * data[2] * dif + 3 * data[3] * dif * dif + 4 * data[4] * dif * dif * dif;
        diag += m->dof_invweight0[m->jnt_dofadr[j2]];
      }
      int r = add_row(m, d, EFC_EQUALITY, e, pos, 0, 0, diag);
      if (r >= 0) { d->efc_J[(size_t)r * nv + d1] = 1; if (j2 >= 0) d->efc_J[(size_t)r * nv + m->jnt_dofadr[j2]] = -deriv; }
    } else { fprintf(stderr, "rg_oracle: equality type %d not implemented\n", m->eq_type[e]); abort(); }
  }
  d->ne = d->nefc;
  /* friction loss: dofs, then tendons */
  for (int i = 0; i < nv; i++)
    if (m->dof_frictionloss[i] > 0) {
      int r = add_row(m, d, EFC_FRICTION_DOF, i, 0, 0, m->dof_frictionloss[i], m->dof_invweight0[i]);
      if (r >= 0) d->efc_J[(size_t)r * nv + i] = 1;
    }
  for (int t = 0; t < m->ntendon; t++)
    if (m->tendon_frictionloss[t] > 0) {
      int r = add_row(m, d, EFC_FRICTION_TENDON, t, 0, 0, m->tendon_frictionloss[t], m->tendon_invweight0[t]);
      if (r >= 0) memcpy(d->efc_J + (size_t)r * nv, d->ten_J + (size_t)t * nv, nv * sizeof(real));
    }
  d->nf = d->nefc - d->ne;
  /* limits: joints (hinge/slide), then tendons */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    int t = m->jnt_type[j];
    if (t != JNT_HINGE && t != JNT_SLIDE) continue; /* ball limits are not used by the robogym models */
    real q = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      real dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - q);
      if (dist < m->jnt_margin[j]) {
        int r = add_row(m, d, EFC_LIMIT_JOINT, j, dist, m->jnt_margin[j], 0, m->dof_invweight0[m->jnt_dofadr[j]]);
        if (r >= 0) d->efc_J[(size_t)r * nv + m->jnt_dofadr[j]] = -side;
      }
    }
  }
  for (int t = 0; t < m->ntendon; t++) {
    if (!m->tendon_limited[t]) continue;
    for (int side = -1; side <= 1; side += 2) {
      real dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - d->ten_length[t]);
      if (dist < m->tendon_margin[t]) {
        int r = add_row(m, d, EFC_LIMIT_TENDON, t, dist, m->tendon_margin[t], 0, m->tendon_invweight0[t]);
        if (r >= 0) for (int k = 0; k < nv; k++) d->efc_J[(size_t)r * nv + k] = -side * d->ten_J[(size_t)t * nv + k];
      }
    }
  }
  d->nl = d->nefc - d->nf - d->ne;
  /* contacts (pyramidal or elliptic cones) */
  real* jbuf = (real*)malloc(12 * nv * sizeof(real));
  real *jp1 = jbuf, *jr1 = jbuf + 3 * nv, *jp2 = jbuf + 6 * nv, *jr2 = jbuf + 9 * nv;
  for (int ci = 0; ci < d->ncon; ci++) {
    ro_contact* c = &d->contact[ci];
    c->efc_address = -1;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    ro_jac(m, d, jp1, jr1, c->pos, b1); ro_jac(m, d, jp2, jr2, c->pos, b2);
    /* contact-frame Jacobian rows: 3 translational then 3 rotational, difference body2 - body1 */
    real* Jc = (real*)malloc(6 * nv * sizeof(real));
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < nv; k++) {
        Jc[r * nv + k] = c->frame[3 * r] * (jp2[k] - jp1[k]) + c->frame[3 * r + 1] * (jp2[nv + k] - jp1[nv + k]) + c->frame[3 * r + 2] * (jp2[2 * nv + k] - jp1[2 * nv + k]);
        Jc[(3 + r) * nv + k] = c->frame[3 * r] * (jr2[k] - jr1[k]) + c->frame[3 * r + 1] * (jr2[nv + k] - jr1[nv + k]) + c->frame[3 * r + 2] * (jr2[2 * nv + k] - jr1[2 * nv + k]);
      }
    real tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    real rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
    if (c->dim == 1) {
      int r = add_row(m, d, EFC_CONTACT_PYRAMIDAL, ci, c->dist, c->includemargin, 0, tran);
      if (r >= 0) { memcpy(d->efc_J + (size_t)r * nv, Jc, nv * sizeof(real)); c->efc_address = r; }
    } else if (m->cone == 1) {
      /* elliptic cone (ur16e/base.xml:3 cone="elliptic"): one row per contact dimension, the rows of the contact-frame Jacobian
       * themselves (normal, two tangents, torsion, two rolling axes); only the normal row carries the distance */
      if (d->nefc + c->dim > m->njmax) { d->warn_efc_full = 1; free(Jc); continue; }
      for (int k = 0; k < c->dim; k++) {
        int r = add_row(m, d, EFC_CONTACT_ELLIPTIC, ci, k == 0 ? c->dist : 0, k == 0 ? c->includemargin : 0, 0, k < 3 ? tran : rot);
        if (k == 0) c->efc_address = r;
        memcpy(d->efc_J + (size_t)r * nv, Jc + (size_t)k * nv, nv * sizeof(real));
      }
    } else {
      for (int k = 0; k < c->dim - 1; k++) {
        real fri = c->friction[k], diag = tran + fri * fri * (k < 2 ? tran : rot);
        for (int s = 1; s >= -1; s -= 2) {
          int r = add_row(m, d, EFC_CONTACT_PYRAMIDAL, ci, c->dist, c->includemargin, 0, diag);
          if (r < 0) continue;
          if (c->efc_address < 0) c->efc_address = r;
          for (int i = 0; i < nv; i++) d->efc_J[(size_t)r * nv + i] = Jc[i] + s * fri * Jc[(k + 1) * nv + i];
        }
      }
    }
    free(Jc);
  }
  free(jbuf);
}

static void ro_make_impedance(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  for (int r = 0; r < d->nefc; r++) {
    const real *solref, *solimp;
    int id = d->efc_id[r], type = d->efc_type[r];
    switch (type) {
      case EFC_FRICTION_DOF: solref = m->dof_solref + 2 * id; solimp = m->dof_solimp + 5 * id; break;
      case EFC_FRICTION_TENDON: solref = m->tendon_solref_fri + 2 * id; solimp = m->tendon_solimp_fri + 5 * id; break;
      case EFC_LIMIT_JOINT: solref = m->jnt_solref + 2 * id; solimp = m->jnt_solimp + 5 * id; break;
      case EFC_LIMIT_TENDON: solref = m->tendon_solref_lim + 2 * id; solimp = m->tendon_solimp_lim + 5 * id; break;
      case EFC_EQUALITY: solref = m->eq_solref + 2 * id; solimp = m->eq_solimp + 5 * id; break;
      default: solref = d->contact[id].solref; solimp = d->contact[id].solimp; break;
    }
    real imp = get_impedance(solimp, d->efc_pos[r], d->efc_margin[r]);
    d->efc_R[r] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[r] / imp);
    real dmax = clampd(solimp[1], 1e-4, 0.9999), K, B;
    if (solref[0] > 0) {
      real tc = fmax(solref[0], 2 * m->timestep), dr = solref[1];
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr); B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -solref[0] / fmax(MINVAL, dmax * dmax); B = -solref[1] / fmax(MINVAL, dmax); }
    if (type == EFC_FRICTION_DOF || type == EFC_FRICTION_TENDON) K = 0;
    d->efc_KBIP[4 * r] = K; d->efc_KBIP[4 * r + 1] = B; d->efc_KBIP[4 * r + 2] = imp; d->efc_KBIP[4 * r + 3] = 0;
  }
  /* pyramidal contacts: all rows share R = 2 mu^2 R_first */
  for (int ci = 0; ci < d->ncon; ci++) {
    ro_contact* c = &d->contact[ci];
    if (c->efc_address < 0 || c->dim == 1) continue;
    if (m->cone == 1) {
      /* mj_makeImpedance, elliptic: R[1] = R[0] / impratio, mu of the regularised cone = friction[0] sqrt(R[1] / R[0]),
       * R[j] mu_j^2 = R[1] mu_1^2 for the remaining friction dimensions */
      int a = c->efc_address;
      d->efc_R[a + 1] = d->efc_R[a] / fmax(MINVAL, m->impratio);
      c->mu = c->friction[0] * sqrt(d->efc_R[a + 1] / d->efc_R[a]);
      for (int j = 2; j < c->dim; j++) d->efc_R[a + j] = d->efc_R[a + 1] * c->friction[0] * c->friction[0] / (c->friction[j - 1] * c->friction[j - 1]);
      continue;
    }
    c->mu = c->friction[0] * sqrt(1 / m->impratio);
    real Rpy = 2 * c->mu * c->mu * d->efc_R[c->efc_address];
    for (int k = 0; k < 2 * (c->dim - 1) && c->efc_address + k < d->nefc; k++) d->efc_R[c->efc_address + k] = Rpy;
  }
  for (int r = 0; r < d->nefc; r++) {
    d->efc_D[r] = 1 / d->efc_R[r];
    real v = 0; const real* J = d->efc_J + (size_t)r * nv;
    for (int k = 0; k < nv; k++) v += J[k] * d->qvel[k];
    d->efc_vel[r] = v;
    d->efc_aref[r] = -d->efc_KBIP[4 * r + 1] * v - d->efc_KBIP[4 * r] * d->efc_KBIP[4 * r + 2] * (d->efc_pos[r] - d->efc_margin[r]);
  }
}

/* ------------------------------------------------------------------------------------------ velocity stage
 * engine_core_smooth.c: mj_comVel, mj_passive, mj_rne */
static void ro_fwd_velocity(const ro_model* m, ro_data* d) {
  int nv = m->nv, nb = m->nbody;
  for (int t = 0; t < m->ntendon; t++) {
    real v = 0; for (int k = 0; k < nv; k++) v += d->ten_J[(size_t)t * nv + k] * d->qvel[k];
    d->ten_velocity[t] = v;
  }
  for (int i = 0; i < m->nu; i++) {
    real v = 0; for (int k = 0; k < nv; k++) v += d->actuator_moment[(size_t)i * nv + k] * d->qvel[k];
    d->actuator_velocity[i] = v;
  }
  memset(d->cvel, 0, 6 * sizeof(real));
  for (int b = 1; b < nb; b++) {
    real cv[6]; memcpy(cv, d->cvel + 6 * m->body_parentid[b], sizeof cv);
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, da = m->jnt_dofadr[j], t = m->jnt_type[j];
      if (t == JNT_FREE) {
        for (int i = 0; i < 3; i++) { memset(d->cdof_dot + 6 * (da + i), 0, 6 * sizeof(real)); for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i]; }
        da += 3;
      }
      if (t == JNT_FREE || t == JNT_BALL) {
        for (int i = 0; i < 3; i++) cross_motion(d->cdof_dot + 6 * (da + i), cv, d->cdof + 6 * (da + i));
        for (int i = 0; i < 3; i++) for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i];
      } else {
        cross_motion(d->cdof_dot + 6 * da, cv, d->cdof + 6 * da);
        for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * da + c] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * b, cv, sizeof cv);
  }
  /* passive: joint springs, dof damping, tendon spring-dampers */
  memset(d->qfrc_passive, 0, nv * sizeof(real));
  for (int j = 0; j < m->njnt; j++) {
    real k = m->jnt_stiffness[j];
    if (k == 0) continue;
    int t = m->jnt_type[j], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (t == JNT_HINGE || t == JNT_SLIDE) d->qfrc_passive[da] -= k * (d->qpos[qa] - m->qpos_spring[qa]);
    else { fprintf(stderr, "rg_oracle: ball/free joint springs not implemented\n"); abort(); }
  }
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
  for (int t = 0; t < m->ntendon; t++) {
    real f = m->tendon_stiffness[t] * (m->tendon_lengthspring[t] - d->ten_length[t]) - m->tendon_damping[t] * d->ten_velocity[t];
    if (f != 0) for (int k = 0; k < nv; k++) d->qfrc_passive[k] += d->ten_J[(size_t)t * nv + k] * f;
  }
  /* RNE with zero acceleration: Coriolis, centrifugal and gravity */
  real* cacc = (real*)calloc(6 * nb, sizeof(real));
  real* cfrc = (real*)calloc(6 * nb, sizeof(real));
  cacc[3] = -m->gravity[0]; cacc[4] = -m->gravity[1]; cacc[5] = -m->gravity[2];
  for (int b = 1; b < nb; b++) {
    real* a = cacc + 6 * b; memcpy(a, cacc + 6 * m->body_parentid[b], 6 * sizeof(real));
    for (int k = 0; k < m->body_dofnum[b]; k++) { int i = m->body_dofadr[b] + k; for (int c = 0; c < 6; c++) a[c] += d->cdof_dot[6 * i + c] * d->qvel[i]; }
    real t1[6], t2[6], t3[6];
    mul_inert_vec(t1, d->cinert + 10 * b, a);
    mul_inert_vec(t2, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(t3, d->cvel + 6 * b, t2);
    for (int c = 0; c < 6; c++) cfrc[6 * b + c] = t1[c] + t3[c];
  }
  for (int b = nb - 1; b > 0; b--) { int p = m->body_parentid[b]; if (p > 0) for (int c = 0; c < 6; c++) cfrc[6 * p + c] += cfrc[6 * b + c]; }
  for (int i = 0; i < nv; i++) {
    const real *c = d->cdof + 6 * i, *f = cfrc + 6 * m->dof_bodyid[i];
    d->qfrc_bias[i] = c[0] * f[0] + c[1] * f[1] + c[2] * f[2] + c[3] * f[3] + c[4] * f[4] + c[5] * f[5];
  }
  free(cacc); free(cfrc);
}

/* ------------------------------------------------------------------------------------------ actuation
 * engine_forward.c: mj_fwdActuation with mujoco-py's mjpid.pyx callbacks (gain = 0, bias = PID force).
 * gainprm = [kp, ti, iclamp, td, dsmooth, deadband] (/root/reference/robogym/mujoco/constants.py:35-52,
 * robot/shadow_hand/mujoco/parameter_manager.py:23-47).  Controller state per actuator:
 * {integral error, last error, last smoothed derivative}.  The callback runs in EVERY mj_forward,
 * including the three state-less forward() calls the reference makes per env step. */
/* The cascaded-PI controller's bias feed-forward (gravity + Coriolis compensation: the force of the velocity loop plus qfrc_bias of the actuated dof).  mjpid.pyx
 * is not in the tree; the term is inferred from the reference's own pins (envs/rearrange/tests/test_rearrange_sim.py:135-230): with it all four impulse-response
 * cases hold at the stated 1e-3 (0.0362 / 0.0364 / 0.0365 for 0.036, 0.02197 / 0.02209 / 0.02206 for 0.022, ...), without it the wrist joints J5 / J6 -- P-only
 * velocity loops, gainprm ti_v = 0 -- creep under the wrist camera's weight and two cases miss by 1.4-1.6e-3; test_rearrange_robots.py:306-378 (wrist isolation:
 * J1 .. J5 within 0.7 deg over 100 wrist-only steps) likewise holds with it and fails without (tests/test_rearrange_oracle.py keeps both runs of each).
 * ro_set_cascade_bias_ff(0) switches it off for that comparison. */
static int ro_cascade_bias_ff = 1;
void ro_set_cascade_bias_ff(int on) { ro_cascade_bias_ff = on; }
static void ro_fwd_actuation(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  real dt = m->timestep;
  memset(d->qfrc_actuator, 0, nv * sizeof(real));
  for (int i = 0; i < m->nu; i++) {
    const real* gp = m->actuator_gainprm + 10 * i;
    real force;
    if (m->actuator_biastype[i] == 2 && m->actuator_user[i] == 1) {
      /* mujoco-py mjpid.pyx, cascaded PI controller (actuator user[0] == 1; ur16e/jointspec/calibrations/cascaded_pi/
       * joint_actuations.xml:4-10): gainprm = [kp, ti, iclamp, td, dsmooth | kp_v, ti_v, iclamp_v, ema_smooth, max_vel].
       * The position set-point is EMA-smoothed (warm-started with ctrl at time 0), an outer P(ID) loop on position produces a
       * velocity set-point clamped to +-max_vel, an inner PI loop on actuator_velocity produces the force, the dof's bias force is added (see
       * ro_cascade_bias_ff above), the sum is clamped to forcerange.
       * State per actuator: {position integral, velocity integral, smoothed set-point}.  PROVENANCE: recalled from mujoco-py
       * 2.0.2.13 except for the bias feed-forward, which is inferred; pinned through the reference's impulse-response / gripper-sync / rest tests
       * re-expressed on this oracle at their own tolerances (tests/test_rearrange_oracle.py). */
      real* st = d->pid + 3 * i;
      real ema = gp[8], max_vel = gp[9];
      real setp = d->time == 0 ? d->ctrl[i] : ema * st[2] + (1 - ema) * d->ctrl[i];
      st[2] = setp;
      real des_vel;
      if (gp[0] != 0) {
        real err = setp - d->actuator_length[i];
        real integ = clampd(st[0] + err * dt, -gp[2], gp[2]);
        des_vel = gp[0] * (err + (gp[1] != 0 ? integ / gp[1] : 0));   /* (td = dsmooth = 0 in the calibration: no derivative term is carried) */
        st[0] = integ;
      } else des_vel = d->ctrl[i];
      des_vel = clampd(des_vel, -max_vel, max_vel);
      real errv = des_vel - d->actuator_velocity[i];
      real integv = clampd(st[1] + errv * dt, -gp[7], gp[7]);
      force = gp[5] * (errv + (gp[6] != 0 ? integv / gp[6] : 0));
      st[1] = integv;
      if (ro_cascade_bias_ff && m->actuator_trntype[i] == TRN_JOINT) {   /* joint transmission: the moment row has the single entry `gear` at the joint's dof */
        int k = m->jnt_dofadr[m->actuator_trnid[i]];
        force += d->qfrc_bias[k] / d->actuator_moment[(size_t)i * nv + k];
      }
      real lo = m->actuator_forcerange[2 * i], hi = m->actuator_forcerange[2 * i + 1];
      if (lo != 0 || hi != 0) force = clampd(force, lo, hi);
    } else if (m->actuator_biastype[i] == 2) {
      real kp = gp[0], ti = gp[1], iclamp = gp[2], td = gp[3], smooth = gp[4], deadband = gp[5];
      real err = d->ctrl[i] - d->actuator_length[i];
      if (fabs(err) < deadband) err = 0;
      real* st = d->pid + 3 * i;
      real integ = clampd(st[0] + err * dt, -iclamp, iclamp);
      real deriv = (1 - smooth) * st[2] + smooth * (err - st[1]) / dt;
      force = kp * (err + (ti != 0 ? integ / ti : 0) + td * deriv);
      st[0] = integ; st[1] = err; st[2] = deriv;
      real lo = m->actuator_forcerange[2 * i], hi = m->actuator_forcerange[2 * i + 1];
      if (lo != 0 || hi != 0) force = clampd(force, lo, hi);
    } else {
      real ctrl = d->ctrl[i];
      if (m->actuator_ctrllimited[i]) ctrl = clampd(ctrl, m->actuator_ctrlrange[2 * i], m->actuator_ctrlrange[2 * i + 1]);
      const real* bp = m->actuator_biasprm + 10 * i;
      force = gp[0] * ctrl + (m->actuator_biastype[i] == 1 ? bp[0] + bp[1] * d->actuator_length[i] + bp[2] * d->actuator_velocity[i] : 0);
    }
    if (m->actuator_forcelimited[i]) force = clampd(force, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
    d->actuator_force[i] = force;
    for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += d->actuator_moment[(size_t)i * nv + k] * force;
  }
}

/* engine_forward.c: mj_fwdAcceleration */
static void ro_fwd_acceleration(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  /* engine_forward.c: mj_xfrcAccumulate — qfrc += J(com of body)' force + J_rot' torque for every body with xfrc_applied */
  for (int b = 1; b < m->nbody; b++) {
    const real* w = d->xfrc_applied + 6 * b;
    if (w[0] == 0 && w[1] == 0 && w[2] == 0 && w[3] == 0 && w[4] == 0 && w[5] == 0) continue;
    real* jp = dalloc(3 * nv); real* jr = dalloc(3 * nv);
    ro_jac(m, d, jp, jr, d->xipos + 3 * b, b);
    for (int i = 0; i < nv; i++)
      for (int k = 0; k < 3; k++) d->qfrc_smooth[i] += jp[k * nv + i] * w[k] + jr[k * nv + i] * w[3 + k];
    free(jp); free(jr);
  }
  memcpy(d->qacc_smooth, d->qfrc_smooth, nv * sizeof(real));
  chol_solve(d->qL, d->qacc_smooth, nv);
}

/* ------------------------------------------------------------------------------------------ solver
 * engine_solver.c: mj_solNewton — primal Newton on
 *     cost(a) = 1/2 (a - a_s)' M (a - a_s) + sum_i s_i(J_i a - aref_i)
 * with exact line search.  (MuJoCo updates the Cholesky factor incrementally; refactoring each
 * iteration gives the same iterates.) */
typedef struct { real cost, grad, hess; } ls_pt;

/* Elliptic cone, one contact (engine_core_constraint.c mj_constraintUpdate, "elliptic" branch).  In the scaled variables
 * U0 = mu jar_0, Uj = friction_(j-1) jar_j, N = U0, T = |U_1..|:  top zone (N >= mu T): no force;  bottom zone (mu N + T <= 0): every
 * row quadratic with its own D;  middle zone: cost = 1/2 Dm (N - mu T)^2 with Dm = D_0 / (mu^2 (1 + mu^2)).
 * zone: 0 top, 1 bottom, 2 middle.  `hess` (dim x dim, may be NULL) is the cone's Hessian with respect to jar in the middle zone. */
static int cone_eval(const ro_data* d, const ro_contact* c, const real* jar, real* force, real* cost, real* hess) {
  int a = c->efc_address, dim = c->dim;
  real mu = c->mu, U[6], N, T = 0;
  U[0] = jar[a] * mu;
  for (int j = 1; j < dim; j++) { U[j] = jar[a + j] * c->friction[j - 1]; T += U[j] * U[j]; }
  N = U[0]; T = sqrt(T);
  if (N >= mu * T || (T <= 0 && N >= 0)) { if (force) for (int j = 0; j < dim; j++) force[a + j] = 0; *cost = 0; return 0; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    real cc = 0;
    for (int j = 0; j < dim; j++) { real D = d->efc_D[a + j]; if (force) force[a + j] = -D * jar[a + j]; cc += 0.5 * D * jar[a + j] * jar[a + j]; }
    *cost = cc; return 1;
  }
  real Dm = d->efc_D[a] / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
  *cost = 0.5 * Dm * NT * NT;
  if (force) {
    force[a] = -Dm * NT * mu;
    for (int j = 1; j < dim; j++) force[a + j] = -force[a] / T * U[j] * c->friction[j - 1];
  }
  if (hess) {
    /* second derivatives of 1/2 Dm (N - mu T)^2 in U, then the chain rule through U = diag(mu, friction) jar */
    real H[36], sc[6];
    sc[0] = mu; for (int j = 1; j < dim; j++) sc[j] = c->friction[j - 1];
    H[0] = 1;
    for (int j = 1; j < dim; j++) H[j] = H[j * dim] = -mu * U[j] / T;
    for (int j = 1; j < dim; j++) for (int k = 1; k < dim; k++) H[j * dim + k] = mu * N / (T * T * T) * U[j] * U[k] + (j == k ? mu * mu - mu * N / T : 0);
    for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) hess[j * dim + k] = Dm * sc[j] * sc[k] * H[j * dim + k];
  }
  return 2;
}
static void constraint_update(const ro_model* m, const ro_data* d, const real* jar, real* force, int* active, real* cost) {
  real c = 0;
  for (int r = 0; r < d->nefc; r++) {
    real D = d->efc_D[r], R = d->efc_R[r], x = jar[r];
    int type = d->efc_type[r];
    if (type == EFC_EQUALITY) { force[r] = -D * x; active[r] = 1; c += 0.5 * D * x * x; continue; }
    if (type == EFC_CONTACT_ELLIPTIC) {
      const ro_contact* con = d->contact + d->efc_id[r];
      real cc; int zone = cone_eval(d, con, jar, force, &cc, NULL);
      for (int j = 0; j < con->dim; j++) active[r + j] = zone == 1 ? 1 : (zone == 2 ? 2 : 0);   /* 2: coupled through the cone Hessian */
      c += cc; r += con->dim - 1; continue;
    }
    if (type == EFC_FRICTION_DOF || type == EFC_FRICTION_TENDON) {
      real f = d->efc_frictionloss[r];
      if (x <= -R * f) { force[r] = f; active[r] = 0; c += f * (-0.5 * R * f - x); }
      else if (x >= R * f) { force[r] = -f; active[r] = 0; c += f * (-0.5 * R * f + x); }
      else { force[r] = -D * x; active[r] = 1; c += 0.5 * D * x * x; }
    } else {
      if (x >= 0) { force[r] = 0; active[r] = 0; }
      else { force[r] = -D * x; active[r] = 1; c += 0.5 * D * x * x; }
    }
  }
  *cost = c;
}
static ls_pt ls_eval(const ro_data* d, real alpha, const real* jar, const real* jv, const real* quadGauss) {
  ls_pt p = {alpha * alpha * quadGauss[2] + alpha * quadGauss[1] + quadGauss[0], 2 * alpha * quadGauss[2] + quadGauss[1], 2 * quadGauss[2]};
  for (int r = 0; r < d->nefc; r++) {
    real D = d->efc_D[r], R = d->efc_R[r], x = jar[r] + alpha * jv[r];
    int type = d->efc_type[r];
    if (type == EFC_EQUALITY) { p.cost += 0.5 * D * x * x; p.grad += D * x * jv[r]; p.hess += D * jv[r] * jv[r]; continue; }
    if (type == EFC_CONTACT_ELLIPTIC) {
      /* the cone's cost along the search direction and its first two derivatives in alpha */
      const ro_contact* con = d->contact + d->efc_id[r];
      int dim = con->dim; real mu = con->mu, U[6], V[6], N, T = 0, UV = 0, VV = 0;
      U[0] = (jar[r] + alpha * jv[r]) * mu; V[0] = jv[r] * mu;
      for (int j = 1; j < dim; j++) {
        U[j] = (jar[r + j] + alpha * jv[r + j]) * con->friction[j - 1]; V[j] = jv[r + j] * con->friction[j - 1];
        T += U[j] * U[j]; UV += U[j] * V[j]; VV += V[j] * V[j];
      }
      N = U[0]; T = sqrt(T);
      if (N >= mu * T || (T <= 0 && N >= 0)) { /* top zone: nothing */ }
      else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        for (int j = 0; j < dim; j++) { real Dj = d->efc_D[r + j], xj = jar[r + j] + alpha * jv[r + j]; p.cost += 0.5 * Dj * xj * xj; p.grad += Dj * xj * jv[r + j]; p.hess += Dj * jv[r + j] * jv[r + j]; }
      } else {
        real Dm = d->efc_D[r] / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
        real N1 = V[0], T1 = UV / T, T2 = (VV - T1 * T1) / T;
        p.cost += 0.5 * Dm * NT * NT; p.grad += Dm * NT * (N1 - mu * T1); p.hess += Dm * ((N1 - mu * T1) * (N1 - mu * T1) - NT * mu * T2);
      }
      r += dim - 1; continue;
    }
    if (type == EFC_FRICTION_DOF || type == EFC_FRICTION_TENDON) {
      real f = d->efc_frictionloss[r];
      if (x <= -R * f) { p.cost += f * (-0.5 * R * f - x); p.grad += -f * jv[r]; }
      else if (x >= R * f) { p.cost += f * (-0.5 * R * f + x); p.grad += f * jv[r]; }
      else { p.cost += 0.5 * D * x * x; p.grad += D * x * jv[r]; p.hess += D * jv[r] * jv[r]; }
    } else if (x < 0) { p.cost += 0.5 * D * x * x; p.grad += D * x * jv[r]; p.hess += D * jv[r] * jv[r]; }
  }
  return p;
}
static int ro_trace = 0;   /* diagnostics of the Newton iterations (ro_set_trace) */
void ro_set_trace(int on) { ro_trace = on; }
/* exact minimiser of the convex piecewise-quadratic 1-D restriction: safeguarded Newton on its derivative */
static real line_search(const ro_data* d, const real* jar, const real* jv, const real* quadGauss, real gtol, int maxit) {
  ls_pt p0 = ls_eval(d, 0, jar, jv, quadGauss);
  if (p0.grad >= 0 || p0.hess <= 0) return 0;
  real lo = 0, hi = -1, glo = p0.grad, hlo = p0.hess, ghi = 0, hhi = 0;
  real a = -p0.grad / p0.hess;
  real best_a = 0, best_cost = p0.cost;     /* the best point seen: what is returned when the iteration limit ends the search */
  real wprev = 1e300; int since = 0;
  for (int it = 0; it < maxit; it++) {
    ls_pt p = ls_eval(d, a, jar, jv, quadGauss);
    if (ro_trace) fprintf(stderr, "      ls it %d a %.9g grad %.4g hess %.4g lo %.9g hi %.9g gtol %.3g\n", it, (double)a, (double)p.grad, (double)p.hess, (double)lo, (double)hi, (double)gtol);
    if (p.cost < best_cost) { best_cost = p.cost; best_a = a; }
    if (fabs(p.grad) < gtol) return a;
    if (p.grad < 0) { lo = a; glo = p.grad; hlo = p.hess; } else { hi = a; ghi = p.grad; hhi = p.hess; }
    real cand = lo - glo / hlo;               /* Newton step from the left end */
    if (hi >= 0) {
      if (!(cand > lo && cand < hi)) {        /* fall back: Newton from the right end, then bisection */
        cand = hi - ghi / hhi;
        if (!(cand > lo && cand < hi)) cand = 0.5 * (lo + hi);
      }
      /* The derivative is increasing but only piecewise smooth: at a kink where a stiff row switches on, the Newton steps from the two ends can
       * alternate between two points on either side of it for ever (seen with the gripper driven into a joint limit: slope ratio 10 at the kink).
       * Safeguard: three steps in a row that are not at least halving (a converging Newton iteration shrinks them much faster) -> bisect. */
      real step = fabs(cand - a);
      if (step > 0.5 * wprev) since++; else since = 0;      /* (wprev: the previous step's length) */
      if (since >= 3) { cand = 0.5 * (lo + hi); since = 0; step = fabs(cand - a); }
      wprev = step;
    }
    if (cand == a) return a;
    a = cand;
  }
  return best_a;
}

static void ro_solve(const ro_model* m, ro_data* d) {
  int nv = m->nv, ne = d->nefc;
  memset(d->qfrc_constraint, 0, nv * sizeof(real));
  d->solver_iter = 0;
  if (ne == 0) { memcpy(d->qacc, d->qacc_smooth, nv * sizeof(real)); return; }
  real *jar = dalloc(ne), *jv = dalloc(ne), *force = dalloc(ne), *Ma = dalloc(nv), *grad = dalloc(nv), *search = dalloc(nv),
         *Mv = dalloc(nv), *H = dalloc((size_t)nv * nv), *Lh = dalloc((size_t)nv * nv), *qa = dalloc(nv);
  int* active = (int*)calloc(ne, sizeof(int));
  real scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  /* warm start: pick the better of qacc_warmstart and qacc_smooth */
  real cost_ws = 0, cost_sm = 0;
  for (int pass = 0; pass < 2; pass++) {
    const real* a = pass == 0 ? d->qacc_warmstart : d->qacc_smooth;
    real g = 0;
    for (int i = 0; i < nv; i++) { real s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * a[k]; Ma[i] = s; }
    for (int i = 0; i < nv; i++) g += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (a[i] - d->qacc_smooth[i]);
    for (int r = 0; r < ne; r++) { real s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * a[k]; jar[r] = s - d->efc_aref[r]; }
    real c; constraint_update(m, d, jar, force, active, &c);
    if (pass == 0) cost_ws = g + c; else cost_sm = g + c;
  }
  memcpy(qa, cost_ws < cost_sm ? d->qacc_warmstart : d->qacc_smooth, nv * sizeof(real));
  real cost = 0, oldcost;
  for (int iter = 0;; iter++) {
    /* Ma, jar, cost, gradient */
    for (int i = 0; i < nv; i++) { real s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * qa[k]; Ma[i] = s; }
    for (int r = 0; r < ne; r++) { real s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * qa[k]; jar[r] = s - d->efc_aref[r]; }
    real cc; constraint_update(m, d, jar, force, active, &cc);
    real gauss = 0;
    for (int i = 0; i < nv; i++) gauss += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (qa[i] - d->qacc_smooth[i]);
    oldcost = cost; cost = gauss + cc;
    for (int i = 0; i < nv; i++) { real s = Ma[i] - d->qfrc_smooth[i]; for (int r = 0; r < ne; r++) s -= d->efc_J[(size_t)r * nv + i] * force[r]; grad[i] = s; }
    real gn = 0; for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
    gn = sqrt(gn) * scale;
    if (ro_trace) fprintf(stderr, "newton iter %d cost %.12g improvement*scale %.3e gn %.3e nefc %d\n", iter, (double)cost, iter ? (double)(scale * (oldcost - cost)) : 0.0, (double)gn, ne);
    if (iter > 0) { real improvement = scale * (oldcost - cost); if (improvement < m->tolerance) break; }
    if (gn < m->tolerance || iter >= m->iterations) break;
    d->solver_iter = iter + 1;
    /* Hessian H = M + J' diag(D_active) J */
    memcpy(H, d->qM, (size_t)nv * nv * sizeof(real));
    for (int r = 0; r < ne; r++) {
      if (active[r] != 1) continue;
      const real* J = d->efc_J + (size_t)r * nv; real D = d->efc_D[r];
      for (int i = 0; i < nv; i++) { if (J[i] == 0) continue; real di = D * J[i]; for (int k = 0; k < nv; k++) H[i * nv + k] += di * J[k]; }
    }
    for (int ci = 0; ci < d->ncon; ci++) {   /* middle-zone elliptic contacts: J_c' H_cone J_c (engine_solver.c HessianCone) */
      const ro_contact* con = d->contact + ci;
      if (m->cone != 1 || con->dim == 1 || con->efc_address < 0 || active[con->efc_address] != 2) continue;
      real hc[36], cc; int dim = con->dim;
      cone_eval(d, con, jar, NULL, &cc, hc);
      const real* Jc = d->efc_J + (size_t)con->efc_address * nv;
      for (int a = 0; a < dim; a++) for (int b = 0; b < dim; b++) {
        real h = hc[a * dim + b];
        if (h == 0) continue;
        for (int i = 0; i < nv; i++) { real ja = Jc[(size_t)a * nv + i]; if (ja == 0) continue; real t = h * ja; for (int k = 0; k < nv; k++) H[i * nv + k] += t * Jc[(size_t)b * nv + k]; }
      }
    }
    if (cholesky(Lh, H, nv) != 0) { d->warn_bad |= 2; break; }
    for (int i = 0; i < nv; i++) search[i] = -grad[i];
    chol_solve(Lh, search, nv);
    /* line search */
    for (int i = 0; i < nv; i++) { real s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * nv + k] * search[k]; Mv[i] = s; }
    for (int r = 0; r < ne; r++) { real s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * search[k]; jv[r] = s; }
    real quadGauss[3] = {gauss, 0, 0};
    for (int i = 0; i < nv; i++) { quadGauss[1] += search[i] * (Ma[i] - d->qfrc_smooth[i]); quadGauss[2] += 0.5 * search[i] * Mv[i]; }
    real snorm = 0; for (int i = 0; i < nv; i++) snorm += search[i] * search[i];
    snorm = sqrt(snorm);
    if (snorm < MINVAL) break;
    real gtol = m->tolerance * m->ls_tolerance * snorm / scale * 1e-3; /* much tighter than MuJoCo's: "exact" */
    real alpha = line_search(d, jar, jv, quadGauss, gtol, 60);
    if (ro_trace) {
      ls_pt pa = ls_eval(d, alpha, jar, jv, quadGauss), pz = ls_eval(d, 0, jar, jv, quadGauss);
      fprintf(stderr, "   alpha %.6g |search| %.3e  model cost(0) %.12g grad(0) %.4g hess(0) %.4g  model cost(alpha) %.12g grad %.4g\n", (double)alpha, (double)snorm, (double)pz.cost, (double)pz.grad, (double)pz.hess, (double)pa.cost, (double)pa.grad);
      for (real t = 0.1; t < 1.25; t += 0.1) { ls_pt q = ls_eval(d, t, jar, jv, quadGauss); fprintf(stderr, "      t %.1f cost %.10g grad %.4g\n", (double)t, (double)q.cost, (double)q.grad); }
    }
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) qa[i] += alpha * search[i];
  }
  /* final forces at the solution */
  for (int r = 0; r < ne; r++) { real s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * qa[k]; jar[r] = s - d->efc_aref[r]; }
  real cc; constraint_update(m, d, jar, force, active, &cc);
  memcpy(d->efc_force, force, ne * sizeof(real));
  memcpy(d->qacc, qa, nv * sizeof(real));
  for (int r = 0; r < ne; r++) for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[(size_t)r * nv + k] * force[r];
  free(jar); free(jv); free(force); free(Ma); free(grad); free(search); free(Mv); free(H); free(Lh); free(qa); free(active);
}

/* ------------------------------------------------------------------------------------------ independent cross-check solver
 * engine_solver.c: mj_solPGS — projected Gauss-Seidel on the DUAL of the same convex problem (VERDICT r03 item 6 iii):
 *     minimise 1/2 f' (A + R) f + f' (J qacc_smooth - aref),  A = J M^-1 J',  subject to the rows' force bounds
 *     (equality: free; friction loss: |f| <= frictionloss; limits and pyramidal contact edges: f >= 0;
 *      an elliptic contact's rows jointly: f_0 >= 0, sum_j (f_j / friction_(j-1))^2 <= f_0^2),
 * then qacc = qacc_smooth + M^-1 J' f.  It shares NOTHING with the Newton path but the constraint rows: no Hessian, no line search, no warm
 * start, no zone formulas -- the Newton solver's three-zone cone cost (cone_eval) is the closed form of exactly this cone-constrained problem in
 * the metric R (with R_j friction_(j-1)^2 equal for all friction rows the cone is circular in the variables f_j sqrt(R_j), slope `mu`), so at the
 * optimum the two coincide (strong duality), which is what tests/test_oracle.py and tests/test_rearrange_oracle.py assert.  An elliptic contact is
 * one Gauss-Seidel block: its dim x dim subproblem is minimised over the cone by projected gradient in the variables y_0 = f_0,
 * y_j = f_j / friction_(j-1), where the cone is the second-order cone |y_1..| <= y_0 and the Euclidean projection is closed form.
 * Uses the rows of the last ro_forward; returns the number of sweeps. */
static void soc_project(real* y, int dim) {
  real t = y[0], n = 0;
  for (int j = 1; j < dim; j++) n += y[j] * y[j];
  n = sqrt(n);
  if (n <= t) return;
  if (n <= -t) { for (int j = 0; j < dim; j++) y[j] = 0; return; }
  real beta = 0.5 * (t + n);
  y[0] = beta;
  for (int j = 1; j < dim; j++) y[j] *= beta / n;
}
int ro_solve_pgs(const ro_model* m, ro_data* d, int max_sweeps, double tol, double* qacc_out) {
  int nv = m->nv, ne = d->nefc;
  if (ne < 0 || ne > MAXEFC || nv <= 0) return -2;
  const size_t un = (size_t)(ne > 0 ? ne : 1);
  real* MinvJt = dalloc((size_t)nv * un);   /* column r = M^-1 J_r' */
  real* A = dalloc(un * un);
  real *b = dalloc(un), *f = dalloc(un), *col = dalloc((size_t)nv);
  for (int r = 0; r < ne; r++) {
    memcpy(col, d->efc_J + (size_t)r * nv, nv * sizeof(real));
    chol_solve(d->qL, col, nv);
    for (int i = 0; i < nv; i++) MinvJt[(size_t)i * ne + r] = col[i];
  }
  for (int r = 0; r < ne; r++) {
    const real* J = d->efc_J + (size_t)r * nv;
    for (int c = 0; c < ne; c++) { real s = 0; for (int i = 0; i < nv; i++) s += J[i] * MinvJt[(size_t)i * ne + c]; A[(size_t)r * ne + c] = s; }
    real s = 0; for (int i = 0; i < nv; i++) s += J[i] * d->qacc_smooth[i];
    b[r] = s - d->efc_aref[r];
  }
  int sweep = 0;
  for (; sweep < max_sweeps; sweep++) {
    real change = 0;
    for (int r = 0; r < ne; r++) {
      int type = d->efc_type[r];
      if (type == EFC_CONTACT_ELLIPTIC) {
        const ro_contact* con = d->contact + d->efc_id[r];
        int dim = con->dim;
        real H[36], c[6], y[6], y0[6], sc[6], tr = 0;
        sc[0] = 1; for (int j = 1; j < dim; j++) sc[j] = con->friction[j - 1];
        for (int j = 0; j < dim; j++) {
          real g = b[r + j];
          for (int k = 0; k < ne; k++) if (k < r || k >= r + dim) g += A[(size_t)(r + j) * ne + k] * f[k];
          c[j] = g * sc[j];                                                   /* linear term, in y */
          for (int k = 0; k < dim; k++) H[j * dim + k] = (A[(size_t)(r + j) * ne + r + k] + (j == k ? d->efc_R[r + j] : 0)) * sc[j] * sc[k];
          tr += H[j * dim + j];
          y[j] = y0[j] = f[r + j] / sc[j];
        }
        for (int it = 0; it < 2000; it++) {                                   /* projected gradient, step 1 / trace(H) <= 1 / lambda_max */
          real g[6], step = 0;
          for (int j = 0; j < dim; j++) { g[j] = c[j]; for (int k = 0; k < dim; k++) g[j] += H[j * dim + k] * y[k]; }
          real yn[6] = {0};
          for (int j = 0; j < dim; j++) yn[j] = y[j] - g[j] / tr;
          soc_project(yn, dim);
          for (int j = 0; j < dim; j++) { step = fmax(step, fabs(yn[j] - y[j]) * tr); y[j] = yn[j]; }
          if (step < 0.01 * tol) break;
        }
        for (int j = 0; j < dim; j++) { change = fmax(change, fabs(y[j] - y0[j]) * H[j * dim + j]); f[r + j] = y[j] * sc[j]; }
        r += dim - 1;
        continue;
      }
      real g = b[r] + d->efc_R[r] * f[r];
      for (int c = 0; c < ne; c++) g += A[(size_t)r * ne + c] * f[c];
      real fn = f[r] - g / (A[(size_t)r * ne + r] + d->efc_R[r]);
      if (type == EFC_FRICTION_DOF || type == EFC_FRICTION_TENDON) fn = clampd(fn, -d->efc_frictionloss[r], d->efc_frictionloss[r]);
      else if (type != EFC_EQUALITY && fn < 0) fn = 0;
      change = fmax(change, fabs(fn - f[r]) * (A[(size_t)r * ne + r] + d->efc_R[r]));
      f[r] = fn;
    }
    if (change < tol) { sweep++; break; }
  }
  for (int i = 0; i < nv; i++) { real s = d->qacc_smooth[i]; for (int r = 0; r < ne; r++) s += MinvJt[(size_t)i * ne + r] * f[r]; qacc_out[i] = (double)s; }
  free(MinvJt); free(A); free(b); free(f); free(col);
  return sweep;
}

/* ------------------------------------------------------------------------------------------ forward / step */
/* engine_sensor.c: mj_sensorAcc, mjSENS_TOUCH.  A touch sensor sums the NORMAL forces of the contacts that involve the
 * body of its site and whose contact point "sees" the site's volume along the contact normal (mju_rayGeom(site, contact
 * position, +-normal) >= 0: only the sign of the ray test matters, i.e. "does the ray from the contact point along the
 * normal meet the site's shape").  The normal force of a pyramidal contact is the sum of its pyramid edge forces
 * (mju_decodePyramid).  The ray test is stated geometrically (exact for sphere, capsule = cylinder segment + two spheres,
 * ellipsoid, cylinder, box), not as MuJoCo's root bookkeeping. */
static int ray_hits_sphere(const real* c, real r, const real* p, const real* v) {
  real w[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]}, vv = dot3(v, v);
  real t = vv > 0 ? -dot3(w, v) / vv : 0; if (t < 0) t = 0;
  real q[3] = {w[0] + t * v[0], w[1] + t * v[1], w[2] + t * v[2]};
  return dot3(q, q) <= r * r;
}
/* the part of the ray (t >= 0) inside the infinite cylinder x^2 + y^2 <= r^2: [t0, t1]; returns 0 if empty */
static int ray_in_cylinder(real r, const real* p, const real* v, real* t0, real* t1) {
  real a = v[0] * v[0] + v[1] * v[1], b = p[0] * v[0] + p[1] * v[1], c = p[0] * p[0] + p[1] * p[1] - r * r;
  if (a < 1e-30) { if (c > 0) return 0; *t0 = 0; *t1 = 1e300; return 1; }
  real det = b * b - a * c; if (det < 0) return 0;
  real sq = sqrt(det); *t0 = (-b - sq) / a; *t1 = (-b + sq) / a;
  if (*t1 < 0) return 0;
  if (*t0 < 0) *t0 = 0;
  return 1;
}
static int ray_hits_site(int type, const real* size, const real* p, const real* v) {   /* p, v in the site frame */
  const real zero[3] = {0, 0, 0};
  if (type == GEOM_SPHERE) return ray_hits_sphere(zero, size[0], p, v);
  if (type == GEOM_CAPSULE || type == GEOM_CYLINDER) {
    real r = size[0], h = size[1], t0, t1;
    if (ray_in_cylinder(r, p, v, &t0, &t1)) {   /* somewhere on [t0, t1] the height must be within +-h */
      real z0 = p[2] + t0 * v[2], z1 = t1 > 1e299 ? (v[2] > 0 ? 1e300 : (v[2] < 0 ? -1e300 : p[2])) : p[2] + t1 * v[2];
      real lo = z0 < z1 ? z0 : z1, hi = z0 < z1 ? z1 : z0;
      if (lo <= h && hi >= -h) return 1;
    }
    if (type == GEOM_CYLINDER) return 0;        /* (flat ends: entering through an end disk also puts a ray point inside the side test above) */
    real ct[3] = {0, 0, h}, cb[3] = {0, 0, -h};
    return ray_hits_sphere(ct, r, p, v) || ray_hits_sphere(cb, r, p, v);
  }
  if (type == GEOM_ELLIPSOID) {
    real ps[3] = {p[0] / size[0], p[1] / size[1], p[2] / size[2]}, vs[3] = {v[0] / size[0], v[1] / size[1], v[2] / size[2]};
    return ray_hits_sphere(zero, 1.0, ps, vs);
  }
  if (type == GEOM_BOX) {
    real t0 = 0, t1 = 1e300;
    for (int k = 0; k < 3; k++) {
      if (fabs(v[k]) < 1e-30) { if (fabs(p[k]) > size[k]) return 0; continue; }
      real a = (-size[k] - p[k]) / v[k], b = (size[k] - p[k]) / v[k];
      if (a > b) { real t = a; a = b; b = t; }
      if (a > t0) t0 = a;
      if (b < t1) t1 = b;
    }
    return t0 <= t1;
  }
  return 0;
}
/* engine_core_smooth.c: mj_rnePostConstraint — body accelerations with the final qacc, external forces (xfrc_applied + contacts)
 * and the interaction force of every body with its parent, all in the com-based frame (rotational part first). */
static void contact_force_world(const ro_model* m, const ro_data* d, const ro_contact* c, real* f, real* t) {
  real cf[6] = {0, 0, 0, 0, 0, 0};
  int a = c->efc_address;
  if (c->dim == 1) cf[0] = d->efc_force[a];
  else if (m->cone == 1) for (int j = 0; j < c->dim; j++) cf[j] = d->efc_force[a + j];
  else for (int j = 0; j < c->dim - 1; j++) {   /* mju_decodePyramid */
    real fp = d->efc_force[a + 2 * j], fn = d->efc_force[a + 2 * j + 1];
    cf[0] += fp + fn; cf[1 + j] = (fp - fn) * c->friction[j];
  }
  for (int k = 0; k < 3; k++) { f[k] = c->frame[k] * cf[0] + c->frame[3 + k] * cf[1] + c->frame[6 + k] * cf[2]; t[k] = c->frame[k] * cf[3] + c->frame[3 + k] * cf[4] + c->frame[6 + k] * cf[5]; }
}
static void ro_rne_post_constraint(const ro_model* m, ro_data* d) {
  int nb = m->nbody;
  memset(d->cfrc_ext, 0, 6 * nb * sizeof(real));
  for (int b = 1; b < nb; b++) {   /* xfrc_applied: force / torque at the body's com, moved to the com-frame origin of its tree */
    const real* w = d->xfrc_applied + 6 * b;
    real off[3], tq[3];
    sub3(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]); cross3(tq, off, w);
    for (int k = 0; k < 3; k++) { d->cfrc_ext[6 * b + k] += w[3 + k] + tq[k]; d->cfrc_ext[6 * b + 3 + k] += w[k]; }
  }
  for (int i = 0; i < d->ncon; i++) {
    const ro_contact* c = d->contact + i;
    if (c->efc_address < 0) continue;
    real f[3], t[3]; contact_force_world(m, d, c, f, t);   /* force body1 exerts on body2 is +f along the frame (geom1 -> geom2) */
    for (int side = 0; side < 2; side++) {
      int b = m->geom_bodyid[side ? c->geom2 : c->geom1]; real sg = side ? 1.0 : -1.0;
      if (b == 0) continue;
      real off[3], tq[3]; sub3(off, c->pos, d->subtree_com + 3 * m->body_rootid[b]); cross3(tq, off, f);
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6 * b + k] += sg * (t[k] + tq[k]); d->cfrc_ext[6 * b + 3 + k] += sg * f[k]; }
    }
  }
  memset(d->cacc, 0, 6 * sizeof(real)); d->cacc[3] = -m->gravity[0]; d->cacc[4] = -m->gravity[1]; d->cacc[5] = -m->gravity[2];
  memset(d->cfrc_int, 0, 6 * sizeof(real));
  for (int b = 1; b < nb; b++) {
    real* a = d->cacc + 6 * b; memcpy(a, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(real));
    for (int k = 0; k < m->body_dofnum[b]; k++) { int i = m->body_dofadr[b] + k; for (int c = 0; c < 6; c++) a[c] += d->cdof_dot[6 * i + c] * d->qvel[i] + d->cdof[6 * i + c] * d->qacc[i]; }
    real t1[6], t2[6], t3[6];
    mul_inert_vec(t1, d->cinert + 10 * b, a); mul_inert_vec(t2, d->cinert + 10 * b, d->cvel + 6 * b); cross_force(t3, d->cvel + 6 * b, t2);
    for (int c = 0; c < 6; c++) d->cfrc_int[6 * b + c] = t1[c] + t3[c] - d->cfrc_ext[6 * b + c];
  }
  for (int b = nb - 1; b > 0; b--) { int p = m->body_parentid[b]; if (p > 0) for (int c = 0; c < 6; c++) d->cfrc_int[6 * p + c] += d->cfrc_int[6 * b + c]; }
}
void ro_sensor(const ro_model* m, ro_data* d) {
  int need_rne = 0;
  for (int k = 0; k < m->nsensor; k++) if (m->sensor_type[k] == SENS_FORCE || m->sensor_type[k] == SENS_TORQUE) need_rne = 1;
  if (need_rne) ro_rne_post_constraint(m, d);
  for (int k = 0; k < m->nsensor; k++) {
    int adr = m->sensor_adr ? m->sensor_adr[k] : k;
    if (m->sensor_type[k] == SENS_JOINTPOS) { d->sensordata[adr] = d->qpos[m->jnt_qposadr[m->sensor_objid[k]]]; continue; }
    if (m->sensor_type[k] == SENS_FORCE || m->sensor_type[k] == SENS_TORQUE) {
      /* engine_sensor.c: cfrc_int of the site's body, moved from the com-frame origin to the site and rotated into the site frame */
      int site = m->sensor_objid[k], body = m->site_bodyid[site];
      const real* cf = d->cfrc_int + 6 * body; const real* R = d->site_xmat + 9 * site;
      real off[3], tq[3], v[3];
      sub3(off, d->site_xpos + 3 * site, d->subtree_com + 3 * m->body_rootid[body]);
      cross3(tq, off, cf + 3);
      if (m->sensor_type[k] == SENS_FORCE) copy3(v, cf + 3); else sub3(v, cf, tq);
      mulmatT3(d->sensordata + adr, R, v);
      continue;
    }
    d->sensordata[adr] = 0;
    if (m->sensor_type[k] != SENS_TOUCH) continue;
    int site = m->sensor_objid[k], body = m->site_bodyid[site];
    for (int i = 0; i < d->ncon; i++) {
      const ro_contact* c = d->contact + i;
      if (c->efc_address < 0) continue;
      int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
      if (b1 != body && b2 != body) continue;
      real nf = 0;
      if (c->dim == 1 || m->cone == 1) nf = d->efc_force[c->efc_address];
      else for (int q = 0; q < 2 * (c->dim - 1); q++) nf += d->efc_force[c->efc_address + q];
      if (nf <= 0) continue;
      real sgn = body == b2 ? -1.0 : 1.0, ray[3] = {sgn * c->frame[0], sgn * c->frame[1], sgn * c->frame[2]};
      real rel[3], lp[3], lv[3]; const real* R = d->site_xmat + 9 * site;
      sub3(rel, c->pos, d->site_xpos + 3 * site);
      for (int a = 0; a < 3; a++) { lp[a] = R[a] * rel[0] + R[3 + a] * rel[1] + R[6 + a] * rel[2]; lv[a] = R[a] * ray[0] + R[3 + a] * ray[1] + R[6 + a] * ray[2]; }
      if (ray_hits_site(m->site_type[site], m->site_size + 3 * site, lp, lv)) d->sensordata[adr] += nf;
    }
  }
}

void ro_fwd_position(const ro_model* m, ro_data* d) {
  ro_kinematics(m, d); ro_com_pos(m, d); ro_tendon(m, d); ro_transmission(m, d); ro_crb(m, d);
  ro_collision(m, d); ro_make_constraint(m, d);
}
void ro_forward(const ro_model* m, ro_data* d) {
  ro_fwd_position(m, d);
  ro_fwd_velocity(m, d);
  ro_make_impedance(m, d); /* needs efc_vel: reference acceleration */
  ro_fwd_actuation(m, d);
  ro_fwd_acceleration(m, d);
  ro_solve(m, d);
  /* engine_forward.c, end of mj_fwdConstraint: "save result for next step warmstart" -- EVERY forward stores it, so the
   * state-less mj_forward calls of an env.step (simulation_interface.py:185, robot_env.py:677, observation/mujoco.py:22-27)
   * overwrite the warm start with the solution at the state the next mj_step starts from. */
  memcpy(d->qacc_warmstart, d->qacc, m->nv * sizeof(real));
  ro_sensor(m, d);
}
static int bad(const real* x, int n) { for (int i = 0; i < n; i++) if (!(fabs(x[i]) < 1e10)) return 1; return 0; }

/* engine_forward.c: mj_step with the Euler integrator (implicit in joint damping), mj_Euler / mj_advance */
void ro_step(const ro_model* m, ro_data* d) {
  int nv = m->nv;
  if (bad(d->qpos, m->nq) || bad(d->qvel, nv)) { d->warn_bad |= 4; return; }
  ro_forward(m, d);
  if (bad(d->qacc, nv)) { d->warn_bad |= 8; return; }
  d->stat_ncon += d->ncon; d->stat_nefc += d->nefc; d->stat_iter += d->solver_iter; d->stat_steps++;
  real* qacc = dalloc(nv);
  int damped = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  if (!damped) memcpy(qacc, d->qacc, nv * sizeof(real));
  else {
    real *H = dalloc((size_t)nv * nv), *L = dalloc((size_t)nv * nv);
    memcpy(H, d->qM, (size_t)nv * nv * sizeof(real));
    for (int i = 0; i < nv; i++) { H[i * nv + i] += m->timestep * m->dof_damping[i]; qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    cholesky(L, H, nv); chol_solve(L, qacc, nv);
    free(H); free(L);
  }
  real h = m->timestep;
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j], t = m->jnt_type[j];
    if (t == JNT_FREE) { for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k]; qa += 3; da += 3; }
    if (t == JNT_FREE || t == JNT_BALL) {
      real w[3] = {d->qvel[da], d->qvel[da + 1], d->qvel[da + 2]}, ang = norm3(w) * h;
      if (ang > 0) {
        real q[4], qn[4]; normalize3(w); axisangle2quat(q, w, ang);
        mulquat(qn, d->qpos + qa, q); normalize4(qn); memcpy(d->qpos + qa, qn, sizeof qn);
      }
    } else d->qpos[qa] += h * d->qvel[da];
  }
  d->time += h;   /* (qacc_warmstart = qacc was stored by the forward pass above) */
  free(qacc);
}

/* SimulationInterface.step (simulation_interface.py:176-189): nsubsteps x mj_step, then mj_forward */
void ro_sim_step(const ro_model* m, ro_data* d, int nsubsteps) {
  for (int i = 0; i < nsubsteps; i++) ro_step(m, d);
  ro_forward(m, d);
}

/* ------------------------------------------------------------------------------------------ accessors */
#define FIELD(name, cnt) if (strcmp(field, #name) == 0) { *n = (cnt); return d->name; }
real* ro_field(const ro_model* m, ro_data* d, const char* field, int* n) {
  int nv = m->nv, nb = m->nbody;
  FIELD(qpos, m->nq) FIELD(qvel, nv) FIELD(ctrl, m->nu) FIELD(pid, 3 * m->nu) FIELD(qacc_warmstart, nv) FIELD(xfrc_applied, 6 * nb)
  FIELD(xpos, 3 * nb) FIELD(xquat, 4 * nb) FIELD(xmat, 9 * nb) FIELD(xipos, 3 * nb) FIELD(ximat, 9 * nb)
  FIELD(xanchor, 3 * m->njnt) FIELD(xaxis, 3 * m->njnt) FIELD(geom_xpos, 3 * m->ngeom) FIELD(geom_xmat, 9 * m->ngeom)
  FIELD(site_xpos, 3 * m->nsite) FIELD(site_xmat, 9 * m->nsite) FIELD(subtree_com, 3 * nb) FIELD(cinert, 10 * nb)
  FIELD(cdof, 6 * nv) FIELD(ten_length, m->ntendon) FIELD(ten_J, m->ntendon * nv) FIELD(actuator_length, m->nu)
  FIELD(actuator_moment, m->nu * nv) FIELD(qM, nv * nv) FIELD(efc_J, d->nefc * nv) FIELD(efc_pos, d->nefc)
  FIELD(efc_margin, d->nefc) FIELD(efc_R, d->nefc) FIELD(efc_D, d->nefc) FIELD(efc_aref, d->nefc) FIELD(efc_force, d->nefc)
  FIELD(efc_vel, d->nefc) FIELD(efc_frictionloss, d->nefc) FIELD(efc_diagApprox, d->nefc)
  FIELD(ten_velocity, m->ntendon) FIELD(actuator_velocity, m->nu) FIELD(cvel, 6 * nb) FIELD(cdof_dot, 6 * nv)
  FIELD(qfrc_passive, nv) FIELD(qfrc_bias, nv) FIELD(actuator_force, m->nu) FIELD(qfrc_actuator, nv) FIELD(qfrc_smooth, nv)
  FIELD(qacc_smooth, nv) FIELD(qfrc_constraint, nv) FIELD(qacc, nv) FIELD(sensordata, m->nsensordata) FIELD(mocap_pos, 3 * m->nmocap) FIELD(mocap_quat, 4 * m->nmocap) FIELD(eq_data, 7 * m->neq)
  FIELD(cfrc_int, 6 * nb) FIELD(cfrc_ext, 6 * nb) FIELD(cacc, 6 * nb)
  *n = 0;
  return NULL;
}
int ro_int(const ro_model* m, const ro_data* d, const char* field) {
  if (!strcmp(field, "ncon")) return d->ncon;
  if (!strcmp(field, "nefc")) return d->nefc;
  if (!strcmp(field, "nf")) return d->nf;
  if (!strcmp(field, "nl")) return d->nl;
  if (!strcmp(field, "ne")) return d->ne;
  if (!strcmp(field, "neq")) return m->neq;
  if (!strcmp(field, "nmocap")) return m->nmocap;
  if (!strcmp(field, "solver_iter")) return d->solver_iter;
  if (!strcmp(field, "warn_bad")) return d->warn_bad;
  if (!strcmp(field, "warn_contact_full")) return d->warn_contact_full;
  if (!strcmp(field, "warn_efc_full")) return d->warn_efc_full;
  return -1;
}
int* ro_efc_type(ro_data* d) { return d->efc_type; }
int* ro_eq_active(ro_data* d) { return d->eq_active; }
int* ro_efc_id(ro_data* d) { return d->efc_id; }
int ro_real_size(void) { return (int)sizeof(real); }
double ro_time(const ro_data* d) { return d->time; }
void ro_set_time(ro_data* d, double t) { d->time = t; }
/* contact i -> out[0..]: dist, pos3, frame9, includemargin, friction5, dim, geom1, geom2, efc_address  (23 doubles) */
void ro_contact_get(const ro_data* d, int i, double* out) {
  const ro_contact* c = &d->contact[i];
  out[0] = c->dist; out[13] = c->includemargin;
  for (int k = 0; k < 3; k++) out[1 + k] = c->pos[k];
  for (int k = 0; k < 9; k++) out[4 + k] = c->frame[k];
  for (int k = 0; k < 5; k++) out[14 + k] = c->friction[k];
  out[19] = c->dim; out[20] = c->geom1; out[21] = c->geom2; out[22] = c->efc_address;
}
void ro_stats(const ro_data* d, double* out) {
  double n = d->stat_steps > 0 ? (double)d->stat_steps : 1;
  out[0] = d->stat_ncon / n; out[1] = d->stat_nefc / n; out[2] = d->stat_iter / n; out[3] = (double)d->stat_steps;
  out[4] = d->stat_mpr_calls / n; out[5] = d->stat_mpr_iter / n;
}
void ro_stats_reset(ro_data* d) { d->stat_ncon = d->stat_nefc = d->stat_iter = d->stat_steps = d->stat_mpr_calls = d->stat_mpr_iter = 0; }
/* standalone MPR call between two geoms of the current configuration (for collision unit tests) */
int ro_mpr_pair(const ro_model* m, ro_data* d, int g1, int g2, double margin, double* out /* depth, dir3, pos3 */) {
  ccd_obj o1 = {m, d, g1, (real)(0.5 * margin)}, o2 = {m, d, g2, (real)(0.5 * margin)};
  real r[7];
  int rc = mpr_penetration(&o1, &o2, m->mpr_iterations, m->mpr_tolerance, r, r + 1, r + 4, NULL);
  for (int k = 0; k < 7; k++) out[k] = r[k];
  return rc;
}

/* ------------------------------------------------------------------------------------------ cpu_baseline driver
 * bench.py's cpu_baseline leg: `nthreads` independent dactyl/locked envs, one per thread, each running env.steps
 * (robot_env.py:804-844 reduced to its arithmetic: action -> ctrl, nsub x mj_step + mj_forward, two more mj_forward)
 * on iid U(-1,1) relative actions for `seconds` of wall clock.  Plain C threads, so the timing is free of interpreter
 * start-up and of 256 Python processes competing for memory. */
typedef struct {
  const ro_model* m; const double *P, *lo, *hi; const int* hand_q; int nhand, nsub, cube_z_q; double seconds; uint64_t seed;
  long steps; double elapsed;
} ro_bench_arg;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void* ro_bench_thread(void* p) {
  ro_bench_arg* a = (ro_bench_arg*)p;
  const ro_model* m = a->m;
  ro_data* d = ro_data_new(m);
  for (int i = 0; i < 30; i++) ro_step(m, d);
  uint64_t x = a->seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  double t0 = now_s();
  long n = 0;
  for (;;) {
    for (int u = 0; u < m->nu; u++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      double act = 2.0 * (double)(x >> 11) / 9007199254740992.0 - 1.0, c = 0;
      for (int j = 0; j < a->nhand; j++) c += a->P[u * a->nhand + j] * d->qpos[a->hand_q[j]];
      c += act * 0.5 * (a->hi[u] - a->lo[u]);
      d->ctrl[u] = c < a->lo[u] ? a->lo[u] : (c > a->hi[u] ? a->hi[u] : c);
    }
    ro_sim_step(m, d, a->nsub); ro_forward(m, d); ro_forward(m, d);
    n++;
    if (d->qpos[a->cube_z_q] < -0.1 || d->warn_bad) { ro_reset(m, d); d->warn_bad = 0; for (int i = 0; i < 30; i++) ro_step(m, d); }
    if (now_s() - t0 >= a->seconds) break;
  }
  a->steps = n; a->elapsed = now_s() - t0;
  ro_data_free(d);
  return NULL;
}
int ro_bench_locked(const ro_model* m, int nthreads, double seconds, const double* P, const int* hand_q, int nhand, const double* lo,
                    const double* hi, int nsub, int cube_z_q, uint64_t seed, long* steps_out, double* secs_out) {
  ro_bench_arg* args = (ro_bench_arg*)calloc(nthreads, sizeof(ro_bench_arg));
  pthread_t* th = (pthread_t*)calloc(nthreads, sizeof(pthread_t));
  int started = 0;
  for (int i = 0; i < nthreads; i++) {
    ro_bench_arg a = {m, P, lo, hi, hand_q, nhand, nsub, cube_z_q, seconds, seed + 7919ull * i, 0, 0};
    args[i] = a;
    if (pthread_create(&th[i], NULL, ro_bench_thread, &args[i]) != 0) break;
    started++;
  }
  for (int i = 0; i < started; i++) { pthread_join(th[i], NULL); steps_out[i] = args[i].steps; secs_out[i] = args[i].elapsed; }
  free(args); free(th);
  return started;
}
