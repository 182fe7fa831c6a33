// rg_types.h — device-side model / batch descriptors shared by the C-ABI host code (rg_api.hip)
// and the stepper kernel (rg_kernel.h).  All device arrays are fp32 / int32.
#pragma once
#include <stdint.h>

// compile-time capacities of the "hand" kernel configuration (dactyl/locked, dactyl/reach); the contact and
// candidate capacities (RG_MAXCON, RG_CPOOL, RG_MAXCAND, RG_MAXCAND2) belong to the kernel configurations of rg_api.hip
#define RG_MAXNQ 40
#define RG_MAXNV 36
#define RG_MAXBODY 32
#define RG_MAXJNT 32
#define RG_MAXGEOM 66
#define RG_MAXSITE 40
#define RG_MAXTEN 12
#define RG_MAXU 20
#define RG_MAXFRIC 32   // friction-loss rows (dofs + tendons)
#define RG_MAXSROW 96   // static row slots: friction rows + 2 per limited joint / tendon
#define RG_MAXNVC 32    // dofs in constrained kinematic trees (the Newton space)
#define RG_HWORDS 1124  // solver work matrix: max(per-tree inertia blocks, (nvc + 1) x hs: the extra row carries the right-hand side through the dense factorisation)
#define RG_MAXNM 160    // non-zeros of the tree-sparse inertia matrix: (dof, ancestor) pairs
#define RG_CELLN 8      // direction cells: cube map, RG_CELLN x RG_CELLN per face (kernel_tables.py CELL_N)
#define RG_NCELL (6 * RG_CELLN * RG_CELLN)
#define RG_LTDL_TRI_ROUNDS 12   // caps on the descriptor rounds of the L'DL passes (registers per lane)
#define RG_LTDL_PAIR_ROUNDS 8
#define RG_KINREC 20    // words per kinematics record
#define RG_MAXDEPTH 8   // moving bodies on a root-to-leaf chain
#define RG_MAXSENSOR 8
#define RG_PAIRREC 28   // words per pair record
#define RG_PAIR_SCALED1 (1 << 28)   // pair record header: geom 1 / geom 2 takes the env's RG_PRM_GEOM_SCALE
#define RG_PAIR_SCALED2 (1 << 29)
#define RG_TLIST 320    // pairs whose distance bound ran out, queued for the sphere/box tests (drained in chunks)
#define RG_MAXROW 64    // friction-loss + limit rows
#define RG_W 14         // max nonzeros of a sparse constraint row
#define RG_WAVE 64

// Per-env model parameters (what the reference's simulation randomizers write into `sim.model` per
// episode, randomization/sim.py:115-589, wrappers/randomizations.py:72-310,562-746): one row of RG_NPRM floats per env,
// laid out by the compile-time capacities; the model carries a default row (its own values) for batches without overrides.
enum {
  RG_PRM_GRAVITY = 0,                                      // 3   opt.gravity
  RG_PRM_TIMESTEP = 3,                                     // 1   opt.timestep
  RG_PRM_DOF_DAMPING = 4,                                  // nv
  RG_PRM_DOF_ARMATURE = RG_PRM_DOF_DAMPING + RG_MAXNV,
  RG_PRM_DOF_FRICTIONLOSS = RG_PRM_DOF_ARMATURE + RG_MAXNV,
  RG_PRM_DOF_INVWEIGHT0 = RG_PRM_DOF_FRICTIONLOSS + RG_MAXNV,   // (mj_setConst output: follows mass / inertia / armature)
  RG_PRM_BODY_MASS = RG_PRM_DOF_INVWEIGHT0 + RG_MAXNV,     // nbody
  RG_PRM_BODY_INERTIA = RG_PRM_BODY_MASS + RG_MAXBODY,     // 3 nbody
  RG_PRM_BODY_INVWEIGHT0 = RG_PRM_BODY_INERTIA + 3 * RG_MAXBODY,   // 2 nbody (mj_setConst output)
  RG_PRM_JNT_RANGE = RG_PRM_BODY_INVWEIGHT0 + 2 * RG_MAXBODY,      // 2 njnt
  RG_PRM_TENDON_RANGE = RG_PRM_JNT_RANGE + 2 * RG_MAXJNT,  // 2 ntendon
  RG_PRM_TENDON_INVWEIGHT0 = RG_PRM_TENDON_RANGE + 2 * RG_MAXTEN,  // ntendon (mj_setConst output)
  RG_PRM_ACT_GAINPRM = RG_PRM_TENDON_INVWEIGHT0 + RG_MAXTEN,       // 10 nu: kp, ti, imax_clamp, td, dsmooth, error_deadband, ...
  RG_PRM_ACT_CTRLRANGE = RG_PRM_ACT_GAINPRM + 10 * RG_MAXU,        // 2 nu
  RG_PRM_ACT_FORCERANGE = RG_PRM_ACT_CTRLRANGE + 2 * RG_MAXU,      // 2 nu
  RG_PRM_GEOM_FRICTION = RG_PRM_ACT_FORCERANGE + 2 * RG_MAXU,      // 3 ngeom: sliding, torsional, rolling (a contact takes the element-wise max of its geoms)
  RG_PRM_XFRC = RG_PRM_GEOM_FRICTION + 3 * RG_MAXGEOM,     // 6 nbody: data.xfrc_applied (force, torque in world coordinates at the body's com)
  RG_PRM_SITE_POS = RG_PRM_XFRC + 6 * RG_MAXBODY,          // 3 nsite: model.site_pos (marker placement, wrappers/dactyl.py:14-50)
  RG_PRM_GEOM_SCALE = RG_PRM_SITE_POS + 3 * RG_MAXSITE,    // 1   size factor of the geoms flagged in k_geom_scaled (the cube: wrappers/cube.py:12-53)
  RG_PRM_JNT_MARGIN = RG_PRM_GEOM_SCALE + 1,               // njnt     model.jnt_margin (JointMarginRandomizer, randomization/sim.py:163-180)
  RG_PRM_GEOM_SOLREF = RG_PRM_JNT_MARGIN + RG_MAXJNT,      // 2 ngeom  model.geom_solref (GeomSolrefRandomizer, :271-315): a contact mixes its two geoms' by solmix
  RG_PRM_GEOM_SOLIMP = RG_PRM_GEOM_SOLREF + 2 * RG_MAXGEOM, // 5 ngeom  model.geom_solimp (GeomSolimpRandomizer, :183-268)
  RG_NPRM = RG_PRM_GEOM_SOLIMP + 5 * RG_MAXGEOM
};

// `sim.data` readout row (mujoco_shadow_hand.py:18-61 reads site_xpos / actuator_force, simulation/base.py and
// sensor_utils.py read contacts, robot_env.py observation providers read body poses): per env, compile-time layout
enum {
  RG_XD_XPOS = 0,                                   // 3 nbody   data.xpos (body_xpos)
  RG_XD_XQUAT = RG_XD_XPOS + 3 * RG_MAXBODY,        // 4 nbody   data.xquat
  RG_XD_SITE_XPOS = RG_XD_XQUAT + 4 * RG_MAXBODY,   // 3 nsite   data.site_xpos
  RG_XD_ACT_FORCE = RG_XD_SITE_XPOS + 3 * RG_MAXSITE,   // nu    data.actuator_force (of the last state-less forward)
  RG_XD_NCON = RG_XD_ACT_FORCE + RG_MAXU,           // 1         data.ncon of the last mj_step
  RG_XD_CONTACT = RG_XD_NCON + 1,                   // RG_DBG_MAXCON x (geom1, geom2, dist)
  RG_XD_SENSOR = RG_XD_CONTACT + 3 * 32,            // RG_MAXSENSOR  data.sensordata (touch sensors; launches with flags bit 5)
  RG_XDATA = RG_XD_SENSOR + 8
};

// per-env sticky status bits (replace MuJoCo's warning callback, warning_buffer.py:27-83)
#define RG_STATUS_BAD_STATE 1u   // NaN/inf or |x|>1e10 in qpos/qvel/qacc
#define RG_STATUS_CON_FULL 2u    // more contacts than RG_MAXCON
#define RG_STATUS_CAND_FULL 4u   // more broadphase candidates than RG_MAXCAND
#define RG_STATUS_ROW_FULL 8u    // more friction/limit rows than RG_MAXROW
#define RG_STATUS_BAD_FACTOR 16u // Cholesky pivot <= 0
#define RG_STATUS_BAD_ACTION 32u // non-finite entry in the env's action row (the row is ignored: ctrl keeps its value)
#define RG_STATUS_SCHED 64u      // substep-granular dispatch: work items of this env were left undrawn; the env.step was completed by the fallback (include/rgstep.h)

enum { RG_JNT_FREE = 0, RG_JNT_BALL = 1, RG_JNT_SLIDE = 2, RG_JNT_HINGE = 3 };
enum { RG_GEOM_PLANE = 0, RG_GEOM_SPHERE = 2, RG_GEOM_CAPSULE = 3, RG_GEOM_ELLIPSOID = 4, RG_GEOM_CYLINDER = 5, RG_GEOM_BOX = 6, RG_GEOM_MESH = 7 };
enum { RG_WRAP_JOINT = 1, RG_WRAP_PULLEY = 2, RG_WRAP_SITE = 3, RG_WRAP_SPHERE = 4, RG_WRAP_CYLINDER = 5 };

struct RgModelDev {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nwrap, nmesh;
  int nlevel, ndoflevel, nM, npair, nstatic, nfric_dof, nfric_ten, nlim_jnt, nlim_ten;
  int nvc, hs, blkwords, ntree, maxtree;  // Newton space size, its row stride, inertia block words, trees, largest tree
  const int *dof_blk, *dof_blk2, *c_blk, *d2c, *c2d;
  int iterations, mpr_iterations, cone;
  float timestep, gravity[3], tolerance, impratio, mpr_tolerance, meaninertia;
  // bodies
  const int *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_lastdof;
  const float *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0;
  const int *ltdl_tri, *ltdl_pair;   // tree-sparse L'DL passes of M: rounds of 64 descriptors (kernel_tables.py)
  int n_tri_rounds, n_pair_rounds;
  const int *ltdl_tri_c, *ltdl_pair_c;   // the same for the constrained trees, vectors indexed by compact dof (Newton Hessian with tree pattern)
  int n_tri_rounds_c, n_pair_rounds_c, tree_newton_ok;
  const int* subtree_mask;      // [nbody] bit c: body c belongs to the subtree rooted at the body (self included)
  const float* body_rec;        // [nbody][RG_KINREC] per moving body: body, parent, first joint and their constants (rg_api.hip)
  const int* body_depth;        // [nbody] moving bodies on the chain from the root to this body, itself included (0: static body) | bit i of the next byte: chain element i has a free joint (its frame is absolute) << 8 | the body the chain hangs off (world / static) << 16
  const int* body_chain;        // [nbody][2] that chain, root first, one byte per body
  const float *static_xpos, *static_xquat;
  const int *root_origin_body, *body_orgslot;
  const float* root_origin_const;
  const float* org_rec;         // [4][4] per com-frame slot: origin body (int; -1: constant, -2: slot unused), then that body's ipos or the constant (rg_api.hip)
  const uint32_t* body_dofmask;  // [nbody][2]
  // joints / dofs
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid;
  const float *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  const int *dof_bodyid, *dof_jntid, *dof_parentid;
  const float *dof_armature, *dof_damping, *dof_frictionloss, *dof_solref, *dof_solimp, *dof_invweight0;
  const float *qpos0, *qpos_spring;
  const int *lvl_dof, *lvl_dof_adr, *M_i, *M_j, *M_lvl_adr, *desc_adr, *desc;
  const float* prm_default;   // [RG_NPRM] the model's own values in the per-env parameter layout (RG_PRM_*)
  const int* M_ent;   // [nM][2]: block-layout word of (i,j) | word of (j,i) << 16 ; compact i | compact j << 8 | constrained-tree << 16
  // geoms / sites / meshes
  const int *geom_type, *geom_bodyid, *geom_dataid, *body_geomadr, *body_geomnum;
  const float *geom_size, *geom_rbound, *geom_pos, *geom_quat, *geom_aabb;
  const int* site_bodyid;
  const float *site_pos;
  int nsensor;                  // touch sensors (mjSENS_TOUCH): site ids, and the shape of every site
  const int *sensor_site, *site_type;
  const float *site_size, *site_quat;
  const int *mesh_vertadr, *mesh_vertnum;
  const float* mesh_vert;
  // hull vertices that can be the support point for a direction in the cell (ascending vertex index):
  const int* mesh_cell_adr;     // [nmesh][RG_NCELL] first overflow record << 8 | count
  const float* mesh_cell_blk;   // [nmesh][RG_NCELL][4] 16-byte records x, y, z, vertex index (int bits): the first four (padded with the last)
  const float* mesh_cell_ovf;   // records 5.. of the cells that have more than four
  // everything the collision stages need about a pair in one place (one load latency instead of a chain)
  const int* pair_gg;           // [npair] g1 | g2 << 8
  const float* pair_rec;        // [npair][RG_PAIRREC] see rg_api.hip build_pair_records
  const int* pair_geom;   // [npair][3] g1, g2, condim
  const float* pair_prm;  // [npair][12] margin, gap, friction3, solref2, solimp5
  const float* pair_mix;  // [npair] solmix weight of geom 1 in the pair's solref / solimp (mj_contactParam)
  // tendons
  const int *tendon_adr, *tendon_num, *wrap_type, *wrap_objid, *ten_dofs;
  const int *ten_path, *ten_path_adr;   // per tendon: 8-word path records (kernel_tables.py k_ten_path), [ntendon + 1] offsets
  const float *wrap_prm, *tendon_range, *tendon_margin, *tendon_stiffness, *tendon_damping, *tendon_frictionloss,
      *tendon_lengthspring, *tendon_solref_lim, *tendon_solimp_lim, *tendon_solref_fri, *tendon_solimp_fri, *tendon_invweight0;
  const int *dof_ten_adr, *dof_ten, *dof_act_adr, *dof_act;
  const float* srow_rec;        // [static rows][16] friction-loss / limit rows: Jacobian descriptor, what they read and their solver parameters (rg_api.hip)
  const float* dof_rec;         // [nv][8] per dof: body, tendon / actuator entry ranges, joint stiffness, qposadr, spring reference (rg_api.hip)
  const int* M_ijb;             // [nM] dof i | dof j << 8 | body of dof i << 16
  // actuators
  const int *actuator_trntype, *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited, *actuator_biastype;
  const float *actuator_gear, *actuator_ctrlrange, *actuator_forcerange, *actuator_gainprm, *actuator_biasprm;
  // constraint row sources
  const int *fric_dof, *fric_ten, *lim_jnt, *lim_ten;
};

// env-level description of the dactyl "cube in hand" task family (obs readout, goal, reward)
struct RgEnvDev {
  int hand_qposadr;      // first of the 24 hand joint angles in qpos
  int n_hand_jnt;        // 24
  int cube_pos_qposadr;  // 3 slides
  int cube_quat_qposadr; // ball
  int target_qposadr, target_nq, target_dofadr, target_nv;  // zeroed in the qpos/qvel observations
  int cube_body;         // body whose xpos is the cube_pos observation
  int ref_site[3];       // phasespace reference sites
  int tip_site[5];       // fingertip sites
  int relative_action;   // 1: action centred on current joint positions
  const float* pos_to_ctrl;  // [nu][n_hand_jnt] joint positions -> control (device)
  float success_threshold;   // cube_quat < 0.4 rad
};

// per-batch device buffers, row-major [B][n]
struct RgBatchDev {
  int B;
  float *qpos, *qvel, *ctrl, *pid, *qacc_warmstart, *time;
  uint32_t* status;
  float* sepdir;        // [B][npair][4] cached separating direction per candidate pair (pure cache, not state)
  float* pairlb;        // [B][npair] lower bound on the distance of the pair's (margin-inflated) geoms, decremented by
                        // a motion bound every substep; pairs with a positive bound skip all collision tests.
                        // Zeroed (= unknown) whenever qpos is written from outside.
  // env-step I/O
  const float* action;  // [B][nu] in [-1,1]   (may be null: ctrl used as is)
  const float* goal_quat;  // [B][4]
  const int* active;    // [B] or null: envs with 0 are skipped by this launch
  const int* hold;      // [B] or null: envs with != 0 ignore their action row and keep the stored ctrl row (scripted resets)
  const int* nticks;    // [B] or null: per-env override of nforward_ticks (the reset recipe's sim.step has 1, env.step 3)
  const int* order;     // [B] or null: workgroup -> env permutation (longest-expected-first dispatch)
  float* cost;          // [B] or null: shader cycles this launch spent on the env (feeds `order` of the next step)
  float* xdata;         // [B][RG_XDATA] or null: the mjData fields in-tree callers read after a step (RG_XD_*)
  const float* envprm;  // [B][RG_NPRM] or null: per-env model parameters (null: every env uses the model's)
  int* preticks;        // [B] or null: state-less forwards owed from the previous step's goal reset (run before the action is applied, then zeroed)
  int* redo;            // [B] or null: an env that exceeds this configuration's contact / candidate capacities is left
                        // untouched and flagged here, to be stepped again by a launch of the large configuration
  float* obs;           // [B][obs_dim]
  float* goal_dist;     // [B]
  float* stats;         // [B][4]: sum ncon, sum nefc, sum newton iters, substeps
  float* dbg;           // optional [B][RG_DBG_SIZE] stage dump of the first forward pass
  int* sched;           // [RG_SCHED_PROG + B] substep-granular dispatch (rg_step_items_kernel): queue heads (one per XCD, 64 B apart), exit counter,
                        // then per env the number of substeps of the current launch that are complete (| RG_SCHED_BAD)
};
#define RG_SCHED_FIN 128
#define RG_SCHED_PROG 192
#define RG_SCHED_BAD (1 << 16)

// debug-dump layout (floats)
#define RG_DBG_XPOS 0                                   // nbody*3
#define RG_DBG_XQUAT (RG_DBG_XPOS + 32 * 3)     // nbody*4
#define RG_DBG_SITE (RG_DBG_XQUAT + 32 * 4)     // nsite*3
#define RG_DBG_M (RG_DBG_SITE + 40 * 3)         // nv*nv dense
#define RG_DBG_TENLEN (RG_DBG_M + 40 * 40)  // ntendon
#define RG_DBG_TENJ (RG_DBG_TENLEN + RG_MAXTEN)         // ntendon*4
#define RG_DBG_BIAS (RG_DBG_TENJ + RG_MAXTEN * 4)       // nv
#define RG_DBG_PASSIVE (RG_DBG_BIAS + 40)         // nv
#define RG_DBG_ACTFRC (RG_DBG_PASSIVE + 40)       // nv  qfrc_actuator
#define RG_DBG_QACCS (RG_DBG_ACTFRC + 40)         // nv  qacc_smooth
#define RG_DBG_QACC (RG_DBG_QACCS + 40)           // nv
#define RG_DBG_NCON (RG_DBG_QACC + 40)            // 1: ncon, +1: nefc, +2: iters, +3: ncand
#define RG_DBG_CON (RG_DBG_NCON + 4)                    // MAXCON * 8: dist, pos3, normal3, pair index
#define RG_DBG_MAXCON 32                               // contacts in the dump (a configuration may keep more)
#define RG_DBG_SIZE (RG_DBG_CON + RG_DBG_MAXCON * 8)
