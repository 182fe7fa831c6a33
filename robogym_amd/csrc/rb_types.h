// rb_types.h — descriptors of the LARGE-MODEL stepper (rb_kernel.h): models beyond the compile-time layout of the
// Shadow-hand kernel (rg_kernel.h), e.g. dactyl/full_perpendicular (BASELINE.json configs[2]: nv 168, 135 bodies,
// condim-6 contacts; /root/reference/robogym/envs/dactyl/full_perpendicular.py:92-136, cube_env.py:239-242).
// One 256-thread workgroup per env, every size a run-time number of the model, per-env stage arrays in an HBM scratch row
// (names as MuJoCo's mjData), the dense blocks of the solver in LDS.
#pragma once
#include <stdint.h>

// Two configurations of rb_kernel.h are compiled (rg_api.hip); a model runs on the smallest one that holds it (rb_model_create):
//   large: 256 threads (4 waves) per env, 4 workgroups per CU    small: 64 threads (1 wave) per env, 16 workgroups per CU
//   medium: one wave per env as well, capacities for 56 dofs (rearrange with 8 objects), 10 workgroups per CU
#define RB_T_LARGE 256
#define RB_MAXGROUP_LARGE 96    // dofs of the largest constraint-coupled group of trees: its dense block lives in LDS
#define RB_MAXNV_LARGE 192      // LDS vectors
#define RB_MAXNQ_LARGE 192
#define RB_T_SMALL 64
#define RB_MAXGROUP_SMALL 40
#define RB_MAXNV_SMALL 40
#define RB_MAXNQ_SMALL 48
#define RB_T_MEDIUM 64
#define RB_MAXGROUP_MEDIUM 56
#define RB_MAXNV_MEDIUM 56
#define RB_MAXNQ_MEDIUM 64
#define RB_TENW 8         // dofs a tendon can depend on (big_tables.py TEN_W)
#define RB_CONW 24        // dofs a contact can depend on (big_tables.py CON_W)
#define RB_STARB 5        // longest chain of a star tree (big_tables.py STAR_B)
#define RB_MLONG 8        // descendant lists of M longer than this are summed by a wave (big_tables.py MLONG)
#define RB_NW 22         // weight words of a contact in the Hessian assembly: 21 (lower triangle of a 6 x 6) + the mode

// arrays of the model blob that are uploaded as they are (field name = blob name)
#define RB_INT_ARRAYS(X) \
  X(body_parentid) X(body_rootid) X(body_weldid) X(body_jntadr) X(body_jntnum) X(body_dofadr) X(body_dofnum) \
  X(jnt_type) X(jnt_qposadr) X(jnt_dofadr) X(jnt_bodyid) \
  X(dof_bodyid) X(dof_jntid) X(dof_parentid) \
  X(geom_type) X(geom_bodyid) X(geom_dataid) X(site_bodyid) \
  X(mesh_vertadr) X(mesh_vertnum) \
  X(tendon_adr) X(tendon_num) X(wrap_type) X(wrap_objid) \
  X(actuator_trntype) X(actuator_trnid) X(actuator_forcelimited) X(actuator_biastype) \
  X(body_mocapid) X(eq_type) X(eq_obj1id) X(eq_obj2id) X(sensor_type) X(sensor_objid) X(sensor_adr) \
  X(b_lvl_body) X(b_lvl_adr) X(b_body_level) X(b_body_lastdof) X(b_subtree_adr) X(b_subtree) X(b_root_list) X(b_M_i) X(b_M_j) X(b_M_adr) \
  X(b_group_adr) X(b_group_dofs) X(b_dof_group) X(b_dof_local) X(b_pair_geom) X(b_ten_dofs) X(b_fric_dof) X(b_fric_ten) X(b_lim_jnt) X(b_lim_ten) X(b_cell_adr) X(b_Mdesc_adr) X(b_Mdesc_ent) X(b_Mdesc_dof) X(b_dof_fricrow) X(b_star_grp) X(b_tree_adr) X(b_tree_desc) X(b_tree_branch) X(b_tree_brn_end) X(b_Mlong) X(b_tree8)
#define RB_FLT_ARRAYS(X) \
  X(body_pos) X(body_quat) X(body_ipos) X(body_iquat) X(body_mass) X(body_inertia) X(body_subtreemass) X(body_invweight0) \
  X(jnt_pos) X(jnt_axis) X(jnt_stiffness) X(jnt_range) X(jnt_margin) X(jnt_solref) X(jnt_solimp) \
  X(dof_armature) X(dof_damping) X(dof_frictionloss) X(dof_solref) X(dof_solimp) X(dof_invweight0) \
  X(qpos0) X(qpos_spring) \
  X(geom_size) X(geom_rbound) X(geom_pos) X(geom_quat) X(site_pos) \
  X(wrap_prm) X(tendon_range) X(tendon_margin) X(tendon_stiffness) X(tendon_damping) X(tendon_frictionloss) X(tendon_lengthspring) \
  X(tendon_solref_lim) X(tendon_solimp_lim) X(tendon_solref_fri) X(tendon_solimp_fri) X(tendon_invweight0) \
  X(actuator_gear) X(actuator_ctrlrange) X(actuator_forcerange) X(actuator_gainprm) X(actuator_user) \
  X(eq_solref) X(eq_solimp) X(site_quat) \
  X(geom_margin) X(geom_gap) X(geom_friction) X(geom_solref) X(geom_solimp) X(geom_solmix) X(opt_gravity) \
  X(b_pair_prm) X(b_geom_aabb) X(b_mesh_rec) X(b_cell_blk) X(b_cell_ovf)

// per-env scratch row: offsets (in 4-byte words) of the stage arrays
enum {
  RB_O_XPOS, RB_O_XQUAT, RB_O_XIPOS, RB_O_XIQUAT, RB_O_XANCHOR, RB_O_XAXIS, RB_O_GPOS, RB_O_GQUAT, RB_O_SPOS, RB_O_ROOTCOM,
  RB_O_CINERT, RB_O_CRB, RB_O_CDOF, RB_O_CDOFDOT, RB_O_CVEL, RB_O_CACC, RB_O_CFRC,
  RB_O_TENLEN, RB_O_TENJ, RB_O_TENVEL, RB_O_MSP,
  RB_O_CAND, RB_O_CON, RB_O_CONJ, RB_O_CONIDX, RB_O_ROW, RB_O_DOFCON_ADR, RB_O_DOFCON, RB_O_CONF,
  RB_O_DBG, RB_O_CFRCEXT, RB_O_CONLOC, RB_O_PRM, RB_NOFF
};
// Per-env model parameters (SURVEY 8f rank 2 on this stepper; what `sim.model.<field>` writes of the reference's randomizers reach:
// /root/reference/robogym/envs/rearrange/common/base.py:1008-1092, randomization/sim.py:115-589).  A model switched on by rb_model_enable_env_params gives every
// env a block RB_O_PRM of its scratch row that holds these fields (initialised with the model's values, written by the host through rb_batch_field_ptr);
// the kernel then reads them from the block instead of from the model's arrays.  Offsets of the fields inside the block: RbModelDev.prm_off.
enum {
  RB_P_GRAVITY, RB_P_DOF_DAMPING, RB_P_DOF_ARMATURE, RB_P_DOF_FRICTIONLOSS, RB_P_DOF_INVWEIGHT0, RB_P_JNT_STIFFNESS, RB_P_JNT_MARGIN, RB_P_JNT_RANGE,
  RB_P_BODY_POS, RB_P_BODY_MASS, RB_P_BODY_INERTIA, RB_P_BODY_INVWEIGHT0, RB_P_ACT_GAINPRM, RB_P_ACT_FORCERANGE, RB_P_ACT_CTRLRANGE,
  RB_P_GEOM_POS, RB_P_GEOM_MARGIN, RB_P_GEOM_GAP, RB_P_GEOM_FRICTION, RB_P_GEOM_SOLREF, RB_P_GEOM_SOLIMP, RB_P_TENDON_RANGE, RB_P_TENDON_INVWEIGHT0, RB_NPRMF
};
// per contact record (floats): dist, pos3, frame9, includemargin, friction5, solref2, solimp5, dim, geom1, geom2, efc_address, nnz, kind
// kind: 0 pyramidal contact, 1 elliptic contact (ur16e/base.xml:3), 2 equality constraint (weld: dim 6, joint coupling: dim 1) — an
// equality is a "contact" whose rows are its basis Jacobian rows themselves: it shares the Jacobian / J'f / Hessian machinery.
// Equality records hold their residuals in frame[0..5], their diagApprox in friction[0..1] (translational, rotational) and the
// equality's index in geom1.
#define RB_CONREC 32
#define RB_CR_KIND 31
#define RB_KIND_PYRAMID 0
#define RB_KIND_ELLIPTIC 1
#define RB_KIND_EQUALITY 2
#define RB_CR_DIST 0
#define RB_CR_POS 1
#define RB_CR_FRAME 4
#define RB_CR_INCL 13
#define RB_CR_FRIC 14
#define RB_CR_SOLREF 19
#define RB_CR_SOLIMP 21
#define RB_CR_DIM 26
#define RB_CR_G1 27
#define RB_CR_G2 28
#define RB_CR_ADR 29
#define RB_CR_NNZ 30
// per constraint row (floats): D, aref, jar, jv, floss, type (0 friction dof, 1 friction tendon, 2 joint limit, 3 tendon limit, 4 pyramid edge,
// 5 basis row of an elliptic contact or of an equality), id, aux (side | pyramid edge k, sign | basis row k), then for type 5 the
// state the per-contact cone pass leaves: force, zone (0 none, 1 quadratic, 2 cone), cost (whole contact's on its first row)
#define RB_ROWREC 12
#define RB_RR_FORCE 8
#define RB_RR_ZONE 9
#define RB_RR_COST 10
#define RB_RR_D 0
#define RB_RR_AREF 1
#define RB_RR_JAR 2
#define RB_RR_JV 3
#define RB_RR_FLOSS 4
#define RB_RR_TYPE 5
#define RB_RR_ID 6
#define RB_RR_AUX 7

struct RbModelDev {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nmesh;
  int nlevel, nM, npair, ngroup, gmax, nroot, conw;
  int nfric_dof, nfric_ten, nlim_jnt, nlim_ten;
  int maxcon, maxrow, maxcand;       // capacities of the scratch row (nconmax, njmax of the model)
  int iterations, mpr_iterations, ls_iterations;
  int cone, neq, nmocap, nsensor, nsensordata;   // rearrange models: elliptic cones, equality constraints, mocap bodies, sensors
  float timestep, gravity[3], tolerance, impratio, mpr_tolerance, meaninertia;
  int off[RB_NOFF];
  int scratch_words;                 // row length
  // LDS residency of stage arrays (round 5): arrays with lds_off[k] >= 0 live in the workgroup's LDS arena (behind RbLds; offsets in words) for the whole launch
  // instead of the env's HBM scratch row, and are copied out to the row (lds_len[k] words) when the launch ends, for the env kernel and the host readers.
  int lds_off[RB_NOFF], lds_len[RB_NOFF];
  int lds_words;                     // arena size
  int prm_on;                        // per-env parameter blocks in the scratch rows (rb_model_enable_env_params)
  int prm_off[RB_NPRMF];             // word offset of each field from the start of the scratch row
  int prm_words;
#define X(n) const int* n;
  RB_INT_ARRAYS(X)
#undef X
#define X(n) const float* n;
  RB_FLT_ARRAYS(X)
#undef X
};

// env-level description: hand joint block and position -> control matrix of the action map (robot_interface.py:247-278)
struct RbEnvDev {
  int hand_qposadr, n_hand_jnt, relative_action;
  const float* pos_to_ctrl;   // [nu][n_hand_jnt]
  // relative actions of a composite robot (robot_interface.py:220-231, composite_robot.py): actuation range capped by max_position_change (> 0), and the
  // actuators whose centre is their stored control instead of a joint position (bit u: the gripper, whose range is not capped)
  float max_position_change;
  unsigned ctrl_centre_mask;
};

struct RbBatchDev {
  int B;
  float *qpos, *qvel, *ctrl, *pid, *qacc_warmstart, *time;
  float *mocap, *eq_data, *sensordata;   // [B][7 nmocap] (pos3 quat4 per mocap body), [B][7 neq], [B][nsensordata]
  int* eq_active;                        // [B][neq]
  uint32_t* status;
  float* stats;          // [B][4] accumulated ncon, nefc, Newton iterations, substeps
  float* scratch;        // [B][scratch_words]
  const float* action;   // [B][nu] or null
  const int* active;     // [B] or null
  const int* hold;       // [B] or null: envs that keep their stored ctrl row (the scripted controls of the reset recipe)
  const int* nticks;     // [B] or null: per-env count of state-less forwards (PID ticks) after the substeps
};

// The TCP solver hook (rearrange, tcp+roll+yaw through the mocap_ik dual simulation): what JointControlledTcpArm.set_position_control
// (/root/reference/robogym/robot/ur16e/mujoco/joint_controlled_tcp_arm.py:89-97) does around the solver simulation's step, run INSIDE that
// simulation's launch: arm joints <- main simulation (sync_to + forward), mocap target <- TCP pose + the denormalised action
// (free_dof_tcp_arm.py:161-206, mocap_solver.py:33-57, gym's mocap_set_action), 40 x mj_step, then main ctrl <- the solver's joint angles and
// the gripper's own position target (robot_interface.py:247-278 around the current ctrl).
struct RbTcpHook {
  int enabled, sync;                 // sync = arm_reset_controller_error
  const float* action;               // [B][6] normalised: xyz, roll, pitch(wrist), gripper
  const float* main_qpos;            // [B][main_nq]
  float* main_ctrl;                  // [B][main_nu]
  int main_nq, main_nu, main_arm_q[6], arm_q[6], main_grip_act, tcp_body, wrist_jnt;
  float max_position_change, speed[2], drift_threshold, grip_lo, grip_hi;
  // the rearrange wrapper stack's action path (envs/rearrange/common/base.py:986-996), optional
  const int* action_index;           // [B][6] bin indices (DiscretizeActionWrapper); null: `action` holds the continuous action
  const float* bins;                 // [6][nbins]
  int nbins;
  float ema_alpha;                   // SmoothActionWrapper's step-adjusted alpha
  float* ema_value; int* ema_t;      // IncrementalExpAvg state [B][6], [B]; null: no smoothing
  float* action_out;                 // [B][6] the action that reached the env (= obs["action_ema"]); may be null
  const int* hold; const float* scripted;   // [B], [B][6]: envs inside their reset recipe take a scripted continuous action (and leave the filter alone); may be null
  int wrist_only;                    // control_mode tcp+wrist: no roll, the commanded orientation aligned with the vertical (MocapSolver.align_axis)
  int self_world;                    // tcp_solver_mode mocap: this world IS the env's world (no sync, no hand-over; the gripper target goes into its own ctrl before the steps)
  const int* skip;                   // [B] or null: envs whose robot gets no command in this launch
};
struct RbLaunch { RbEnvDev env; RbBatchDev bt; int nsubsteps, nforward_ticks, flags; RbTcpHook tcp; };
// several batches of several models stepped by ONE launch (rb_step_multi_kernel; one-wave configurations): the whole record travels in the kernel argument segment
#define RB_MAXMULTI 8
struct RbMultiLaunch { int n, group_size; const RbModelDev* m[RB_MAXMULTI]; RbLaunch L[RB_MAXMULTI]; };
static_assert(sizeof(RbMultiLaunch) <= 4096, "RbMultiLaunch must fit the kernel argument segment");
