/*
 * rgstep.h — C ABI of the MI355X-native batched physics stepper ("librgstep").
 *
 * The reference (openai/robogym) has no FFI of its own: its seam to native code is class
 * substitution around a mujoco_py.MjSim (robogym/mujoco/mujoco_xml.py:249-260 builds it,
 * robogym/mujoco/simulation_interface.py:25-250 wraps it).  Each entry point below names the
 * reference interface it replaces.  All pointers are plain C pointers; `*_dev` arguments are
 * device (HBM) addresses, e.g. torch.Tensor.data_ptr().  No torch types cross this boundary.
 *
 * Ownership: the library owns models and batches (state rows live in HBM inside the batch);
 * callers own the I/O buffers they pass in.  Errors: negative return codes + rg_last_error().
 * Per-env problems (NaN state, contact/row overflow) are sticky status bits (rg_batch_copy with
 * RG_F_STATUS), replacing MuJoCo's process-global warning callback
 * (robogym/mujoco/warning_buffer.py:27-83).  Threading: a batch may be used from one thread at a
 * time; work is asynchronous on the HIP stream passed to rg_batch_step.
 */
#ifndef RGSTEP_H
#define RGSTEP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_model rg_model;
typedef struct rg_batch rg_batch;

/* state / diagnostic fields addressable through rg_batch_copy */
enum {
  RG_F_QPOS = 0,      /* float [B][nq]     sim.data.qpos  (simulation_interface.py:128-150 get/set_qpos)   */
  RG_F_QVEL = 1,      /* float [B][nv]     sim.data.qvel                                                   */
  RG_F_CTRL = 2,      /* float [B][nu]     sim.data.ctrl  (mujoco_shadow_hand.py:120-137)                  */
  RG_F_PID = 3,       /* float [B][3*nu]   sim.data.userdata slice used by mjpid (integral, last error, last derivative) */
  RG_F_WARMSTART = 4, /* float [B][nv]     sim.data.qacc_warmstart                                         */
  RG_F_TIME = 5,      /* float [B]         sim.data.time                                                   */
  RG_F_STATUS = 6,    /* uint32 [B]        sticky per-env status bits (RG_STATUS_*)                        */
  RG_F_STATS = 7,     /* float [B][4]      accumulated ncon, nefc, Newton iterations, substeps             */
  RG_F_DEBUG = 8,     /* float [B][rg_debug_size()] stage dump of the first substep (flags & 1)            */
  RG_F_COST = 9,      /* float [B]         shader cycles the last rg_batch_step_ex spent on the env (dispatch ordering) */
  RG_F_PAIRLB = 10,   /* float [B][npair]  collision cache: lower bounds on the pair distances (0 = unknown). Not state. */
  RG_F_ENVPRM = 11    /* float [B][rg_prm_layout()[0]] per-env model parameters, after rg_batch_enable_env_params          */
};

#define RG_STATUS_BAD_STATE 1u
#define RG_STATUS_CON_FULL 2u
#define RG_STATUS_CAND_FULL 4u
#define RG_STATUS_ROW_FULL 8u
#define RG_STATUS_BAD_FACTOR 16u
#define RG_STATUS_BAD_ACTION 32u /* a non-finite entry in the env's action row: the row was ignored (ctrl kept) */
#define RG_STATUS_SCHED 64u      /* substep-granular dispatch (flags bit 7): the rollout launch ended with work items of this env undrawn (its XCD's queue was not
                                    served); the env.step was completed by the large-configuration launch behind it (redo_dev).  Informational, sticky. */

/* Replaces mujoco_py.load_model_from_xml (mujoco_xml.py:259): `blob` is the "RGMODEL1" flat model
 * produced by the host-side MJCF compiler (robogym_amd/mujoco/model_blob.py). Returns NULL on error. */
/* ---- The model blob (format "RGMODEL1"; what mujoco_py.load_model_from_xml's result is to the reference, mujoco/mujoco_xml.py:249-260).
 * MJCF compilation is host-side (robogym_amd/mujoco/mjcf_compiler.py + setconst.py + kernel_tables.py / big_tables.py; reachable through this ABI as
 * rg_compile_mjcf / rb_compile_mjcf below, which run it as a helper process) and its product is this self-describing container, which any language can assemble:
 *   bytes 0..7   "RGMODEL1"        bytes 8..11  uint32 entry count n        bytes 12..15  reserved (0)
 *   then n directory entries of 56 bytes: char name[40] (NUL padded) | uint32 dtype (0 = float64, 1 = int32, 2 = float32) | uint32 element count |
 *   uint64 byte offset of the data from the start of the blob (8-byte aligned); the arrays follow, little endian, C order.
 * Arrays carry mjModel's names and layouts (body_pos [nbody][3], jnt_range [njnt][2], ...; `dims` = nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon,
 * nwrap, nmesh, nmeshvert, nexclude, nsensor) plus the derived execution tables k_* (rg_model_create) / b_* (rb_model_create).  The exact key set a model
 * kind needs is reported by the reader itself: rg_model_blob_keys / rb_model_blob_keys return the comma-separated names the create call asked for
 * (return value: bytes needed incl. NUL); a missing or mistyped key fails the create with an error that names it.  rg_blob_entry enumerates a blob's
 * directory (index < 0: only the entry count is returned) with the same bounds checks the creates apply. */
int rg_blob_entry(const void* blob, size_t nbytes, int index, char* name40, int* dtype, unsigned* count);
/* ---- MJCF across the boundary (SURVEY 8b `rg_compile_mjcf`).  Replaces MujocoXML.build -> mujoco_py.load_model_from_xml(xml_string)
 * (/root/reference/robogym/mujoco/mujoco_xml.py:249-260): `xml` is the merged MJCF document as a NUL-terminated string, `meshdir` the directory mesh file
 * names are relative to (NULL or "": the document's <compiler meshdir>).  The MJCF compiler is the package's Python module: the library runs it as a helper
 * process -- `$RGSTEP_PYTHON` (default "python3") `-m robogym_amd.mujoco.compile_cli`, with the package root (three directories above the library file, or
 * `$RGSTEP_PYTHONPATH`) prepended to PYTHONPATH -- which writes the RGMODEL1 blob to a private temporary directory; the library reads it back, removes the
 * directory and creates the model from it.  Blocking, host only (no GPU work before the create), seconds for a Shadow-hand document (mesh hulls).
 * Errors (compiler unavailable, a feature outside the supported MJCF subset, a missing mesh) return NULL with the compiler's message in `err`.
 *   rg_compile_mjcf       -> a model of the Shadow-hand layout (as rg_model_create)
 *   rb_compile_mjcf       -> a model of the large-model stepper (as rb_model_create)
 *   rg_compile_mjcf_blob  -> only the blob (kind 0: with the k_* tables rg_model_create reads; 1: with the b_* tables of rb_model_create), malloc'ed,
 *                            released with rg_blob_free: compile once, create on several devices (rg_model_create_on). */
rg_model* rg_compile_mjcf(const char* xml, const char* meshdir, char* err, int errlen);
int rg_compile_mjcf_blob(const char* xml, const char* meshdir, int kind, void** blob_out, size_t* nbytes_out, char* err, int errlen);
void rg_blob_free(void* blob);
int rg_model_blob_keys(const rg_model* m, char* out, int outlen);
rg_model* rg_model_create(const void* blob, size_t nbytes, char* err, int errlen);
/* The same with the HIP device ordinal that shall hold the model tables (rg_model_create uses the calling
 * thread's current device).  A batch must be created on the device of its model. */
rg_model* rg_model_create_on(const void* blob, size_t nbytes, int device, char* err, int errlen);
void rg_model_free(rg_model* m);
/* sizes: out[0..4] = nq, nv, nu, nbody, nsite */
int rg_model_dims(const rg_model* m, int* out);

/* Replaces constructing B independent MjSim objects (robot_env.py:328-350: one env <-> one sim).
 * State is initialised as MjSim.reset() would (qpos0, zeros). `device` is the HIP device ordinal. */
rg_batch* rg_batch_create(const rg_model* m, int B, int device);
void rg_batch_free(rg_batch* b);

/* Per-env model parameters: the batched form of what the reference's simulation randomizers and physics wrappers write
 * into `sim.model` / `sim.data` of ONE env per episode (randomization/sim.py:115-589, wrappers/randomizations.py:72-310 and
 * 562-746, wrappers/cube.py:12-85): opt.gravity, opt.timestep, dof_damping / armature / frictionloss, body_mass / inertia,
 * jnt_range, tendon_range, actuator_gainprm / ctrlrange / forcerange, geom_friction, data.xfrc_applied, site_pos
 * (marker placement, wrappers/dactyl.py:14-50), one size factor for the geoms the model flags in `k_geom_scaled` (the cube,
 * wrappers/cube.py:12-53; boxes only), plus the
 * mj_setConst outputs that follow mass changes (dof / body / tendon _invweight0: recomputed on the device by
 * rg_batch_set_constants), jnt_margin, and geom_solref / geom_solimp (JointMarginRandomizer, GeomSolrefRandomizer,
 * GeomSolimpRandomizer: randomization/sim.py:163-315; a contact mixes its two geoms' values by solmix).  rg_batch_enable_env_params allocates one row per env, initialised with the model's own
 * values; rows are read and written through rg_batch_field_ptr(RG_F_ENVPRM) (device pointer).  rg_prm_layout fills
 * out[0] = row length, then the offsets of gravity, timestep, dof_damping, dof_armature, dof_frictionloss, dof_invweight0,
 * body_mass, body_inertia, body_invweight0, jnt_range, tendon_range, tendon_invweight0, actuator_gainprm (10 per
 * actuator), actuator_ctrlrange, actuator_forcerange, geom_friction (3 per geom), xfrc_applied (6 per body: force, torque),
 * site_pos (3 per site), geom_scale (1), jnt_margin, geom_solref (2 per geom), geom_solimp (5 per geom) and returns the number of entries.  Arrays are dense by the model's own counts (e.g. dof_damping[d] at offset + d). */
int rg_batch_enable_env_params(rg_batch* b);
int rg_prm_layout(int* out, int n);
/* Task description for the dactyl cube-in-hand family: which qpos slices / sites feed the action
 * map and the observation row (robot_env.py:497-504, robot_interface.py:247-278,
 * hand_interface.py:245-266 and 399-405, envs/dactyl/observation/ *.py).
 * ints[16]: hand_qposadr, n_hand_jnt, cube_pos_qposadr, cube_quat_qposadr, target_qposadr, target_nq,
 *           target_dofadr, target_nv, cube_body, ref_site[3], tip_site[5] (=19 ints), relative_action
 * pos_to_ctrl: host float [nu][n_hand_jnt]. */
int rg_batch_set_env(rg_batch* b, const int* ints, int nints, const float* pos_to_ctrl_host, float success_threshold);

/* MjSim.reset / set_state / get_state (simulation_interface.py:154-172,191-197).
 * to_batch != 0 copies ptr -> batch field, else batch field -> ptr; ptr_is_device selects HBM vs host.
 * Both are slow-path, synchronous calls: they wait for all work queued on the device (whatever the stream) and return when
 * the copy is complete, so they are ordered with step launches on any stream.  The hot path writes rows through
 * rg_batch_field_ptr views / rg_batch_copy_rows (stream-ordered, asynchronous). */
int rg_batch_copy(rg_batch* b, int field, void* ptr, int to_batch, int ptr_is_device);
int rg_batch_reset(rg_batch* b);
/* Masked row copy, device to device and asynchronous on `stream`: for every env e with mask_dev[e] != 0 the
 * columns [col0, col0 + ncols) of its `field` row are replaced by row e of src_dev ([B][ncols]; ncols <= 0: the
 * whole row, src_dev laid out as for rg_batch_copy).  Writing RG_F_QPOS voids the collision caches of those
 * envs only.  This is what a per-env `MjSim.set_state` / `mj_resetData` is for
 * a batch (simulation_interface.py:128-172, cube_env.py:330-355) without stalling the other envs. */
int rg_batch_copy_rows(rg_batch* b, int field, const void* src_dev, const int* mask_dev, int col0, int ncols, void* stream);

/* One env.step of the whole batch (robot_env.py:804-844 + simulation_interface.py:176-189):
 *   action_dev   float [B][nu] in [-1,1] or NULL (then the stored ctrl row is used unchanged); a row whose first
 *                entry is NaN keeps that env's stored ctrl row (envs that are being reset follow a scripted ctrl)
 *   goal_quat_dev float [B][4] or NULL
 *   obs_dev      float [B][rg_obs_dim] or NULL:  cube_pos3 cube_quat4 qpos[nq] qvel[nv] hand_angle[nh] fingertip_pos15
 *   goal_dist_dev float [B] or NULL
 *   active_dev   int [B] or NULL: envs whose entry is 0 are left untouched by this call (masked
 *                resets, per-env goal resets; the reference simply does not call step on those envs)
 *   nsubsteps    mj_step calls (MjSim.nsubsteps, mujoco_xml.py:249-260 / robot_env.py:132)
 *   nforward_ticks number of state-less mj_forward calls the reference makes afterwards (3 per env.step)
 *   flags        bit0: write the RG_F_DEBUG stage dump of the first substep
 *                bit1: accumulate per-stage cycle counters into the RG_F_DEBUG contact region (profiling)
 *                bit2: ignore the cached pair distance bounds (every pair goes through the sphere/box tests
 *                      each substep; results are bit-identical either way — test hook for that claim)
 *                bit3: hull support points by scanning every vertex instead of the per-direction-cell
 *                      candidate lists (bit-identical as well; test hook)
 *                bit4: contact depth / normal of convex pairs from the final MPR portal's PLANE instead of libccd's closest
 *                      point on the final portal TRIANGLE (the default, = MuJoCo 2.0's mjc_Convex).  The two agree whenever
 *                      the origin projects inside the triangle; the plane variant does not depend on the rounding-level
 *                      tie breaks that pick the triangle on flat contacts.  The flag selects round 1's contact generation as a
 *                      whole: pairs of two boxes then go through MPR too (one contact) instead of the multi-point box-box
 *                      routine (the default, = mjc_BoxBox: face clipping, up to 8 contacts per pair).
 *                bit5: evaluate data.sensordata (touch sensors, mj_sensorAcc) into the xdata row: the last of the state-less
 *                      forwards then runs in full (collision and constraint solve at the final state, as mj_forward does),
 *                      about one more substep of work.  Needs xdata_dev and nforward_ticks >= 1.
 *                bit6: dense Newton step with the forward substitution as its own pass instead of inside the
 *                      factorisation (same result to rounding; test hook)
 *                bit7: substep-granular dispatch (RG_CFG_ROLLOUT, nsubsteps >= 2, no bit 0 / 1 / 5): the launch is a grid of
 *                      persistent workgroups that draw (env, substep) work items in substep-major order instead of one workgroup
 *                      per env.step, so the device stays full to the end of the launch; bit-identical results.  An env that
 *                      exceeds the rollout capacities in substep s gets redo_dev[e] = s + 1.  Ignored where the device probe
 *                      at rg_batch_create found it unavailable (rg_batch_items_info).
 *                bit9: test hook for the hand-over: a rollout-configuration substep with more than 5 contacts counts as exceeding the
 *                      capacities (the env goes to redo_dev although it would fit)
 *                bit8: active_dev is the redo_dev array of a rollout launch: entry - 1 is the first substep still to do
 *                      (the env.step is resumed there; entry 1 = from the start)
 *   stream       hipStream_t (NULL = default stream).  Asynchronous. */
int rg_batch_step(rg_batch* b, const float* action_dev, const float* goal_quat_dev, float* obs_dev, float* goal_dist_dev,
                  const int* active_dev, int nsubsteps, int nforward_ticks, int flags, void* stream);
/* The same launch with every optional per-env input spelled out.  Arrays are device pointers or NULL.
 *   hold_dev    int [B]: envs with != 0 ignore their action row and keep the stored ctrl row (what the reference
 *               does while a reset recipe scripts ctrl, locked.py:197-225); replaces the in-band NaN convention
 *   nticks_dev  int [B]: per-env number of state-less forwards after the substeps (overrides nforward_ticks):
 *               `SimulationInterface.step` has one (simulation_interface.py:185), `RobotEnv.step` three
 *   order_dev   int [B]: permutation, workgroup i steps env order_dev[i]; with RG_F_COST of the previous step
 *               sorted descending the longest envs are dispatched first (shorter tail of the launch)
 * A row of action_dev with a non-finite entry raises RG_STATUS_BAD_ACTION for that env and is ignored. */
typedef struct rg_step_args {
  const float* action_dev; const float* goal_quat_dev; float* obs_dev; float* goal_dist_dev;
  const int* active_dev; const int* hold_dev; const int* nticks_dev; const int* order_dev;
  int nsubsteps, nforward_ticks, flags;
  void* stream;
  /* Kernel configuration: RG_CFG_ROLLOUT holds 24 contacts / 768 Jacobian words per env in LDS (rollouts: mean 3.5
   * contacts, P(> 21) < 2e-6 per mj_step; 12 envs in flight per CU), RG_CFG_LARGE 64 / 2048 (8 per CU; the reset recipe, where the
   * hand closes around a freshly dropped cube; MuJoCo's nconmax for these models is 100).
   * redo_dev int [B] (RG_CFG_ROLLOUT only, may be NULL): an env that exceeds the rollout capacities is left untouched
   * and gets redo_dev[e] = 1 instead of dropping contacts; pass the array as active_dev of an RG_CFG_LARGE launch. */
  int config;
  int* redo_dev;
  /* int [B] or NULL: state-less forwards owed to the env from the goal reset of its previous step (rg_env_post_step);
   * executed before the action is applied, then zeroed. */
  int* preticks_dev;
  /* float [B][rg_xdata_layout()[0]] or NULL: the mjData fields in-tree code reads after a step — body xpos / xquat and
   * site_xpos of the final state (mujoco_shadow_hand.py:18-61, observation providers), actuator_force of the last
   * state-less forward, ncon and contact[i].{geom1, geom2, dist} of the last mj_step (simulation/base.py:562-635,
   * utils/sensor_utils.py:18-38).  rg_xdata_layout: out = row length, offsets of xpos, xquat, site_xpos, actuator_force,
   * ncon, contact triples, number of contact slots, sensordata (touch sensors; filled by launches with flags bit 5), number of
   * sensor slots. */
  float* xdata_dev;
} rg_step_args;
int rg_xdata_layout(int* out, int n);
enum { RG_CFG_ROLLOUT = 0, RG_CFG_LARGE = 1 };
int rg_batch_step_ex(rg_batch* b, const rg_step_args* args);
/* Device address of a field's [B][n] buffer inside the batch (library-owned; valid until rg_batch_free) and its
 * row length in 4-byte words: zero-copy views for callers that live on the same device (torch tensors over
 * `sim.data.*`, simulation_interface.py:128-172).  Writers of RG_F_QPOS must zero the env's RG_F_PAIRLB row. */
void* rg_batch_field_ptr(rg_batch* b, int field, int* row_words);
/* number of static collision pairs (row length of RG_F_PAIRLB) */
int rg_model_npair(const rg_model* m);

/* ---- the env-level half of RobotEnv.step, one launch for the whole batch -------------------------------------------
 * After rg_batch_step(_ex) wrote the observation rows and goal distances: goal-distance reward and success flag
 * (robot_env.py:550-625), MultiGoalTracker.process / reset / reset_goal_steps (utils/multi_goal_tracker.py:83-241, dactyl
 * settings: one successful step counts, min_timesteps_per_goal 0, no reachability check), goal resampling
 * (envs/dactyl/goals/locked_parallel.py:32-46), reset_goal's bookkeeping (robot_env.py:893-909; its two state-less
 * forwards are booked in `preticks` and executed by the env's next step launch, RG args preticks_dev) and, with
 * `pipelined`, the per-env phase machine of the reset recipe (cube_env.py:330-355, locked.py:197-225) including its
 * masked state writes (MjSim.reset, cube pose perturbation, scripted ctrl).  All arrays are device pointers owned by the
 * caller, int32 / float32, [B] unless noted.  Randomness: `draws` ([B][RG_POST_NDRAW]: u_angle, u_choice in [0,1),
 * 4 + 3 standard normals, nu actions in [-1,1]) or, when NULL, a counter-based generator keyed by (seed, step, env). */
#define RG_POST_NDRAW (2 + 4 + 3 + 20)
typedef struct rg_post_args {
  // ---- physics outputs of this step (read)
  const float* goal_dist;      // [B] distance to the goal the env had during the step
  float* obs;                  // [B][obs_dim] observation row (zeroed for crashed envs)
  int obs_dim;
  // ---- env state (read / write), all [B] unless noted
  int *t, *phase, *tries;                      // env clock; reset-recipe phase (0 = live) and retry counter
  int *steps, *steps_since_last_goal, *successes_so_far, *goals_so_far, *consecutive;   // MultiGoalTracker
  float* prev_dist; int* prev_valid; int* is_successful;
  float* goal_quat;            // [B][4]
  float* qpos_goal;            // [B][nq]
  int* preticks;               // [B] state-less forwards owed to the env (consumed by its next step launch)
  // ---- outputs
  float* reward;               // [B][3]
  unsigned char *done, *goal_reset, *trial_success, *sub_goal_ok, *env_crash, *resetting, *episode_started;   /* 0 / 1 bytes (torch.bool storage) */
  int* info_ssl;               /* [B] steps_since_last_goal as the reference's info reports it */
  int *nticks_next, *reset_mask, *live_mask;   /* [B] inputs of the NEXT step launch: per-env forward ticks (3 live, 1 / 2 in the recipe), envs in the recipe (hold + large configuration), the others */
  float* goal_dist_before;     // [B]
  float* packed;               // [B][obs_dim + 3 + 4 + nq + 1 + 3 + 1] or null: obs | goal_pos | goal_quat | qpos_goal | achieved | reward | done
  // ---- randomness: either the caller's draws or a counter-based generator (seed, step, env)
  const float* draws;          // [B][RG_POST_NDRAW] or null: u_angle, u_choice, n_quat[4], n_wiggle[3], u_action[nu]
  const float* goal_override;  // [B][4] or null: the goal an env receives if it needs a new one (scripted goals; default: LockedParallelGoal sampling)
  unsigned seed, step;
  // ---- constants
  const float* parallel_quats; // [24][4]
  const float* qpos0;          // [nq]
  const float* zero_ctrl;      // [nu] ctrl of the zero action (absolute)
  const float* ctrl_lo, *ctrl_hi;   // [nu]
  float success_threshold, success_reward, wiggle_std, cube_body_z;
  int max_timesteps_per_goal, successes_needed, use_goal_distance_reward;
  int pipelined, reset_initial_steps, n_random_initial_steps, max_pose_resets;
  int cube_pos_col, cube_quat_col;
  int stop_on_fall;            /* StopOnFallWrapper (wrappers/cube.py:106-156) folded in: a live env whose cube centre is below z = 0.04 reports done (and, pipelined, restarts) */
} rg_post_args;
int rg_env_post_step(rg_batch* b, const rg_post_args* args, void* stream);
int rg_post_args_size(void);   /* sizeof(rg_post_args) as compiled: a binding checks its own struct against it */
/* mj_setConst for the envs of `mask_dev` (int [B] device pointer, NULL: all) -- replaces `MjSim.set_constants()` =
 * `mujoco_simulation.set_constants()`, which the reference calls in every `_reset` after the randomizers have written
 * the model (/root/reference/robogym/envs/dactyl/common/cube_env.py:346-349,
 * /root/reference/robogym/mujoco/simulation_interface.py:199-201): recomputes dof_invweight0 / body_invweight0 /
 * tendon_invweight0 in each env's parameter row (RG_F_ENVPRM) from that row's body_mass / body_inertia / dof_armature /
 * site_pos at qpos0.  Asynchronous on `stream`; needs rg_batch_enable_env_params. */
int rg_batch_set_constants(rg_batch* b, const int* mask_dev, void* stream);
/* Collision unit-test hook (no reference counterpart; mjc_Convex is internal to MuJoCo): runs the
 * kinematics of every env's stored qpos and one MPR penetration query between geoms g1, g2 inflated
 * by margin/2 each.  out_dev float [B][8] = hit, depth, direction3 (g1 -> g2), position3. */
int rg_batch_mpr_pair(rg_batch* b, int g1, int g2, float margin, float* out_dev, void* stream);
/* substep-granular dispatch (flags bit 7): persistent workgroups per launch (0: unavailable on this device) and work queues (XCDs) */
int rg_batch_items_info(const rg_batch* b, int* slots, int* queues);
int rg_obs_dim(const rg_batch* b);
int rg_debug_size(void);
/* bytes of LDS one env occupies in the rollout configuration (diagnostic); rg_lds_bytes_cfg: any configuration */
int rg_lds_bytes(void);
int rg_lds_bytes_cfg(int config);
/* ---- Large-model path (rb_kernel.h): models beyond the compile-time layout of the Shadow-hand kernel -- BASELINE.json
 * configs[2], dactyl/full_perpendicular (/root/reference/robogym/envs/dactyl/full_perpendicular.py:92-136: nv 168, 135 bodies,
 * condim-6 mesh hulls, njmax 2000 / nconmax 200 from cube_env.py:239-242).  Same blob format (plus the b_* tables of
 * robogym_amd/mujoco/big_tables.py), same ownership / error / stream conventions as the rg_* entry points they mirror:
 *   rb_model_create / rb_model_free        mujoco_py.load_model_from_xml (mujoco_xml.py:259)
 *   rb_batch_create / _free / _reset       B x MjSim, MjSim.reset (mujoco_xml.py:260, simulation_interface.py:191-197)
 *   rb_batch_field_ptr(field)              device address of a state field (RG_F_QPOS ... RG_F_STATS; RG_F_DEBUG = the env's
 *                                          scratch row of stage arrays under their mjData names, offsets from rb_scratch_offset)
 *   rb_batch_set_env                       hand joint block + position -> control matrix of the action map (robot_interface.py:247-278)
 *   rb_batch_step                          RobotEnv._set_action + SimulationInterface.step + the PID ticks of the state-less forwards
 *                                          (robot_env.py:497-504,677, simulation_interface.py:176-189); flags bit 0: stage dump, bit 4:
 *                                          portal-plane contact depth (as rg_batch_step)
 *   rb_model_info: out = nq, nv, nu, nbody, njnt, ngeom, nsite, ntendon, nM, npair, ngroup, gmax, maxcon, maxrow, scratch words,
 *                  contact record words, row record words, contact dof width, tendon dof width, LDS bytes per workgroup, threads per
 *                  workgroup (the kernel configuration the model runs on: 256 = large, 64 = small / one wave per env; the small one is used
 *                  when the model fits it, RB_CONFIG=large in the environment forces the large one) */
typedef struct rb_model rb_model;
typedef struct rb_batch rb_batch;
rb_model* rb_model_create(const void* blob, size_t nbytes, char* err, int errlen);
rb_model* rb_compile_mjcf(const char* xml, const char* meshdir, char* err, int errlen);   /* MJCF string -> model (see rg_compile_mjcf) */
/* Per-env model parameters on the large-model stepper (what the reference's simulation randomizers write into `sim.model` per episode,
 * /root/reference/robogym/envs/rearrange/common/base.py:1008-1092, randomization/sim.py:115-589).  rb_model_enable_env_params switches the MODEL (before its first
 * rb_batch_create): every batch then carries one parameter block per env inside the env's scratch row (RG_F_DEBUG through rb_batch_field_ptr), initialised with the
 * model's values, which rb_step_kernel reads instead of the model's arrays: opt.gravity, dof_damping / armature / frictionloss / invweight0, jnt_stiffness / margin /
 * range, body_pos / mass / inertia / invweight0, actuator_gainprm / forcerange / ctrlrange, geom_pos / margin / gap / friction / solref / solimp (a contact mixes its
 * two geoms' values as mj_contactParam does), tendon_range / invweight0.  The *_invweight0 rows are mj_setConst outputs: the reference's rearrange envs never call
 * set_constants after randomizing (only dactyl's cube_env.py:349 does), so they stay at the compiled model's values unless the host writes them.
 * rb_prm_layout: out[0] = 1 if enabled, out[1] = words per block, then per field (order above) its word offset in the scratch row and its length. */
int rb_model_enable_env_params(rb_model* m);
int rb_prm_layout(const rb_model* m, int* out, int n);
int rb_model_blob_keys(const rb_model* m, char* out, int outlen);   /* the blob arrays rb_model_create read (see rg_blob_entry) */
void rb_model_free(rb_model* m);
int rb_model_info(const rb_model* m, int* out, int n);
int rb_scratch_offset(const rb_model* m, int which);
rb_batch* rb_batch_create(const rb_model* m, int B);
void rb_batch_free(rb_batch* b);
int rb_batch_reset(rb_batch* b);
int rb_batch_set_env(rb_batch* b, int hand_qposadr, int n_hand_jnt, int relative_action, const float* pos_to_ctrl);
/* relative actions of a composite robot (/root/reference/robogym/robot/robot_interface.py:220-231 Robot.actuation_range, robot/composite/composite_robot.py): the
 * per-step actuation range of the actuators that centre on a joint position is min((hi - lo) / 2, max_position_change) (0 = not capped); the actuators named by
 * `ctrl_centre_mask` (bit u) centre on their stored control and keep the full range (the gripper of the UR16e + Robotiq robot, joint control mode:
 * robot/ur16e/mujoco/joint_controlled_arm.py:89-200, robot/gripper/mujoco/mujoco_robotiq_gripper.py:142-172).  After rb_batch_set_env. */
int rb_batch_set_action_limits(rb_batch* b, float max_position_change, unsigned ctrl_centre_mask);
void* rb_batch_field_ptr(rb_batch* b, int field, int* row_words);
int rb_batch_step(rb_batch* b, const float* action_dev, const int* active_dev, int nsubsteps, int nforward_ticks, int flags, void* stream);
/* the same with per-env `hold_dev` (int [B] or NULL: envs that keep their stored ctrl row, the reset recipe's scripted controls) and `nticks_dev` (int [B] or
 * NULL: per-env count of state-less forwards) -- as rg_step_args.hold_dev / nticks_dev */
int rb_batch_step_ex(rb_batch* b, const float* action_dev, const int* active_dev, const int* hold_dev, const int* nticks_dev, int nsubsteps, int nforward_ticks, int flags, void* stream);
/* ---- rearrange (UR16e + 2f-85 gripper; BASELINE.json configs[3]).  The large-model stepper also runs the rearrange worlds: elliptic friction
 * cones with impratio (assets/xmls/robot/ur16e/base.xml:3), weld-to-mocap and joint-coupling equality constraints (base.xml:52-54,
 * gripper_actuators.xml:3), mujoco-py's cascaded-PI actuators (jointspec/calibrations/cascaded_pi/joint_actuations.xml:4-10), jointpos / force /
 * torque sensors (base.xml:38-49).  Extra per-env rows, through rb_batch_field_ptr: */
enum {
  RB_F_MOCAP = 12,      /* float [B][7 nmocap]  data.mocap_pos | mocap_quat per mocap body (gym.envs.robotics.utils.mocap_set_action writes them) */
  RB_F_EQ_DATA = 13,    /* float [B][7 neq]     model.eq_data (reset_mocap_welds writes it; joint coupling: polycoef[5])                    */
  RB_F_EQ_ACTIVE = 14,  /* int32 [B][neq]       model.eq_active (envs/rearrange/common/base.py:452-455)                                     */
  RB_F_SENSORDATA = 15  /* float [B][nsensordata] data.sensordata of the last full forward (step flags bit 5)                               */
};
/* step flags bit 5 (32): the LAST state-less forward of the call runs in full (mj_forward: collision, constraint solve) and the sensors are
 * evaluated from it (mj_sensorPos / mj_sensorAcc with mj_rnePostConstraint); the stage arrays of the scratch row (body frames, velocities,
 * contact list) then describe the final state, which is what the rearrange observation reads (envs/rearrange/common/base.py:376-421).
 *
 * rb_batch_step_tcp: `JointControlledTcpArm.set_position_control` (robot/ur16e/mujoco/joint_controlled_tcp_arm.py:89-97) as ONE launch of the TCP
 * solver's own simulation (`solver`: arm + gripper world with the mocap weld, build_solver_sim, robot/composite/ur_gripper_arm.py:143-160):
 *   controller_arm.sync_to(main joint positions) + mj_forward          free_dof_tcp_arm.py:214-225        (reset_controller_error)
 *   FreeDOFTcpArm.denormalize / constrain_quat_ctrl / set_position_control   free_dof_tcp_arm.py:161-206   action[B][6] = xyz, roll, wrist, gripper
 *   MocapSolver.get_tcp_quat + gym mocap_set_action                      robot/control/tcp/mocap_solver.py:33-57
 *   solver simulation: nsubsteps x mj_step                                (controller autostep)
 *   main ctrl[:6] <- solver joint angles                                  joint_controlled_arm.py:180-184
 *   main ctrl[gripper] <- clip(ctrl + a * range / 2)                      robot_interface.py:247-278 around the current ctrl (mujoco_robotiq_gripper.py:139-172)
 * `main` is stepped afterwards by rb_batch_step_ex with a NULL action (its ctrl row is the input). */
typedef struct rb_tcp_args {
  int arm_qposadr[6], main_arm_qposadr[6];   /* qpos addresses of robot0:J1..J6 in the solver / main model */
  int main_gripper_actuator, tcp_body, wrist_joint, reset_controller_error;
  float max_position_change, speed_roll, speed_pitch, joint_drift_threshold, gripper_ctrl_lo, gripper_ctrl_hi;
  /* the rearrange env's default wrapper stack on the action side (RearrangeEnv.apply_wrappers, envs/rearrange/common/base.py:986-996), all optional:
   * DiscretizeActionWrapper (wrappers/util.py:36-70): action_index[B][6] bin indices into bins[6][nbins] replace `action_dev` (which may then be NULL);
   * SmoothActionWrapper (util.py:192-218): ema_value[B][6] / ema_t[B] = IncrementalExpAvg state (zero at reset), ema_alpha = alpha ^ (timestep nsubsteps / 0.08);
   * action_out[B][6] receives the action that reached the env (the wrapper's "action_ema" observation).  device pointers; NULL = not used */
  const int* action_index; const float* bins; int nbins; float ema_alpha;
  float* ema_value; int* ema_t; float* action_out;
  /* scripted actions (pipelined resets: the recipe's random-action and settle steps, RearrangeEnv._randomize_robot_initial_position, common/base.py:498-510):
   * hold[B] != 0 -> that env takes scripted[B][6] (continuous, unwrapped) instead of its action, its smoothing filter is left alone; NULL = not used */
  const int* hold; const float* scripted;
  /* control_mode tcp+wrist (robot_interface.py:9-20; FreeWristTcpArm, robot/ur16e/mujoco/free_dof_tcp_arm.py:238-246): the roll number (index 3 of the six) is ignored
   * = 0, and the commanded TCP orientation is rotated back onto the vertical before the difference goes to the mocap body (MocapSolver.align_axis with ALIGN_AXIS =
   * PITCH, robot/control/tcp/mocap_solver.py:41-74).  0: tcp+roll+yaw (FreeRollYawTcpArm) */
  int wrist_only;
  /* tcp_solver_mode = mocap (robot_interface.py:22-29, 54-58): the arm of the MAIN world hangs on the mocap weld itself (MujocoIdealURGripperCompositeRobot =
   * IdealJointControlledTcpArm + MujocoRobotiqGripper with solver_simulation = simulation, robot/composite/ur_gripper_arm.py:126-128; robot/ur16e/mujoco/
   * ideal_joint_controlled_tcp_arm.py:67-76): `solver` IS the env's world, `main` is ignored (may be NULL).  No arm synchronisation and no hand-over; the gripper's
   * relative target goes into this world's own ctrl[main_gripper_actuator] BEFORE the mj_steps; the launch then continues like rb_batch_step_ex with `nforward_ticks`
   * state-less forwards (flags bit 5: the last in full) or the per-env `nticks`.  skip[B] (or NULL): envs whose robot gets no command this step (their mocap target and
   * ctrl stay: the objects' stabilisation steps of the reset recipe). */
  int self_world, nforward_ticks;
  const int* nticks; const int* skip;
} rb_tcp_args;
int rb_batch_step_tcp(rb_batch* solver, rb_batch* main, const float* action_dev, const int* active_dev /* [B] or NULL */, const rb_tcp_args* args, int nsubsteps, int flags, void* stream);
int rb_tcp_args_size(void);   /* sizeof(rb_tcp_args): a binding checks its own layout against it */
/* Several batches in ONE launch (heterogeneous object sets side by side: /root/reference/robogym/envs/rearrange/ycb.py:58-84 rebuilds the simulation with new objects per
 * episode -- here a batch per compiled object set).  Between rb_multi_begin and rb_multi_launch the rb_batch_step / _step_ex / _step_tcp calls of the calling thread
 * validate their arguments and RECORD the launch; rb_multi_launch issues ONE kernel over all recorded batches (same batch size, same one-wave kernel configuration;
 * at most 8), or one launch each, in order, when they do not match.  Results are those of the separate launches, bit for bit.
 * While recording: the `stream` argument of the recorded calls is IGNORED (rb_multi_launch's stream is the one); a batch may be recorded once only
 * (rb_multi_launch refuses a duplicate: two workgroups would step the same rows); the entry points that are not recordable -- rb_env_post_step, ra_env_post_step,
 * ra_env_recipe_step, rb_cube_ops -- fail instead of running ahead of the recorded physics. */
int rb_multi_begin(void);
int rb_multi_launch(void* stream);
/* ---- the env-level half of RearrangeEnv.step (one launch after the two physics launches; robogym_amd/csrc/ra_env_kernel.h lists the
 * reference call sites): the 24-key observation of envs/rearrange/common/base.py:376-421 as one packed row per env — obj_pos 3N | obj_rel_pos 3N |
 * obj_vel_pos 3N | obj_rot 3N | obj_vel_rot 3N | robot_joint_pos 6 | gripper_pos 3 | gripper_velp 3 | gripper_controls 1 | gripper_qpos 1 |
 * gripper_vel 1 | qpos nq | qpos_goal nq | goal_obj_pos 3N | goal_obj_rot 3N | is_goal_achieved 1 | rel_goal_obj_pos 3N | rel_goal_obj_rot 3N |
 * obj_gripper_contact 2N | obj_bbox_size 3N | obj_colors 4N | safety_stop 1 | tcp_force 3 | tcp_torque 3 (= 36 N + 23 + 2 nq scalars; 289 for blocks
 * with 5 objects) followed by reward[3] and done — plus reward / done (common/base.py:768-795, 824-848), ObjectStateGoal's distances
 * (goals/object_state.py:492-599), MultiGoalTracker.process and the gripper hand-over to the solver world (joint_controlled_tcp_arm.py:114-129).
 * The main batch must have been stepped with flags bit 5 (the row is read from the final forward's stage arrays).  goal_reset[e] = 1 asks the
 * host for a new goal (ObjectStateGoal.next_goal is placement sampling: host work). */
#define RA_MAXOBJ 16
typedef struct ra_post_args {
  float* obs; int obs_dim;                       /* [B][obs_dim + 4] */
  int num_objects;
  int *t, *steps, *steps_since_last_goal, *successes_so_far, *consecutive;
  float* prev_nsucc; int* prev_valid;            /* previous count of objects within both thresholds (x goal_reward_per_object) */
  const float* goal;                             /* [B][N][7]: goal position, goal orientation as a quaternion (euler2quat of goal_obj_rot) */
  const float* goal_rot;                         /* [B][N][3]: goal_obj_rot (Euler angles, as observed) */
  const float* qpos_goal;                        /* [B][nq] */
  const float* static_obs;                       /* [B][N][7]: obj_bbox_size 3, obj_colors 4 */
  float* reward;                                 /* [B][3] env, goal, success */
  float* goal_dist;                              /* [B][2] sums over the objects of obj_pos / obj_rot distances */
  unsigned char *done, *goal_reset, *trial_success, *sub_goal_ok, *env_crash, *objects_off_table;
  int* info_ssl;
  int obj_body[RA_MAXOBJ], tcp_body, arm_qposadr[6], grip_qposadr, grip_dofadr, grip_act, finger_geom[2], table_plane_geom, force_adr, torque_adr;
  unsigned long long gripper_geom_mask;          /* geoms of the gripper's bodies (ur16e/mujoco/simulation/base.py:40-52): bit g = geom g, so they need ids < 64 (robot geoms come first) */
  float table_min[2], table_max[2], table_height, pos_threshold, rot_threshold, goal_pos_offset, goal_rot_weight, goal_reward_per_object, success_reward,
      penalty_table_collision, penalty_objects_off_table, penalty_safety_stop, safety_stop_force;
  int max_timesteps_per_goal, successes_needed, use_goal_distance_reward;
  float *solver_qpos, *solver_ctrl;              /* rows of the solver batch (filled by ra_env_post_step), NULL: no hand-over */
  int solver_nq, solver_nu, solver_grip_qposadr, solver_grip_act;
  const unsigned char* frozen;                   /* [B] or NULL.  1: an env inside its reset recipe (pipelined resets) or being re-observed after a goal change — observation
                                                    row and gripper hand-over only: no reward, no tracker step, done = 0.  2: the env is skipped altogether.  3: the observation entries only (a live env
                                                    whose goal was just replaced): reward / done / flags / counters of the step stay */
  float reward_clip;                             /* ClipRewardWrapper (wrappers/util.py:115-126; 100 in RearrangeEnv.apply_wrappers): every reward entry clipped to +- this; 0 = off */
} ra_post_args;
int ra_env_post_step(rb_batch* main, rb_batch* solver, const ra_post_args* args, void* stream);
int ra_post_args_size(void);
/* ---- the reset recipe and the goal sampling of the rearrange envs on the device (pipelined resets: an ended episode restarts INSIDE the following step calls),
 * one launch after ra_env_post_step; no host readback.  Per env: the recipe's stage machine — stabilise (stabilize_steps, the stored controls held) -> one random
 * action for n_random_initial_steps (RearrangeEnv._randomize_robot_initial_position, /root/reference/robogym/envs/rearrange/common/base.py:498-510) -> settle_steps
 * of the zero action -> the episode starts: tracker / wrapper state zeroed, first goal —; for an episode that ended on this step the state RearrangeEnv._reset writes
 * before anything is simulated (common/base.py:897-932: both worlds as freshly made, the arm's start pose, object rotations about z, `place_objects_in_grid`,
 * common/utils.py:719-829, with `place_objects_with_no_constraint`'s rejection sampling, :829-880, when the grid has fewer cells than objects; bounding boxes and colours
 * of the static observation); for a live env whose goal was reached `ObjectStateGoal.next_goal` (goals/object_state.py:355-418: a new placement, the goal keeps the
 * objects' initial yaw).  Draws come from a counter-based generator (seed, step, env, k).  Outputs for the NEXT step's launches: hold / hold_ctrl / scripted /
 * frozen / solver_active / nticks; `reobserve` is the `frozen` array of a second ra_env_post_step launch (1 = first observation of a new episode, 3 = a live env with
 * a new goal, 2 = skip); `ended` / `stabilised` / `episode_started` are masks for the host's tensor ops on the per-env parameter rows (stabilize_objects' damping,
 * the simulation randomizers).  solver may be NULL (control_mode joint). */
typedef struct ra_recipe_args {
  int num_objects, action_dim;
  int *stage, *left;                             /* [B]: 0 live, 1 stabilise, 2 random action, 3 settle; steps left in the stage */
  float* yaw;                                    /* [B][N] the episode's object rotations about z */
  const unsigned char *done, *goal_reset;        /* [B] as ra_env_post_step left them */
  int *hold, *hold_ctrl, *solver_active, *nticks;
  float* scripted;                               /* [B][action_dim] */
  unsigned char *frozen, *resetting, *episode_started, *reobserve, *ended, *stabilised;
  int* placement_failed;                         /* [B] += 1 when the rejection sampling ran out of restarts (the last proposal is used) */
  int *t, *steps, *steps_since_last_goal, *successes_so_far, *consecutive, *prev_valid, *ema_t;
  float *ema_value, *action_ema;                 /* [B][action_dim] */
  float *goal, *goal_rot, *qpos_goal, *static_obs;   /* as ra_post_args */
  int obj_qposadr[RA_MAXOBJ], arm_qposadr[6], solver_arm_qposadr[6];
  float arm_start[6];                            /* TABLETOP_EXPERIMENT_INITIAL_POS (robot/ur16e/arm_interface.py:27) */
  float obj_center[RA_MAXOBJ][3], obj_half[RA_MAXOBJ][3];   /* bounding box of each object's vertices in its body frame */
  float area_offset[2], area_size[2], table_pos[3], table_size[3];   /* get_placement_area (simulation/base.py:980-1010), offset from the table's low corner */
  int stabilize_steps, n_random_initial_steps, settle_steps;
  unsigned seed, step;
} ra_recipe_args;
int ra_env_recipe_step(rb_batch* main, rb_batch* solver, const ra_recipe_args* args, void* stream);
int ra_recipe_args_size(void);
/* ---- the env-level half of RobotEnv.step for the full cube (dactyl/full_perpendicular), one launch after rb_batch_step:
 * FaceFreeGoal.goal_distance / relative_goal / next_goal (/root/reference/robogym/envs/dactyl/goals/face_free.py:61-189, with
 * cube_utils.py:26-181), the target cube's joint manipulation that next_goal entails (full_perpendicular.py:138-155 ->
 * cube_manipulator.py:148-187, 377-409: clone, soft_align_faces, rotate_face), reward / success of robot_env.py:550-625 (sum over
 * the keys cube_quat and cube_face_angle), MultiGoalTracker.process (utils/multi_goal_tracker.py:157-241), reset_goal
 * (robot_env.py:893-909; its two state-less forwards are executed as PID ticks inside this launch) and the observation row of
 * full_perpendicular.py:177-192.  Arrays are device pointers owned by the caller, int32 / float32, [B] unless noted.
 * goal row (RB_GOAL_WORDS floats): cube_quat[4] (w >= 0), cube_face_angle[6], goal_type (1 rotation, 0 flip), axis_nr, axis_sign, pad.
 * observation row: cube_pos 3 | cube_quat 4 | cube_face_angle 6 | hand_angle n_hand | fingertip_pos 15 | goal_pos 3 | goal_quat 4 |
 * goal_face_angle 6.  Randomness: `draws` ([B][RB_POST_NDRAW] = u_reorient, u_round, k_direction (index; the fraction of the range
 * when round_target_face < 1), k_face (index), z angle in [-pi, pi)) or, when NULL, a counter-based generator keyed by (seed, step, env).
 * force_new_goal ([B] or NULL): when given, ONLY reset_goal runs, for the flagged envs (the tail of RobotEnv.reset, robot_env.py:787-792). */
#define RB_POST_NDRAW 5
#define RB_GOAL_WORDS 16
typedef struct rb_post_args {
  float* obs; int obs_dim;
  int *t, *steps, *steps_since_last_goal, *successes_so_far, *goals_so_far, *consecutive;   /* env clock, MultiGoalTracker */
  float* prev_dist;            /* [B][2] previous distances (cube_quat, cube_face_angle) */
  int *prev_valid, *is_successful;
  float* goal;                 /* [B][RB_GOAL_WORDS] */
  float* reward;               /* [B][3] */
  float* goal_dist;            /* [B][2] distances to the goal the env had during the step */
  unsigned char *done, *goal_reset, *trial_success, *sub_goal_ok, *env_crash;   /* 0 / 1 bytes (torch.bool storage) */
  int* info_ssl;
  const int* force_new_goal;
  const float* draws;
  unsigned seed, step;
  const int* cube_tab;         /* [20][6]: per cubelet the offsets of its rotx / roty / rotz joint in the 66-joint block, its coordinates in {-1,0,1}^3 */
  const float* face_up_quats;  /* [6][4]: per face the axis-aligned orientation that puts it up (cube_utils.face_up_quats) */
  int face_geom[6];            /* geom ids of cube:cubelet:neg_x ... pos_z */
  int tip_site[5], ref_site[3], center_site;
  int cube_pos_col, cube_quat_col, cube_block_col, target_block_col, hand_col, n_hand;   /* qpos columns; *_block_col: first of the 66 cubelet joints (6 drivers, 20 x 3 hinges) */
  float quat_threshold, face_threshold, success_reward, p_face_flip, round_target_face;
  int directions;              /* bit 0 "cw", bit 1 "ccw" (goal_directions) */
  int max_timesteps_per_goal, successes_needed, use_goal_distance_reward, stop_on_fall;
  int goal_mode;               /* 0: FaceFreeGoal (goals/face_free.py); 1: FullUnconstrainedGoal (goals/full_unconstrained.py:55-117: any face gets a turn --
                                  draws k_face, u_round, k_direction -- the goal quaternion is zero and its distance 0); 2: FaceCurriculumGoal
                                  (goals/face_curriculum.py:57-170: as 0, but a turn's orientation goal is round_to_straight_quat(cube_quat) and the orientation
                                  distance is always the plain quaternion difference) */
  /* ---- pipelined resets (pipelined = 0: unused).  An env whose episode ends restarts by itself: the reset recipe of
   * /root/reference/robogym/envs/dactyl/common/cube_env.py:330-355 + full_perpendicular.py:286-345 as a per-env phase counter (0 = live; k > 0: k - 1
   * recipe steps done) with its state writes done here: MjSim.reset + the zero-action ctrl; after reset_initial_steps steps the cube pose perturbation,
   * the scramble (num_scramble_steps face turns on signed permutation matrices -> hinge angles, as from_pycuber), the face-driver angles, two
   * CubeManipulator.rotate_face calls, the random action's ctrl; after n_random_initial_steps more steps the on-palm test (retry up to max_pose_resets) and
   * the first goal.  hold_next / nticks_next are the inputs of the NEXT rb_batch_step_ex (envs on scripted ctrl; per-env forward ticks: 3 live, 1 in the
   * recipe, 2 on its steps reset_initial_steps and reset_initial_steps + n_random_initial_steps).  reset_draws ([B][RB_RESET_NDRAW] or NULL -> the hash
   * generator): 3 + 4 standard normals (position wiggle, orientation), num_scramble_steps <= 50 action indices 0..11 (L, L', R, R', F, F', B, B', D, D', U, U'),
   * 6 driver multiples of pi / 2 in -2..2, 2 face angles (rad), the face axis 0..2, nu <= 20 actions in [-1, 1]. */
  int pipelined;
  int *phase, *tries;
  int *nticks_next, *hold_next;
  unsigned char *resetting, *episode_started;
  const float* reset_draws;
  const float *qpos0, *ctrl_lo, *ctrl_hi;     /* [nq], [nu], [nu] */
  float wiggle_std;
  int reset_initial_steps, n_random_initial_steps, max_pose_resets, num_scramble_steps, scramble_face_angles, randomize_face_angles;
} rb_post_args;
#define RB_RESET_NDRAW (3 + 4 + 50 + 6 + 2 + 1 + 20)
int rb_env_post_step(rb_batch* b, const rb_post_args* args, void* stream);
int rb_post_args_size(void);
/* CubeManipulator operations on one of the two cubes of every env in `active_dev` (NULL: all): ops_dev float [B][nops][4] =
 * {axis, side, angle, code}; code 0 rotate_face (cube_manipulator.py:148-187), 1 the same without the driver joint, 2 soft_align_faces
 * (:377-409), negative: nothing.  block_col = the qpos column of the cube's first cubelet joint.  Used by the reset recipe
 * (full_perpendicular.py:318-332: randomize_face_angles) and by the parity tests. */
int rb_cube_ops(rb_batch* b, int block_col, const int* cube_tab_dev, const float* ops_dev, int nops, const int* active_dev, void* stream);
int rg_sync(void* stream);
const char* rg_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
