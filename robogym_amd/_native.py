"""ctypes binding of librgstep.so (the gfx950 HIP stepper behind the C ABI of include/rgstep.h).

There is no CPU fallback: if the library is missing or no MI355X is visible the product raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGSTEP_LIB") or os.path.join(_HERE, "csrc", "librgstep.so")   # (RGSTEP_LIB: A/B builds of the same ABI, tools/ only)

RG_F_QPOS, RG_F_QVEL, RG_F_CTRL, RG_F_PID, RG_F_WARMSTART, RG_F_TIME, RG_F_STATUS, RG_F_STATS, RG_F_DEBUG, RG_F_COST, RG_F_PAIRLB, RG_F_ENVPRM = range(12)
RB_F_MOCAP, RB_F_EQ_DATA, RB_F_EQ_ACTIVE, RB_F_SENSORDATA = 12, 13, 14, 15
RG_STATUS_BAD_STATE, RG_STATUS_CON_FULL, RG_STATUS_CAND_FULL, RG_STATUS_ROW_FULL, RG_STATUS_BAD_FACTOR, RG_STATUS_BAD_ACTION = 1, 2, 4, 8, 16, 32


class StepArgs(ctypes.Structure):
    """`rg_step_args` of include/rgstep.h."""

    _fields_ = [("action_dev", ctypes.c_void_p), ("goal_quat_dev", ctypes.c_void_p), ("obs_dev", ctypes.c_void_p), ("goal_dist_dev", ctypes.c_void_p),
                ("active_dev", ctypes.c_void_p), ("hold_dev", ctypes.c_void_p), ("nticks_dev", ctypes.c_void_p), ("order_dev", ctypes.c_void_p),
                ("nsubsteps", ctypes.c_int), ("nforward_ticks", ctypes.c_int), ("flags", ctypes.c_int), ("stream", ctypes.c_void_p),
                ("config", ctypes.c_int), ("redo_dev", ctypes.c_void_p), ("preticks_dev", ctypes.c_void_p), ("xdata_dev", ctypes.c_void_p)]


RG_CFG_ROLLOUT, RG_CFG_LARGE = 0, 1
RG_POST_NDRAW = 2 + 4 + 3 + 20


def _post_fields():
    p_, i_, f_, u_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
    ptrs1 = ("t", "phase", "tries", "steps", "steps_since_last_goal", "successes_so_far", "goals_so_far", "consecutive",
             "prev_dist", "prev_valid", "is_successful", "goal_quat", "qpos_goal", "preticks", "reward",
             "done", "goal_reset", "trial_success", "sub_goal_ok", "env_crash", "resetting", "episode_started",
             "info_ssl", "nticks_next", "reset_mask", "live_mask", "goal_dist_before", "packed", "draws", "goal_override")
    return ([("goal_dist", p_), ("obs", p_), ("obs_dim", i_)] + [(n, p_) for n in ptrs1] + [("seed", u_), ("step", u_)]
            + [(n, p_) for n in ("parallel_quats", "qpos0", "zero_ctrl", "ctrl_lo", "ctrl_hi")]
            + [(n, f_) for n in ("success_threshold", "success_reward", "wiggle_std", "cube_body_z")]
            + [(n, i_) for n in ("max_timesteps_per_goal", "successes_needed", "use_goal_distance_reward", "pipelined", "reset_initial_steps",
                                 "n_random_initial_steps", "max_pose_resets", "cube_pos_col", "cube_quat_col", "stop_on_fall")])


class PostArgs(ctypes.Structure):
    """`rg_post_args` of include/rgstep.h (field order and types must match; bind() checks the size)."""

    _fields_ = _post_fields()


RB_POST_NDRAW, RB_GOAL_WORDS, RB_RESET_NDRAW = 5, 16, 3 + 4 + 50 + 6 + 2 + 1 + 20


class RbPostArgs(ctypes.Structure):
    """`rb_post_args` of include/rgstep.h (field order and types must match; bind() checks the size)."""

    _fields_ = ([("obs", ctypes.c_void_p), ("obs_dim", ctypes.c_int)]
                + [(n, ctypes.c_void_p) for n in ("t", "steps", "steps_since_last_goal", "successes_so_far", "goals_so_far", "consecutive", "prev_dist", "prev_valid",
                                                  "is_successful", "goal", "reward", "goal_dist", "done", "goal_reset", "trial_success", "sub_goal_ok", "env_crash",
                                                  "info_ssl", "force_new_goal", "draws")]
                + [("seed", ctypes.c_uint), ("step", ctypes.c_uint), ("cube_tab", ctypes.c_void_p), ("face_up_quats", ctypes.c_void_p),
                   ("face_geom", ctypes.c_int * 6), ("tip_site", ctypes.c_int * 5), ("ref_site", ctypes.c_int * 3), ("center_site", ctypes.c_int)]
                + [(n, ctypes.c_int) for n in ("cube_pos_col", "cube_quat_col", "cube_block_col", "target_block_col", "hand_col", "n_hand")]
                + [(n, ctypes.c_float) for n in ("quat_threshold", "face_threshold", "success_reward", "p_face_flip", "round_target_face")]
                + [(n, ctypes.c_int) for n in ("directions", "max_timesteps_per_goal", "successes_needed", "use_goal_distance_reward", "stop_on_fall", "goal_mode", "pipelined")]
                + [(n, ctypes.c_void_p) for n in ("phase", "tries", "nticks_next", "hold_next", "resetting", "episode_started", "reset_draws", "qpos0", "ctrl_lo", "ctrl_hi")]
                + [("wiggle_std", ctypes.c_float)]
                + [(n, ctypes.c_int) for n in ("reset_initial_steps", "n_random_initial_steps", "max_pose_resets", "num_scramble_steps", "scramble_face_angles", "randomize_face_angles")])


class RbTcpArgs(ctypes.Structure):
    """`rb_tcp_args` of include/rgstep.h."""

    _fields_ = [("arm_qposadr", ctypes.c_int * 6), ("main_arm_qposadr", ctypes.c_int * 6), ("main_gripper_actuator", ctypes.c_int), ("tcp_body", ctypes.c_int),
                ("wrist_joint", ctypes.c_int), ("reset_controller_error", ctypes.c_int), ("max_position_change", ctypes.c_float), ("speed_roll", ctypes.c_float),
                ("speed_pitch", ctypes.c_float), ("joint_drift_threshold", ctypes.c_float), ("gripper_ctrl_lo", ctypes.c_float), ("gripper_ctrl_hi", ctypes.c_float),
                ("action_index", ctypes.c_void_p), ("bins", ctypes.c_void_p), ("nbins", ctypes.c_int), ("ema_alpha", ctypes.c_float),
                ("ema_value", ctypes.c_void_p), ("ema_t", ctypes.c_void_p), ("action_out", ctypes.c_void_p), ("hold", ctypes.c_void_p), ("scripted", ctypes.c_void_p), ("wrist_only", ctypes.c_int), ("self_world", ctypes.c_int), ("nforward_ticks", ctypes.c_int),
                ("nticks", ctypes.c_void_p), ("skip", ctypes.c_void_p)]


RA_MAXOBJ = 16


def _ra_post_fields():
    _p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    return ([("obs", _p), ("obs_dim", _i), ("num_objects", _i)]
            + [(n, _p) for n in ("t", "steps", "steps_since_last_goal", "successes_so_far", "consecutive", "prev_nsucc", "prev_valid", "goal", "goal_rot", "qpos_goal", "static_obs",
                                 "reward", "goal_dist", "done", "goal_reset", "trial_success", "sub_goal_ok", "env_crash", "objects_off_table", "info_ssl")]
            + [("obj_body", _i * RA_MAXOBJ), ("tcp_body", _i), ("arm_qposadr", _i * 6), ("grip_qposadr", _i), ("grip_dofadr", _i), ("grip_act", _i), ("finger_geom", _i * 2),
               ("table_plane_geom", _i), ("force_adr", _i), ("torque_adr", _i), ("gripper_geom_mask", ctypes.c_ulonglong), ("table_min", _f * 2), ("table_max", _f * 2)]
            + [(n, _f) for n in ("table_height", "pos_threshold", "rot_threshold", "goal_pos_offset", "goal_rot_weight", "goal_reward_per_object", "success_reward",
                                 "penalty_table_collision", "penalty_objects_off_table", "penalty_safety_stop", "safety_stop_force")]
            + [(n, _i) for n in ("max_timesteps_per_goal", "successes_needed", "use_goal_distance_reward")]
            + [("solver_qpos", _p), ("solver_ctrl", _p), ("solver_nq", _i), ("solver_nu", _i), ("solver_grip_qposadr", _i), ("solver_grip_act", _i), ("frozen", _p), ("reward_clip", _f)])


class RaPostArgs(ctypes.Structure):
    """`ra_post_args` of include/rgstep.h (field order and types must match; bind() checks the size)."""

    _fields_ = _ra_post_fields()


def _ra_recipe_fields():
    _p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    return ([("num_objects", _i), ("action_dim", _i)]
            + [(n, _p) for n in ("stage", "left", "yaw", "done", "goal_reset", "hold", "hold_ctrl", "solver_active", "nticks", "scripted", "frozen", "resetting", "episode_started",
                                 "reobserve", "ended", "stabilised", "placement_failed", "t", "steps", "steps_since_last_goal", "successes_so_far", "consecutive", "prev_valid",
                                 "ema_t", "ema_value", "action_ema", "goal", "goal_rot", "qpos_goal", "static_obs")]
            + [("obj_qposadr", _i * RA_MAXOBJ), ("arm_qposadr", _i * 6), ("solver_arm_qposadr", _i * 6), ("arm_start", _f * 6), ("obj_center", (_f * 3) * RA_MAXOBJ),
               ("obj_half", (_f * 3) * RA_MAXOBJ), ("area_offset", _f * 2), ("area_size", _f * 2), ("table_pos", _f * 3), ("table_size", _f * 3), ("stabilize_steps", _i),
               ("n_random_initial_steps", _i), ("settle_steps", _i), ("seed", ctypes.c_uint), ("step", ctypes.c_uint)])


class RaRecipeArgs(ctypes.Structure):
    """`ra_recipe_args` of include/rgstep.h (field order and types must match; bind() checks the size)."""

    _fields_ = _ra_recipe_fields()


EXPORTS = [
    "rg_model_create", "rg_model_free", "rg_model_dims", "rg_batch_create", "rg_batch_free", "rg_batch_set_env",
    "rg_batch_copy", "rg_batch_reset", "rg_batch_step", "rg_obs_dim", "rg_debug_size", "rg_lds_bytes", "rg_sync",
    "rg_last_error", "rg_batch_mpr_pair", "rg_batch_copy_rows", "rg_batch_step_ex", "rg_batch_field_ptr", "rg_model_create_on", "rg_model_npair",
    "rg_lds_bytes_cfg", "rg_env_post_step", "rg_post_args_size", "rg_batch_enable_env_params", "rg_prm_layout", "rg_xdata_layout", "rg_batch_set_constants", "rg_batch_items_info",
    "rb_model_create", "rb_model_free", "rb_model_info", "rb_scratch_offset", "rb_batch_create", "rb_batch_free", "rb_batch_reset", "rb_batch_set_env",
    "rb_batch_field_ptr", "rb_batch_step", "rb_batch_step_ex", "rb_env_post_step", "rb_post_args_size", "rb_cube_ops", "rb_batch_step_tcp", "ra_env_post_step", "ra_post_args_size",
    "rg_blob_entry", "rg_model_blob_keys", "rb_model_blob_keys", "rg_compile_mjcf", "rb_compile_mjcf", "rg_compile_mjcf_blob", "rg_blob_free",
    "rb_model_enable_env_params", "rb_prm_layout", "rb_batch_set_action_limits", "ra_env_recipe_step", "ra_recipe_args_size", "rb_tcp_args_size", "rb_multi_begin", "rb_multi_launch",
]


class NativeError(RuntimeError):
    pass


def bind(path):
    """Load a build of the C ABI and declare its signatures."""
    if not os.path.exists(path):
        raise NativeError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path
        )
    L = ctypes.CDLL(path)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.rg_model_create.restype = vp
    L.rg_model_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ci]
    L.rg_model_free.argtypes = [vp]
    L.rg_model_create_on.restype = vp
    L.rg_model_create_on.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ci, ctypes.c_char_p, ci]
    L.rg_model_npair.argtypes = [vp]
    L.rg_batch_step_ex.argtypes = [vp, ctypes.POINTER(StepArgs)]
    L.rg_batch_field_ptr.restype = vp
    L.rg_batch_field_ptr.argtypes = [vp, ci, ctypes.POINTER(ci)]
    L.rg_model_dims.argtypes = [vp, ctypes.POINTER(ci)]
    L.rg_batch_create.restype = vp
    L.rg_batch_create.argtypes = [vp, ci, ci]
    L.rg_batch_free.argtypes = [vp]
    L.rg_batch_set_env.argtypes = [vp, ctypes.POINTER(ci), ci, ctypes.POINTER(cf), cf]
    L.rg_batch_copy.argtypes = [vp, ci, vp, ci, ci]
    L.rg_batch_reset.argtypes = [vp]
    L.rg_batch_copy_rows.argtypes = [vp, ci, vp, vp, ci, ci, vp]
    L.rg_batch_step.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
    L.rg_batch_mpr_pair.argtypes = [vp, ci, ci, cf, vp, vp]
    L.rg_obs_dim.argtypes = [vp]
    L.rg_debug_size.restype = ci
    L.rg_lds_bytes.restype = ci
    L.rg_lds_bytes_cfg.restype = ci
    L.rg_lds_bytes_cfg.argtypes = [ci]
    L.rg_batch_enable_env_params.argtypes = [vp]
    L.rg_prm_layout.argtypes = [ctypes.POINTER(ci), ci]
    L.rg_xdata_layout.argtypes = [ctypes.POINTER(ci), ci]
    L.rg_env_post_step.argtypes = [vp, ctypes.POINTER(PostArgs), vp]
    L.rg_post_args_size.restype = ci
    if L.rg_post_args_size() != ctypes.sizeof(PostArgs):
        raise NativeError("rg_post_args layout mismatch between include/rgstep.h and robogym_amd/_native.py")
    L.rg_batch_set_constants.argtypes = [vp, vp, vp]
    L.rg_batch_items_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.rb_model_create.restype = vp
    L.rb_model_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ci]
    L.rb_model_free.argtypes = [vp]
    L.rb_model_info.argtypes = [vp, ctypes.POINTER(ci), ci]
    L.rg_blob_entry.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ci, ctypes.c_char_p, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_uint)]
    L.rg_model_blob_keys.argtypes = [vp, ctypes.c_char_p, ci]
    L.rb_model_blob_keys.argtypes = [vp, ctypes.c_char_p, ci]
    L.rg_compile_mjcf.restype = vp
    L.rg_compile_mjcf.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ci]
    L.rb_compile_mjcf.restype = vp
    L.rb_compile_mjcf.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ci]
    L.rg_compile_mjcf_blob.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ci, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ci]
    L.rg_blob_free.argtypes = [vp]
    L.rb_model_enable_env_params.argtypes = [vp]
    L.rb_prm_layout.argtypes = [vp, ctypes.POINTER(ci), ci]
    L.rb_scratch_offset.argtypes = [vp, ci]
    L.rb_batch_create.restype = vp
    L.rb_batch_create.argtypes = [vp, ci]
    L.rb_batch_free.argtypes = [vp]
    L.rb_batch_reset.argtypes = [vp]
    L.rb_batch_set_env.argtypes = [vp, ci, ci, ci, ctypes.POINTER(cf)]
    L.rb_batch_set_action_limits.argtypes = [vp, cf, ctypes.c_uint]
    L.rb_batch_field_ptr.restype = vp
    L.rb_batch_field_ptr.argtypes = [vp, ci, ctypes.POINTER(ci)]
    L.rb_batch_step.argtypes = [vp, vp, vp, ci, ci, ci, vp]
    L.rb_batch_step_ex.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, vp]
    L.rb_batch_step_tcp.argtypes = [vp, vp, vp, vp, ctypes.POINTER(RbTcpArgs), ci, ci, vp]
    L.rb_multi_launch.argtypes = [vp]
    L.rb_tcp_args_size.restype = ci
    if L.rb_tcp_args_size() != ctypes.sizeof(RbTcpArgs):
        raise NativeError("rb_tcp_args layout mismatch between include/rgstep.h and robogym_amd/_native.py")
    L.ra_env_post_step.argtypes = [vp, vp, ctypes.POINTER(RaPostArgs), vp]
    L.ra_post_args_size.restype = ci
    if L.ra_post_args_size() != ctypes.sizeof(RaPostArgs):
        raise NativeError("ra_post_args layout mismatch between include/rgstep.h and robogym_amd/_native.py")
    L.ra_env_recipe_step.argtypes = [vp, vp, ctypes.POINTER(RaRecipeArgs), vp]
    L.ra_recipe_args_size.restype = ci
    if L.ra_recipe_args_size() != ctypes.sizeof(RaRecipeArgs):
        raise NativeError("ra_recipe_args layout mismatch between include/rgstep.h and robogym_amd/_native.py")
    L.rb_env_post_step.argtypes = [vp, ctypes.POINTER(RbPostArgs), vp]
    L.rb_post_args_size.restype = ci
    if L.rb_post_args_size() != ctypes.sizeof(RbPostArgs):
        raise NativeError("rb_post_args layout mismatch between include/rgstep.h and robogym_amd/_native.py")
    L.rb_cube_ops.argtypes = [vp, ci, vp, vp, ci, vp, vp]
    L.rg_sync.argtypes = [vp]
    L.rg_last_error.restype = ctypes.c_char_p
    return L


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = bind(LIB_PATH)
    return _LIB


def check(L, rc, what):
    if rc != 0:
        raise NativeError("%s failed: %s" % (what, L.rg_last_error().decode()))


RG_FLAG_MPR_PLANE_DEPTH = 16   # rg_step_args.flags bit 4 (include/rgstep.h)
RG_FLAG_SENSORS = 32           # bit 5: evaluate data.sensordata (the last state-less forward runs in full)
RG_FLAG_SEPARATE_FORWARD_SUBSTITUTION = 64   # bit 6: test hook (include/rgstep.h)
RG_FLAG_SUBSTEP_ITEMS = 128    # bit 7: substep-granular dispatch of a rollout launch (rg_step_items_kernel)
RG_FLAG_CAPACITY_TEST_HOOK = 512   # bit 9: test hook (include/rgstep.h)
RG_FLAG_DESERT_QUEUE0 = 1024       # bit 10: test hook: the substep-granular dispatch leaves its queue 0 unserved (the fallback must complete those envs)
RG_STATUS_SCHED = 64
RG_FLAG_RESUME = 256           # bit 8: active_dev is a redo array: entry - 1 = first substep still to do

RB_PRM_NAMES = ["gravity", "dof_damping", "dof_armature", "dof_frictionloss", "dof_invweight0", "jnt_stiffness", "jnt_margin", "jnt_range", "body_pos", "body_mass", "body_inertia",
                "body_invweight0", "actuator_gainprm", "actuator_forcerange", "actuator_ctrlrange", "geom_pos", "geom_margin", "geom_gap", "geom_friction", "geom_solref", "geom_solimp",
                "tendon_range", "tendon_invweight0"]      # rb_types.h RB_P_* order (rb_prm_layout)
PRM_NAMES = ["row", "gravity", "timestep", "dof_damping", "dof_armature", "dof_frictionloss", "dof_invweight0", "body_mass", "body_inertia", "body_invweight0",
             "jnt_range", "tendon_range", "tendon_invweight0", "actuator_gainprm", "actuator_ctrlrange", "actuator_forcerange", "geom_friction", "xfrc_applied", "site_pos", "geom_scale", "jnt_margin", "geom_solref", "geom_solimp"]


def prm_layout(L):
    """rg_prm_layout as a dict: 'row' -> floats per env, then field -> offset."""
    buf = (ctypes.c_int * 32)()
    n = L.rg_prm_layout(buf, 32)
    assert n == len(PRM_NAMES), "rg_prm_layout and robogym_amd/_native.py disagree"
    return {k: int(buf[i]) for i, k in enumerate(PRM_NAMES)}
