"""Python mirror of ``okvis::Estimator`` over the C ABI of ``libsvin_ba.so``.

Method names / argument meaning follow the reference interface
(okvis_ceres/include/okvis/Estimator.hpp:81): ``addStates`` -> :meth:`add_states`,
``addLandmark`` -> :meth:`add_landmark`, ``addObservation`` -> :meth:`add_observation`,
``optimize``, ``applyMarginalizationStrategy`` -> :meth:`apply_marginalization`, getters/setters.
Everything numerical happens inside the HIP library; this module only marshals arguments.
The library is required: importing without it (or creating an estimator without a GPU) raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64, u32, f64, i32 = C.c_uint64, C.c_uint32, C.c_double, C.c_int
pd = C.POINTER(C.c_double)
pu64 = C.POINTER(C.c_uint64)
COST_FUNCTION = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(pd), pd, C.POINTER(pd))   # svin_cost_function (include/svin_ba.h)
pi32 = C.POINTER(C.c_int32)


class ImuParams(C.Structure):
    _fields_ = [(n, f64) for n in ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c",
                                   "sigma_aw_c", "tau", "g")] + [("a0", f64 * 3)]


class ImuSample(C.Structure):
    _fields_ = [("sec", u32), ("nsec", u32), ("gyr", f64 * 3), ("acc", f64 * 3)]


class LandmarkInfo(C.Structure):
    _fields_ = [("point", f64 * 4), ("quality", f64), ("distance", f64), ("num_observations", C.c_int32),
                ("initialized", C.c_int32)]


class SummaryStruct(C.Structure):
    _fields_ = [("initial_cost", f64), ("final_cost", f64), ("iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("termination", C.c_int32), ("total_time_s", f64), ("upload_time_s", f64), ("solve_time_s", f64),
                ("download_time_s", f64)]


IMU_SAMPLE_DTYPE = np.dtype([("sec", np.uint32), ("nsec", np.uint32), ("gyr", np.float64, 3), ("acc", np.float64, 3)],
                            align=True)

EXPORTS = [
    "svin_ba_create", "svin_ba_destroy", "svin_ba_last_error", "svin_ba_new_id", "svin_ba_add_camera", "svin_ba_add_imu",
    "svin_ba_set_sonar_extrinsics", "svin_ba_add_states", "svin_ba_add_landmark", "svin_ba_add_observation",
    "svin_ba_add_observations", "svin_ba_keyframe_points",
    "svin_ba_remove_observation", "svin_ba_remove_observation_by_id", "svin_ba_optimize", "svin_ba_prepare",
    "svin_ba_solve_prepared", "svin_ba_finish", "svin_ba_invalidate_preintegration",
    "svin_ba_set_optimization_time_limit", "svin_ba_apply_marginalization_strategy", "svin_ba_get_summary",
    "svin_ba_set_solver_tolerances", "svin_ba_set_distributed", "svin_ba_get_T_WS", "svin_ba_get_speed_and_bias", "svin_ba_get_camera_sensor_states",
    "svin_ba_get_landmark", "svin_ba_is_landmark_added", "svin_ba_set_T_WS", "svin_ba_set_speed_and_bias",
    "svin_ba_set_camera_sensor_states", "svin_ba_set_landmark", "svin_ba_num_frames", "svin_ba_num_landmarks",
    "svin_ba_current_keyframe_id", "svin_ba_current_frame_id", "svin_ba_frame_id_by_age", "svin_ba_is_keyframe",
    "svin_ba_is_in_imu_window", "svin_ba_frame_ids", "svin_ba_landmark_ids", "svin_ba_imu_propagation",
    "svin_ba_eval_reprojection", "svin_ba_observation_ids", "svin_ba_eval_factors", "svin_ba_linearize", "svin_ba_debug_reduced_solve", "svin_ba_debug_reduced_solve_ex", "svin_ba_debug_set_switch", "svin_ba_debug_set_option", "svin_ba_debug_get_option", "svin_ba_debug_sym_eig", "svin_ba_get_path_counters", "svin_ba_wait_idle", "svin_ba_debug_peek_solver_scratch",
    "svin_ba_get_prior", "svin_ba_describe_block", "svin_ba_bench_jacobian_eval", "svin_ba_bench_jacobian_eval_b2b", "svin_ba_set_pack_mode", "svin_ba_debug_csr", "svin_ba_residual_info",
    "svin_ba_map_add_parameter_block", "svin_ba_set_parameter_block", "svin_ba_map_remove_parameter_block", "svin_ba_map_add_pose_error",
    "svin_ba_map_add_speed_and_bias_error", "svin_ba_map_add_relative_pose_error", "svin_ba_map_add_reprojection_error",
    "svin_ba_map_add_imu_error", "svin_ba_map_add_sonar_error", "svin_ba_map_add_depth_error", "svin_ba_map_add_host_residual",
    "svin_ba_map_remove_residual_block", "svin_ba_bench_kernel_times",
    "svin_ba_set_id_provider", "svin_ba_reserve_ids", "svin_ba_set_camera_geometry", "svin_ba_clear_cameras",
    "svin_ba_clear_imus", "svin_ba_is_landmark_initialized", "svin_ba_set_landmark_initialized", "svin_ba_get_landmarks",
    "svin_ba_set_keyframe", "svin_ba_timestamp", "svin_ba_state_count", "svin_ba_get_imu_preintegral",
    "svin_ba_set_imu_preintegral", "svin_ba_init_pose_from_imu", "svin_ba_imu_propagation_integrals",
    "svin_ba_rccl_unique_id", "svin_ba_set_distributed_rccl", "svin_ba_solve_prepared_batch", "svin_ba_optimize_batch",
    "svin_ba_parameter_block_exists", "svin_ba_set_parameter_block_constant", "svin_ba_is_parameter_block_constant",
    "svin_ba_reset_parameterization", "svin_ba_get_parameterization",
    "svin_ba_residuals_of", "svin_ba_parameters_of", "svin_ba_get_landmark_observations",
    "svin_host_imu_propagation", "svin_host_reprojection_error", "svin_host_homogeneous_point_error",
    "svin_ba_add_homogeneous_point_error", "svin_ba_remove_homogeneous_point_error",
    "svin_host_pose_information", "svin_host_pose_error", "svin_host_manifold_dims", "svin_host_manifold_plus",
    "svin_host_manifold_minus", "svin_host_manifold_plus_jacobian", "svin_host_manifold_lift_jacobian",
    "svin_host_manifold_minus_jacobian", "svin_ba_get_parameter_block", "svin_ba_parameter_block_ids",
    "svin_ba_get_all_landmark_observations", "svin_ba_bench_allreduce", "svin_ba_get_marg_pre", "svin_ba_get_marg_pre_blocks",
]

ID_PROVIDER_FN = C.CFUNCTYPE(C.c_uint64, C.c_void_p)


def library_path():
    # SVIN_BA_LIB: developer override (A/B runs of two builds of the same ABI); there is still no non-HIP path
    return os.environ.get("SVIN_BA_LIB") or os.path.join(_HERE, "libsvin_ba.so")


def load_library():
    """Load ``libsvin_ba.so`` (built in-tree by ``__graft_entry__.build()``); raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("libsvin_ba.so is missing: run __graft_entry__.build() (there is no fallback path)")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME)
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)
    sig("svin_ba_create", vp, i32)
    sig("svin_ba_destroy", None, vp)
    sig("svin_ba_last_error", C.c_char_p)
    sig("svin_ba_new_id", u64, vp)
    sig("svin_ba_add_camera", i32, vp, i32, pd, pd, i32, i32, i32, pd)
    sig("svin_ba_add_imu", i32, vp, C.POINTER(ImuParams))
    sig("svin_ba_set_sonar_extrinsics", i32, vp, pd)
    sig("svin_ba_add_states", i32, vp, u64, u32, u32, u64, pd, i32, C.c_void_p, i32, i32, pd, i32, pd, i32, f64)
    sig("svin_ba_add_landmark", i32, vp, u64, pd)
    sig("svin_ba_add_observation", u64, vp, u64, u64, u64, u64, pd, f64)
    sig("svin_ba_add_observations", i32, vp, i32, vp, vp, vp, vp, vp, vp, vp)   # (plain addresses: the call sits in the frame loop)
    sig("svin_ba_keyframe_points", i32, vp, u64, u64, i32, C.POINTER(u64), pd, C.POINTER(u64), pd, C.POINTER(i32), i32,
        C.POINTER(u64), C.POINTER(i32))
    sig("svin_ba_remove_observation", i32, vp, u64, u64, u64, u64)
    sig("svin_ba_remove_observation_by_id", i32, vp, u64)
    sig("svin_ba_add_homogeneous_point_error", u64, vp, u64, pd, pd)
    sig("svin_ba_remove_homogeneous_point_error", i32, vp, u64)
    sig("svin_host_homogeneous_point_error", i32, pd, pd, pd, pd, pd, pd)
    sig("svin_ba_optimize", i32, vp, u64, u64, i32)
    sig("svin_ba_prepare", i32, vp)
    sig("svin_ba_solve_prepared", i32, vp, u64, i32)
    sig("svin_ba_solve_prepared_batch", i32, C.POINTER(vp), i32, u64, i32, C.POINTER(i32))
    sig("svin_ba_optimize_batch", i32, C.POINTER(vp), i32, u64, i32, C.POINTER(i32))
    sig("svin_ba_finish", i32, vp)
    sig("svin_ba_invalidate_preintegration", i32, vp)
    sig("svin_ba_set_optimization_time_limit", i32, vp, f64, i32)
    sig("svin_ba_apply_marginalization_strategy", i32, vp, u64, u64, pu64, i32, C.POINTER(C.c_int))
    sig("svin_ba_get_summary", i32, vp, C.POINTER(SummaryStruct))
    sig("svin_ba_set_solver_tolerances", i32, vp, f64, f64, f64)
    sig("svin_ba_set_distributed", i32, vp, i32, i32, C.c_void_p, C.c_void_p)
    sig("svin_ba_get_T_WS", i32, vp, u64, pd)
    sig("svin_ba_get_speed_and_bias", i32, vp, u64, u64, pd)
    sig("svin_ba_get_camera_sensor_states", i32, vp, u64, u64, pd)
    sig("svin_ba_get_landmark", i32, vp, u64, C.POINTER(LandmarkInfo))
    sig("svin_ba_is_landmark_added", i32, vp, u64)
    sig("svin_ba_set_T_WS", i32, vp, u64, pd)
    sig("svin_ba_set_speed_and_bias", i32, vp, u64, u64, pd)
    sig("svin_ba_set_camera_sensor_states", i32, vp, u64, u64, pd)
    sig("svin_ba_set_landmark", i32, vp, u64, pd)
    sig("svin_ba_num_frames", u64, vp)
    sig("svin_ba_num_landmarks", u64, vp)
    sig("svin_ba_current_keyframe_id", u64, vp)
    sig("svin_ba_current_frame_id", u64, vp)
    sig("svin_ba_frame_id_by_age", u64, vp, u64)
    sig("svin_ba_is_keyframe", i32, vp, u64)
    sig("svin_ba_is_in_imu_window", i32, vp, u64)
    sig("svin_ba_frame_ids", i32, vp, pu64, i32)
    sig("svin_ba_landmark_ids", i32, vp, pu64, i32)
    sig("svin_ba_imu_propagation", i32, vp, C.c_void_p, i32, C.POINTER(ImuParams), pd, pd, u32, u32, u32, u32, pd, pd)
    sig("svin_ba_eval_reprojection", i32, vp, i32, pd, pd, pd, pd, i32)
    sig("svin_ba_observation_ids", i32, vp, pu64, pu64, pu64, pi32, i32)
    sig("svin_ba_eval_factors", i32, vp, pi32, pi32, pi32, pd, pd, pu64, pu64, i32)
    sig("svin_ba_linearize", i32, vp, f64, pd, pd, pu64, pi32, pi32, i32, pd)
    sig("svin_ba_debug_reduced_solve", i32, vp, f64, pd, i32)
    sig("svin_ba_debug_reduced_solve_ex", i32, vp, f64, i32, pd, i32)
    sig("svin_ba_debug_set_switch", i32, C.c_char_p, i32)
    sig("svin_ba_debug_set_option", i32, C.c_char_p, i32)
    sig("svin_ba_debug_get_option", i32, C.c_char_p, pi32)
    sig("svin_ba_debug_sym_eig", i32, i32, pd, pd, pd, pd)
    sig("svin_ba_get_path_counters", i32, vp, C.POINTER(C.c_int64))
    sig("svin_ba_wait_idle", i32, vp)
    sig("svin_ba_debug_peek_solver_scratch", i32, vp, u64, u64, pd)
    sig("svin_ba_get_prior", i32, vp, pd, pd, pd, pd, pu64, pi32, pi32, pi32, i32)
    sig("svin_ba_describe_block", i32, vp, u64, pu64, pi32, pi32)
    sig("svin_ba_bench_jacobian_eval", i32, vp, i32, i32, pd, pd)
    sig("svin_ba_bench_jacobian_eval_b2b", i32, vp, i32, i32, pd, pd, pd)
    sig("svin_ba_set_pack_mode", i32, vp, i32)
    sig("svin_ba_residual_info", i32, vp, i32, pu64, pi32, pi32, pi32, pi32)
    sig("svin_ba_map_add_parameter_block", i32, vp, u64, i32, pd)
    sig("svin_ba_set_parameter_block", i32, vp, u64, pd)
    sig("svin_ba_map_remove_parameter_block", i32, vp, u64)
    sig("svin_ba_map_add_pose_error", u64, vp, u64, pd, pd)
    sig("svin_ba_map_add_speed_and_bias_error", u64, vp, u64, pd, pd)
    sig("svin_ba_map_add_relative_pose_error", u64, vp, u64, u64, pd)
    sig("svin_ba_map_add_imu_error", u64, vp, pu64, vp, i32, vp, u32, u32, u32, u32)
    sig("svin_ba_map_add_sonar_error", u64, vp, u64, f64, f64, f64, pd, i32)
    sig("svin_ba_map_add_depth_error", u64, vp, u64, f64, f64, f64)
    sig("svin_ba_map_add_host_residual", u64, vp, pu64, i32, i32, COST_FUNCTION, vp)
    sig("svin_ba_map_add_reprojection_error", u64, vp, u64, u64, u64, u64, pd, pd)
    sig("svin_ba_map_remove_residual_block", i32, vp, u64)
    sig("svin_ba_debug_csr", i32, vp, pi32, pi32, pi32, pi32, C.POINTER(C.c_uint32), pd, pd, pd, pi32, pi32)
    sig("svin_ba_bench_kernel_times", i32, vp, i32, pd, pd, pd)
    sig("svin_host_imu_propagation", i32, C.c_void_p, i32, C.POINTER(ImuParams), pd, pd, u32, u32, u32, u32, pd, pd, pd)
    sig("svin_host_reprojection_error", i32, i32, pd, pd, i32, pd, pd, pd, pd, pd, pd, pd, pd, pd, pd, pd, pd)
    sig("svin_ba_get_landmark_observations", i32, vp, u64, pu64, pu64, pu64, pu64, i32)
    sig("svin_ba_parameter_block_exists", i32, vp, u64)
    sig("svin_ba_set_parameter_block_constant", i32, vp, u64, i32)
    sig("svin_ba_is_parameter_block_constant", i32, vp, u64)
    sig("svin_ba_reset_parameterization", i32, vp, u64, i32)
    sig("svin_ba_get_parameterization", i32, vp, u64)
    sig("svin_ba_residuals_of", i32, vp, u64, pu64, i32)
    sig("svin_ba_parameters_of", i32, vp, u64, pu64, i32, pi32)
    sig("svin_ba_rccl_unique_id", i32, C.c_char_p)
    sig("svin_ba_set_distributed_rccl", i32, vp, i32, i32, C.c_char_p)
    sig("svin_ba_set_id_provider", i32, vp, C.c_void_p, C.c_void_p)
    sig("svin_ba_reserve_ids", i32, vp, u64)
    sig("svin_ba_set_camera_geometry", i32, vp, u64, i32, pd, pd, i32, i32, i32)
    sig("svin_ba_clear_cameras", i32, vp)
    sig("svin_ba_clear_imus", i32, vp)
    sig("svin_ba_is_landmark_initialized", i32, vp, u64)
    sig("svin_ba_set_landmark_initialized", i32, vp, u64, i32)
    sig("svin_ba_get_landmarks", i32, vp, pu64, C.POINTER(LandmarkInfo), i32)
    sig("svin_ba_set_keyframe", i32, vp, u64, i32)
    sig("svin_ba_timestamp", i32, vp, u64, C.POINTER(u32), C.POINTER(u32))
    sig("svin_ba_state_count", i32, vp)
    sig("svin_ba_get_imu_preintegral", i32, vp, u64, pd, pd, pd)
    sig("svin_ba_set_imu_preintegral", i32, vp, u64, pd, pd, f64)
    sig("svin_ba_init_pose_from_imu", i32, C.c_void_p, i32, pd)
    sig("svin_ba_imu_propagation_integrals", i32, vp, C.c_void_p, i32, C.POINTER(ImuParams), pd, pd, u32, u32, u32, u32, pd, pd,
        pd)
    sig("svin_host_pose_information", i32, pd, pd, pd)
    sig("svin_host_pose_error", i32, pd, pd, pd, pd, pd, pd)
    sig("svin_host_manifold_dims", i32, i32, pi32, pi32)
    sig("svin_host_manifold_plus", i32, i32, pd, pd, pd)
    sig("svin_host_manifold_minus", i32, i32, pd, pd, pd)
    for name in ("plus_jacobian", "lift_jacobian", "minus_jacobian"):
        sig("svin_host_manifold_" + name, i32, i32, pd, pd)
    sig("svin_ba_get_parameter_block", i32, vp, u64, pi32, pd, C.POINTER(u32), C.POINTER(u32), pi32, pi32)
    sig("svin_ba_parameter_block_ids", i32, vp, pu64, i32)
    sig("svin_ba_bench_allreduce", i32, vp, u64, i32, pd)
    sig("svin_ba_get_marg_pre", i32, vp, pi32, pi32, pd, pd, pd, pd, pd, pi32, i32, i32)
    sig("svin_ba_get_marg_pre_blocks", i32, vp, pu64, pi32, pi32, i32, pu64, i32)
    sig("svin_ba_get_all_landmark_observations", i32, vp, i32, pu64, C.POINTER(LandmarkInfo), pi32, i32, pu64, pu64, pu64, pu64, pi32)
    _LIB = L
    return L


def _batch_call(fn_name, estimators, num_iter, verbose):
    L = load_library()
    n = len(estimators)
    arr = (C.c_void_p * max(n, 1))(*[e.h for e in estimators])
    nb = C.c_int32(0)
    rc = getattr(L, fn_name)(arr, n, num_iter, 1 if verbose else 0, C.byref(nb))
    if rc != 1:
        raise RuntimeError("%s failed (%d): %s" % (fn_name, rc, L.svin_ba_last_error().decode()))
    return int(nb.value)


def solve_prepared_batch(estimators, num_iter, verbose=False):
    """svin_ba_solve_prepared_batch: the trust-region iterations of several prepared windows (Estimator.prepare() on each) through
    one launch sequence per round; returns the number of windows that ran in a batch (the others ran one after the other)"""
    return _batch_call("svin_ba_solve_prepared_batch", estimators, num_iter, verbose)


def optimize_batch(estimators, num_iter, verbose=False):
    """svin_ba_optimize_batch: optimize() of several windows at once (prepare, batched solve, finish)"""
    return _batch_call("svin_ba_optimize_batch", estimators, num_iter, verbose)


def rccl_unique_id():
    """128-byte ncclUniqueId (call on rank 0, hand to every rank)"""
    L = load_library()
    buf = C.create_string_buffer(128)
    if L.svin_ba_rccl_unique_id(buf) != 1:
        raise RuntimeError("svin_ba_rccl_unique_id failed: " + L.svin_ba_last_error().decode())
    return buf.raw


def _d(a):
    return None if a is None else a.ctypes.data_as(pd)


def host_imu_propagation(imu_t, imu_m, params, T, sb, t0, t1, want_cov=False, want_jac=False):
    """CPU twin of ImuError::propagation (svin_host_imu_propagation): returns (n, T, sb, cov, jac, integrals)"""
    L = load_library()
    s = pack_imu(imu_t, imu_m)
    q = make_imu_params(params)
    T, sb = _arr(T).copy(), _arr(sb).copy()
    cov = np.zeros((15, 15)) if want_cov else None
    jac = np.zeros((15, 15)) if want_jac else None
    integ = np.zeros(7)
    n = L.svin_host_imu_propagation(s.ctypes.data_as(C.c_void_p), len(s), C.byref(q), _d(T), _d(sb), t0[0], t0[1], t1[0], t1[1],
                                    _d(cov), _d(jac), _d(integ))
    return n, T, sb, cov, jac, integ


def host_homogeneous_point_error(hp, measurement, information):
    """CPU twin of HomogeneousPointError::EvaluateWithMinimalJacobians: (residual[3], J_min 3x3, J 3x4)"""
    L = load_library()
    r, Jm, J = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 4))
    rc = L.svin_host_homogeneous_point_error(_d(_arr(hp)), _d(_arr(measurement)), _d(_arr(np.asarray(information, float).reshape(3, 3))),
                                             _d(r), _d(Jm), _d(J))
    if rc != 1:
        raise RuntimeError("svin_host_homogeneous_point_error failed (%d)" % rc)
    return r, Jm, J


MANIFOLD_HPOINT, MANIFOLD_POSE6D, MANIFOLD_POSE3D, MANIFOLD_POSE4D, MANIFOLD_POSE2D = range(5)


def host_pose_information(information):
    """PoseError::setInformation (svin_host_pose_information): (sqrt_information 6x6 upper, covariance 6x6)"""
    L = load_library()
    W, cov = np.zeros((6, 6)), np.zeros((6, 6))
    if L.svin_host_pose_information(_d(_arr(np.asarray(information, float).reshape(6, 6))), _d(W), _d(cov)) != 1:
        raise RuntimeError("svin_host_pose_information failed")
    return W, cov


def host_pose_error(measurement, information, T_WS):
    """CPU twin of PoseError::EvaluateWithMinimalJacobians (svin_host_pose_error): (residual[6], J_min 6x6, J 6x7)"""
    L = load_library()
    W, _ = host_pose_information(information)
    r, Jm, J = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 7))
    if L.svin_host_pose_error(_d(_arr(measurement)), _d(W), _d(_arr(T_WS)), _d(r), _d(Jm), _d(J)) != 1:
        raise RuntimeError("svin_host_pose_error failed")
    return r, Jm, J


def host_manifold(kind, x, delta=None):
    """the parameter-block manifolds (svin_host_manifold_*): dict with dims, plus / minus (if delta is given) and the Jacobians"""
    L = load_library()
    na, nt = C.c_int32(), C.c_int32()
    if L.svin_host_manifold_dims(kind, C.byref(na), C.byref(nt)) != 1:
        raise RuntimeError("unknown manifold kind %r" % (kind,))
    na, nt = na.value, nt.value
    x = _arr(x)
    out = dict(ambient=na, tangent=nt, J_plus=np.zeros((na, nt)), J_lift=np.zeros((nt, na)), J_minus=np.zeros((nt, na)))
    L.svin_host_manifold_plus_jacobian(kind, _d(x), _d(out["J_plus"]))
    L.svin_host_manifold_lift_jacobian(kind, _d(x), _d(out["J_lift"]))
    L.svin_host_manifold_minus_jacobian(kind, _d(x), _d(out["J_minus"]))
    if delta is not None:
        xp, d = np.zeros(na), np.zeros(nt)
        L.svin_host_manifold_plus(kind, _d(x), _d(_arr(delta)), _d(xp))
        L.svin_host_manifold_minus(kind, _d(xp), _d(x), _d(d))
        out["plus"], out["minus"] = xp, d
    return out


def host_reprojection_error(model, intr, dist, T_WS, hp, T_SC, uv, information):
    """CPU twin of one ReprojectionError::EvaluateWithMinimalJacobians (svin_host_reprojection_error)"""
    L = load_library()
    intr, dist, T_WS, hp, T_SC, uv, info = (_arr(x) for x in (intr, dist, T_WS, hp, T_SC, uv, information))
    r, Jp, Jl, Je = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 6))
    Ap, Al, Ae = np.zeros((2, 7)), np.zeros((2, 4)), np.zeros((2, 7))
    rc = L.svin_host_reprojection_error(model, _d(intr), _d(dist) if len(dist) else None, len(dist), _d(T_WS), _d(hp), _d(T_SC), _d(uv),
                                        _d(info.reshape(-1)), _d(r), _d(Jp), _d(Jl), _d(Je), _d(Ap), _d(Al), _d(Ae))
    if rc != 1:
        raise RuntimeError("svin_host_reprojection_error failed (%d)" % rc)
    return dict(r=r, Jp=Jp, Jl=Jl, Je=Je, J_pose=Ap, J_lm=Al, J_ext=Ae)


def _arr(x, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


def pack_imu(imu_t, imu_m):
    imu_t = _arr(imu_t, np.uint32).reshape(-1, 2)
    imu_m = _arr(imu_m).reshape(-1, 6)
    s = np.zeros(len(imu_t), IMU_SAMPLE_DTYPE)
    s["sec"], s["nsec"] = imu_t[:, 0], imu_t[:, 1]
    s["gyr"], s["acc"] = imu_m[:, :3], imu_m[:, 3:]
    assert s.itemsize == C.sizeof(ImuSample)
    return s


def make_imu_params(p):
    q = ImuParams()
    for k in ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c", "tau", "g"):
        setattr(q, k, float(p[k]))
    a0 = p.get("a0", [0.0, 0.0, 0.0])
    q.a0[0], q.a0[1], q.a0[2] = float(a0[0]), float(a0[1]), float(a0[2])
    return q


class Estimator:
    """MI355X-native drop-in for ``okvis::Estimator`` (hot path only)."""

    def __init__(self, device=0):
        self.L = load_library()
        h = self.L.svin_ba_create(device)
        if not h:
            raise RuntimeError("svin_ba_create failed: " + self.L.svin_ba_last_error().decode())
        self.h = C.c_void_p(h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.svin_ba_destroy(self.h)
            self.h = None

    def _check(self, r, what):
        if r < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, r, self.L.svin_ba_last_error().decode()))
        return r

    # -- construction ---------------------------------------------------------------------------
    def new_id(self):
        return int(self.L.svin_ba_new_id(self.h))

    def set_id_provider(self, fn):
        """fn() -> int: the host's id source (okvis::IdProvider); None restores the internal counter"""
        self._id_cb = ID_PROVIDER_FN(lambda _u: int(fn())) if fn is not None else None
        self._check(self.L.svin_ba_set_id_provider(self.h, C.cast(self._id_cb, C.c_void_p) if fn is not None else None, None),
                    "set_id_provider")

    def reserve_ids(self, largest):
        self._check(self.L.svin_ba_reserve_ids(self.h, int(largest)), "reserve_ids")

    def set_camera_geometry(self, cam, model, intr, dist, w, h):
        intr, dist = _arr(intr), _arr(dist)
        return self.L.svin_ba_set_camera_geometry(self.h, cam, model, _d(intr), _d(dist) if len(dist) else None, len(dist), w, h) == 1

    def clear_cameras(self):
        self.L.svin_ba_clear_cameras(self.h)

    def clear_imus(self):
        self.L.svin_ba_clear_imus(self.h)

    def add_camera(self, model, intr, dist, w, h, sigmas):
        intr, dist, sig = _arr(intr), _arr(dist), _arr(sigmas)
        return self._check(self.L.svin_ba_add_camera(self.h, model, _d(intr), _d(dist) if len(dist) else None, len(dist), w, h,
                                                     _d(sig)), "add_camera")

    def add_imu(self, params):
        q = make_imu_params(params)
        return self._check(self.L.svin_ba_add_imu(self.h, C.byref(q)), "add_imu")

    def set_sonar_extrinsics(self, T):
        T = _arr(T)
        return self._check(self.L.svin_ba_set_sonar_extrinsics(self.h, _d(T)), "set_sonar_extrinsics")

    def add_states(self, fid, stamp, num_keypoints, T_SC, imu_t, imu_m, as_keyframe, sonar=None, depth=None, first_depth=0.0):
        T_SC = _arr(T_SC).reshape(-1, 7)
        s = pack_imu(imu_t, imu_m)
        sonar = _arr(sonar if sonar is not None else np.zeros((0, 2))).reshape(-1, 2)
        depth = _arr(depth if depth is not None else np.zeros(0)).reshape(-1)
        r = self.L.svin_ba_add_states(self.h, fid, stamp[0], stamp[1], num_keypoints, _d(T_SC), len(T_SC),
                                      s.ctypes.data_as(C.c_void_p), len(s), 1 if as_keyframe else 0,
                                      _d(sonar) if len(sonar) else None, len(sonar), _d(depth) if len(depth) else None,
                                      len(depth), first_depth)
        return bool(self._check(r, "add_states"))

    def add_landmark(self, lid, hp):
        hp = _arr(hp)
        return bool(self._check(self.L.svin_ba_add_landmark(self.h, lid, _d(hp)), "add_landmark"))

    def add_observation(self, lid, pose, cam, kp, uv, size):
        uv = _arr(uv)
        return int(self.L.svin_ba_add_observation(self.h, lid, pose, cam, kp, _d(uv), size))

    def add_observations(self, lids, poses, cams, kps, uvs, sizes):
        """batched add_observation (svin_ba_add_observations); returns the residual ids (0 = duplicate)"""
        n = len(lids)
        a = [np.ascontiguousarray(x, np.uint64) for x in (lids, poses, cams, kps)]
        uvs = np.ascontiguousarray(uvs, np.float64).reshape(n, 2)
        sizes = np.ascontiguousarray(sizes, np.float64)
        out = np.zeros(n, np.uint64)
        self._check(self.L.svin_ba_add_observations(self.h, n, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                                    uvs.ctypes.data, sizes.ctypes.data, out.ctypes.data), "add_observations")
        return out

    def keyframe_points(self, frame_id, cam=0):
        """estimator-side content of the keyframe message for pose_graph (ThreadedKFVio.cpp:1147-1240): landmark ids,
        Euclidean points, keypoint indices, qualities and per point the frame ids of its other observations"""
        p64, p32 = C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
        nt = C.c_int32(0)
        n = self._check(self.L.svin_ba_keyframe_points(self.h, frame_id, cam, 0, None, None, None, None, None, 0, None,
                                                       C.byref(nt)), "keyframe_points")
        ids, kps = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        xyz, q = np.zeros((n, 3)), np.zeros(n)
        ptr, fr = np.zeros(n + 1, np.int32), np.zeros(max(nt.value, 1), np.uint64)
        self._check(self.L.svin_ba_keyframe_points(self.h, frame_id, cam, n, ids.ctypes.data_as(p64), _d(xyz),
                                                   kps.ctypes.data_as(p64), _d(q), ptr.ctypes.data_as(p32), nt.value,
                                                   fr.ctypes.data_as(p64), C.byref(nt)), "keyframe_points")
        return ids, xyz, kps, q, [fr[ptr[i]:ptr[i + 1]].copy() for i in range(n)]

    def remove_observation(self, lid, pose, cam, kp):
        return bool(self._check(self.L.svin_ba_remove_observation(self.h, lid, pose, cam, kp), "remove_observation"))

    def remove_observation_by_id(self, rid):
        return bool(self._check(self.L.svin_ba_remove_observation_by_id(self.h, rid), "remove_observation_by_id"))

    def add_homogeneous_point_error(self, lid, measurement, information=None, variance=None):
        """HomogeneousPointError on a landmark; information 3x3 or variance (information = I / variance); returns the residual id"""
        info = np.eye(3) / variance if information is None else np.asarray(information, float).reshape(3, 3)
        m, info = _arr(measurement), _arr(info)
        return int(self.L.svin_ba_add_homogeneous_point_error(self.h, lid, _d(m), _d(info)))

    def remove_homogeneous_point_error(self, rid):
        return bool(self._check(self.L.svin_ba_remove_homogeneous_point_error(self.h, rid), "remove_homogeneous_point_error"))

    # -- hot path ---------------------------------------------------------------------------------
    def optimize(self, num_iter, num_threads=1, verbose=False):
        self._check(self.L.svin_ba_optimize(self.h, num_iter, num_threads, 1 if verbose else 0), "optimize")

    def prepare(self):
        self._check(self.L.svin_ba_prepare(self.h), "prepare")

    def solve_prepared(self, num_iter, verbose=False):
        self._check(self.L.svin_ba_solve_prepared(self.h, num_iter, 1 if verbose else 0), "solve_prepared")

    def finish(self):
        self._check(self.L.svin_ba_finish(self.h), "finish")

    def invalidate_preintegration(self):
        self.L.svin_ba_invalidate_preintegration(self.h)

    def set_distributed(self, rank, world, allreduce_cb):
        """landmark-sharded mode; `allreduce_cb` is a distributed.ALLREDUCE_FN instance (kept alive here)"""
        self._allreduce_cb = allreduce_cb
        self._check(self.L.svin_ba_set_distributed(self.h, rank, world, C.cast(allreduce_cb, C.c_void_p), None),
                    "set_distributed")

    def set_distributed_rccl(self, rank, world, unique_id):
        """landmark-sharded mode with native RCCL on the solver's stream; `unique_id` = 128 bytes from rccl_unique_id()
        on rank 0, distributed by the caller (svin_amd.distributed.init_rccl does it over torch.distributed)"""
        assert len(unique_id) == 128
        self._check(self.L.svin_ba_set_distributed_rccl(self.h, rank, world, bytes(unique_id)), "set_distributed_rccl")

    def set_time_limit(self, tl, min_iter):
        return bool(self.L.svin_ba_set_optimization_time_limit(self.h, tl, min_iter))

    def apply_marginalization(self, num_kf, num_imu):
        ids = getattr(self, "_removed_buf", None)   # (a fresh 512 KB array per call cost more than the policy loop it serves)
        if ids is None:
            ids = self._removed_buf = np.zeros(1 << 16, np.uint64)
            self._removed_ptr = ids.ctypes.data_as(pu64)
            self._removed_n = C.c_int()
        n = self._removed_n
        r = self._check(self.L.svin_ba_apply_marginalization_strategy(self.h, num_kf, num_imu, self._removed_ptr, len(ids), C.byref(n)),
                        "apply_marginalization")
        return bool(r), ids[:n.value].copy()

    def summary(self):
        s = SummaryStruct()
        self.L.svin_ba_get_summary(self.h, C.byref(s))
        return dict(initial_cost=s.initial_cost, final_cost=s.final_cost, iterations=s.iterations,
                    successful=s.num_successful_steps, termination=s.termination, time=s.total_time_s,
                    upload_time=s.upload_time_s, solve_time=s.solve_time_s, download_time=s.download_time_s)

    def set_solver_options(self, function_tol=1e-6, gradient_tol=1e-10, parameter_tol=1e-8, jacobi_scaling=True):
        assert jacobi_scaling, "the device solver always applies Ceres' default Jacobi scaling"
        self.L.svin_ba_set_solver_tolerances(self.h, function_tol, gradient_tol, parameter_tol)

    # -- getters / setters --------------------------------------------------------------------------
    def get_T_WS(self, fid):
        T = np.zeros(7)
        return T if self.L.svin_ba_get_T_WS(self.h, fid, _d(T)) == 1 else None

    def get_speed_and_bias(self, fid, imu=0):
        sb = np.zeros(9)
        return sb if self.L.svin_ba_get_speed_and_bias(self.h, fid, imu, _d(sb)) == 1 else None

    def get_camera_sensor_states(self, fid, cam):
        T = np.zeros(7)
        return T if self.L.svin_ba_get_camera_sensor_states(self.h, fid, cam, _d(T)) == 1 else None

    def get_landmark(self, lid):
        info = LandmarkInfo()
        if self.L.svin_ba_get_landmark(self.h, lid, C.byref(info)) != 1:
            return None
        return dict(point=np.array(info.point[:]), quality=info.quality, distance=info.distance,
                    n_obs=info.num_observations, initialized=bool(info.initialized))

    def get_landmarks(self):
        """Estimator::getLandmarks: {id: dict} in PointMap order"""
        n = self.L.svin_ba_get_landmarks(self.h, None, None, 0)
        ids, infos = np.zeros(max(n, 1), np.uint64), (LandmarkInfo * max(n, 1))()
        self.L.svin_ba_get_landmarks(self.h, ids.ctypes.data_as(pu64), infos, n)
        return {int(ids[i]): dict(point=np.array(infos[i].point[:]), quality=infos[i].quality, distance=infos[i].distance,
                                  n_obs=infos[i].num_observations, initialized=bool(infos[i].initialized)) for i in range(n)}

    def landmark_observations(self, lid):
        """MapPoint::observations: [(frame id, camera, keypoint, residual id)] in KeypointIdentifier order"""
        n = self._check(self.L.svin_ba_get_landmark_observations(self.h, lid, None, None, None, None, 0), "landmark_observations")
        a = [np.zeros(max(n, 1), np.uint64) for _ in range(4)]
        self.L.svin_ba_get_landmark_observations(self.h, lid, *[x.ctypes.data_as(pu64) for x in a], n)
        return [tuple(int(x[i]) for x in a) for i in range(n)]

    def is_landmark_added(self, lid):
        return self.L.svin_ba_is_landmark_added(self.h, lid) == 1

    def is_landmark_initialized(self, lid):
        return self._check(self.L.svin_ba_is_landmark_initialized(self.h, lid), "is_landmark_initialized") == 1

    def set_landmark_initialized(self, lid, flag):
        return self.L.svin_ba_set_landmark_initialized(self.h, lid, 1 if flag else 0) == 1

    def set_keyframe(self, fid, flag):
        return self.L.svin_ba_set_keyframe(self.h, fid, 1 if flag else 0) == 1

    def is_keyframe(self, fid):
        return self._check(self.L.svin_ba_is_keyframe(self.h, fid), "is_keyframe") == 1

    def timestamp(self, fid):
        a, b = u32(), u32()
        return (a.value, b.value) if self.L.svin_ba_timestamp(self.h, fid, C.byref(a), C.byref(b)) == 1 else None

    def state_count(self):
        return int(self.L.svin_ba_state_count(self.h))

    def get_imu_preintegral(self, fid):
        a, b, dt = np.zeros(3), np.zeros(3), np.zeros(1)
        return (a, b, float(dt[0])) if self.L.svin_ba_get_imu_preintegral(self.h, fid, _d(a), _d(b), _d(dt)) == 1 else None

    def set_imu_preintegral(self, fid, adi, ai, dt):
        adi, ai = _arr(adi), _arr(ai)
        return self.L.svin_ba_set_imu_preintegral(self.h, fid, _d(adi), _d(ai), float(dt)) == 1

    def init_pose_from_imu(self, imu_t, imu_m):
        s = pack_imu(imu_t, imu_m)
        T = np.zeros(7)
        ok = self.L.svin_ba_init_pose_from_imu(s.ctypes.data_as(C.c_void_p), len(s), _d(T))
        return ok == 1, T

    # -- okvis::ceres::Map graph queries ------------------------------------------------------------
    def parameter_block_exists(self, bid):
        return self.L.svin_ba_parameter_block_exists(self.h, bid) == 1

    def set_parameter_block_constant(self, bid, constant=True):
        return self._check(self.L.svin_ba_set_parameter_block_constant(self.h, bid, 1 if constant else 0), "set_parameter_block_constant") == 1

    # Map::Parameterization (Map.hpp:97-105)
    HOMOGENEOUS_POINT, POSE6D, POSE3D, POSE4D, POSE2D, TRIVIAL = range(6)

    def reset_parameterization(self, bid, parameterization):
        """Map::resetParameterization (Map.cpp:513-543): False for an unknown block, RuntimeError for a manifold the block cannot take"""
        return self._check(self.L.svin_ba_reset_parameterization(self.h, bid, int(parameterization)), "reset_parameterization") == 1

    def parameterization(self, bid):
        return self._check(self.L.svin_ba_get_parameterization(self.h, bid), "get_parameterization")

    def is_parameter_block_constant(self, bid):
        return self._check(self.L.svin_ba_is_parameter_block_constant(self.h, bid), "is_parameter_block_constant") == 1

    def residuals_of(self, bid):
        n = self._check(self.L.svin_ba_residuals_of(self.h, bid, None, 0), "residuals_of")
        out = np.zeros(max(n, 1), np.uint64)
        self.L.svin_ba_residuals_of(self.h, bid, out.ctypes.data_as(pu64), n)
        return [int(v) for v in out[:n]]

    def parameters_of(self, rid):
        kind = C.c_int32()
        n = self._check(self.L.svin_ba_parameters_of(self.h, rid, None, 0, C.byref(kind)), "parameters_of")   # the count first: a prior lists every block it touches
        out = np.zeros(max(n, 1), np.uint64)
        self.L.svin_ba_parameters_of(self.h, rid, out.ctypes.data_as(pu64), n, C.byref(kind))
        return [int(v) for v in out[:n]], int(kind.value)

    def parameter_block(self, bid):
        """Map::parameterBlockPtr as a value: dict(type 0 pose / 1 extrinsics / 2 speed-bias / 3 landmark, values, stamp, fixed, initialized)"""
        t, fx, ini, sec, nsec, x = C.c_int32(), C.c_int32(), C.c_int32(), u32(), u32(), np.zeros(9)
        dim = self._check(self.L.svin_ba_get_parameter_block(self.h, bid, C.byref(t), _d(x), C.byref(sec), C.byref(nsec), C.byref(fx), C.byref(ini)),
                          "get_parameter_block")
        return dict(type=t.value, values=x[:dim].copy(), stamp=(sec.value, nsec.value), fixed=bool(fx.value), initialized=bool(ini.value))

    def parameter_block_ids(self):
        n = self.L.svin_ba_parameter_block_ids(self.h, None, 0)
        out = np.zeros(max(n, 1), np.uint64)
        self.L.svin_ba_parameter_block_ids(self.h, out.ctypes.data_as(pu64), n)
        return [int(v) for v in out[:n]]

    def all_landmark_observations(self):
        """getLandmarks with the observation maps in ONE call: {id: (info dict, [(frame, cam, keypoint, residual id)])}"""
        tot = C.c_int32()
        n = self.L.svin_ba_get_all_landmark_observations(self.h, 0, None, None, None, 0, None, None, None, None, C.byref(tot))
        m = tot.value
        ids, infos, ptr = np.zeros(max(n, 1), np.uint64), (LandmarkInfo * max(n, 1))(), np.zeros(n + 1, np.int32)
        a = [np.zeros(max(m, 1), np.uint64) for _ in range(4)]
        self.L.svin_ba_get_all_landmark_observations(self.h, n, ids.ctypes.data_as(pu64), infos, ptr.ctypes.data_as(pi32), m,
                                                     *[x.ctypes.data_as(pu64) for x in a], C.byref(tot))
        out = {}
        for i in range(n):
            info = dict(point=np.array(infos[i].point[:]), quality=infos[i].quality, distance=infos[i].distance,
                        n_obs=infos[i].num_observations, initialized=bool(infos[i].initialized))
            out[int(ids[i])] = (info, [tuple(int(x[k]) for x in a) for k in range(ptr[i], ptr[i + 1])])
        return out

    def current_keyframe_id(self):
        return int(self.L.svin_ba_current_keyframe_id(self.h))

    def current_frame_id(self):
        return int(self.L.svin_ba_current_frame_id(self.h))

    def frame_id_by_age(self, age):
        return int(self.L.svin_ba_frame_id_by_age(self.h, age))

    def is_in_imu_window(self, fid):
        return self.L.svin_ba_is_in_imu_window(self.h, fid) == 1

    def set_camera_sensor_states(self, fid, cam, T):
        T = _arr(T)
        return self.L.svin_ba_set_camera_sensor_states(self.h, fid, cam, _d(T)) == 1

    def set_T_WS(self, fid, T):
        T = _arr(T)
        return self.L.svin_ba_set_T_WS(self.h, fid, _d(T)) == 1

    def set_speed_and_bias(self, fid, sb, imu=0):
        sb = _arr(sb)
        return self.L.svin_ba_set_speed_and_bias(self.h, fid, imu, _d(sb)) == 1

    def set_landmark(self, lid, hp):
        hp = _arr(hp)
        return self.L.svin_ba_set_landmark(self.h, lid, _d(hp)) == 1

    def frame_ids(self):
        ids = np.zeros(4096, np.uint64)
        n = self.L.svin_ba_frame_ids(self.h, ids.ctypes.data_as(pu64), len(ids))
        return [int(i) for i in ids[:n]]

    def landmark_ids(self):
        n = int(self.L.svin_ba_num_landmarks(self.h))
        ids = np.zeros(max(n, 1), np.uint64)
        self.L.svin_ba_landmark_ids(self.h, ids.ctypes.data_as(pu64), len(ids))
        return [int(i) for i in ids[:n]]

    def num_frames(self):
        return int(self.L.svin_ba_num_frames(self.h))

    def num_landmarks(self):
        return int(self.L.svin_ba_num_landmarks(self.h))

    def imu_propagation(self, imu_t, imu_m, params, T, sb, t0, t1, want_cov=False, want_jac=False, want_integrals=False):
        s = pack_imu(imu_t, imu_m)
        q = make_imu_params(params)
        T, sb = _arr(T).copy(), _arr(sb).copy()
        cov = np.zeros((15, 15)) if want_cov else None
        jac = np.zeros((15, 15)) if want_jac else None
        if want_integrals:   # second overload: acc_doubleintegral, acc_integral, Delta_t
            integ = np.zeros(7)
            n = self._check(self.L.svin_ba_imu_propagation_integrals(self.h, s.ctypes.data_as(C.c_void_p), len(s), C.byref(q),
                                                                     _d(T), _d(sb), t0[0], t0[1], t1[0], t1[1], _d(cov),
                                                                     _d(jac), _d(integ)) + 1, "imu_propagation") - 1
            return n, T, sb, cov, jac, integ
        n = self._check(self.L.svin_ba_imu_propagation(self.h, s.ctypes.data_as(C.c_void_p), len(s), C.byref(q), _d(T), _d(sb),
                                                       t0[0], t0[1], t1[0], t1[1], _d(cov), _d(jac)) + 1, "imu_propagation") - 1
        return n, T, sb, cov, jac

    # -- inspection hooks -----------------------------------------------------------------------------
    def eval_reprojection(self, robust=False):
        n = self._check(self.L.svin_ba_eval_reprojection(self.h, 1 if robust else 0, None, None, None, None, 0), "eval")
        r, Jp, Jl, Je = np.zeros((n, 2)), np.zeros((n, 2, 6)), np.zeros((n, 2, 3)), np.zeros((n, 2, 6))
        rid, lm, pose, cam = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.int32)
        if n:
            self.L.svin_ba_eval_reprojection(self.h, 1 if robust else 0, _d(r), _d(Jp), _d(Jl), _d(Je), n)
            self.L.svin_ba_observation_ids(self.h, rid.ctypes.data_as(pu64), lm.ctypes.data_as(pu64),
                                           pose.ctypes.data_as(pu64), cam.ctypes.data_as(pi32), n)
        return dict(r=r, Jp=Jp, Jl=Jl, Je=Je, res_id=rid, lm_id=lm, pose_id=pose, cam=cam)

    def eval_factors(self):
        n = self._check(self.L.svin_ba_eval_factors(self.h, None, None, None, None, None, None, None, 0), "eval_factors")
        kind, m, nc = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        r, J, blk, rid = np.zeros((n, 15)), np.zeros((n, 450)), np.zeros((n, 4), np.uint64), np.zeros(n, np.uint64)
        if n:
            self.L.svin_ba_eval_factors(self.h, kind.ctypes.data_as(pi32), m.ctypes.data_as(pi32), nc.ctypes.data_as(pi32),
                                        _d(r), _d(J), blk.ctypes.data_as(pu64), rid.ctypes.data_as(pu64), n)
        out = []
        for i in range(n):
            out.append(dict(kind=int(kind[i]), m=int(m[i]), res_id=int(rid[i]), r=r[i, :m[i]].copy(),
                            J=J[i, :m[i] * nc[i]].reshape(m[i], nc[i]).copy(), blocks=[int(b) for b in blk[i] if b]))
        return out

    def linearize(self, mu=0.0, cap=4096):
        S, g = np.zeros((cap, cap)), np.zeros(cap)
        ids, off, nb, cost = np.zeros(cap, np.uint64), np.zeros(cap, np.int32), C.c_int32(), np.zeros(1)
        # first query the size
        d = self.L.svin_ba_linearize(self.h, mu, None, None, ids.ctypes.data_as(pu64), off.ctypes.data_as(pi32), C.byref(nb),
                                     cap, _d(cost))
        d = self._check(d, "linearize")
        S, g = np.zeros((d, d)), np.zeros(d)
        self.L.svin_ba_linearize(self.h, mu, _d(S), _d(g), ids.ctypes.data_as(pu64), off.ctypes.data_as(pi32), C.byref(nb), d,
                                 _d(cost))
        return dict(d=d, S=S, g=g, block_ids=ids[:nb.value].copy(), block_off=off[:nb.value].copy(), cost=float(cost[0]))

    def wait_idle(self):
        """blocks until the device work the handle has enqueued (the last marginalisation job) has run"""
        self._check(self.L.svin_ba_wait_idle(self.h), "wait_idle")

    def debug_reduced_solve(self, mu=0.0, cap=4096, fused=False):
        """the Gauss-Newton step of linearize(mu)'s system as the device solver computes it (fused: metric and damping applied in
        the solver's load phase, the form optimize() runs)"""
        y = np.zeros(cap)
        d = self._check(self.L.svin_ba_debug_reduced_solve_ex(self.h, float(mu), 1 if fused else 0, _d(y), cap), "debug_reduced_solve")
        return y[:d].copy()

    def path_counters(self):
        """dict(resident_solves, host_pack_solves, device_gathered_marginalisations, host_assembled_marginalisations)"""
        out = (C.c_int64 * 4)()
        self._check(self.L.svin_ba_get_path_counters(self.h, out), "path_counters")
        return dict(resident_solves=out[0], host_pack_solves=out[1], device_gathered_marginalisations=out[2],
                    host_assembled_marginalisations=out[3])

    @staticmethod
    def debug_sym_eig(A):
        """the prior's eigen-solver (tridiagonalisation + divide and conquer, one workgroup) on a symmetric matrix:
        (eigenvalues ascending, eigenvectors as columns, device milliseconds)"""
        A = np.ascontiguousarray(A, dtype=np.float64)
        n = A.shape[0]
        lam, X, ms = np.zeros(n), np.zeros((n, n)), np.zeros(1)
        rc = load_library().svin_ba_debug_sym_eig(n, _d(A), _d(lam), _d(X), _d(ms))
        if rc != 1:
            raise RuntimeError("svin_ba_debug_sym_eig: %d" % rc)
        return lam, X, float(ms[0])

    _MARG_EIG = {"": 0, None: 0, "direct": 1, "cholesky": 2, "jacobi": 3}

    @staticmethod
    def debug_set_option(name, value):
        """process-wide debug / A-B option of the library, named after the environment variable that initialises it
        (include/svin_ba.h: svin_ba_debug_set_option); "SVIN_MARG_EIG" also takes "direct" / "cholesky" / "jacobi" / None"""
        if name == "SVIN_MARG_EIG" and not isinstance(value, (int, bool)):
            value = Estimator._MARG_EIG[value]
        if load_library().svin_ba_debug_set_option(name.encode(), int(value)) != 1:
            raise KeyError(name)

    @staticmethod
    def debug_get_option(name):
        v = np.zeros(1, dtype=np.int32)
        if load_library().svin_ba_debug_get_option(name.encode(), v.ctypes.data_as(C.POINTER(C.c_int32))) != 1:
            raise KeyError(name)
        return int(v[0])

    @staticmethod
    def debug_set_switch(name, value):
        """round-4 spelling of debug_set_option for the on / off switches of the reduced solve"""
        Estimator.debug_set_option(name, 1 if value else 0)

    def debug_peek_solver_scratch(self, offset, count):
        out = np.zeros(int(count))
        if self.L.svin_ba_debug_peek_solver_scratch(self.h, int(offset), int(count), _d(out)) != 1:
            raise RuntimeError("peek outside the solver scratch")
        return out

    def describe_block(self, bid):
        f, k, ix = C.c_uint64(), C.c_int32(), C.c_int32()
        if self.L.svin_ba_describe_block(self.h, int(bid), C.byref(f), C.byref(k), C.byref(ix)) != 1:
            return None
        return int(f.value), int(k.value), int(ix.value)

    def marg(self, cap=2048):
        ids, ordr, md, nb = np.zeros(512, np.uint64), np.zeros(512, np.int32), np.zeros(512, np.int32), C.c_int32()
        n = self.L.svin_ba_get_prior(self.h, None, None, None, None, ids.ctypes.data_as(pu64), ordr.ctypes.data_as(pi32),
                                     md.ctypes.data_as(pi32), C.byref(nb), cap)
        if n <= 0:
            return None
        H, b0, J, e0 = np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)
        self.L.svin_ba_get_prior(self.h, _d(H), _d(b0), _d(J), _d(e0), ids.ctypes.data_as(pu64), ordr.ctypes.data_as(pi32),
                                 md.ctypes.data_as(pi32), C.byref(nb), n)
        blocks = []
        for i in range(nb.value):
            d = self.describe_block(ids[i])
            blocks.append(dict(id=int(ids[i]), ordering=int(ordr[i]), mdim=int(md[i]), frame=d[0] if d else None,
                               kind=d[1] if d else None, index=d[2] if d else None))
        return dict(n=n, H=H, b0=b0, J=J, e0=e0, blocks=blocks)

    # -- measurement hooks -------------------------------------------------------------------------------
    def bench_jacobian_eval(self, copies, iters):
        ms, by = np.zeros(1), np.zeros(1)
        self._check(self.L.svin_ba_bench_jacobian_eval(self.h, copies, iters, _d(ms), _d(by)), "bench_jacobian_eval")
        return float(ms[0]), float(by[0])

    # -- okvis::ceres::Map as a graph builder (Map.cpp:255-376) ---------------------------------------------
    BLOCK_POSE, BLOCK_SPEED_AND_BIAS, BLOCK_HOMOGENEOUS_POINT = 0, 2, 3

    def map_add_parameter_block(self, bid, btype, values):
        v = _arr(values)
        return self._check(self.L.svin_ba_map_add_parameter_block(self.h, bid, btype, _d(v)), "map_add_parameter_block") == 1

    def set_parameter_block(self, bid, values):
        v = _arr(values)
        return self._check(self.L.svin_ba_set_parameter_block(self.h, bid, _d(v)), "set_parameter_block") == 1

    def get_parameter_block(self, bid):
        t, fx, ini = C.c_int32(), C.c_int32(), C.c_int32()
        x = np.zeros(9)
        d = self.L.svin_ba_get_parameter_block(self.h, bid, C.byref(t), _d(x), None, None, C.byref(fx), C.byref(ini))
        return None if d < 0 else x[:d].copy()

    def map_remove_parameter_block(self, bid):
        return self._check(self.L.svin_ba_map_remove_parameter_block(self.h, bid), "map_remove_parameter_block") == 1

    def map_add_pose_error(self, bid, measurement, information):
        m, i = _arr(measurement), _arr(np.asarray(information, float).reshape(6, 6))
        return int(self.L.svin_ba_map_add_pose_error(self.h, bid, _d(m), _d(i)))

    def map_add_speed_and_bias_error(self, bid, measurement, information):
        m, i = _arr(measurement), _arr(np.asarray(information, float).reshape(9, 9))
        return int(self.L.svin_ba_map_add_speed_and_bias_error(self.h, bid, _d(m), _d(i)))

    def map_add_relative_pose_error(self, b0, b1, information):
        i = _arr(np.asarray(information, float).reshape(6, 6))
        return int(self.L.svin_ba_map_add_relative_pose_error(self.h, b0, b1, _d(i)))

    def map_add_imu_error(self, blocks4, imu_t, imu_m, params, t0, t1):
        """ImuError on (pose_0, speed/bias_0, pose_1, speed/bias_1); imu_t (n, 2) uint32 stamps, imu_m (n, 6) gyr | acc"""
        ids = np.asarray(blocks4, np.uint64)
        imu = pack_imu(imu_t, imu_m)
        p = make_imu_params(params)
        return int(self.L.svin_ba_map_add_imu_error(self.h, ids.ctypes.data_as(pu64), imu.ctypes.data_as(C.c_void_p), len(imu), C.byref(p),
                                                    int(t0[0]), int(t0[1]), int(t1[0]), int(t1[1])))

    def map_add_sonar_error(self, pose, rng_m, heading, information, patch):
        pt = _arr(np.asarray(patch, float).reshape(-1, 3))
        return int(self.L.svin_ba_map_add_sonar_error(self.h, pose, float(rng_m), float(heading), float(information), _d(pt), len(pt)))

    def map_add_depth_error(self, pose, depth, information, first_depth):
        return int(self.L.svin_ba_map_add_depth_error(self.h, pose, float(depth), float(information), float(first_depth)))

    def map_add_host_residual(self, block_ids, block_dims, residual_dim, fn):
        """svin_ba_map_add_host_residual: a residual block whose cost function Python evaluates.  fn(params) gets the blocks as a list
        of numpy arrays (7 or 9 numbers each, in the order of block_ids) and returns (residuals, [J_0, J_1, ...]) with J_b of shape
        residual_dim x (6 | 9) in minimal coordinates; block_dims = the 7 / 9 of each block (for unpacking the pointers)"""
        dims = [int(d) for d in block_dims]
        mins = [6 if d == 7 else 9 for d in dims]
        m = int(residual_dim)

        def trampoline(user, params, residuals, jacobians):
            try:
                ps = [np.ctypeslib.as_array(params[b], shape=(dims[b],)).copy() for b in range(len(dims))]
                r, Js = fn(ps)
                r = np.asarray(r, float).reshape(m)
                for a in range(m):
                    residuals[a] = r[a]
                for b in range(len(dims)):
                    J = np.asarray(Js[b], float).reshape(m, mins[b])
                    out = np.ctypeslib.as_array(jacobians[b], shape=(m * mins[b],))
                    out[:] = J.reshape(-1)
                return 1
            except Exception:   # (an exception must not cross the C boundary: reported as the cost function's failure)
                import traceback
                traceback.print_exc()
                return 0
        cb = COST_FUNCTION(trampoline)
        ids = np.asarray(block_ids, np.uint64)
        rid = int(self.L.svin_ba_map_add_host_residual(self.h, ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(dims), m, cb, None))
        if rid:
            if not hasattr(self, "_host_callbacks"):
                self._host_callbacks = {}
            self._host_callbacks[rid] = cb   # (the library keeps the function pointer: the object must outlive the residual)
        return rid

    def map_add_reprojection_error(self, pose, landmark, ext, cam, uv, information):
        u, i = _arr(uv), _arr(np.asarray(information, float).reshape(2, 2))
        return int(self.L.svin_ba_map_add_reprojection_error(self.h, pose, landmark, ext, cam, _d(u), _d(i)))

    def map_remove_residual_block(self, rid):
        return self._check(self.L.svin_ba_map_remove_residual_block(self.h, rid), "map_remove_residual_block") == 1

    def residual_info(self, rids):
        """[(kind, residual dimension, [block dimensions])] for a list of residual ids (one call)"""
        r = np.ascontiguousarray(rids, np.uint64)
        n = len(r)
        kind, m, nb, dims = (np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32),
                             np.zeros(4 * max(n, 1), np.int32))
        p32 = C.POINTER(C.c_int32)
        self._check(self.L.svin_ba_residual_info(self.h, n, r.ctypes.data_as(pu64), kind.ctypes.data_as(p32), m.ctypes.data_as(p32),
                                                 nb.ctypes.data_as(p32), dims.ctypes.data_as(p32)), "residual_info")
        return [(int(kind[i]), int(m[i]), [int(x) for x in dims[4 * i:4 * i + min(int(nb[i]), 4)]]) for i in range(n)]

    def set_pack_mode(self, mode):
        """0: device-resident window whenever it qualifies (default); 1: always the host graph -> array pass + full upload"""
        self._check(self.L.svin_ba_set_pack_mode(self.h, int(mode)), "set_pack_mode")

    def debug_csr(self):
        """the observation CSR optimize() would solve on, copied back from the device"""
        L, N, res = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.L.svin_ba_debug_csr(self.h, C.byref(L), C.byref(N), None, None, None, None, None, None, None, C.byref(res)),
                    "debug_csr")
        L, N = L.value, N.value
        out = dict(L=L, N=N, lm_ptr=np.zeros(L + 1, np.int32), obs_lm=np.zeros(N, np.int32), obs_idx=np.zeros(N, np.uint32),
                   uv=np.zeros((N, 2)), w=np.zeros(N), lm=np.zeros((L, 4)), obs_order=np.zeros(N, np.int32))
        L2, N2 = C.c_int32(), C.c_int32()
        self._check(self.L.svin_ba_debug_csr(
            self.h, C.byref(L2), C.byref(N2), out["lm_ptr"].ctypes.data_as(C.POINTER(C.c_int32)),
            out["obs_lm"].ctypes.data_as(C.POINTER(C.c_int32)), out["obs_idx"].ctypes.data_as(C.POINTER(C.c_uint32)),
            _d(out["uv"]), _d(out["w"]), _d(out["lm"]), out["obs_order"].ctypes.data_as(C.POINTER(C.c_int32)), C.byref(res)), "debug_csr")
        assert (L2.value, N2.value) == (L, N)
        out["resident"] = bool(res.value)
        return out

    def bench_jacobian_eval_b2b(self, copies, iters):
        """(mean ms per launch with one event pair per launch, ms per launch with the launches back to back under one pair, bytes)"""
        ms, b2b, by = np.zeros(1), np.zeros(1), np.zeros(1)
        self._check(self.L.svin_ba_bench_jacobian_eval_b2b(self.h, copies, iters, _d(ms), _d(b2b), _d(by)), "bench_jacobian_eval_b2b")
        return float(ms[0]), float(b2b[0]), float(by[0])

    def marg_pre(self):
        """system of the last marginalisation after M1, before M2 (needs SVIN_MARG_KEEP_PRE=1): dict(H, b0, lm ranges, dense ranges)
        in the GPU's ordering [dense rows | landmarks], the form tests/mp_marg.py takes"""
        m, L = C.c_int32(), C.c_int32()
        self.L.svin_ba_get_marg_pre(self.h, C.byref(m), C.byref(L), None, None, None, None, None, None, 0, 0)
        m, L = m.value, L.value
        if m == 0:
            return None
        U, ba, W, V, bb = np.zeros((m, m)), np.zeros(m), np.zeros((m, 3 * L)), np.zeros((L, 3, 3)), np.zeros(3 * L)
        rows = np.zeros(m, np.int32)
        assert self.L.svin_ba_get_marg_pre(self.h, None, None, _d(U), _d(ba), _d(W), _d(V), _d(bb), rows.ctypes.data_as(pi32), m, L) == 1
        n = m + 3 * L
        H, b = np.zeros((n, n)), np.zeros(n)
        H[:m, :m], H[:m, m:], H[m:, :m] = U, W, W.T
        for l in range(L):
            H[m + 3 * l:m + 3 * l + 3, m + 3 * l:m + 3 * l + 3] = V[l]
        b[:m], b[m:] = ba, bb
        dense, i = [], 0
        while i < m:   # maximal runs of marginalised dense rows
            if rows[i]:
                j = i
                while j < m and rows[j]:
                    j += 1
                dense.append((i, j - i))
                i = j
            else:
                i += 1
        nd = self.L.svin_ba_get_marg_pre_blocks(self.h, None, None, None, 0, None, 0)
        ids, od, md, lids = np.zeros(max(nd, 1), np.uint64), np.zeros(max(nd, 1), np.int32), np.zeros(max(nd, 1), np.int32), np.zeros(max(L, 1), np.uint64)
        self.L.svin_ba_get_marg_pre_blocks(self.h, ids.ctypes.data_as(pu64), od.ctypes.data_as(pi32), md.ctypes.data_as(pi32), nd,
                                           lids.ctypes.data_as(pu64), L)
        rows_of = {int(ids[i]): (int(od[i]), int(md[i])) for i in range(nd)}
        rows_of.update({int(lids[l]): (m + 3 * l, 3) for l in range(L)})
        return dict(H=H, b0=b, lm=[(m + 3 * l, 3) for l in range(L)], dense=dense, m=m, n_landmarks=L, rows_of=rows_of)

    def bench_allreduce(self, n_doubles, iters=20):
        """mean microseconds of one native RCCL all-reduce of n_doubles FP64 values on the solver stream (collective call)"""
        us = np.zeros(1)
        self._check(self.L.svin_ba_bench_allreduce(self.h, int(n_doubles), int(iters), _d(us)), "bench_allreduce")
        return float(us[0])

    def bench_kernel_times(self, iters):
        a, b, c = np.zeros(1), np.zeros(1), np.zeros(1)
        self._check(self.L.svin_ba_bench_kernel_times(self.h, iters, _d(a), _d(b), _d(c)), "bench_kernel_times")
        return float(a[0]), float(b[0]), float(c[0])
