/* svin_ba.h -- C ABI of the MI355X-native sliding-window bundle-adjustment backend.
 *
 * Drop-in boundary for the hot path of AutonomousFieldRoboticsLab/SVIn: every entry point
 * replaces one member of okvis::Estimator / okvis::ceres::Map (paths below are relative to
 * /root/reference/okvis_ros/okvis/okvis_ceres/).  The header-only C++ shim that re-declares
 * okvis::Estimator on top of this ABI is described in INTEGRATION.md.
 *
 * Conventions
 *   - opaque handle, externally synchronised (one caller at a time, like the reference under
 *     ThreadedKFVio::estimator_mutex_, okvis_multisensor_processing/src/ThreadedKFVio.cpp:1083)
 *   - poses are 7 doubles [x y z qx qy qz qw] (src/PoseParameterBlock.cpp:64-74), speed/bias 9
 *     doubles [v bg ba], homogeneous landmarks 4 doubles
 *   - timestamps are (sec, nsec) uint32 pairs (okvis_time/include/okvis/Time.hpp:128)
 *   - return: 1 = true / ok, 0 = the reference's benign `false`, <0 = error (SVIN_ERR_*); nothing throws
 *   - all compute runs on the GPU selected at creation; there is no CPU fallback: without a
 *     usable HIP device svin_ba_create() returns NULL and svin_ba_last_error() says why.
 */
#ifndef SVIN_BA_H_
#define SVIN_BA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svin_ba svin_ba;

#define SVIN_ERR_INVALID_ARG (-1)
#define SVIN_ERR_NOT_FOUND (-2)
#define SVIN_ERR_DEVICE (-3)
#define SVIN_ERR_UNSUPPORTED (-4)

/* distortion models (okvis_cv/include/okvis/cameras/ ...Distortion.hpp) */
#define SVIN_DIST_NONE 0
#define SVIN_DIST_RADTAN 1
#define SVIN_DIST_EQUIDISTANT 2
#define SVIN_DIST_RADTAN8 3

/* okvis::ImuParameters (okvis_common/include/okvis/Parameters.hpp) */
typedef struct svin_imu_params {
  double a_max, g_max, sigma_g_c, sigma_a_c, sigma_bg, sigma_ba, sigma_gw_c, sigma_aw_c, tau, g;
  double a0[3];
} svin_imu_params;

/* one IMU sample: okvis::ImuMeasurement */
typedef struct svin_imu_sample {
  uint32_t sec, nsec;
  double gyr[3], acc[3];
} svin_imu_sample;

/* okvis::MapPoint (okvis_common/include/okvis/FrameTypedefs.hpp) without the observation map */
typedef struct svin_landmark_info {
  double point[4];
  double quality;
  double distance;
  int32_t num_observations;
  int32_t initialized;
} svin_landmark_info;

/* ::ceres::Solver::Summary subset filled by optimize() (Map.hpp:344) */
typedef struct svin_summary {
  double initial_cost, final_cost;
  int32_t iterations, num_successful_steps;
  int32_t termination; /* 0 convergence, 1 max iterations, 2 time limit (callback), 3 failure */
  double total_time_s;
  double upload_time_s, solve_time_s, download_time_s;
} svin_summary;

/* ---- lifetime ---------------------------------------------------------------------------- */
svin_ba* svin_ba_create(int device);            /* Estimator::Estimator()  src/Estimator.cpp:67-73 */
void svin_ba_destroy(svin_ba* h);
const char* svin_ba_last_error(void);
uint64_t svin_ba_new_id(svin_ba* h);            /* IdProvider::instance().newId()  src/IdProvider.cpp */
/* ONE id space.  Upstream, frame ids (FrameSynchronizer.cpp:97), landmark ids (Frontend.cpp:599) and the estimator's
 * internal extrinsics / speed-bias block ids (src/Estimator.cpp:217,234) all come from the process-wide
 * okvis::IdProvider.  A host that owns such a provider installs it here (the shim passes a trampoline to
 * IdProvider::instance().newId()); the core then draws its internal ids -- and svin_ba_new_id() -- from it.
 * Without a provider the core counts upwards from the largest id it has seen (svin_ba_reserve_ids() raises that
 * mark) and steps over any id a frame / landmark / block of the window already uses.  With a provider installed, an
 * id it hands out that collides with a known frame / landmark / block id makes add_states fail with
 * SVIN_ERR_INVALID_ARG and leaves the window -- state count and IMU integrals included -- unchanged. */
typedef uint64_t (*svin_id_provider_fn)(void* user);
int svin_ba_set_id_provider(svin_ba* h, svin_id_provider_fn fn, void* user);
int svin_ba_reserve_ids(svin_ba* h, uint64_t largest_id_seen);

/* ---- sensors (Estimator.cpp:77-96) -------------------------------------------------------- */
/* intr = fu fv cu cv; dist = up to 8 coefficients (radtan k1 k2 p1 p2; equidistant k1..k4;
 * radtan8 k1 k2 p1 p2 k3 k4 k5 k6); sigmas = sigma_absolute_translation, sigma_absolute_orientation,
 * sigma_c_relative_translation, sigma_c_relative_orientation (ExtrinsicsEstimationParameters) */
int svin_ba_add_camera(svin_ba* h, int distortion_model, const double intr[4], const double* dist, int n_dist,
                       int width, int height, const double sigmas[4]);
/* Estimator::addCamera only registers the extrinsics-estimation parameters (:77-81); the geometry reaches the reference
 * with every observation (implementation/Estimator.hpp:62-66, multiFrame->geometryAs<GEOMETRY>(camIdx)).  A host in that
 * position calls svin_ba_add_camera with SVIN_DIST_NONE / zero intrinsics and hands the geometry over once it has the
 * first multi-frame. */
int svin_ba_set_camera_geometry(svin_ba* h, uint64_t cam_idx, int distortion_model, const double intr[4],
                                const double* dist, int n_dist, int width, int height);
int svin_ba_add_imu(svin_ba* h, const svin_imu_params* p);
int svin_ba_clear_cameras(svin_ba* h);          /* Estimator::clearCameras :91 */
int svin_ba_clear_imus(svin_ba* h);             /* Estimator::clearImus    :94 */
int svin_ba_set_sonar_extrinsics(svin_ba* h, const double T_SSo[7]); /* Estimator.hpp:617 sonarParameters_ */

/* ---- window construction ------------------------------------------------------------------ */
/* Estimator::addStates  src/Estimator.cpp:98-411.  T_SC: n_cam x 7 (multiFrame->T_SC(i));
 * sonar: n_sonar x {range, heading}; depth: n_depth values. */
int svin_ba_add_states(svin_ba* h, uint64_t frame_id, uint32_t sec, uint32_t nsec, uint64_t num_keypoints,
                       const double* T_SC, int n_cam, const svin_imu_sample* imu, int n_imu, int as_keyframe,
                       const double* sonar, int n_sonar, const double* depth, int n_depth, double first_depth);
int svin_ba_add_landmark(svin_ba* h, uint64_t landmark_id, const double hp[4]);       /* :414-429 */
/* Estimator::addObservation<GEOMETRY>  include/okvis/implementation/Estimator.hpp:47-87.
 * Returns the residual id (non-zero) or 0 for a duplicate (the reference returns NULL). */
uint64_t svin_ba_add_observation(svin_ba* h, uint64_t landmark_id, uint64_t pose_id, uint64_t cam_idx,
                                 uint64_t keypoint_idx, const double uv[2], double keypoint_size);
/* the same call for n matches held as arrays (uv = n x 2); out_ids (optional) receives the residual ids, 0 for
 * duplicates; returns the number of observations added.  The reference adds matches one by one from
 * VioKeyframeWindowMatchingAlgorithm::setBestMatch; a host that has them as arrays saves n - 1 boundary crossings. */
int svin_ba_add_observations(svin_ba* h, int n, const uint64_t* landmark_ids, const uint64_t* pose_ids,
                             const uint64_t* cam_idx, const uint64_t* keypoint_idx, const double* uv,
                             const double* keypoint_sizes, uint64_t* out_ids);
int svin_ba_remove_observation(svin_ba* h, uint64_t landmark_id, uint64_t pose_id, uint64_t cam_idx,
                               uint64_t keypoint_idx);                                 /* :452-474 */
int svin_ba_remove_observation_by_id(svin_ba* h, uint64_t residual_id);               /* :432-449 */
/* HomogeneousPointError on a landmark (okvis_ceres/src/HomogeneousPointError.cpp:48-117, <3,4>): error = landmark -
 * measurement (first three homogeneous components), weighted by the upper Cholesky factor of the 3x3 information
 * (row-major; the variance constructor :52-55 is information = I / variance).  okvis::Estimator never adds one; the
 * entry point is for callers that build such priors themselves (the reference does it through Map::addResidualBlock).
 * It takes part in the Schur elimination of its landmark, the cost and the landmark quality like any residual of the
 * block.  Returns the residual id (0: unknown landmark / information not positive definite).  A landmark that carries
 * one does not take part in svin_ba_apply_marginalization_strategy (error if it is observed from a leaving frame). */
uint64_t svin_ba_add_homogeneous_point_error(svin_ba* h, uint64_t landmark_id, const double measurement[4],
                                             const double information[9]);
int svin_ba_remove_homogeneous_point_error(svin_ba* h, uint64_t residual_id);

/* ---- the hot path -------------------------------------------------------------------------- */
int svin_ba_optimize(svin_ba* h, uint64_t num_iter, uint64_t num_threads_ignored, int verbose); /* :876-929 */
/* Blocks until everything the handle has enqueued has run -- the device part of the last
 * svin_ba_apply_marginalization_strategy in particular, which that call only enqueues (okvis runs the whole of
 * applyMarginalizationStrategy inside the call, src/Estimator.cpp:472-813; here the ~1 ms of device work overlap whatever the
 * caller does until its next call on the handle, and every call waits for what it needs by itself).  Never required for
 * correctness: a caller that wants to TIME a later call on its own uses it.  Returns 1. */
int svin_ba_wait_idle(svin_ba* h);
int svin_ba_set_optimization_time_limit(svin_ba* h, double time_limit, int min_iterations);    /* :932-951 */
/* Estimator::applyMarginalizationStrategy :495-814; removed landmark ids are written to
 * removed_ids (capacity cap), *n_removed receives the full count. */
int svin_ba_apply_marginalization_strategy(svin_ba* h, uint64_t num_keyframes, uint64_t num_imu_frames,
                                           uint64_t* removed_ids, int cap, int* n_removed);
/* optimize() split in three for measurement with HBM-resident inputs: prepare = pack + upload the window,
 * solve_prepared = the trust-region iterations only (device-synchronised on return), finish = download
 * states + landmark quality.  optimize() == prepare; solve_prepared; finish. */
int svin_ba_prepare(svin_ba* h);
int svin_ba_solve_prepared(svin_ba* h, uint64_t num_iter, int verbose);
int svin_ba_finish(svin_ba* h);
/* SEVERAL windows at once (the reference has no counterpart: okvis::Estimator::optimize, Estimator.cpp:876-929, is one window per
 * call and ThreadedKFVio runs one estimator; SURVEY 8(e) names "independent replicas processing different windows" as the other
 * way to fill the hardware).  Handles of ONE device; windows with the same launch geometry (numbers of states, landmarks,
 * observations and factors) that the LDS-resident solver takes (reduced system of at most 176 rows, no marginalisation-only
 * restrictions) share one launch sequence per trust-region round, the window as a grid dimension; the others -- and a window
 * without a partner -- are optimised one after the other by the ordinary path.  Every window ends exactly (bit for bit) where
 * svin_ba_optimize / svin_ba_solve_prepared would leave it on its own; *n_batched (may be NULL) = windows that ran in a batch.
 * solve_prepared_batch expects svin_ba_prepare on every handle (measurement form); optimize_batch = prepare, solve, finish.
 * The call owns the handles while it runs (no other thread may use them); the batched windows run on up to four streams of the
 * library's own (one pool per device, SVIN_BATCH_LANES), every handle's stream is synchronised before and the pool's streams after.
 * Time limits (svin_ba_set_optimization_time_limit) are honoured per window.  Measured: 7.1 x / 8.4 x / 8.8 x one window's rate at
 * 16 / 32 / 64 windows of BASELINE configs[1] (profiles/r06_bench_v3.json). */
int svin_ba_solve_prepared_batch(svin_ba* const* handles, int n, uint64_t num_iter, int verbose, int* n_batched);
int svin_ba_optimize_batch(svin_ba* const* handles, int n, uint64_t num_iter, int verbose, int* n_batched);
/* forces every IMU factor to re-preintegrate at its next evaluation (ImuError::redo_ = true) */
int svin_ba_invalidate_preintegration(svin_ba* h);
int svin_ba_get_summary(svin_ba* h, svin_summary* out);
/* Landmark-sharded multi-GPU solve (one process per GPU; BASELINE config "64 KF / 50 000 landmarks").  Every
 * rank adds ALL states (add_states on every rank) and only ITS landmarks + observations; the reduced camera
 * system and a handful of scalars are summed over ranks through `fn`, which must all-reduce `count` doubles at
 * the DEVICE address `ptr` in place (op 0 = sum, 1 = max) and return 0 -- e.g. RCCL via torch.distributed.
 * The reference has no counterpart (single process, Ceres threads: Estimator.cpp:889). */
typedef int (*svin_allreduce_fn)(void* ptr, uint64_t count, int op, void* user);
int svin_ba_set_distributed(svin_ba* h, int rank, int world, svin_allreduce_fn fn, void* user);
/* The same mode with RCCL called natively: the all-reduces are enqueued on the handle's stream (ncclAllReduce, FP64, in
 * place), no host synchronisation and no callback inside the iteration.  Rank 0 obtains a 128-byte ncclUniqueId with
 * svin_ba_rccl_unique_id() and the host hands it to every rank by whatever means it has (torch.distributed broadcast,
 * MPI, a file); each rank then calls svin_ba_set_distributed_rccl() -- collectively, it blocks until all ranks have
 * joined (ncclCommInitRank).  One process per GPU.  RCCL (librccl.so.1) is resolved with dlopen at this point, a
 * single-GPU user never needs it. */
int svin_ba_rccl_unique_id(unsigned char id_out[128]);
int svin_ba_set_distributed_rccl(svin_ba* h, int rank, int world, const unsigned char id[128]);
/* solver tolerances (::ceres::Solver::Options defaults: 1e-6, 1e-10, 1e-8) */
int svin_ba_set_solver_tolerances(svin_ba* h, double function_tol, double gradient_tol, double parameter_tol);

/* ---- getters / setters (Estimator.cpp:955-1316) -------------------------------------------- */
int svin_ba_get_T_WS(svin_ba* h, uint64_t pose_id, double T[7]);
int svin_ba_get_speed_and_bias(svin_ba* h, uint64_t pose_id, uint64_t imu_idx, double sb[9]);
int svin_ba_get_camera_sensor_states(svin_ba* h, uint64_t pose_id, uint64_t cam_idx, double T[7]);
int svin_ba_get_landmark(svin_ba* h, uint64_t landmark_id, svin_landmark_info* out);
int svin_ba_is_landmark_added(svin_ba* h, uint64_t landmark_id);
/* Estimator::isLandmarkInitialized :966-969 / setLandmarkInitialized :1126-1129 (HomogeneousPointParameterBlock::
 * initialized_, true after addLandmark).  is_: 1 / 0, SVIN_ERR_NOT_FOUND for an unknown landmark. */
int svin_ba_is_landmark_initialized(svin_ba* h, uint64_t landmark_id);
int svin_ba_set_landmark_initialized(svin_ba* h, uint64_t landmark_id, int initialized);
/* Estimator::getLandmarks :974-990: all landmarks in PointMap order (ascending id); ids / infos may be NULL;
 * returns the number of landmarks (numLandmarks()). */
int svin_ba_get_landmarks(svin_ba* h, uint64_t* ids, svin_landmark_info* infos, int cap);
/* MapPoint::observations of one landmark (okvis_common FrameTypedefs.hpp: map KeypointIdentifier -> residual id), in
 * KeypointIdentifier order (frame, camera, keypoint); any output may be NULL; returns the number of observations */
int svin_ba_get_landmark_observations(svin_ba* h, uint64_t landmark_id, uint64_t* frame_ids, uint64_t* cam_idx,
                                      uint64_t* keypoint_idx, uint64_t* residual_ids, int cap);
/* svin_ba_get_landmarks + svin_ba_get_landmark_observations for ALL landmarks in one call (the PointMap with its
 * observation maps that Estimator::getLandmarks :974-990 copies and applyMarginalizationStrategy's caller walks): landmarks
 * in ascending id order, obs_ptr (cap_landmarks + 1 entries) is a CSR into the per-observation arrays, each landmark's
 * entries in KeypointIdentifier order.  Any output may be NULL.  Returns the number of landmarks, *n_obs_total the number
 * of observations (either may exceed its capacity: call again with larger buffers). */
int svin_ba_get_all_landmark_observations(svin_ba* h, int cap_landmarks, uint64_t* ids, svin_landmark_info* infos,
                                          int32_t* obs_ptr, int cap_obs, uint64_t* frame_ids, uint64_t* cam_idx,
                                          uint64_t* keypoint_idx, uint64_t* residual_ids, int32_t* n_obs_total);
int svin_ba_set_T_WS(svin_ba* h, uint64_t pose_id, const double T[7]);
int svin_ba_set_speed_and_bias(svin_ba* h, uint64_t pose_id, uint64_t imu_idx, const double sb[9]);
int svin_ba_set_camera_sensor_states(svin_ba* h, uint64_t pose_id, uint64_t cam_idx, const double T[7]);
int svin_ba_set_landmark(svin_ba* h, uint64_t landmark_id, const double hp[4]);
uint64_t svin_ba_num_frames(svin_ba* h);
uint64_t svin_ba_num_landmarks(svin_ba* h);
uint64_t svin_ba_current_keyframe_id(svin_ba* h);
uint64_t svin_ba_current_frame_id(svin_ba* h);
uint64_t svin_ba_frame_id_by_age(svin_ba* h, uint64_t age);
int svin_ba_is_keyframe(svin_ba* h, uint64_t frame_id);
int svin_ba_set_keyframe(svin_ba* h, uint64_t frame_id, int is_keyframe);              /* Estimator.hpp:444 */
int svin_ba_timestamp(svin_ba* h, uint64_t frame_id, uint32_t* sec, uint32_t* nsec);   /* Estimator.hpp:373 */
int svin_ba_state_count(svin_ba* h);            /* public member stateCount_, Estimator.hpp:450 (read by Frontend.cpp:269) */
/* Estimator::getImuPreIntegral :1001-1014 / setImuPreIntegral :1081-1087 (imuIntegralsMap_, filled by addStates :165
 * from the second ImuError::propagation overload): acc_doubleintegral(3), acc_integral(3), Delta_t */
int svin_ba_get_imu_preintegral(svin_ba* h, uint64_t pose_id, double acc_doubleintegral[3], double acc_integral[3],
                                double* delta_t);
int svin_ba_set_imu_preintegral(svin_ba* h, uint64_t pose_id, const double acc_doubleintegral[3],
                                const double acc_integral[3], double delta_t);
/* static Estimator::initPoseFromImu :848-873 (no handle: a few scalar operations on the mean accelerometer reading);
 * returns 0 for an empty deque like the reference */
int svin_ba_init_pose_from_imu(const svin_imu_sample* imu, int n_imu, double T_WS[7]);
int svin_ba_is_in_imu_window(svin_ba* h, uint64_t frame_id);
int svin_ba_frame_ids(svin_ba* h, uint64_t* ids, int cap);      /* returns the number of frames */
int svin_ba_landmark_ids(svin_ba* h, uint64_t* ids, int cap);   /* returns the number of landmarks */

/* ---- okvis::ceres::Map surface that survives (Map.hpp:65-420): the graph queries Estimator / callers use,
 * answered from the core's own graph.  Block ids: frame id (pose block), the ids add_states drew (extrinsics,
 * speed/bias: svin_ba_describe_block tells which), landmark ids.  Residual ids: what add_observation returned, the
 * ids of the non-reprojection factors (svin_ba_eval_factors lists them) and of the marginalisation prior. */
int svin_ba_parameter_block_exists(svin_ba* h, uint64_t block_id);                    /* Map::parameterBlockExists */
/* Map::setParameterBlockConstant / Variable (src/Map.cpp:495-510) on any block, landmarks included (a window with a constant
 * landmark is packed by the host and cannot marginalise a frame that sees it); unknown id: 0 */
int svin_ba_set_parameter_block_constant(svin_ba* h, uint64_t block_id, int constant);
int svin_ba_is_parameter_block_constant(svin_ba* h, uint64_t block_id);               /* ParameterBlock::fixed() */
/* Map::resetParameterization (src/Map.cpp:513-543) with the values of Map::Parameterization (include/okvis/ceres/Map.hpp:97-105):
 * 0 HomogeneousPoint (landmarks), 1 Pose6d, 2 Pose3d (orientation only: PoseManifold3d, src/PoseManifold.cpp:173-272),
 * 3 Pose4d (position + yaw: :276-368), 4 Pose2d (roll / pitch: :372-466), 5 Trivial (speed / bias).  A pose or extrinsics block
 * on a reduced manifold keeps its six rows in the reduced system; the rows of the held directions are struck out before the solve
 * (same iterates as with the Jacobian columns removed).  1 = done, 0 = unknown block, SVIN_ERR_INVALID_ARG = a manifold the
 * block's type cannot take.  svin_ba_linearize reports the 6-DoF system; a window holding such a block cannot marginalise. */
int svin_ba_reset_parameterization(svin_ba* h, uint64_t block_id, int parameterization);
int svin_ba_get_parameterization(svin_ba* h, uint64_t block_id);   /* the value above, SVIN_ERR_NOT_FOUND for an unknown block */
/* Map::residuals(id) (src/Map.cpp:576-587): ids of every residual touching the block, in insertion order; returns the
 * count (may exceed cap), SVIN_ERR_NOT_FOUND for an unknown block */
int svin_ba_residuals_of(svin_ba* h, uint64_t block_id, uint64_t* residual_ids, int cap);
/* Map::parameters(residual) (src/Map.cpp:602-620): block ids in the cost function's parameter order (reprojection:
 * pose, landmark, extrinsics); *kind receives 100 reprojection / 101 marginalisation prior
 * / 102 HomogeneousPointError on a landmark / the factor kind (0 imu, 1 pose prior, 2 speed-bias prior, 3 relative
 * pose, 4 sonar, 5 depth); returns the count (it may exceed cap: the prior lists every block it touches) */
int svin_ba_parameters_of(svin_ba* h, uint64_t residual_id, uint64_t* block_ids, int cap, int32_t* kind);

/* Map::parameterBlockPtr (Map.hpp:166-170) / id2parameterBlockMap (:188) as values: *type 0 pose (PoseParameterBlock),
 * 1 extrinsics (PoseParameterBlock), 2 speed/bias (SpeedAndBiasParameterBlock), 3 landmark
 * (HomogeneousPointParameterBlock); values (up to 9 doubles) = ParameterBlock::parameters(), (sec, nsec) = timestamp(),
 * fixed = fixed(), initialized = HomogeneousPointParameterBlock::initialized().  Any output may be NULL.  Returns the
 * ambient dimension (7 / 9 / 4), SVIN_ERR_NOT_FOUND for an unknown id.  _ids: every parameter block id, ascending;
 * returns the count (may exceed cap). */
int svin_ba_get_parameter_block(svin_ba* h, uint64_t block_id, int32_t* type, double* values, uint32_t* sec,
                                uint32_t* nsec, int32_t* fixed, int32_t* initialized);
int svin_ba_parameter_block_ids(svin_ba* h, uint64_t* ids, int cap);

/* ErrorInterface::residualDim / parameterBlocks / parameterBlockDim and the type of a LIST of residuals in one call (what
 * Map::residuals(block) needs per entry of its collection, Map.cpp:576-587): kind as svin_ba_parameters_of reports it, the
 * residual dimension, the number of blocks and the ambient dimensions of the first four (the marginalisation prior, kind 101,
 * lists its blocks through svin_ba_parameters_of).  Unknown ids: kind -1.  Returns the number of known residuals. */
int svin_ba_residual_info(svin_ba* h, int n, const uint64_t* residual_ids, int32_t* kind, int32_t* residual_dim,
                          int32_t* n_blocks, int32_t* block_dims /* 4 per residual */);

/* ---- okvis::ceres::Map as a graph BUILDER (Map.cpp:255-376, :322-333, :467-492): parameter blocks and residual blocks added one
 * by one, outside any frame -- what the reference's own tests do (okvis_ceres/test/TestMap.cpp, TestHomogeneousPointError.cpp,
 * TestPoseError ...).  The residual kinds are the reference's error-term classes; a device solver cannot call a caller's virtual
 * cost function, so each class has its entry point.  type: 0 = pose (T_WS, or extrinsics T_SC from its first use as the third
 * block of a reprojection residual), 2 = speed and bias, 3 = homogeneous point.  Returns 1 / 0 (id in use, Map.cpp:257-260). */
int svin_ba_map_add_parameter_block(svin_ba* h, uint64_t block_id, int type, const double* values);
/* ParameterBlock::setParameters on any block of the graph (pose / extrinsics 7, speed and bias 9, landmark 4) */
int svin_ba_set_parameter_block(svin_ba* h, uint64_t block_id, const double* values);
/* Map::removeParameterBlock: the block and every residual on it (blocks of a frame: applyMarginalizationStrategy instead -> 0) */
int svin_ba_map_remove_parameter_block(svin_ba* h, uint64_t block_id);
/* PoseError(measurement, information)  src/PoseError.cpp:49-132; returns the residual id (0 = refused) */
uint64_t svin_ba_map_add_pose_error(svin_ba* h, uint64_t block_id, const double measurement[7], const double information[36]);
/* SpeedAndBiasError(measurement, information)  src/SpeedAndBiasError.cpp:47-113 */
uint64_t svin_ba_map_add_speed_and_bias_error(svin_ba* h, uint64_t block_id, const double measurement[9], const double information[81]);
/* RelativePoseError(information) between two pose (or two extrinsics) blocks  src/RelativePoseError.cpp:48-147 */
uint64_t svin_ba_map_add_relative_pose_error(svin_ba* h, uint64_t block0, uint64_t block1, const double information[36]);
/* ImuError(measurements, parameters, t_0, t_1) on blocks = (pose_0, speed/bias_0, pose_1, speed/bias_1)  src/ImuError.cpp:58-75,
 * Map::addResidualBlock src/Map.cpp:341-376: the factor okvis::Estimator::addStates creates between two frames, here between any
 * four blocks of the map.  Its pre-integration is redone on the device whenever the bias estimate moves (ImuError.cpp:702-706). */
uint64_t svin_ba_map_add_imu_error(svin_ba* h, const uint64_t blocks[4], const svin_imu_sample* imu, int n_imu, const svin_imu_params* params,
                                   uint32_t t0_sec, uint32_t t0_nsec, uint32_t t1_sec, uint32_t t1_nsec);
/* SonarError(range, heading, information, landmark patch) on a pose block  src/SonarError.cpp:57-183 (T_SSo: svin_ba_set_sonar_extrinsics,
 * identity in the reference); patch_xyz: n_patch Euclidean points, the residual uses their mean (:124-131) */
uint64_t svin_ba_map_add_sonar_error(svin_ba* h, uint64_t pose_block, double range, double heading, double information,
                                     const double* patch_xyz, int n_patch);
/* DepthError(depth, information, first depth) on a pose block  src/DepthError.cpp:50-139 */
uint64_t svin_ba_map_add_depth_error(svin_ba* h, uint64_t pose_block, double depth, double information, double first_depth);
/* Map::addResidualBlock (src/Map.cpp:341-376) with a cost function the library has no kernel for: evaluated by the HOST.  The
 * callback gets the n_blocks parameter blocks in the order given (pose / extrinsics: r(3), q(xyzw); speed / bias: 9) and writes
 * residual_dim residuals and, per block, the residual_dim x (6 | 9) row-major Jacobian in MINIMAL coordinates (what
 * ErrorInterface::EvaluateWithMinimalJacobians delivers: pose delta = (dr, dalpha), q <- exp(dalpha) * q); it returns non-zero on
 * success (::ceres::CostFunction::Evaluate's bool).  Called from the thread that optimises, before every evaluation launch, with one
 * stream synchronisation each time: a slow path for graphs third parties build through okvis::ceres::Map.  No loss function; no
 * landmark blocks; residual_dim <= 15; at most 4 blocks and 30 minimal columns.  Such a window is neither batched nor
 * marginalised nor sharded.  Returns the residual id, 0 if refused. */
typedef int (*svin_cost_function)(void* user, const double* const* parameters, double* residuals, double** jacobians_minimal);
uint64_t svin_ba_map_add_host_residual(svin_ba* h, const uint64_t* block_ids, int n_blocks, int residual_dim, svin_cost_function fn, void* user);
/* ReprojectionError<geometry of camera cam_idx>(uv, information) under CauchyLoss(1) on (pose, landmark, extrinsics)
 * (ReprojectionErrorBase.hpp:50-54, Estimator.cpp:69).  information: 2 x 2 row-major, a positive multiple of the identity
 * (the device stores one weight per residual, as Estimator::addObservation's 64 / size^2 * I needs). */
uint64_t svin_ba_map_add_reprojection_error(svin_ba* h, uint64_t pose_block, uint64_t landmark_id, uint64_t extrinsics_block,
                                            uint64_t cam_idx, const double uv[2], const double information[4]);
/* Map::removeResidualBlock for any residual of the graph (Map.cpp:467-492) */
int svin_ba_map_remove_residual_block(svin_ba* h, uint64_t residual_id);

/* ---- keyframe hand-off to pose_graph: the estimator-side content of the keyframe message that
 * ThreadedKFVio::optimizationLoop assembles (okvis_multisensor_processing/src/ThreadedKFVio.cpp:1147-1240).  For every
 * landmark whose first observation in `frame_id` (MapPoint::observations order: frame, camera, keypoint) is in camera
 * `cam_idx` (:1167, :1183): landmark id, Euclidean point (:1171-1176), keypoint index and quality (:1196-1199), and the
 * frame ids of all its other observations (:1210-1226; the caller maps them to keyframe indices with kf_f_map_ and
 * adds the cv::KeyPoint fields it owns).  obs_ptr has cap_points + 1 entries (CSR into obs_frame_ids).  Returns the
 * number of points (it may exceed cap_points: call again with larger buffers); *n_obs_total the number of list entries. */
int svin_ba_keyframe_points(svin_ba* h, uint64_t frame_id, uint64_t cam_idx, int cap_points, uint64_t* landmark_ids,
                            double* xyz, uint64_t* keypoint_idx, double* quality, int32_t* obs_ptr, int cap_obs,
                            uint64_t* obs_frame_ids, int32_t* n_obs_total);

/* ---- CPU-callable prediction kept for the frontend (ImuError::propagation, ImuError.cpp:266-476):
 * runs on the GPU like everything else; T (7) and sb (9) are in/out; cov / jac are 15x15 or NULL. */
int svin_ba_imu_propagation(svin_ba* h, const svin_imu_sample* imu, int n_imu, const svin_imu_params* p, double T[7],
                            double sb[9], uint32_t sec0, uint32_t nsec0, uint32_t sec1, uint32_t nsec1, double* cov,
                            double* jac);
/* the second overload (ImuError.cpp:479-697): additionally acc_doubleintegral(3), acc_integral(3), Delta_t in integrals[7] */
int svin_ba_imu_propagation_integrals(svin_ba* h, const svin_imu_sample* imu, int n_imu, const svin_imu_params* p,
                                      double T[7], double sb[9], uint32_t sec0, uint32_t nsec0, uint32_t sec1,
                                      uint32_t nsec1, double* cov, double* jac, double integrals[7]);

/* ---- host-side twins for the callers that stay on the CPU.  No handle, no GPU, no allocation: these are
 * NOT a fall-back of the window solve (it has none).
 * svin_host_imu_propagation: ImuError::propagation, both overloads (src/ImuError.cpp:266-476, :479-697) -- the call
 * ThreadedKFVio::imuConsumerLoop makes per IMU sample (ThreadedKFVio.cpp:808-819) and :599 / Frontend.cpp:258 per frame.
 * Same arguments and return value as svin_ba_imu_propagation_integrals; cov / jac / integrals may be NULL. */
int svin_host_imu_propagation(const svin_imu_sample* imu, int n_imu, const svin_imu_params* p, double T[7], double sb[9],
                              uint32_t sec0, uint32_t nsec0, uint32_t sec1, uint32_t nsec1, double* cov, double* jac,
                              double* integrals);
/* svin_host_reprojection_error: ONE ReprojectionError<GEOMETRY>::EvaluateWithMinimalJacobians
 * (include/okvis/ceres/implementation/ReprojectionError.hpp:85-229) as ProbabilisticStereoTriangulator.cpp:266-300 uses it.
 * information = 2x2 (row-major), its Cholesky factor weights the error (ReprojectionErrorBase setInformation).
 * Outputs (any Jacobian pointer may be NULL): residual[2]; minimal 2x6 / 2x3 / 2x6; ambient 2x7 / 2x4 / 2x7. */
int svin_host_reprojection_error(int distortion_model, const double intr[4], const double* dist, int n_dist,
                                 const double T_WS[7], const double hp_W[4], const double T_SC[7], const double uv[2],
                                 const double information[4], double residual[2], double* J_pose_min, double* J_lm_min,
                                 double* J_ext_min, double* J_pose, double* J_lm, double* J_ext);

/* svin_host_homogeneous_point_error: HomogeneousPointError::EvaluateWithMinimalJacobians (HomogeneousPointError.cpp:85-117):
 * residual[3]; J_min 3x3 (= the square-root information); J 3x4 (last column zero).  Jacobian pointers may be NULL. */
int svin_host_homogeneous_point_error(const double hp_W[4], const double measurement[4], const double information[9],
                                      double residual[3], double* J_min, double* J);

/* svin_host_pose_information / svin_host_pose_error: okvis::ceres::PoseError (src/PoseError.cpp:52-132, <6,7>) as
 * ProbabilisticStereoTriangulator.cpp:87-99 uses it.  _information does what PoseError::setInformation does (:70-76):
 * sqrt_information (6x6 row-major) = the upper Cholesky factor with Eigen::LLT's stop-at-a-non-positive-pivot
 * behaviour, covariance = information^-1 by LU with partial pivoting; either output may be NULL.  _error: residual[6],
 * J_min 6x6, J 6x7 (= J_min * PoseManifold::liftJacobian) row-major, either may be NULL.  The unweighted error and
 * Jacobian are the function the factor kernel evaluates for its pose priors (dmath.hpp poseErrorEval). */
int svin_host_pose_information(const double information[36], double* sqrt_information, double* covariance);
int svin_host_pose_error(const double measurement[7], const double sqrt_information[36], const double T_WS[7],
                         double residual[6], double* J_min, double* J);

/* The parameter-block manifolds (src/PoseManifold.cpp, src/HomogeneousPointManifold.cpp) for the shim's
 * ParameterBlock / Manifold classes; kinds in the order of Map::Parameterization (Map.hpp:97-105).  Jacobians row-major:
 * plus ambient x tangent, lift and minus tangent x ambient.  Pose6d plus IS the retraction the device applies
 * (dmath.hpp poseOplus).  Returns 1, SVIN_ERR_INVALID_ARG for a NULL pointer or an unknown kind. */
#define SVIN_MANIFOLD_HPOINT 0
#define SVIN_MANIFOLD_POSE6D 1
#define SVIN_MANIFOLD_POSE3D 2
#define SVIN_MANIFOLD_POSE4D 3
#define SVIN_MANIFOLD_POSE2D 4
int svin_host_manifold_dims(int kind, int* ambient, int* tangent);
int svin_host_manifold_plus(int kind, const double* x, const double* delta, double* x_plus_delta);
int svin_host_manifold_minus(int kind, const double* x_plus_delta, const double* x, double* delta);
int svin_host_manifold_plus_jacobian(int kind, const double* x, double* J);
int svin_host_manifold_lift_jacobian(int kind, const double* x, double* J);
int svin_host_manifold_minus_jacobian(int kind, const double* x, double* J);

/* ---- inspection / parity hooks (ErrorInterface::EvaluateWithMinimalJacobians, Map::getLhs) ---- */
/* Evaluates every reprojection residual of the window on the GPU at the current estimates.
 * Outputs are per observation in the order given by svin_ba_observation_ids(); any may be NULL.
 *   r: n x 2, Jpose: n x 12 (2x6 row-major), Jlm: n x 6 (2x3), Jext: n x 12; robust != 0 applies the
 *   Cauchy corrector exactly as the solver sees it. Returns n. */
int svin_ba_eval_reprojection(svin_ba* h, int robust, double* r, double* Jpose, double* Jlm, double* Jext, int cap);
int svin_ba_observation_ids(svin_ba* h, uint64_t* residual_ids, uint64_t* landmark_ids, uint64_t* pose_ids,
                            int32_t* cam_idx, int cap);
/* small factors (IMU, priors, relative pose, sonar, depth): kind, residual dim, residual, stacked
 * minimal Jacobian (m x ncols, row-major), block ids.  Returns the number of factors. */
int svin_ba_eval_factors(svin_ba* h, int32_t* kind, int32_t* m, int32_t* ncols, double* r15, double* J15x30,
                         uint64_t* block_ids4, uint64_t* residual_ids, int cap);
/* reduced (Schur) system at the current estimates with damping mu: S (d x d), g (d); block_ids / offsets
 * describe the ordering. Returns d. */
int svin_ba_linearize(svin_ba* h, double mu, double* S, double* g, uint64_t* block_ids, int32_t* block_offsets,
                      int32_t* n_blocks, int cap_d, double* cost);
/* the Gauss-Newton step y of that system, (S + mu D) y = g, as the device solver computes it (the path the size of the window
 * selects: LDS-resident, left-looking, blocked, blocked behind the speed / bias chain elimination).  Returns d (or -d if
 * cap_d < d). */
int svin_ba_debug_reduced_solve(svin_ba* h, double mu, double* y, int cap_d);
/* the same with the choice svin_ba_optimize makes: fuse_finalize != 0 applies the metric and the damping inside the solver's
 * load phase (the fused form every trust-region iteration runs), 0 is svin_ba_debug_reduced_solve. */
int svin_ba_debug_reduced_solve_ex(svin_ba* h, double mu, int fuse_finalize, double* y, int cap_d);
/* Which path the handle's work took since it was created, so that no fall-back is silent: out[0] optimisations on the
 * device-resident window (per frame only the new / removed observation records travel), out[1] optimisations that
 * re-packed the whole window on the host (wide windows with their panel work lists, landmark priors, constant landmarks,
 * sharded mode, svin_ba_set_pack_mode), out[2] marginalisation jobs whose observation tables were gathered on the device,
 * out[3] jobs whose tables the host assembled.  Returns 1. */
int svin_ba_get_path_counters(svin_ba* h, int64_t out[4]);
/* the eigen-solver of the marginalisation prior (M3, MarginalizationError.cpp:725-758: Eigen::SelfAdjointEigenSolver there;
 * here Householder tridiagonalisation + divide and conquer in one workgroup, svin_amd/csrc/symeig.hpp) on an arbitrary
 * symmetric matrix A (n x n, row-major, n <= 128), on the current HIP device: eigenvalues ascending, eigenvectors[i * n + j] =
 * component i of vector j, device_ms (may be NULL) = the fastest of three launches.  Returns 1, 0 if n is out of range,
 * -1 if the result is not finite.  Test hook. */
int svin_ba_debug_sym_eig(int n, const double* A, double* eigenvalues, double* eigenvectors, double* device_ms);
/* Debug / A-B options of the library (svin_amd/csrc/options.hpp; INTEGRATION.md §5 lists them).  Each option is named after the
 * environment variable that initialises it -- the library looks at the environment ONCE, when the first of them is asked for
 * (svin_ba_create at the latest) -- and only changes through svin_ba_debug_set_option afterwards; a later setenv() of the host
 * process is never seen.  None changes a result beyond rounding: they select between implementations of the same arithmetic
 * ("SVIN_SCHUR_PAIRWISE", "SVIN_NO_LL", "SVIN_NO_SB_ELIM", "SVIN_NO_LDS_BORDER", "SVIN_PANELS_OLD", "SVIN_MARG_EIG" = 0 default
 * chain / 1 direct / 2 cholesky / 3 jacobi, ...), run the sharded code path with a one-rank communicator
 * ("SVIN_FORCE_DISTRIBUTED") or switch diagnostics on ("SVIN_MARG_KEEP_PRE", "SVIN_*_TIMING").  Options that shape a handle's
 * buffers ("SVIN_NO_MAILBOX") are read when the handle is created; the others at the next pack() / solve() / marginalisation.
 * Test hooks: set returns 1, or 0 for an unknown name; get returns 1 and *value, or 0.  svin_ba_debug_set_switch is the
 * round-4 spelling of set (value != 0 -> 1). */
int svin_ba_debug_set_option(const char* name, int value);
int svin_ba_debug_get_option(const char* name, int* value);
int svin_ba_debug_set_switch(const char* name, int value);
/* doubles [offset, offset + count) of the reduced-system solver's scratch buffer after the last solve (tests of the solver
 * kernels' intermediate results).  Returns 1, 0 if the range is outside the buffer. */
int svin_ba_debug_peek_solver_scratch(svin_ba* h, uint64_t offset, uint64_t count, double* out);
/* marginalisation prior: returns its dimension m; H (m x m), b0 (m), J (m x m), e0 (m) may be NULL.
 * J and e0 are defined up to a left-orthogonal factor: what the solver (and Ceres in the reference) consumes is J^T J, J^T e0
 * and e0 . e0, and those equal the reference's (MarginalizationError.cpp:746-750).  When the prior has full numerical rank the
 * default route (k_marg_final_chol, certified: the rank rule of MarginalizationError.cpp:1055-1065 drops nothing) returns the
 * triangular square root J = L^T P, e0 = -L^-1 P^-1 b0 instead of the eigenbasis form U sqrt(S); a caller that compares J / e0
 * ROW BY ROW with the reference selects the eigen-solve with svin_ba_debug_set_option("SVIN_MARG_EIG", 1). */
int svin_ba_get_prior(svin_ba* h, double* H, double* b0, double* J, double* e0, uint64_t* block_ids,
                      int32_t* block_ordering, int32_t* block_mdim, int32_t* n_blocks, int cap_m);
/* the linear system of the last svin_ba_apply_marginalization_strategy AFTER MarginalizationError::addResidualBlock (M1,
 * src/MarginalizationError.cpp:126-397) and BEFORE marginalizeOut, kept only when SVIN_MARG_KEEP_PRE is set in the
 * environment (inspection: lets a test arbitrate M1 and M2 / M3 separately): H = [U W; W^T blockdiag(V)], b = [ba; bb] with
 * U m x m (dense blocks, old prior first), W m x 3 n_landmarks, V 3x3 per landmark; marg_rows[i] = 1 for the dense rows that
 * are marginalised.  *m / *n_landmarks always receive the sizes; returns 1 when the arrays were filled (capacities suffice). */
int svin_ba_get_marg_pre(svin_ba* h, int32_t* m, int32_t* n_landmarks, double* U, double* ba, double* W, double* V,
                         double* bb, int32_t* marg_rows, int cap_m, int cap_landmarks);
/* ... and its ordering: the dense blocks (id, first row, number of rows; 0 rows = a fixed block) and the landmark ids in the
 * order of their 3x3 blocks.  Returns the number of dense blocks. */
int svin_ba_get_marg_pre_blocks(svin_ba* h, uint64_t* dense_ids, int32_t* dense_ord, int32_t* dense_mdim, int cap_dense,
                                uint64_t* landmark_ids, int cap_landmarks);
/* semantic description of an internal parameter-block id: kind 0 pose / 1 extrinsics / 2 speed-bias */
int svin_ba_describe_block(svin_ba* h, uint64_t block_id, uint64_t* frame_id, int32_t* kind, int32_t* index);

/* ---- device-resident window (design in svin_amd/csrc/resident.hpp).  A narrow window (<= 42 pose blocks, one
 * GPU, no HomogeneousPointError) keeps its landmark-major observation table on the device from frame to frame: optimize()
 * sends the observations added / removed since the last call and one kernel rebuilds the table (the reference re-reads
 * its four hash containers, Map.cpp:341-376, :467-492).  mode 0 (default): resident whenever the window qualifies;
 * mode 1: always the host path (graph -> arrays on the host, full upload) -- the form the tests compare the resident one with. */
int svin_ba_set_pack_mode(svin_ba* h, int mode);
/* inspection: builds the device tables exactly as optimize() would and copies the observation CSR back: sizes first (call
 * with null arrays), then lm_ptr[L + 1], obs_lm[N], obs_idx[N] (pose slot | extrinsics slot << 12 | camera << 24), uv[2 N],
 * w[N], lm[4 L], obs_order[N] (-1 when the window needs no per-chunk pose order); *resident = 1 when the resident path built it */
int svin_ba_debug_csr(svin_ba* h, int32_t* n_landmarks, int32_t* n_observations, int32_t* lm_ptr, int32_t* obs_lm,
                      uint32_t* obs_idx, double* uv, double* w, double* lm, int32_t* obs_order, int32_t* resident);

/* ---- measurement hook: the Jacobian-evaluation kernel on `copies` replicas of the current window's
 * observation set (HBM-resident working set). Runs `iters` launches, returns the mean kernel time in
 * milliseconds measured with HIP events on the handle's stream; *bytes_per_launch receives the
 * algorithmic byte count of one launch. */
int svin_ba_bench_jacobian_eval(svin_ba* h, int copies, int iters, double* mean_ms, double* bytes_per_launch);
/* the same, plus *back_to_back_ms: the `iters` launches enqueued back to back between ONE pair of HIP events, divided by
 * `iters` (the write-back of launch i overlaps launch i + 1; a per-launch bracket can stop the clock while up to an
 * Infinity Cache's worth of stores is still on the die) */
int svin_ba_bench_jacobian_eval_b2b(svin_ba* h, int copies, int iters, double* mean_ms, double* back_to_back_ms,
                                    double* bytes_per_launch);
/* the collective of the sharded solve on its own: `iters` in-place FP64 sum all-reduces of n_doubles values through the
 * communicator of svin_ba_set_distributed_rccl, on the handle's stream, timed with HIP events (mean microseconds per
 * all-reduce).  Collective: every rank calls it with the same arguments. */
int svin_ba_bench_allreduce(svin_ba* h, uint64_t n_doubles, int iters, double* mean_us);
/* mean wall time (ms) of one reprojection-evaluation launch on the plain window (cache-resident) */
int svin_ba_bench_kernel_times(svin_ba* h, int iters, double* eval_ms, double* build_ms, double* solve_ms);

#ifdef __cplusplus
}
#endif
#endif /* SVIN_BA_H_ */
