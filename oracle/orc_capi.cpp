// ORACLE -- test infrastructure only (never linked into the product library).
// C API over the CPU restatement, bound from Python with ctypes (tests/, bench.py cpu_baseline).
#include "orc_estimator.hpp"
#include <cstdio>

using namespace orc;

namespace {
Camera makeCamera(int model, const double* intr, const double* dist, int w, int h) {
  Camera c;
  c.model = model;
  c.fu = intr[0]; c.fv = intr[1]; c.cu = intr[2]; c.cv = intr[3];
  for (int i = 0; i < 8; ++i) c.k[i] = dist ? dist[i] : 0.0;
  c.width = w; c.height = h;
  return c;
}
ImuParameters makeImu(const double* p) {
  ImuParameters q;
  q.a_max = p[0]; q.g_max = p[1]; q.sigma_g_c = p[2]; q.sigma_a_c = p[3]; q.sigma_bg = p[4]; q.sigma_ba = p[5];
  q.sigma_gw_c = p[6]; q.sigma_aw_c = p[7]; q.tau = p[8]; q.g = p[9];
  q.a0[0] = p[10]; q.a0[1] = p[11]; q.a0[2] = p[12];
  return q;
}
std::vector<ImuSample> makeImuSamples(int n, const uint32_t* t, const double* m) {
  std::vector<ImuSample> v(n);
  for (int i = 0; i < n; ++i) {
    v[i].t.sec = t[2 * i]; v[i].t.nsec = t[2 * i + 1];
    for (int k = 0; k < 3; ++k) { v[i].gyr[k] = m[6 * i + k]; v[i].acc[k] = m[6 * i + 3 + k]; }
  }
  return v;
}
}  // namespace

extern "C" {

// ------------------------------------------------------------------ estimator
void* orc_create() { return new Estimator(); }
void orc_destroy(void* h) { delete static_cast<Estimator*>(h); }
uint64_t orc_new_id(void* h) { return static_cast<Estimator*>(h)->newId(); }
void* orc_estimator_map(void* h) { return &static_cast<Estimator*>(h)->map(); }

int orc_add_camera(void* h, int model, const double* intr, const double* dist, int w, int hh, const double* sig) {
  ExtrinsicsEstimationParameters p;
  p.sigma_absolute_translation = sig[0]; p.sigma_absolute_orientation = sig[1];
  p.sigma_c_relative_translation = sig[2]; p.sigma_c_relative_orientation = sig[3];
  return static_cast<Estimator*>(h)->addCamera(p, makeCamera(model, intr, dist, w, hh));
}
int orc_add_imu(void* h, const double* p) { return static_cast<Estimator*>(h)->addImu(makeImu(p)); }
void orc_set_sonar_extrinsics(void* h, const double* T) { static_cast<Estimator*>(h)->setSonarExtrinsics(Transformation(T)); }

int orc_add_states(void* h, uint64_t id, uint32_t sec, uint32_t nsec, uint64_t numKeypoints, int nCam, const double* T_SC,
                   int nImu, const uint32_t* imuT, const double* imuM, int asKeyframe, int nSonar, const double* sonar,
                   int nDepth, const double* depth, double firstDepth) {
  std::vector<Transformation> tsc;
  for (int i = 0; i < nCam; ++i) tsc.emplace_back(T_SC + 7 * i);
  std::vector<SonarMeasurement> sm;
  for (int i = 0; i < nSonar; ++i) sm.push_back({sonar[2 * i], sonar[2 * i + 1]});
  std::vector<double> dm(depth, depth + nDepth);
  Time t; t.sec = sec; t.nsec = nsec;
  return static_cast<Estimator*>(h)->addStates(id, t, numKeypoints, tsc, makeImuSamples(nImu, imuT, imuM), asKeyframe != 0,
                                               sm, dm, firstDepth) ? 1 : 0;
}
int orc_add_landmark(void* h, uint64_t id, const double* hp) { return static_cast<Estimator*>(h)->addLandmark(id, hp) ? 1 : 0; }
uint64_t orc_add_observation(void* h, uint64_t lm, uint64_t pose, uint64_t cam, uint64_t kp, const double* uv, double size) {
  return static_cast<Estimator*>(h)->addObservation(lm, pose, cam, kp, uv, size);
}
int orc_remove_observation(void* h, uint64_t lm, uint64_t pose, uint64_t cam, uint64_t kp) {
  return static_cast<Estimator*>(h)->removeObservation(lm, pose, cam, kp) ? 1 : 0;
}
int orc_remove_observation_by_id(void* h, uint64_t rid) { return static_cast<Estimator*>(h)->removeObservation(rid) ? 1 : 0; }
void orc_optimize(void* h, uint64_t numIter, uint64_t numThreads, int verbose) {
  static_cast<Estimator*>(h)->optimize(numIter, numThreads, verbose != 0);
}
int orc_set_time_limit(void* h, double tl, int minIter) { return static_cast<Estimator*>(h)->setOptimizationTimeLimit(tl, minIter) ? 1 : 0; }
int orc_apply_marginalization(void* h, uint64_t numKF, uint64_t numImu, uint64_t* removedIds, int cap, int* nRemoved) {
  std::vector<MapPoint> removed;
  const bool ok = static_cast<Estimator*>(h)->applyMarginalizationStrategy(numKF, numImu, removed);
  if (nRemoved) *nRemoved = (int)removed.size();
  for (int i = 0; i < (int)removed.size() && i < cap; ++i) removedIds[i] = removed[i].id;
  return ok ? 1 : 0;
}
int orc_get_T_WS(void* h, uint64_t id, double* T) { return static_cast<Estimator*>(h)->get_T_WS(id, T) ? 1 : 0; }
int orc_get_speed_and_bias(void* h, uint64_t id, uint64_t imu, double* sb) { return static_cast<Estimator*>(h)->getSpeedAndBias(id, imu, sb) ? 1 : 0; }
int orc_get_camera_sensor_states(void* h, uint64_t id, uint64_t cam, double* T) { return static_cast<Estimator*>(h)->getCameraSensorStates(id, cam, T) ? 1 : 0; }
int orc_get_landmark(void* h, uint64_t id, double* hp, double* quality, double* distance, int* nObs) {
  MapPoint mp;
  if (!static_cast<Estimator*>(h)->getLandmark(id, mp)) return 0;
  std::memcpy(hp, mp.point, 4 * sizeof(double));
  if (quality) *quality = mp.quality;
  if (distance) *distance = mp.distance;
  if (nObs) *nObs = (int)mp.observations.size();
  return 1;
}
int orc_set_T_WS(void* h, uint64_t id, const double* T) { return static_cast<Estimator*>(h)->set_T_WS(id, T) ? 1 : 0; }
int orc_set_camera_sensor_states(void* h, uint64_t id, uint64_t cam, const double* T) { return static_cast<Estimator*>(h)->setCameraSensorStates(id, cam, T) ? 1 : 0; }
int orc_set_speed_and_bias(void* h, uint64_t id, uint64_t imu, const double* sb) { return static_cast<Estimator*>(h)->setSpeedAndBias(id, imu, sb) ? 1 : 0; }
int orc_set_landmark(void* h, uint64_t id, const double* hp) { return static_cast<Estimator*>(h)->setLandmark(id, hp) ? 1 : 0; }
uint64_t orc_num_frames(void* h) { return static_cast<Estimator*>(h)->numFrames(); }
uint64_t orc_num_landmarks(void* h) { return static_cast<Estimator*>(h)->numLandmarks(); }
uint64_t orc_current_keyframe_id(void* h) { return static_cast<Estimator*>(h)->currentKeyframeId(); }
uint64_t orc_current_frame_id(void* h) { return static_cast<Estimator*>(h)->currentFrameId(); }
uint64_t orc_frame_id_by_age(void* h, uint64_t age) { return static_cast<Estimator*>(h)->frameIdByAge(age); }
int orc_is_keyframe(void* h, uint64_t id) { return static_cast<Estimator*>(h)->isKeyframe(id) ? 1 : 0; }
int orc_is_in_imu_window(void* h, uint64_t id) { return static_cast<Estimator*>(h)->isInImuWindow(id) ? 1 : 0; }
int orc_frame_ids(void* h, uint64_t* ids, int cap) {
  int n = 0;
  for (auto& kv : static_cast<Estimator*>(h)->states()) { if (n < cap) ids[n] = kv.first; ++n; }
  return n;
}
int orc_landmark_ids(void* h, uint64_t* ids, int cap) {
  int n = 0;
  for (auto& kv : static_cast<Estimator*>(h)->landmarks()) { if (n < cap) ids[n] = kv.first; ++n; }
  return n;
}
// keyframe message content for pose_graph (ThreadedKFVio.cpp:1147-1240), same contract as svin_ba_keyframe_points
int orc_keyframe_points(void* h, uint64_t frame_id, uint64_t cam_idx, int cap_points, uint64_t* lm_ids, double* xyz,
                        uint64_t* kp_idx, double* quality, int* obs_ptr, int cap_obs, uint64_t* obs_frame_ids, int* n_obs_total) {
  int n = 0, no = 0;
  for (auto& kv : static_cast<Estimator*>(h)->landmarks()) {
    const MapPoint& mp = kv.second;
    for (auto mit = mp.observations.begin(); mit != mp.observations.end(); ++mit) {
      if (std::get<0>(mit->first) != frame_id) continue;
      if (std::get<1>(mit->first) != cam_idx) break;   // :1183 `continue` on the first match, then :1229 never reached: next landmark
      if (n < cap_points) {
        if (lm_ids) lm_ids[n] = kv.first;
        if (xyz) for (int k = 0; k < 3; ++k) xyz[3 * n + k] = mp.point[k] / mp.point[3];
        if (kp_idx) kp_idx[n] = std::get<2>(mit->first);
        if (quality) quality[n] = mp.quality;
        if (obs_ptr) obs_ptr[n] = no;
      }
      for (auto oit = mp.observations.begin(); oit != mp.observations.end(); ++oit) {
        if (std::get<0>(oit->first) == frame_id) continue;
        if (no < cap_obs && obs_frame_ids) obs_frame_ids[no] = std::get<0>(oit->first);
        ++no;
      }
      ++n;
      break;
    }
  }
  if (obs_ptr && n <= cap_points) obs_ptr[n] = no;
  if (n_obs_total) *n_obs_total = no;
  return n;
}
// [initial_cost, final_cost, iterations, successful_steps, termination, total_time]
void orc_summary(void* h, double* out) {
  const SolverSummary& s = static_cast<Estimator*>(h)->map().summary;
  out[0] = s.initial_cost; out[1] = s.final_cost; out[2] = s.iterations; out[3] = s.num_successful_steps;
  out[4] = s.termination; out[5] = s.total_time;
}
int orc_cost_history(void* h, double* out, int cap) {
  const SolverSummary& s = static_cast<Estimator*>(h)->map().summary;
  for (int i = 0; i < (int)s.cost_history.size() && i < cap; ++i) out[i] = s.cost_history[i];
  return (int)s.cost_history.size();
}
void orc_set_solver_options(void* h, double function_tol, double gradient_tol, double parameter_tol, int jacobi_scaling) {
  SolverOptions& o = static_cast<Estimator*>(h)->map().options;
  o.function_tolerance = function_tol; o.gradient_tolerance = gradient_tol; o.parameter_tolerance = parameter_tol;
  o.jacobi_scaling = jacobi_scaling != 0;
}
// marginalisation prior inspection: returns size n; fills H (n*n), b0 (n), J (n*n), e0 (n) if non-null
int orc_marg_size(void* h) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  return m ? m->size() : 0;
}
int orc_marg_get(void* h, double* H, double* b0, double* J, double* e0) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  if (!m) return 0;
  const int n = m->size();
  if (H) std::memcpy(H, m->H().data(), sizeof(double) * n * n);
  if (b0) std::memcpy(b0, m->b0().data(), sizeof(double) * n);
  if (J && (int)m->Jmat().size() == n * n) std::memcpy(J, m->Jmat().data(), sizeof(double) * n * n);
  if (e0 && (int)m->e0().size() == n) std::memcpy(e0, m->e0().data(), sizeof(double) * n);
  return n;
}
// test hook (orc_marg.hpp PreMarg): returns n; H (n x n), b0 (n); ranges = {first, size} pairs, landmark part then dense part
int orc_marg_pre(void* h, double* H, double* b0, int* nLm, int* nDense, int* ranges, int cap) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  if (!m) return 0;
  const auto& p = m->preMarg();
  if (H) std::memcpy(H, p.H.data(), sizeof(double) * p.H.size());
  if (b0) std::memcpy(b0, p.b0.data(), sizeof(double) * p.b0.size());
  if (nLm) *nLm = (int)p.lm.size();
  if (nDense) *nDense = (int)p.dense.size();
  int k = 0;
  if (ranges) {
    for (auto& pr : p.lm) if (k + 2 <= cap) { ranges[k++] = pr.first; ranges[k++] = pr.second; }
    for (auto& pr : p.dense) if (k + 2 <= cap) { ranges[k++] = pr.first; ranges[k++] = pr.second; }
  }
  return p.n;
}
// test hook (orc_marg.hpp PreMarg): the blocks as M1 left them -- id, first row, minimal dimension, type (0 pose, 1 speed/bias,
// 2 landmark), linearisation point (9 doubles) -- and the residuals M1 linearised since the previous marginalizeOut
int orc_marg_pre_blocks(void* h, uint64_t* ids, int* ordering, int* mdim, int* type, double* lin9, int cap, int* hadPrior) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  if (!m) return 0;
  const auto& p = m->preMarg();
  if (hadPrior) *hadPrior = p.hadPrior ? 1 : 0;
  int n = 0;
  for (auto& inf : p.infos) {
    if (n < cap) {
      if (ids) ids[n] = inf.id;
      if (ordering) ordering[n] = inf.orderingIdx;
      if (mdim) mdim[n] = inf.mdim;
      if (type) type[n] = inf.type;
      if (lin9) std::memcpy(lin9 + 9 * n, inf.lin, sizeof(double) * 9);
    }
    ++n;
  }
  return n;
}
int orc_marg_log_count(void* h) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  return m ? (int)m->preMarg().log.size() : 0;
}
// entry i: returns the length of its definition vector (written to def when cap suffices); ids4: its parameter blocks
int orc_marg_log_entry(void* h, int i, uint64_t* resId, int* kind, int* loss, double* lossParam, uint64_t* ids4, int* nIds, double* def,
                       int cap) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  if (!m || i < 0 || i >= (int)m->preMarg().log.size()) return -1;
  const auto& e = m->preMarg().log[i];
  if (resId) *resId = e.resId;
  if (kind) *kind = e.kind;
  if (loss) *loss = e.loss;
  if (lossParam) *lossParam = e.lossParam;
  if (nIds) *nIds = (int)e.ids.size();
  if (ids4) for (size_t k = 0; k < e.ids.size() && k < 4; ++k) ids4[k] = e.ids[k];
  if (def && (int)e.def.size() <= cap) std::memcpy(def, e.def.data(), sizeof(double) * e.def.size());
  return (int)e.def.size();
}
// per connected block: id, orderingIdx, mdim ; returns number of blocks
int orc_marg_blocks(void* h, uint64_t* ids, int* ordering, int* mdim, double* lin9, int cap) {
  auto m = static_cast<Estimator*>(h)->marginalizationError();
  if (!m) return 0;
  int n = 0;
  for (auto& inf : m->infos()) {
    if (n < cap) {
      ids[n] = inf.id; ordering[n] = inf.orderingIdx; mdim[n] = inf.mdim;
      if (lin9) std::memcpy(lin9 + 9 * n, inf.lin, sizeof(double) * 9);
    }
    ++n;
  }
  return n;
}
// semantic description of a parameter block id: kind 0 pose, 1 extrinsics, 2 speedbias; frame id; index
int orc_describe_block(void* h, uint64_t id, uint64_t* frameId, int* kind, int* index) {
  for (auto& kv : static_cast<Estimator*>(h)->states()) {
    const auto& s = kv.second;
    if (s.T_WS.id == id) { *frameId = s.id; *kind = 0; *index = 0; return 1; }
    for (size_t i = 0; i < s.T_SC.size(); ++i) if (s.T_SC[i].id == id) { *frameId = s.id; *kind = 1; *index = (int)i; return 1; }
    for (size_t i = 0; i < s.speedAndBias.size(); ++i) if (s.speedAndBias[i].id == id) { *frameId = s.id; *kind = 2; *index = (int)i; return 1; }
  }
  return 0;
}

// ------------------------------------------------------------------ raw map API
void* orc_map_create() { return new Map(); }
void orc_map_destroy(void* m) { delete static_cast<Map*>(m); }
int orc_map_add_param(void* m, uint64_t id, int type, const double* x) { return static_cast<Map*>(m)->addParameterBlock(id, type, x) ? 1 : 0; }
int orc_map_set_constant(void* m, uint64_t id, int constant) {
  return (constant ? static_cast<Map*>(m)->setParameterBlockConstant(id) : static_cast<Map*>(m)->setParameterBlockVariable(id)) ? 1 : 0;
}
// Map::resetParameterization (Map.cpp:513-543) on a pose block: manifold = 6 (PoseManifold) / 3 / 4 / 2 (PoseManifold3d / 4d / 2d)
int orc_map_reset_parameterization(void* m, uint64_t id, int manifold) { return static_cast<Map*>(m)->resetParameterization(id, manifold) ? 1 : 0; }
int orc_map_get_param(void* m, uint64_t id, double* x) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->parameterBlockExists(id)) return 0;
  const ParamBlock& b = mp->param(id);
  std::memcpy(x, b.x, sizeof(double) * b.dim());
  return b.dim();
}
int orc_map_set_param(void* m, uint64_t id, const double* x) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->parameterBlockExists(id)) return 0;
  ParamBlock& b = mp->param(id);
  std::memcpy(b.x, x, sizeof(double) * b.dim());
  return 1;
}
uint64_t orc_map_add_reproj(void* m, int model, const double* intr, const double* dist, const double* uv,
                            const double* info4, int loss, uint64_t poseId, uint64_t lmId, uint64_t extId) {
  auto e = std::make_shared<ReprojectionError>(makeCamera(model, intr, dist, 0, 0), uv, info4);
  return static_cast<Map*>(m)->addResidualBlock(e, loss, {poseId, lmId, extId});
}
uint64_t orc_map_add_imu(void* m, int nImu, const uint32_t* t, const double* meas, const double* par, uint32_t s0,
                         uint32_t ns0, uint32_t s1, uint32_t ns1, const uint64_t* ids4) {
  Time t0, t1; t0.sec = s0; t0.nsec = ns0; t1.sec = s1; t1.nsec = ns1;
  auto e = std::make_shared<ImuError>(makeImuSamples(nImu, t, meas), makeImu(par), t0, t1);
  return static_cast<Map*>(m)->addResidualBlock(e, LOSS_NONE, {ids4[0], ids4[1], ids4[2], ids4[3]});
}
uint64_t orc_map_add_pose_error(void* m, const double* T, const double* info36, uint64_t id) {
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<PoseError>(Transformation(T), info36), LOSS_NONE, {id});
}
uint64_t orc_map_add_pose_error_var(void* m, const double* T, double tv, double rv, uint64_t id) {
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<PoseError>(Transformation(T), tv, rv), LOSS_NONE, {id});
}
uint64_t orc_map_add_speedbias_error(void* m, const double* meas, double sv, double gv, double av, uint64_t id) {
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<SpeedAndBiasError>(meas, sv, gv, av), LOSS_NONE, {id});
}
uint64_t orc_map_add_relpose_error(void* m, double tv, double rv, uint64_t id0, uint64_t id1) {
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<RelativePoseError>(tv, rv), LOSS_NONE, {id0, id1});
}
uint64_t orc_map_add_sonar_error(void* m, const double* T_SSo, double range, double heading, double info, int k,
                                 const double* patch, uint64_t id) {
  std::vector<double> p(patch, patch + 3 * k);
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<SonarError>(Transformation(T_SSo), range, heading, info, p),
                                                LOSS_NONE, {id});
}
uint64_t orc_map_add_depth_error(void* m, double depth, double info, double firstDepth, uint64_t id) {
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<DepthError>(depth, info, firstDepth), LOSS_NONE, {id});
}
uint64_t orc_map_add_hpoint_error(void* m, const double* meas, double variance, uint64_t id) {
  return static_cast<Map*>(m)->addResidualBlock(std::make_shared<HomogeneousPointError>(meas, variance), LOSS_NONE, {id});
}
// ErrorTerm::Kind of a residual block (0 = reprojection), -1 if it does not exist
int orc_map_residual_kind(void* m, uint64_t rid) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->residualExists(rid)) return -1;
  return (int)mp->residual(rid).err->kind();
}
int orc_map_residual_ids(void* m, uint64_t* ids, int cap) {
  int n = 0;
  for (auto& kv : static_cast<Map*>(m)->residualMap()) { if (n < cap) ids[n] = kv.first; ++n; }
  return n;
}
// Map::residuals(paramId) / Map::parameters(resId) (Map.cpp:576-620)
int orc_map_residuals_of(void* m, uint64_t pid, uint64_t* ids, int cap) {
  const std::vector<uint64_t> v = static_cast<Map*>(m)->residuals(pid);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
  return (int)v.size();
}
int orc_map_parameters_of(void* m, uint64_t rid, uint64_t* ids, int cap) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->residualExists(rid)) return -1;
  const ResidualBlock& rb = mp->residual(rid);
  for (int i = 0; i < (int)rb.params.size() && i < cap; ++i) ids[i] = rb.params[i];
  return (int)rb.params.size();
}
int orc_map_is_constant(void* m, uint64_t pid) { return static_cast<Map*>(m)->param(pid).fixed ? 1 : 0; }
int orc_map_remove_residual(void* m, uint64_t rid) { return static_cast<Map*>(m)->removeResidualBlock(rid) ? 1 : 0; }
int orc_map_remove_param(void* m, uint64_t id) { return static_cast<Map*>(m)->removeParameterBlock(id) ? 1 : 0; }   // Map.cpp:322-333
// dims: m, nb, then per block dim / mdim
int orc_map_residual_dims(void* m, uint64_t rid, int* dims, int cap) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->residualExists(rid)) return 0;
  const ResidualBlock& rb = mp->residual(rid);
  int n = 0;
  auto put = [&](int v) { if (n < cap) dims[n] = v; ++n; };
  put(rb.err->residualDim());
  put((int)rb.params.size());
  for (uint64_t p : rb.params) { put(mp->param(p).dim()); put(mp->param(p).mdim()); }
  return n;
}
// evaluates at the current parameter values. J / Jmin are concatenations of the row-major per-block matrices.
int orc_map_eval(void* m, uint64_t rid, double* r, double* J, double* Jmin) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->residualExists(rid)) return 0;
  const ResidualBlock& rb = mp->residual(rid);
  const int mm = rb.err->residualDim(), nb = (int)rb.params.size();
  std::vector<const double*> P(nb);
  std::vector<double*> Jp(nb, nullptr), Jmp(nb, nullptr);
  size_t o = 0, om = 0;
  for (int i = 0; i < nb; ++i) {
    const ParamBlock& b = mp->param(rb.params[i]);
    P[i] = b.x;
    if (J) { Jp[i] = J + o; o += (size_t)mm * b.dim(); }
    if (Jmin) { Jmp[i] = Jmin + om; om += (size_t)mm * b.mdim(); }
  }
  // the reference only fills minimal Jacobians when full Jacobians are requested too
  std::vector<double> scratch;
  if (!J && Jmin) {
    size_t tot = 0;
    for (int i = 0; i < nb; ++i) tot += (size_t)mm * mp->param(rb.params[i]).dim();
    scratch.assign(tot, 0.0);
    size_t oo = 0;
    for (int i = 0; i < nb; ++i) { Jp[i] = scratch.data() + oo; oo += (size_t)mm * mp->param(rb.params[i]).dim(); }
  }
  rb.err->evaluate(P.data(), r, (J || Jmin) ? Jp.data() : nullptr, Jmin ? Jmp.data() : nullptr);
  return 1;
}
int orc_map_is_jacobian_correct(void* m, uint64_t rid, double relTol, double* worst) {
  return static_cast<Map*>(m)->isJacobianCorrect(rid, relTol, worst) ? 1 : 0;
}
int orc_map_get_lhs(void* m, uint64_t id, double* H) { static_cast<Map*>(m)->getLhs(id, H); return 1; }
void orc_map_solve(void* m, int maxIter, int verbose, double* summary6) {
  Map* mp = static_cast<Map*>(m);
  mp->options.max_num_iterations = maxIter;
  mp->options.verbose = verbose != 0;
  mp->solve();
  if (summary6) {
    const SolverSummary& s = mp->summary;
    summary6[0] = s.initial_cost; summary6[1] = s.final_cost; summary6[2] = s.iterations;
    summary6[3] = s.num_successful_steps; summary6[4] = s.termination; summary6[5] = s.total_time;
  }
}
void orc_map_set_tolerances(void* m, double f, double g, double p) {
  Map* mp = static_cast<Map*>(m);
  mp->options.function_tolerance = f; mp->options.gradient_tolerance = g; mp->options.parameter_tolerance = p;
}
// linearisation snapshot for parity tests
static Map::Linearization g_lin;
int orc_map_linearize(void* m, double mu, int* d, int* nCam, int* nLm, double* cost) {
  static_cast<Map*>(m)->linearize(g_lin, mu);
  *d = g_lin.d; *nCam = (int)g_lin.camIds.size(); *nLm = (int)g_lin.lmIds.size(); *cost = g_lin.cost;
  return 1;
}
void orc_map_linearize_get(uint64_t* camIds, int* camOff, double* S, double* g, double* A, double* b, uint64_t* lmIds,
                           double* V, double* bl) {
  const int d = g_lin.d;
  if (camIds) std::memcpy(camIds, g_lin.camIds.data(), sizeof(uint64_t) * g_lin.camIds.size());
  if (camOff) std::memcpy(camOff, g_lin.camOffsets.data(), sizeof(int) * g_lin.camOffsets.size());
  if (S) std::memcpy(S, g_lin.S.data(), sizeof(double) * d * d);
  if (g) std::memcpy(g, g_lin.g.data(), sizeof(double) * d);
  if (A) std::memcpy(A, g_lin.A.data(), sizeof(double) * d * d);
  if (b) std::memcpy(b, g_lin.b.data(), sizeof(double) * d);
  if (lmIds) std::memcpy(lmIds, g_lin.lmIds.data(), sizeof(uint64_t) * g_lin.lmIds.size());
  if (V) std::memcpy(V, g_lin.V.data(), sizeof(double) * g_lin.V.size());
  if (bl) std::memcpy(bl, g_lin.bl.data(), sizeof(double) * g_lin.bl.size());
}

// ------------------------------------------------------------------ stand-alone math
int orc_project(int model, const double* intr, const double* dist, int w, int h, const double* p, double* kp, double* J) {
  return project(makeCamera(model, intr, dist, w, h), p, kp, J);
}
int orc_project_homogeneous(int model, const double* intr, const double* dist, int w, int h, const double* hp, double* kp, double* Jh) {
  return projectHomogeneous(makeCamera(model, intr, dist, w, h), hp, kp, Jh);
}
int orc_distort(int model, const double* dist, const double* u, double* d, double* J) {
  const double intr[4] = {1, 1, 0, 0};
  return distort(makeCamera(model, intr, dist, 0, 0), u, d, J) ? 1 : 0;
}
void orc_manifold_plus(int type, const double* x, const double* delta, double* xp) { manifoldPlus(type, x, delta, xp); }
void orc_manifold_minus(int type, const double* xp, const double* x, double* d) { manifoldMinus(type, xp, x, d); }
void orc_manifold_plus_jacobian(int type, const double* x, double* J) { manifoldPlusJacobian(type, x, J); }
void orc_manifold_lift_jacobian(int type, const double* x, double* J) { manifoldLiftJacobian(type, x, J); }
void orc_pose_minus_jacobian(const double* x, double* J) { poseMinusJacobian(x, J); }
void orc_sym_eig(const double* A, int n, double* ev, double* V) { sym_eig(A, n, ev, V); }
void orc_right_jacobian(const double* phi, double* J) { rightJacobian(phi, J); }
void orc_transformation_inverse(const double* T, double* Ti) { Transformation t(T); std::memcpy(Ti, t.inverse().p, 56); }
void orc_transformation_compose(const double* A, const double* B, double* AB) {
  Transformation a(A), b(B);
  std::memcpy(AB, (a * b).p, 56);
}
int orc_init_pose_from_imu(int n, const uint32_t* t, const double* m, double* T) {
  Transformation tw;
  const bool ok = Estimator::initPoseFromImu(makeImuSamples(n, t, m), tw);
  std::memcpy(T, tw.p, 56);
  return ok ? 1 : 0;
}
// ImuError::propagation; cov / jac are 15x15 or null.  T (7) and sb (9) are in/out.
int orc_imu_propagation(int n, const uint32_t* t, const double* m, const double* par, double* T, double* sb, uint32_t s0,
                        uint32_t ns0, uint32_t s1, uint32_t ns1, double* cov, double* jac) {
  Time t0, t1; t0.sec = s0; t0.nsec = ns0; t1.sec = s1; t1.nsec = ns1;
  Transformation tw(T);
  const int r = ImuError::propagation(makeImuSamples(n, t, m), makeImu(par), tw, sb, t0, t1, cov, jac);
  std::memcpy(T, tw.p, 56);
  return r;
}
// pre-integration state of an IMU residual in a map (after at least one evaluation):
// out = Delta_q(4) C_integral(9) C_doubleintegral(9) acc_integral(3) acc_doubleintegral(3) dalpha_db_g(9) dv_db_g(9)
//       dp_db_g(9) P_delta(225) information(225) sqrtInformation(225) sb_ref(9) redoCounter(1)
int orc_map_imu_state(void* m, uint64_t rid, double* out) {
  Map* mp = static_cast<Map*>(m);
  if (!mp->residualExists(rid)) return 0;
  auto e = std::dynamic_pointer_cast<ImuError>(mp->residual(rid).err);
  if (!e) return 0;
  double* o = out;
  auto put = [&](const double* s, int n) { std::memcpy(o, s, sizeof(double) * n); o += n; };
  put(e->Delta_q, 4); put(e->C_integral, 9); put(e->C_doubleintegral, 9); put(e->acc_integral, 3);
  put(e->acc_doubleintegral, 3); put(e->dalpha_db_g, 9); put(e->dv_db_g, 9); put(e->dp_db_g, 9);
  put(e->P_delta, 225); put(e->information, 225); put(e->sqrtInformation, 225); put(e->sb_ref, 9);
  *o = e->redoCounter;
  return 1;
}

}  // extern "C"
