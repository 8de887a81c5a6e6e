// svin_amd C ABI (include/svin_ba.h).  Plain pointers and sizes only; nothing throws across the boundary.
#include "../../include/svin_ba.h"

#include <algorithm>
#include "window.hpp"
#include <vector>

using namespace svin;

struct svin_ba {
  Window w;
  explicit svin_ba(int device) : w(device) {}
};

#define GUARD_BEGIN try {
#define GUARD_END(errval)                     \
  }                                           \
  catch (const std::exception& e) {           \
    lastError() = e.what();                   \
    return errval;                            \
  }                                           \
  catch (...) {                               \
    lastError() = "unknown error";            \
    return errval;                            \
  }

// function-try-block tail for the entry points that are host look-ups most of the time but may reach quiesce() /
// syncLandmarks() (which rethrow what the asynchronous marginalisation job threw, or a HIP error): nothing crosses the C ABI
#define CATCH_ALL(errval)                     \
  catch (const std::exception& e) {           \
    lastError() = e.what();                   \
    return errval;                            \
  }                                           \
  catch (...) {                               \
    lastError() = "unknown error";            \
    return errval;                            \
  }

static void splitSamples(const svin_imu_sample* imu, int n, std::vector<uint32_t>& t, std::vector<double>& m) {
  t.resize(2 * (size_t)n);
  m.resize(6 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    t[2 * i] = imu[i].sec; t[2 * i + 1] = imu[i].nsec;
    for (int k = 0; k < 3; ++k) { m[6 * i + k] = imu[i].gyr[k]; m[6 * i + 3 + k] = imu[i].acc[k]; }
  }
}
static ImuParams toParams(const svin_imu_params* p) {
  ImuParams q;
  q.a_max = p->a_max; q.g_max = p->g_max; q.sigma_g_c = p->sigma_g_c; q.sigma_a_c = p->sigma_a_c;
  q.sigma_bg = p->sigma_bg; q.sigma_ba = p->sigma_ba; q.sigma_gw_c = p->sigma_gw_c; q.sigma_aw_c = p->sigma_aw_c;
  q.tau = p->tau; q.g = p->g;
  q.a0[0] = p->a0[0]; q.a0[1] = p->a0[1]; q.a0[2] = p->a0[2];
  return q;
}

extern "C" {

svin_ba* svin_ba_create(int device) {
  try {
    return new svin_ba(device);
  } catch (const std::exception& e) {
    lastError() = e.what();
    return nullptr;
  } catch (...) {
    lastError() = "unknown error";
    return nullptr;
  }
}
void svin_ba_destroy(svin_ba* h) { delete h; }
const char* svin_ba_last_error(void) { return lastError().c_str(); }
uint64_t svin_ba_new_id(svin_ba* h) try { return h ? h->w.newId() : 0; } CATCH_ALL(0)
int svin_ba_set_id_provider(svin_ba* h, svin_id_provider_fn fn, void* user) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  h->w.setIdProvider(fn, user);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_reserve_ids(svin_ba* h, uint64_t largest) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  h->w.reserveIds(largest);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_camera_geometry(svin_ba* h, uint64_t cam, int model, const double intr[4], const double* dist, int n_dist,
                                int width, int height) try {
  if (!h || !intr || (n_dist > 0 && !dist) || n_dist < 0 || n_dist > 8) return SVIN_ERR_INVALID_ARG;
  return h->w.setCameraGeometry(cam, model, intr, dist, n_dist, width, height) ? 1 : SVIN_ERR_NOT_FOUND;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_clear_cameras(svin_ba* h) try { if (!h) return SVIN_ERR_INVALID_ARG; h->w.clearCameras(); return 1; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_clear_imus(svin_ba* h) try { if (!h) return SVIN_ERR_INVALID_ARG; h->w.clearImus(); return 1; } CATCH_ALL(SVIN_ERR_DEVICE)

int svin_ba_add_camera(svin_ba* h, int model, const double intr[4], const double* dist, int n_dist, int width, int height,
                       const double sigmas[4]) {
  if (!h || !intr || !sigmas) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.addCamera(model, intr, dist, n_dist, width, height, sigmas);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_add_imu(svin_ba* h, const svin_imu_params* p) {
  if (!h || !p) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.addImu(toParams(p));
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_set_sonar_extrinsics(svin_ba* h, const double T_SSo[7]) try {
  if (!h || !T_SSo) return SVIN_ERR_INVALID_ARG;
  h->w.setSonarExtrinsics(T_SSo);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_add_states(svin_ba* h, uint64_t frame_id, uint32_t sec, uint32_t nsec, uint64_t num_keypoints,
                       const double* T_SC, int n_cam, const svin_imu_sample* imu, int n_imu, int as_keyframe,
                       const double* sonar, int n_sonar, const double* depth, int n_depth, double first_depth) {
  if (!h || n_imu < 0 || n_cam < 0 || n_sonar < 0 || n_depth < 0 || (n_imu > 0 && !imu) || (n_cam > 0 && !T_SC) ||
      (n_sonar > 0 && !sonar) || (n_depth > 0 && !depth))
    return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  std::vector<uint32_t> t;
  std::vector<double> m;
  splitSamples(imu, n_imu, t, m);
  TimeStamp ts; ts.sec = sec; ts.nsec = nsec;
  return h->w.addStates(frame_id, ts, num_keypoints, T_SC, n_cam, t.data(), m.data(), n_imu, as_keyframe != 0, sonar,
                        n_sonar, depth, n_depth, first_depth);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_add_landmark(svin_ba* h, uint64_t id, const double hp[4]) {
  if (!h || !hp) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.addLandmark(id, hp);
  GUARD_END(SVIN_ERR_DEVICE)
}
uint64_t svin_ba_add_observation(svin_ba* h, uint64_t lm, uint64_t pose, uint64_t cam, uint64_t kp, const double uv[2],
                                 double size) {
  if (!h || !uv) return 0;
  GUARD_BEGIN return h->w.addObservation(lm, pose, cam, kp, uv, size);
  GUARD_END(0)
}
int svin_ba_add_observations(svin_ba* h, int n, const uint64_t* lm, const uint64_t* pose, const uint64_t* cam, const uint64_t* kp,
                             const double* uv, const double* size, uint64_t* out_ids) {
  if (!h || n < 0 || (n > 0 && (!lm || !pose || !cam || !kp || !uv || !size))) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.addObservations(n, lm, pose, cam, kp, uv, size, out_ids);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_remove_observation(svin_ba* h, uint64_t lm, uint64_t pose, uint64_t cam, uint64_t kp) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.removeObservation(lm, pose, cam, kp);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_remove_observation_by_id(svin_ba* h, uint64_t rid) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.removeObservationById(rid);
  GUARD_END(SVIN_ERR_DEVICE)
}
uint64_t svin_ba_add_homogeneous_point_error(svin_ba* h, uint64_t lm, const double* meas, const double* info) {
  if (!h || !meas || !info) return 0;
  try { return h->w.addLandmarkPrior(lm, meas, info); } catch (const std::exception& e) { svin::lastError() = e.what(); return 0; }
}
int svin_ba_remove_homogeneous_point_error(svin_ba* h, uint64_t rid) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.removeLandmarkPrior(rid);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_optimize(svin_ba* h, uint64_t num_iter, uint64_t, int verbose) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.optimize(num_iter, verbose != 0);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_prepare(svin_ba* h) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.prepare();
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_solve_prepared(svin_ba* h, uint64_t num_iter, int verbose) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.solvePrepared(num_iter, verbose != 0);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_solve_prepared_batch(svin_ba* const* hs, int n, uint64_t num_iter, int verbose, int* n_batched) {
  if (n_batched) *n_batched = 0;
  if (n < 0 || (n > 0 && !hs)) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
    std::vector<svin::Window*> ws((size_t)n);
    for (int i = 0; i < n; ++i) {
      if (!hs[i]) return SVIN_ERR_INVALID_ARG;
      ws[(size_t)i] = &hs[i]->w;
    }
    return svin::Window::solvePreparedBatch(ws.data(), n, num_iter, verbose != 0, n_batched);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_optimize_batch(svin_ba* const* hs, int n, uint64_t num_iter, int verbose, int* n_batched) {
  if (n_batched) *n_batched = 0;
  if (n < 0 || (n > 0 && !hs)) return SVIN_ERR_INVALID_ARG;
  for (int i = 0; i < n; ++i)
    if (!hs[i]) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
    for (int i = 0; i < n; ++i) hs[i]->w.prepare();
  GUARD_END(SVIN_ERR_DEVICE)
  const int rc = svin_ba_solve_prepared_batch(hs, n, num_iter, verbose, n_batched);
  if (rc != 1) return rc;
  GUARD_BEGIN
    for (int i = 0; i < n; ++i) hs[i]->w.finish();
    return 1;
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_finish(svin_ba* h) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.finish();
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_invalidate_preintegration(svin_ba* h) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  h->w.invalidatePreintegration();
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_optimization_time_limit(svin_ba* h, double tl, int min_iter) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  return h->w.setOptimizationTimeLimit(tl, min_iter);
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_apply_marginalization_strategy(svin_ba* h, uint64_t nkf, uint64_t nimu, uint64_t* removed, int cap,
                                           int* n_removed) {
  if (!h || cap < 0 || (cap > 0 && !removed)) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  std::vector<uint64_t> rem;
  const int r = h->w.applyMarginalizationStrategy(nkf, nimu, rem);
  if (n_removed) *n_removed = (int)rem.size();
  for (int i = 0; i < (int)rem.size() && i < cap; ++i) removed[i] = rem[i];
  return r;
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_get_summary(svin_ba* h, svin_summary* out) try {
  if (!h || !out) return SVIN_ERR_INVALID_ARG;
  const Summary& s = h->w.summary();
  out->initial_cost = s.initial_cost; out->final_cost = s.final_cost; out->iterations = s.iterations;
  out->num_successful_steps = s.num_successful_steps; out->termination = s.termination;
  out->total_time_s = s.total_time; out->upload_time_s = s.upload_time; out->solve_time_s = s.solve_time;
  out->download_time_s = s.download_time;
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_distributed(svin_ba* h, int rank, int world, svin_allreduce_fn fn, void* user) try {
  if (!h || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return SVIN_ERR_INVALID_ARG;
  if (world > svin::kScalGatherSlots / 2) {   // every rank publishes its (gradient max, factorisation flag) pair in a slot of its own
    svin::lastError() = "set_distributed: at most " + std::to_string(svin::kScalGatherSlots / 2) + " ranks (one node of MI355X)";
    return SVIN_ERR_INVALID_ARG;
  }
  h->w.setDistributed(rank, world, fn, user);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_rccl_unique_id(unsigned char id_out[128]) {
  if (!id_out) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return Window::rcclUniqueId(id_out);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_set_distributed_rccl(svin_ba* h, int rank, int world, const unsigned char id[128]) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return SVIN_ERR_INVALID_ARG;
  if (world > svin::kScalGatherSlots / 2) {
    svin::lastError() = "set_distributed_rccl: at most " + std::to_string(svin::kScalGatherSlots / 2) + " ranks (one node of MI355X)";
    return SVIN_ERR_INVALID_ARG;
  }
  GUARD_BEGIN return h->w.setDistributedRccl(rank, world, id);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_set_solver_tolerances(svin_ba* h, double f, double g, double p) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  h->w.setTolerances(f, g, p);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_T_WS(svin_ba* h, uint64_t id, double T[7]) try { return h ? h->w.get_T_WS(id, T) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_speed_and_bias(svin_ba* h, uint64_t id, uint64_t imu, double sb[9]) try {
  return h ? h->w.getSpeedAndBias(id, imu, sb) : SVIN_ERR_INVALID_ARG;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_camera_sensor_states(svin_ba* h, uint64_t id, uint64_t cam, double T[7]) try {
  return h ? h->w.getCameraSensorStates(id, cam, T) : SVIN_ERR_INVALID_ARG;
} CATCH_ALL(SVIN_ERR_DEVICE)
static void fillInfo(const Landmark& lm, svin_landmark_info* out) {
  for (int k = 0; k < 4; ++k) out->point[k] = lm.hp[k];
  out->quality = lm.quality; out->distance = lm.distance;
  out->num_observations = (int32_t)lm.obs.size();
  out->initialized = lm.initialized ? 1 : 0;
}
int svin_ba_get_landmark(svin_ba* h, uint64_t id, svin_landmark_info* out) try {
  if (!h || !out) return SVIN_ERR_INVALID_ARG;
  const Landmark* lm = h->w.landmark(id);
  if (!lm) return 0;
  fillInfo(*lm, out);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_landmarks(svin_ba* h, uint64_t* ids, svin_landmark_info* infos, int cap) try {
  if (!h || cap < 0) return SVIN_ERR_INVALID_ARG;
  int n = 0;
  for (const auto& kv : h->w.landmarks()) {
    if (n < cap) {
      if (ids) ids[n] = kv.first;
      if (infos) fillInfo(kv.second, infos + n);
    }
    ++n;
  }
  return n;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_landmark_observations(svin_ba* h, uint64_t id, uint64_t* frames, uint64_t* cams, uint64_t* kps, uint64_t* rids,
                                      int cap) try {
  if (!h || cap < 0) return SVIN_ERR_INVALID_ARG;
  const Landmark* lm = h->w.landmarkGraph(id);
  if (!lm) return SVIN_ERR_NOT_FOUND;
  std::vector<const svin::Observation*> sorted;
  for (const svin::Observation& o : lm->obs) sorted.push_back(&o);
  std::sort(sorted.begin(), sorted.end(), [](const svin::Observation* a, const svin::Observation* b) {
    if (a->poseId != b->poseId) return a->poseId < b->poseId;
    if (a->cam != b->cam) return a->cam < b->cam;
    return a->kp < b->kp;
  });
  for (int i = 0; i < (int)sorted.size() && i < cap; ++i) {
    if (frames) frames[i] = sorted[i]->poseId;
    if (cams) cams[i] = (uint64_t)sorted[i]->cam;
    if (kps) kps[i] = sorted[i]->kp;
    if (rids) rids[i] = sorted[i]->resId;
  }
  return (int)sorted.size();
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_all_landmark_observations(svin_ba* h, int cap_landmarks, uint64_t* ids, svin_landmark_info* infos, int32_t* obs_ptr,
                                          int cap_obs, uint64_t* frames, uint64_t* cams, uint64_t* kps, uint64_t* rids, int32_t* n_obs_total) try {
  if (!h || cap_landmarks < 0 || cap_obs < 0) return SVIN_ERR_INVALID_ARG;
  int n = 0, total = 0;
  std::vector<const svin::Observation*> sorted;
  for (const auto& kv : h->w.landmarks()) {
    const Landmark& lm = kv.second;
    if (n < cap_landmarks) {
      if (ids) ids[n] = kv.first;
      if (infos) fillInfo(lm, infos + n);
      if (obs_ptr) obs_ptr[n] = total;
      sorted.clear();
      for (const svin::Observation& o : lm.obs) sorted.push_back(&o);
      std::sort(sorted.begin(), sorted.end(), [](const svin::Observation* a, const svin::Observation* b) {
        if (a->poseId != b->poseId) return a->poseId < b->poseId;
        if (a->cam != b->cam) return a->cam < b->cam;
        return a->kp < b->kp;
      });
      for (size_t i = 0; i < sorted.size(); ++i) {
        const int at = total + (int)i;
        if (at >= cap_obs) break;
        if (frames) frames[at] = sorted[i]->poseId;
        if (cams) cams[at] = (uint64_t)sorted[i]->cam;
        if (kps) kps[at] = sorted[i]->kp;
        if (rids) rids[at] = sorted[i]->resId;
      }
    }
    total += (int)lm.obs.size();
    ++n;
  }
  if (obs_ptr && n <= cap_landmarks) obs_ptr[n] = total;
  if (n_obs_total) *n_obs_total = total;
  return n;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_parameter_block(svin_ba* h, uint64_t id, int32_t* type, double* values, uint32_t* sec, uint32_t* nsec, int32_t* fixed,
                                int32_t* initialized) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  return h->w.getParameterBlock(id, type, values, sec, nsec, fixed, initialized);
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_parameter_block_ids(svin_ba* h, uint64_t* ids, int cap) try {
  if (!h || cap < 0 || (cap > 0 && !ids)) return SVIN_ERR_INVALID_ARG;
  std::vector<uint64_t> v;
  h->w.parameterBlockIds(v);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
  return (int)v.size();
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_is_landmark_initialized(svin_ba* h, uint64_t id) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  const Landmark* lm = h->w.landmarkGraph(id);
  return lm ? (lm->initialized ? 1 : 0) : SVIN_ERR_NOT_FOUND;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_landmark_initialized(svin_ba* h, uint64_t id, int initialized) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  return h->w.setLandmarkInitialized(id, initialized != 0) ? 1 : SVIN_ERR_NOT_FOUND;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_keyframe(svin_ba* h, uint64_t id, int is_kf) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  return h->w.setKeyframe(id, is_kf != 0) ? 1 : SVIN_ERR_NOT_FOUND;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_timestamp(svin_ba* h, uint64_t id, uint32_t* sec, uint32_t* nsec) try {
  if (!h || !sec || !nsec) return SVIN_ERR_INVALID_ARG;
  auto it = h->w.states().find(id);
  if (it == h->w.states().end()) return SVIN_ERR_NOT_FOUND;
  *sec = it->second.stamp.sec; *nsec = it->second.stamp.nsec;
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_state_count(svin_ba* h) try { return h ? h->w.stateCount() : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_imu_preintegral(svin_ba* h, uint64_t pose_id, double adi[3], double ai[3], double* dt) try {
  if (!h || !adi || !ai || !dt) return SVIN_ERR_INVALID_ARG;
  double v[7];
  if (!h->w.getImuPreIntegral(pose_id, v)) return 0;
  for (int k = 0; k < 3; ++k) { adi[k] = v[k]; ai[k] = v[3 + k]; }
  *dt = v[6];
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_imu_preintegral(svin_ba* h, uint64_t pose_id, const double adi[3], const double ai[3], double dt) try {
  if (!h || !adi || !ai) return SVIN_ERR_INVALID_ARG;
  const double v[7] = {adi[0], adi[1], adi[2], ai[0], ai[1], ai[2], dt};
  h->w.setImuPreIntegral(pose_id, v);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_init_pose_from_imu(const svin_imu_sample* imu, int n_imu, double T_WS[7]) {
  if (!T_WS || n_imu < 0 || (n_imu > 0 && !imu)) return SVIN_ERR_INVALID_ARG;
  std::vector<uint32_t> t;
  std::vector<double> m;
  splitSamples(imu, n_imu, t, m);
  return Window::initPoseFromImu(m.data(), n_imu, T_WS) ? 1 : 0;
}
int svin_ba_is_landmark_added(svin_ba* h, uint64_t id) try { return h && h->w.landmarkExists(id) ? 1 : 0; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_T_WS(svin_ba* h, uint64_t id, const double T[7]) try { return h ? h->w.set_T_WS(id, T) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_speed_and_bias(svin_ba* h, uint64_t id, uint64_t imu, const double sb[9]) try {
  return h ? h->w.setSpeedAndBias(id, imu, sb) : SVIN_ERR_INVALID_ARG;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_camera_sensor_states(svin_ba* h, uint64_t id, uint64_t cam, const double T[7]) try {
  return h ? h->w.setCameraSensorStates(id, cam, T) : SVIN_ERR_INVALID_ARG;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_landmark(svin_ba* h, uint64_t id, const double hp[4]) try { return h ? h->w.setLandmark(id, hp) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
uint64_t svin_ba_num_frames(svin_ba* h) try { return h ? h->w.states().size() : 0; } CATCH_ALL(0)
uint64_t svin_ba_num_landmarks(svin_ba* h) try { return h ? h->w.numLandmarks() : 0; } CATCH_ALL(0)
uint64_t svin_ba_current_keyframe_id(svin_ba* h) try { return h ? h->w.currentKeyframeId() : 0; } CATCH_ALL(0)
uint64_t svin_ba_current_frame_id(svin_ba* h) try { return (h && !h->w.states().empty()) ? h->w.states().rbegin()->first : 0; } CATCH_ALL(0)
uint64_t svin_ba_frame_id_by_age(svin_ba* h, uint64_t age) try { return h ? h->w.frameIdByAge(age) : 0; } CATCH_ALL(0)
int svin_ba_is_keyframe(svin_ba* h, uint64_t id) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  auto it = h->w.states().find(id);
  return it == h->w.states().end() ? SVIN_ERR_NOT_FOUND : (it->second.isKeyframe ? 1 : 0);
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_is_in_imu_window(svin_ba* h, uint64_t id) try { return h ? (h->w.isInImuWindow(id) ? 1 : 0) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_frame_ids(svin_ba* h, uint64_t* ids, int cap) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  int n = 0;
  for (auto& kv : h->w.states()) { if (n < cap && ids) ids[n] = kv.first; ++n; }
  return n;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_landmark_ids(svin_ba* h, uint64_t* ids, int cap) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  int n = 0;
  for (auto& kv : h->w.landmarksGraph()) { if (n < cap && ids) ids[n] = kv.first; ++n; }
  return n;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_parameter_block_exists(svin_ba* h, uint64_t id) try { return h ? (h->w.parameterBlockExists(id) ? 1 : 0) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_set_parameter_block_constant(svin_ba* h, uint64_t id, int constant) try {
  return h ? h->w.setParameterBlockConstant(id, constant != 0) : SVIN_ERR_INVALID_ARG;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_reset_parameterization(svin_ba* h, uint64_t id, int parameterization) try {
  return h ? h->w.resetParameterization(id, parameterization) : SVIN_ERR_INVALID_ARG;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_parameterization(svin_ba* h, uint64_t id) try { return h ? h->w.parameterization(id) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_is_parameter_block_constant(svin_ba* h, uint64_t id) try { return h ? h->w.isParameterBlockConstant(id) : SVIN_ERR_INVALID_ARG; } CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_residuals_of(svin_ba* h, uint64_t id, uint64_t* out, int cap) {
  if (!h || cap < 0 || (cap > 0 && !out)) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  std::vector<uint64_t> v;
  if (!h->w.residualsOf(id, v)) return SVIN_ERR_NOT_FOUND;
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_parameters_of(svin_ba* h, uint64_t rid, uint64_t* out, int cap, int32_t* kind) {
  if (!h || cap < 0 || (cap > 0 && !out)) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  std::vector<uint64_t> v;
  if (!h->w.parametersOf(rid, v)) return SVIN_ERR_NOT_FOUND;
  if (kind) *kind = h->w.residualKind(rid);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_keyframe_points(svin_ba* h, uint64_t frame_id, uint64_t cam_idx, int cap_points, uint64_t* lm_ids, double* xyz,
                            uint64_t* kp_idx, double* quality, int32_t* obs_ptr, int cap_obs, uint64_t* obs_frame_ids,
                            int32_t* n_obs_total) {
  if (!h || cap_points < 0 || cap_obs < 0) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  int n = 0, no = 0;
  std::vector<const svin::Observation*> sorted;
  for (const auto& kv : h->w.landmarks()) {   // PointMap order (std::map by landmark id)
    const svin::Landmark& lm = kv.second;
    // MapPoint::observations is ordered by (frame, camera, keypoint); the publisher takes the first entry of the frame
    sorted.clear();
    for (const svin::Observation& o : lm.obs) sorted.push_back(&o);
    std::sort(sorted.begin(), sorted.end(), [](const svin::Observation* a, const svin::Observation* b) {
      if (a->poseId != b->poseId) return a->poseId < b->poseId;
      if (a->cam != b->cam) return a->cam < b->cam;
      return a->kp < b->kp;
    });
    const svin::Observation* first = nullptr;
    for (const svin::Observation* o : sorted)
      if (o->poseId == frame_id) { first = o; break; }
    if (!first || (uint64_t)first->cam != cam_idx) continue;   // ThreadedKFVio.cpp:1167, :1183
    if (n < cap_points) {
      if (lm_ids) lm_ids[n] = lm.id;
      if (xyz) for (int k = 0; k < 3; ++k) xyz[3 * n + k] = lm.hp[k] / lm.hp[3];
      if (kp_idx) kp_idx[n] = first->kp;
      if (quality) quality[n] = lm.quality;
      if (obs_ptr) obs_ptr[n] = no;
    }
    for (const svin::Observation* o : sorted) {
      if (o->poseId == frame_id) continue;   // :1213
      if (no < cap_obs && obs_frame_ids) obs_frame_ids[no] = o->poseId;
      ++no;
    }
    ++n;
  }
  if (obs_ptr && n <= cap_points) obs_ptr[n] = no;
  if (n_obs_total) *n_obs_total = no;
  return n;
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_imu_propagation(svin_ba* h, const svin_imu_sample* imu, int n_imu, const svin_imu_params* p, double T[7],
                            double sb[9], uint32_t s0, uint32_t ns0, uint32_t s1, uint32_t ns1, double* cov, double* jac) {
  if (!h || !imu || !p) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  std::vector<uint32_t> t;
  std::vector<double> m;
  splitSamples(imu, n_imu, t, m);
  TimeStamp a, b; a.sec = s0; a.nsec = ns0; b.sec = s1; b.nsec = ns1;
  return h->w.imuPropagation(t.data(), m.data(), n_imu, toParams(p), T, sb, a, b, cov, jac);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_imu_propagation_integrals(svin_ba* h, const svin_imu_sample* imu, int n_imu, const svin_imu_params* p, double T[7],
                                      double sb[9], uint32_t s0, uint32_t ns0, uint32_t s1, uint32_t ns1, double* cov,
                                      double* jac, double integrals[7]) {
  if (!h || !imu || !p || !T || !sb || n_imu < 0) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN
  std::vector<uint32_t> t;
  std::vector<double> m;
  splitSamples(imu, n_imu, t, m);
  TimeStamp a, b; a.sec = s0; a.nsec = ns0; b.sec = s1; b.nsec = ns1;
  return h->w.imuPropagation(t.data(), m.data(), n_imu, toParams(p), T, sb, a, b, cov, jac, integrals);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_eval_reprojection(svin_ba* h, int robust, double* r, double* Jp, double* Jl, double* Je, int cap) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.evalReprojection(robust != 0, r, Jp, Jl, Je, cap);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_observation_ids(svin_ba* h, uint64_t* rid, uint64_t* lm, uint64_t* pose, int32_t* cam, int cap) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.observationIds(rid, lm, pose, cam, cap);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_eval_factors(svin_ba* h, int32_t* kind, int32_t* m, int32_t* ncols, double* r, double* J, uint64_t* blocks,
                         uint64_t* rids, int cap) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.evalFactors(kind, m, ncols, r, J, blocks, rids, cap);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_linearize(svin_ba* h, double mu, double* S, double* g, uint64_t* ids, int32_t* off, int32_t* nb, int cap_d,
                      double* cost) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.linearize(mu, S, g, ids, off, nb, cap_d, cost);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_wait_idle(svin_ba* h) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN h->w.waitIdle(); return 1;
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_debug_reduced_solve(svin_ba* h, double mu, double* y, int cap_d) {
  if (!h || !y) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.debugReducedSolve(mu, y, cap_d);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_debug_reduced_solve_ex(svin_ba* h, double mu, int fuse_finalize, double* y, int cap_d) {
  if (!h || !y) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.debugReducedSolve(mu, y, cap_d, fuse_finalize != 0);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_get_path_counters(svin_ba* h, int64_t out[4]) try {
  if (!h || !out) return SVIN_ERR_INVALID_ARG;
  const long long* c = h->w.pathCounters();
  for (int i = 0; i < 4; ++i) out[i] = c[i];
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_debug_sym_eig(int n, const double* A, double* eigenvalues, double* eigenvectors, double* device_ms) {
  if (!A || !eigenvalues || !eigenvectors) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return svin::debugSymEig(n, A, eigenvalues, eigenvectors, device_ms);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_debug_set_option(const char* name, int value) { return svin::setDebugOption(name, value); }
int svin_ba_debug_get_option(const char* name, int* value) { return svin::debugOptionByName(name, value); }
int svin_ba_debug_set_switch(const char* name, int value) { return svin::setDebugOption(name, value ? 1 : 0); }
int svin_ba_debug_peek_solver_scratch(svin_ba* h, uint64_t offset, uint64_t count, double* out) {
  if (!h || !out) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.debugPeekSolverScratch(offset, count, out);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_get_prior(svin_ba* h, double* H, double* b0, double* J, double* e0, uint64_t* ids, int32_t* ord,
                      int32_t* mdim, int32_t* nb, int cap_m) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.getPrior(H, b0, J, e0, ids, ord, mdim, nb, cap_m);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_get_marg_pre(svin_ba* h, int32_t* m, int32_t* n_landmarks, double* U, double* ba, double* W, double* V, double* bb,
                         int32_t* marg_rows, int cap_m, int cap_landmarks) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  const auto& q = h->w.margPre();
  if (m) *m = q.m;
  if (n_landmarks) *n_landmarks = q.Lm;
  if (q.m > cap_m || q.Lm > cap_landmarks) return 0;
  if (U) std::memcpy(U, q.U.data(), sizeof(double) * q.U.size());
  if (ba) std::memcpy(ba, q.ba.data(), sizeof(double) * q.m);
  if (W) std::memcpy(W, q.W.data(), sizeof(double) * q.W.size());
  if (V) std::memcpy(V, q.V.data(), sizeof(double) * q.V.size());
  if (bb) std::memcpy(bb, q.bb.data(), sizeof(double) * q.bb.size());
  if (marg_rows) for (size_t i = 0; i < q.margRows.size(); ++i) marg_rows[i] = q.margRows[i];
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_get_marg_pre_blocks(svin_ba* h, uint64_t* dense_ids, int32_t* dense_ord, int32_t* dense_mdim, int cap_dense, uint64_t* landmark_ids,
                                int cap_landmarks) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  const auto& q = h->w.margPre();
  for (size_t i = 0; i < q.denseIds.size() && (int)i < cap_dense; ++i) {
    if (dense_ids) dense_ids[i] = q.denseIds[i];
    if (dense_ord) dense_ord[i] = q.denseOrd[i];
    if (dense_mdim) dense_mdim[i] = q.denseMdim[i];
  }
  for (size_t i = 0; i < q.lmIds.size() && (int)i < cap_landmarks; ++i)
    if (landmark_ids) landmark_ids[i] = q.lmIds[i];
  return (int)q.denseIds.size();
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_describe_block(svin_ba* h, uint64_t id, uint64_t* frame, int32_t* kind, int32_t* index) try {
  if (!h) return SVIN_ERR_INVALID_ARG;
  return h->w.describeBlock(id, frame, kind, index);
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_bench_allreduce(svin_ba* h, uint64_t n_doubles, int iters, double* mean_us) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.benchAllReduce((size_t)n_doubles, iters, mean_us);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_bench_jacobian_eval(svin_ba* h, int copies, int iters, double* mean_ms, double* bytes) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.benchJacobianEval(copies, iters, mean_ms, bytes);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_residual_info(svin_ba* h, int n, const uint64_t* residual_ids, int32_t* kind, int32_t* residual_dim, int32_t* n_blocks,
                          int32_t* block_dims) {
  if (!h || n < 0 || (n > 0 && !residual_ids)) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.residualInfo(n, residual_ids, kind, residual_dim, n_blocks, block_dims);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_map_add_parameter_block(svin_ba* h, uint64_t id, int type, const double* values) {
  if (!h || !values) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.mapAddParameterBlock(id, type, values);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_set_parameter_block(svin_ba* h, uint64_t id, const double* values) {
  if (!h || !values) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.mapSetParameterBlock(id, values);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_map_remove_parameter_block(svin_ba* h, uint64_t id) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.mapRemoveParameterBlock(id);
  GUARD_END(SVIN_ERR_DEVICE)
}
uint64_t svin_ba_map_add_pose_error(svin_ba* h, uint64_t block, const double meas[7], const double information[36]) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddPoseError(block, meas, information);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_speed_and_bias_error(svin_ba* h, uint64_t block, const double meas[9], const double information[81]) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddSpeedAndBiasError(block, meas, information);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_relative_pose_error(svin_ba* h, uint64_t block0, uint64_t block1, const double information[36]) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddRelativePoseError(block0, block1, information);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_imu_error(svin_ba* h, const uint64_t blocks[4], const svin_imu_sample* imu, int n_imu, const svin_imu_params* p,
                                   uint32_t t0_sec, uint32_t t0_nsec, uint32_t t1_sec, uint32_t t1_nsec) {
  if (!h || !blocks || !imu || !p || n_imu < 2) return 0;
  GUARD_BEGIN
  std::vector<uint32_t> t;
  std::vector<double> m;
  splitSamples(imu, n_imu, t, m);
  TimeStamp a, b;
  a.sec = t0_sec; a.nsec = t0_nsec; b.sec = t1_sec; b.nsec = t1_nsec;
  return h->w.mapAddImuError(blocks, t.data(), m.data(), n_imu, toParams(p), a, b);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_sonar_error(svin_ba* h, uint64_t pose_block, double range, double heading, double information,
                                     const double* patch_xyz, int n_patch) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddSonarError(pose_block, range, heading, information, patch_xyz, n_patch);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_depth_error(svin_ba* h, uint64_t pose_block, double depth, double information, double first_depth) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddDepthError(pose_block, depth, information, first_depth);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_host_residual(svin_ba* h, const uint64_t* block_ids, int n_blocks, int residual_dim, svin_cost_function fn, void* user) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddHostResidual(block_ids, n_blocks, residual_dim, fn, user);
  GUARD_END(0)
}
uint64_t svin_ba_map_add_reprojection_error(svin_ba* h, uint64_t pose_block, uint64_t landmark, uint64_t extrinsics_block, uint64_t cam,
                                            const double uv[2], const double information[4]) {
  if (!h) return 0;
  GUARD_BEGIN return h->w.mapAddReprojectionError(pose_block, landmark, extrinsics_block, cam, uv, information);
  GUARD_END(0)
}
int svin_ba_map_remove_residual_block(svin_ba* h, uint64_t rid) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.mapRemoveResidualBlock(rid);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_set_pack_mode(svin_ba* h, int mode) try {
  if (!h || mode < 0 || mode > 1) return SVIN_ERR_INVALID_ARG;
  h->w.setPackMode(mode);
  return 1;
} CATCH_ALL(SVIN_ERR_DEVICE)
int svin_ba_debug_csr(svin_ba* h, int32_t* n_landmarks, int32_t* n_observations, int32_t* lm_ptr, int32_t* obs_lm, uint32_t* obs_idx,
                      double* uv, double* w, double* lm, int32_t* obs_order, int32_t* resident) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.debugCsr(n_landmarks, n_observations, lm_ptr, obs_lm, obs_idx, uv, w, lm, obs_order, resident);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_bench_jacobian_eval_b2b(svin_ba* h, int copies, int iters, double* mean_ms, double* b2b_ms, double* bytes) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.benchJacobianEval(copies, iters, mean_ms, bytes, b2b_ms);
  GUARD_END(SVIN_ERR_DEVICE)
}
int svin_ba_bench_kernel_times(svin_ba* h, int iters, double* e, double* b, double* s) {
  if (!h) return SVIN_ERR_INVALID_ARG;
  GUARD_BEGIN return h->w.benchKernelTimes(iters, e, b, s);
  GUARD_END(SVIN_ERR_DEVICE)
}

}  // extern "C"
