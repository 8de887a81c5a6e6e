// ORACLE -- test infrastructure only (never linked into the product library).
// Restatement of okvis_ceres/src/Estimator.cpp (line numbers cited inline).
#include "orc_estimator.hpp"

namespace orc {

int Estimator::addCamera(const ExtrinsicsEstimationParameters& p, const Camera& geometry) {
  extrinsicsVec_.push_back(p);
  cameras_.push_back(geometry);
  return (int)extrinsicsVec_.size() - 1;
}
int Estimator::addImu(const ImuParameters& p) {
  if (imuVec_.size() > 1) return -1;
  imuVec_.push_back(p);
  return (int)imuVec_.size() - 1;
}

// Estimator.cpp:848-873
bool Estimator::initPoseFromImu(const std::vector<ImuSample>& imu, Transformation& T_WS) {
  T_WS.setIdentity();
  if (imu.empty()) return false;
  double acc_B[3] = {0, 0, 0};
  for (const auto& s : imu) { acc_B[0] += s.acc[0]; acc_B[1] += s.acc[1]; acc_B[2] += s.acc[2]; }
  const double n = static_cast<double>(imu.size());
  acc_B[0] /= n; acc_B[1] /= n; acc_B[2] /= n;
  const double an = norm3(acc_B);
  const double e_acc[3] = {acc_B[0] / an, acc_B[1] / an, acc_B[2] / an};
  // ez_W x e_acc, normalised
  double c[3] = {0.0 * e_acc[2] - 1.0 * e_acc[1], 1.0 * e_acc[0] - 0.0 * e_acc[2], 0.0};
  const double cn = norm3(c);
  if (cn > 0) { c[0] /= cn; c[1] /= cn; c[2] /= cn; }
  const double angle = std::acos(e_acc[2]);
  double inc[6] = {0, 0, 0, -c[0] * angle, -c[1] * angle, -c[2] * angle};
  T_WS.oplus(inc);
  return true;
}

// Estimator.cpp:98-411
bool Estimator::addStates(uint64_t frameId, Time stamp, size_t numKeypoints, const std::vector<Transformation>& T_SC,
                          const std::vector<ImuSample>& imu, bool asKeyframe,
                          const std::vector<SonarMeasurement>& sonar, const std::vector<double>& depth,
                          double firstDepth) {
  Transformation T_WS;
  double speedAndBias[9];
  if (statesMap_.empty()) {
    if (!initPoseFromImu(imu, T_WS)) return false;  // :110-113
    if (!(numKeypoints > 10)) return false;         // :116-122
    for (double& v : speedAndBias) v = 0;
    for (int k = 0; k < 3; ++k) speedAndBias[6 + k] = imuVec_.at(0).a0[k];
  } else {
    const States& last = statesMap_.rbegin()->second;
    T_WS = Transformation(map_->param(last.T_WS.id).x);
    std::memcpy(speedAndBias, map_->param(last.speedAndBias.at(0).id).x, sizeof(speedAndBias));
    const int used = ImuError::propagation(imu, imuVec_.at(0), T_WS, speedAndBias, last.timestamp, stamp, nullptr,
                                           nullptr);
    if (used < 1) return false;  // :159-162
  }
  States states;
  states.isKeyframe = asKeyframe;
  states.id = frameId;
  states.timestamp = stamp;
  if (statesMap_.count(frameId)) return false;
  states.T_WS.exists = true;
  states.T_WS.id = frameId;
  if (statesMap_.empty()) referencePoseId_ = frameId;
  if (!map_->addParameterBlock(frameId, BLOCK_POSE, T_WS.p)) return false;

  const bool first = statesMap_.empty();
  const States* prev = first ? nullptr : &statesMap_.rbegin()->second;
  // cameras (:203-229)
  for (size_t i = 0; i < extrinsicsVec_.size(); ++i) {
    StateInfo info;
    info.exists = true;
    if ((extrinsicsVec_[i].sigma_c_relative_translation < 1e-12 ||
         extrinsicsVec_[i].sigma_c_relative_orientation < 1e-12) && !first) {
      info.id = prev->T_SC.at(i).id;
    } else {
      const uint64_t id = newId();
      if (!map_->addParameterBlock(id, BLOCK_POSE, T_SC.at(i).p)) return false;
      info.id = id;
    }
    states.T_SC.push_back(info);
  }
  // imu (:232-246)
  for (size_t i = 0; i < imuVec_.size(); ++i) {
    StateInfo info;
    info.exists = true;
    info.id = newId();
    if (!map_->addParameterBlock(info.id, BLOCK_SPEEDBIAS, speedAndBias)) return false;
    states.speedAndBias.push_back(info);
  }
  // depth (:248-262)
  if (!depth.empty()) {
    double mean_depth = 0.0;
    for (double dm : depth) mean_depth += dm;
    mean_depth = mean_depth / depth.size();
    const double information_depth = 5.0;
    map_->addResidualBlock(std::make_shared<DepthError>(mean_depth, information_depth, firstDepth), LOSS_NONE,
                           {frameId});
  }
  // sonar (:265-316)
  if (!sonar.empty()) {
    const double range = sonar.back().range, heading = sonar.back().heading;
    const Transformation T_WSo = T_WS * T_SSo_;
    const double rp[3] = {range * std::cos(heading), range * std::sin(heading), 0.0};
    const double qid[4] = {0, 0, 0, 1};
    const Transformation T_WSo_point = T_WSo * Transformation(rp, qid);
    const double* sl = T_WSo_point.r();
    std::vector<double> subset;
    double vl[3] = {0, 0, 0};  // NOTE: reference leaves this uninitialised when |w|<=1e-8; value carries over
    for (auto rit = landmarksMap_.rbegin(); rit != landmarksMap_.rend(); ++rit) {
      const double* pt = rit->second.point;
      if (std::fabs(pt[3]) > 1.0e-8) { vl[0] = pt[0] / pt[3]; vl[1] = pt[1] / pt[3]; vl[2] = pt[2] / pt[3]; }
      if (std::fabs(sl[0] - vl[0]) < 0.1 && std::fabs(sl[1] - vl[1]) < 0.1 && std::fabs(sl[2] - vl[2]) < 0.1) {
        subset.push_back(vl[0]); subset.push_back(vl[1]); subset.push_back(vl[2]);
      }
    }
    if (!subset.empty()) {
      const double information_sonar = 1.0;
      map_->addResidualBlock(std::make_shared<SonarError>(T_SSo_, range, heading, information_sonar, subset), LOSS_NONE,
                             {frameId});
    }
  }
  if (first) {
    // pose prior (:319-327): singular information diag(1e8,1e8,1e8,0,0,1e8)
    double information[36] = {0};
    information[5 * 7] = 1.0e8; information[0] = 1.0e8; information[7] = 1.0e8; information[14] = 1.0e8;
    map_->addResidualBlock(std::make_shared<PoseError>(T_WS, information), LOSS_NONE, {frameId});
    for (size_t i = 0; i < extrinsicsVec_.size(); ++i) {  // :330-350
      const double ts = extrinsicsVec_[i].sigma_absolute_translation, tv = ts * ts;
      const double rs = extrinsicsVec_[i].sigma_absolute_orientation, rv = rs * rs;
      if (tv > 1.0e-16 && rv > 1.0e-16) {
        map_->addResidualBlock(std::make_shared<PoseError>(T_SC.at(i), tv, rv), LOSS_NONE, {states.T_SC[i].id});
      } else {
        map_->setParameterBlockConstant(states.T_SC[i].id);
      }
    }
    for (size_t i = 0; i < imuVec_.size(); ++i) {  // :351-364
      const double sbg = imuVec_.at(0).sigma_bg, sba = imuVec_.at(0).sigma_ba;
      map_->addResidualBlock(std::make_shared<SpeedAndBiasError>(speedAndBias, 1.0, sbg * sbg, sba * sba), LOSS_NONE,
                             {states.speedAndBias[i].id});
    }
  } else {
    for (size_t i = 0; i < imuVec_.size(); ++i) {  // :368-382
      map_->addResidualBlock(std::make_shared<ImuError>(imu, imuVec_[i], prev->timestamp, states.timestamp), LOSS_NONE,
                             {prev->id, prev->speedAndBias.at(i).id, states.id, states.speedAndBias[i].id});
    }
    for (size_t i = 0; i < extrinsicsVec_.size(); ++i) {  // :385-404
      if (prev->T_SC.at(i).id != states.T_SC[i].id) {
        const double dt = dtSec(states.timestamp, prev->timestamp);
        const double tsc = extrinsicsVec_[i].sigma_c_relative_translation, tv = tsc * tsc * dt;
        const double rsc = extrinsicsVec_[i].sigma_c_relative_orientation, rv = rsc * rsc * dt;
        map_->addResidualBlock(std::make_shared<RelativePoseError>(tv, rv), LOSS_NONE,
                               {prev->T_SC.at(i).id, states.T_SC[i].id});
      }
    }
  }
  statesMap_[frameId] = states;
  return true;
}

bool Estimator::addLandmark(uint64_t id, const double* hp) {  // :414-429
  if (!map_->addParameterBlock(id, BLOCK_HPOINT, hp)) return false;
  MapPoint mp;
  mp.id = id;
  std::memcpy(mp.point, hp, sizeof(mp.point));
  mp.quality = 0.0;
  double dist = std::numeric_limits<double>::max();
  if (std::fabs(hp[3]) > 1.0e-8) {
    const double e[3] = {hp[0] / hp[3], hp[1] / hp[3], hp[2] / hp[3]};
    dist = norm3(e);
  }
  mp.distance = dist;
  landmarksMap_[id] = mp;
  return true;
}

uint64_t Estimator::addObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx,
                                   const double* uv, double size) {  // implementation/Estimator.hpp:47-87
  auto kid = std::make_tuple(poseId, camIdx, keypointIdx);
  // the reference asserts (debug only) that landmark and pose exist; a missing one is reported as 0 here
  if (!landmarksMap_.count(landmarkId) || !statesMap_.count(poseId) || camIdx >= cameras_.size()) return 0;
  MapPoint& mp = landmarksMap_.at(landmarkId);
  if (mp.observations.count(kid)) return 0;
  double information[4] = {1, 0, 0, 1};
  const double f = 64.0 / (size * size);
  information[0] *= f; information[3] *= f;
  auto err = std::make_shared<ReprojectionError>(cameras_.at(camIdx), uv, information);
  err->camIdx = camIdx;
  const uint64_t rid =
      map_->addResidualBlock(err, LOSS_CAUCHY, {poseId, landmarkId, statesMap_.at(poseId).T_SC.at(camIdx).id});
  mp.observations[kid] = rid;
  return rid;
}

bool Estimator::removeObservation(uint64_t residualId) {  // :432-449
  if (!map_->residualExists(residualId)) return false;
  const uint64_t landmarkId = map_->residual(residualId).params.at(1);
  MapPoint& mp = landmarksMap_.at(landmarkId);
  for (auto it = mp.observations.begin(); it != mp.observations.end();) {
    if (it->second == residualId) it = mp.observations.erase(it);
    else ++it;
  }
  map_->removeResidualBlock(residualId);
  return true;
}
bool Estimator::removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) {
  auto kid = std::make_tuple(poseId, camIdx, keypointIdx);
  MapPoint& mp = landmarksMap_.at(landmarkId);
  auto it = mp.observations.find(kid);
  if (it == mp.observations.end()) return false;
  map_->removeResidualBlock(it->second);
  mp.observations.erase(it);
  return true;
}

namespace {
template <class T>
bool vectorContains(const std::vector<T>& v, const T& q) {
  for (const T& e : v) if (e == q) return true;
  return false;
}
}  // namespace

// Estimator.cpp:495-814
bool Estimator::applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, std::vector<MapPoint>& removed) {
  auto rit = statesMap_.rbegin();
  for (size_t k = 0; k < numImuFrames; k++) {
    rit++;
    if (rit == statesMap_.rend()) return true;
  }
  if (margPtr_ && margResidualId_) {  // :509-514
    const bool ok = map_->removeResidualBlock(margResidualId_);
    margResidualId_ = 0;
    if (!ok) return false;
  }
  std::vector<uint64_t> toMarginalize;
  if (!margPtr_) margPtr_.reset(new MarginalizationError(map_.get()));

  std::vector<uint64_t> removeFrames, removeAllButPose, allLinearizedFrames;
  size_t countedKeyframes = 0;
  while (rit != statesMap_.rend()) {  // :529-538
    if (!rit->second.isKeyframe || countedKeyframes >= numKeyframes) removeFrames.push_back(rit->second.id);
    else countedKeyframes++;
    removeAllButPose.push_back(rit->second.id);
    allLinearizedFrames.push_back(rit->second.id);
    ++rit;
  }
  auto isReproj = [&](uint64_t rid) { return map_->residual(rid).err->kind() == ErrorTerm::REPROJECTION; };

  // marginalize everything but pose (:541-605); the only non-pose, non-extrinsics state is speed/bias
  for (uint64_t fid : removeAllButPose) {
    auto it = statesMap_.find(fid);
    for (size_t j = 0; j < it->second.speedAndBias.size(); ++j) {
      StateInfo& si = it->second.speedAndBias[j];
      if (!si.exists) continue;
      if (map_->param(si.id).fixed) continue;
      auto checkit = it;
      checkit++;
      if (checkit != statesMap_.end() && checkit->second.speedAndBias.size() > j &&
          checkit->second.speedAndBias[j].exists && checkit->second.speedAndBias[j].id == si.id) continue;
      si.exists = false;
      toMarginalize.push_back(si.id);
      for (uint64_t rid : map_->residuals(si.id))
        if (map_->residualExists(rid) && !isReproj(rid)) margPtr_->addResidualBlock(rid);
    }
  }
  // marginalize ONLY pose now (:607-770)
  bool reDoFixation = false;
  for (uint64_t fid : removeFrames) {
    auto it = statesMap_.find(fid);
    it->second.T_WS.exists = false;
    toMarginalize.push_back(it->second.T_WS.id);
    for (uint64_t rid : map_->residuals(it->second.T_WS.id)) {  // :622-635
      if (!map_->residualExists(rid)) continue;
      if (map_->residual(rid).err->kind() == ErrorTerm::POSE) {
        map_->removeResidualBlock(rid);
        reDoFixation = true;
        continue;
      }
      if (!isReproj(rid)) margPtr_->addResidualBlock(rid);
    }
    for (size_t j = 0; j < it->second.T_SC.size(); ++j) {  // :638-664
      StateInfo& si = it->second.T_SC[j];
      if (!si.exists) continue;
      if (map_->param(si.id).fixed) continue;
      auto checkit = it;
      checkit++;
      if (checkit != statesMap_.end() && checkit->second.T_SC[j].exists && checkit->second.T_SC[j].id == si.id) continue;
      si.exists = false;
      toMarginalize.push_back(si.id);
      for (uint64_t rid : map_->residuals(si.id))
        if (map_->residualExists(rid) && !isReproj(rid)) margPtr_->addResidualBlock(rid);
    }
    // observations (:667-766)
    const uint64_t currentKfId = allLinearizedFrames.at(0);
    for (auto pit = landmarksMap_.begin(); pit != landmarksMap_.end();) {
      std::vector<uint64_t> residuals = map_->residuals(pit->first);
      bool skipLandmark = true, hasNewObservations = false, justDelete = false, marginalize = true,
           errorTermAdded = false;
      size_t obsCount = 0;
      for (uint64_t rid : residuals) {
        if (!isReproj(rid)) continue;
        const uint64_t poseId = map_->residual(rid).params.at(0);
        if (vectorContains(removeFrames, poseId)) skipLandmark = false;
        if (poseId >= currentKfId) { marginalize = false; hasNewObservations = true; }
        if (vectorContains(allLinearizedFrames, poseId)) obsCount++;
      }
      if (residuals.empty()) {
        map_->removeParameterBlock(pit->first);
        removed.push_back(pit->second);
        pit = landmarksMap_.erase(pit);
        continue;
      }
      if (skipLandmark) { pit++; continue; }
      for (size_t r = 0; r < residuals.size(); ++r) {
        const uint64_t rid = residuals[r];
        if (!isReproj(rid)) continue;
        const uint64_t poseId = map_->residual(rid).params.at(0);
        if ((vectorContains(removeFrames, poseId) && hasNewObservations) ||
            (!vectorContains(allLinearizedFrames, poseId) && marginalize)) {
          removeObservation(rid);
          residuals.erase(residuals.begin() + r);
          r--;
        } else if (marginalize && vectorContains(allLinearizedFrames, poseId)) {
          if (obsCount < 2) {
            removeObservation(rid);
            residuals.erase(residuals.begin() + r);
            r--;
          } else {
            errorTermAdded = true;
            margPtr_->addResidualBlock(rid, false);
          }
        }
        if (residuals.size() == 0) { justDelete = true; marginalize = false; }
      }
      if (justDelete) {
        map_->removeParameterBlock(pit->first);
        removed.push_back(pit->second);
        pit = landmarksMap_.erase(pit);
        continue;
      }
      if (marginalize && errorTermAdded) {
        toMarginalize.push_back(pit->first);
        removed.push_back(pit->second);
        pit = landmarksMap_.erase(pit);
        continue;
      }
      pit++;
    }
    statesMap_.erase(it->second.id);
  }
  if (!toMarginalize.empty()) {  // :773-785
    margPtr_->marginalizeOut(toMarginalize);
    margPtr_->updateErrorComputation();
  }
  if (margPtr_->residualDim() == 0) margPtr_.reset();  // :788-790
  if (margPtr_) {
    margResidualId_ = map_->addResidualBlock(margPtr_, LOSS_NONE, margPtr_->parameterBlockIds());
    if (!margResidualId_) return false;
  }
  if (reDoFixation) {  // :799-811
    const uint64_t firstId = statesMap_.begin()->first;
    const Transformation T_WS_0(map_->param(firstId).x);
    double information[36] = {0};
    information[5 * 7] = 1.0e14; information[0] = 1.0e14; information[7] = 1.0e14; information[14] = 1.0e14;
    map_->addResidualBlock(std::make_shared<PoseError>(T_WS_0, information), LOSS_NONE, {firstId});
  }
  return true;
}

// Estimator.cpp:876-929
void Estimator::optimize(size_t numIter, size_t numThreads, bool verbose) {
  map_->options.max_num_iterations = (int)numIter;
  map_->options.num_threads = (int)std::max<size_t>(1, numThreads);  // :889 options.num_threads = numThreads
  map_->options.verbose = verbose;
  map_->solve();
  for (auto& kv : landmarksMap_) {
    double H[9], ev[3], U[9];
    map_->getLhs(kv.first, H);
    sym_eig(H, 3, ev, U);
    const double smallest = ev[0], largest = ev[2];
    if (smallest < 1.0e-12) kv.second.quality = 0.0;
    else kv.second.quality = std::sqrt(smallest) / std::sqrt(largest);
    std::memcpy(kv.second.point, map_->param(kv.first).x, sizeof(double) * 4);
  }
}

bool Estimator::setOptimizationTimeLimit(double timeLimit, int minIterations) {  // :932-951
  if (hasCallback_) {
    if (timeLimit < 0.0) { map_->options.min_iterations = map_->options.max_num_iterations; return true; }
    map_->options.time_limit = timeLimit;
    map_->options.min_iterations = minIterations;
    return true;
  } else if (timeLimit >= 0.0) {
    hasCallback_ = true;
    map_->options.time_limit = timeLimit;
    map_->options.min_iterations = minIterations;
    return true;
  }
  return true;
}

bool Estimator::get_T_WS(uint64_t poseId, double* T) const {
  auto it = statesMap_.find(poseId);
  if (it == statesMap_.end() || !it->second.T_WS.exists) return false;
  std::memcpy(T, map_->param(it->second.T_WS.id).x, 7 * sizeof(double));
  return true;
}
bool Estimator::getSpeedAndBias(uint64_t poseId, size_t imuIdx, double* sb) const {
  auto it = statesMap_.find(poseId);
  if (it == statesMap_.end() || imuIdx >= it->second.speedAndBias.size() || !it->second.speedAndBias[imuIdx].exists)
    return false;
  std::memcpy(sb, map_->param(it->second.speedAndBias[imuIdx].id).x, 9 * sizeof(double));
  return true;
}
bool Estimator::getCameraSensorStates(uint64_t poseId, size_t camIdx, double* T) const {
  auto it = statesMap_.find(poseId);
  if (it == statesMap_.end() || camIdx >= it->second.T_SC.size() || !it->second.T_SC[camIdx].exists) return false;
  std::memcpy(T, map_->param(it->second.T_SC[camIdx].id).x, 7 * sizeof(double));
  return true;
}
bool Estimator::getLandmark(uint64_t id, MapPoint& mp) const {
  auto it = landmarksMap_.find(id);
  if (it == landmarksMap_.end()) return false;
  mp = it->second;
  return true;
}
bool Estimator::set_T_WS(uint64_t poseId, const double* T) {
  auto it = statesMap_.find(poseId);
  if (it == statesMap_.end() || !it->second.T_WS.exists) return false;
  const Transformation t(T);  // PoseParameterBlock::setEstimate stores normalised parameters
  std::memcpy(map_->param(it->second.T_WS.id).x, t.p, 7 * sizeof(double));
  return true;
}
bool Estimator::setCameraSensorStates(uint64_t poseId, size_t camIdx, const double* T) {  // Estimator.cpp:1102-1106
  auto it = statesMap_.find(poseId);
  if (it == statesMap_.end() || camIdx >= it->second.T_SC.size() || !it->second.T_SC[camIdx].exists) return false;
  const Transformation t(T);
  std::memcpy(map_->param(it->second.T_SC[camIdx].id).x, t.p, 7 * sizeof(double));
  return true;
}
bool Estimator::setSpeedAndBias(uint64_t poseId, size_t imuIdx, const double* sb) {
  auto it = statesMap_.find(poseId);
  if (it == statesMap_.end() || imuIdx >= it->second.speedAndBias.size() || !it->second.speedAndBias[imuIdx].exists)
    return false;
  std::memcpy(map_->param(it->second.speedAndBias[imuIdx].id).x, sb, 9 * sizeof(double));
  return true;
}
bool Estimator::setLandmark(uint64_t id, const double* hp) {  // Estimator.cpp setLandmark
  auto it = landmarksMap_.find(id);
  if (it == landmarksMap_.end()) return false;
  std::memcpy(map_->param(id).x, hp, 4 * sizeof(double));
  std::memcpy(it->second.point, hp, 4 * sizeof(double));
  return true;
}
uint64_t Estimator::currentKeyframeId() const {
  for (auto rit = statesMap_.rbegin(); rit != statesMap_.rend(); ++rit)
    if (rit->second.isKeyframe) return rit->first;
  return 0;
}
uint64_t Estimator::frameIdByAge(size_t age) const {
  auto rit = statesMap_.rbegin();
  for (size_t i = 0; i < age; ++i) { ++rit; if (rit == statesMap_.rend()) return 0; }
  return rit->first;
}
bool Estimator::isInImuWindow(uint64_t id) const {
  const States& s = statesMap_.at(id);
  if (s.speedAndBias.empty()) return false;
  return s.speedAndBias[0].exists;
}

}  // namespace orc
