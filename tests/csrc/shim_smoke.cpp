// Drives okvis::Estimator -- the header-only shim in integration/okvis/Estimator.hpp -- the way ThreadedKFVio and the
// frontend do (addCamera / addImu / addStates with a MultiFrame / addLandmark / addObservation<GEOMETRY> / optimize /
// applyMarginalizationStrategy / getters), on a window dumped by tests/test_gpu_shim.py, and prints what it reads back.
// Built against the stand-in okvis headers in tests/csrc/mock_okvis (no Eigen / okvis in this image).
#include <okvis/Estimator.hpp>

#include <cstdio>
#include <fstream>
#include <iostream>

struct PinholeRadTan {};   // plays the GEOMETRY_TYPE template argument of addObservation

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  int nCam;
  in >> nCam;
  std::vector<std::shared_ptr<const okvis::cameras::CameraBase>> geo;
  std::vector<std::shared_ptr<const okvis::kinematics::Transformation>> T_SC;
  okvis::Estimator est(0);
  for (int c = 0; c < nCam; ++c) {
    std::string dist;
    int w, h, nIntr;
    in >> dist >> w >> h >> nIntr;
    std::vector<double> intr(nIntr);
    for (double& v : intr) in >> v;
    double T[7], s[4];
    for (double& v : T) in >> v;
    for (double& v : s) in >> v;
    geo.push_back(std::make_shared<okvis::cameras::CameraBase>(w, h, dist, intr));
    T_SC.push_back(std::make_shared<okvis::kinematics::Transformation>(Eigen::Vector3d(T[0], T[1], T[2]), Eigen::Quaterniond(T[6], T[3], T[4], T[5])));
    okvis::ExtrinsicsEstimationParameters e;
    e.sigma_absolute_translation = s[0]; e.sigma_absolute_orientation = s[1];
    e.sigma_c_relative_translation = s[2]; e.sigma_c_relative_orientation = s[3];
    if (est.addCamera(e) != c) return 3;
  }
  okvis::ImuParameters ip;
  in >> ip.a_max >> ip.g_max >> ip.sigma_g_c >> ip.sigma_a_c >> ip.sigma_bg >> ip.sigma_ba >> ip.sigma_gw_c >> ip.sigma_aw_c >> ip.tau >> ip.g;
  in >> ip.a0[0] >> ip.a0[1] >> ip.a0[2];
  if (est.addImu(ip) != 0) return 4;
  int L;
  in >> L;
  std::vector<uint64_t> lmIds(L);
  for (int l = 0; l < L; ++l) {
    double hp[4];
    for (double& v : hp) in >> v;
    lmIds[l] = okvis::IdProvider::instance().newId();
    if (!est.addLandmark(lmIds[l], Eigen::Vector4d(hp[0], hp[1], hp[2], hp[3]))) return 5;
  }
  int P, numKf, numImu, iters;
  in >> P >> numKf >> numImu >> iters;
  std::vector<uint64_t> frameIds;
  for (int k = 0; k < P; ++k) {
    auto mf = std::make_shared<okvis::MultiFrame>();
    int keyframe, nImu, nObs;
    in >> mf->stamp_.sec >> mf->stamp_.nsec >> keyframe >> nImu;
    okvis::ImuMeasurementDeque imu;
    for (int i = 0; i < nImu; ++i) {
      okvis::ImuMeasurement m;
      in >> m.timeStamp.sec >> m.timeStamp.nsec;
      for (int a = 0; a < 3; ++a) in >> m.measurement.gyroscopes[a];
      for (int a = 0; a < 3; ++a) in >> m.measurement.accelerometers[a];
      imu.push_back(m);
    }
    double Tinit[7], sbInit[9];
    for (double& v : Tinit) in >> v;
    for (double& v : sbInit) in >> v;
    in >> nObs;
    struct Obs { int lm, cam; double u, v, size; };
    std::vector<Obs> obs(nObs);
    mf->kps_.assign(nCam, {});
    for (Obs& o : obs) {
      in >> o.lm >> o.cam >> o.u >> o.v >> o.size;
      mf->kps_[o.cam].push_back({o.u, o.v, o.size});
    }
    // pad with unmatched keypoints so that numKeypoints() > 10 on the first frame (Estimator.cpp:116-122)
    while (mf->numKeypoints() < 400) mf->kps_[0].push_back({0.0, 0.0, 8.0});
    mf->id_ = okvis::IdProvider::instance().newId();
    mf->T_SC_ = T_SC;
    mf->geo_ = geo;
    if (!est.addStates(mf, imu, keyframe != 0)) { std::printf("addStates failed at frame %d\n", k); return 6; }
    frameIds.push_back(mf->id());
    if (k > 0) est.set_T_WS(mf->id(), okvis::kinematics::Transformation(Eigen::Vector3d(Tinit[0], Tinit[1], Tinit[2]),
                                                                        Eigen::Quaterniond(Tinit[6], Tinit[3], Tinit[4], Tinit[5])));
    okvis::SpeedAndBias sb;
    for (int a = 0; a < 9; ++a) sb[a] = sbInit[a];
    est.setSpeedAndBias(mf->id(), 0, sb);
    std::vector<size_t> next(nCam, 0);
    for (const Obs& o : obs) {
      const size_t kp = next[o.cam]++;
      // NULL: the landmark has been marginalised in the meantime (the frontend checks isLandmarkAdded first, Frontend.cpp:928)
      if (est.addObservation<PinholeRadTan>(lmIds[o.lm], mf->id(), o.cam, kp) == nullptr && est.isLandmarkAdded(lmIds[o.lm])) return 7;
    }
    // a duplicate returns NULL (implementation/Estimator.hpp:55-57)
    for (const Obs& o : obs)
      if (est.isLandmarkAdded(lmIds[o.lm])) {
        if (est.addObservation<PinholeRadTan>(lmIds[o.lm], mf->id(), o.cam, 0) != nullptr && o.cam == obs[0].cam && &o == &obs[0]) return 8;
        break;
      }
    if (numKf > 0) {
      est.optimize(iters, 2, false);
      okvis::MapPointVector removed;
      if (!est.applyMarginalizationStrategy(numKf, numImu, removed)) return 9;
      std::printf("frame %d removed %zu stateCount %d frames %zu\n", k, removed.size(), est.stateCount_, est.numFrames());
    }
  }
  if (numKf == 0) est.optimize(iters, 2, false);
  std::printf("summary iterations %d final_cost %.17g termination %d\n", (int)est.map()->summary.iterations.size() - 1,
              est.map()->summary.final_cost, (int)est.map()->summary.termination_type);
  for (size_t age = 0; age < est.numFrames(); ++age) {
    const uint64_t id = est.frameIdByAge(age);
    okvis::kinematics::Transformation T;
    okvis::SpeedAndBias sb;
    est.get_T_WS(id, T);
    const bool hasSb = est.getSpeedAndBias(id, 0, sb);
    std::printf("pose %llu kf %d imu %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", (unsigned long long)id, (int)est.isKeyframe(id),
                (int)est.isInImuWindow(id), T.r()[0], T.r()[1], T.r()[2], T.q().x(), T.q().y(), T.q().z(), T.q().w());
    if (hasSb) std::printf(" sb %.17g %.17g %.17g", sb[0], sb[1], sb[2]);
    std::printf("\n");
  }
  okvis::PointMap lms;
  const size_t n = est.getLandmarks(lms);
  size_t nObsTotal = 0;
  for (auto& kv : lms) nObsTotal += kv.second.observations.size();
  std::printf("landmarks %zu observations %zu currentKeyframe %llu currentFrame %llu\n", n, nObsTotal,
              (unsigned long long)est.currentKeyframeId(), (unsigned long long)est.currentFrameId());
  int shown = 0;
  for (auto& kv : lms) {
    if (shown++ >= 5) break;
    std::printf("lm %llu %.17g %.17g %.17g %.17g q %.17g nobs %zu init %d\n", (unsigned long long)kv.first, kv.second.point[0], kv.second.point[1],
                kv.second.point[2], kv.second.point[3], kv.second.quality, kv.second.observations.size(), (int)est.isLandmarkInitialized(kv.first));
  }
  // Map view: the residuals of the newest pose and the blocks of the first of them
  auto res = est.map()->residuals(est.currentFrameId());
  std::printf("map residuals_of_current %zu first_params %zu exists %d\n", res.size(), res.empty() ? 0 : est.map()->parameters(res[0].residualBlockId).size(),
              (int)est.map()->parameterBlockExists(est.currentFrameId()));
  // snapshots of the window's blocks through the reference's accessors (Map.hpp:166-188)
  auto pb = est.map()->parameterBlockPtr(est.currentFrameId());
  auto all = est.map()->id2parameterBlockMap();
  size_t nPose = 0, nSb = 0, nLm = 0;
  for (auto& kv : all) {
    const std::string t = kv.second->typeInfo();
    nPose += t == "PoseParameterBlock";
    nSb += t == "SpeedAndBiasParameterBlock";
    nLm += t == "HomogeneousPointParameterBlock";
  }
  std::printf("blockptr %s dim %zu fixed %d x0 %.17g qw %.17g all %zu pose %zu sb %zu lm %zu missing %d\n", pb->typeInfo().c_str(), pb->dimension(), (int)pb->fixed(),
              pb->parameters()[0], pb->parameters()[6], all.size(), nPose, nSb, nLm, (int)(est.map()->parameterBlockPtr(999999999ULL) == nullptr));
  if (!res.empty()) {
    auto ei = res.back().errorInterfacePtr;
    auto ps = est.map()->parameters(res.back().residualBlockId);
    std::printf("errif %s dim %zu blocks %zu firstdim %zu snapshot_id %llu\n", ei->typeInfo().c_str(), ei->residualDim(), ei->parameterBlocks(), ei->parameterBlockDim(0),
                (unsigned long long)ps[0].second->id());
  }
  return 0;
}
