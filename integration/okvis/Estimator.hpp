// integration/okvis/Estimator.hpp -- okvis::Estimator on top of libsvin_ba.so (MI355X).
//
// Header-only replacement for okvis_ceres/include/okvis/Estimator.hpp:81-640 (+ src/Estimator.cpp,
// include/okvis/implementation/Estimator.hpp): the same class name, the same public member signatures, the same public
// data members (`stateCount_` :450 is read by Frontend.cpp:269, `imuIntegralsMap_` :404-408), so that okvis_frontend,
// okvis_multisensor_processing and the ROS nodes compile against it unchanged.  Every window operation forwards to the
// C ABI in <svin_ba.h>; the class itself keeps only what the reference keeps outside the optimisation graph:
// the MultiFrame pointers (:600) and the sensor parameter vectors (:605-617).
//
// Error convention: the ABI returns 1 / 0 / <0 and never throws; 0 becomes the reference's `false`,
// <0 becomes okvis::Estimator::Exception (where the reference hits OKVIS_THROW / OKVIS_ASSERT_TRUE).
// Threading: unchanged -- every mutator is called under ThreadedKFVio::estimator_mutex_ (ThreadedKFVio.cpp:737,:1083);
// getLandmark(s) additionally take statesMutex_ like the reference (Estimator.cpp:956,:974,:982).
#ifndef INTEGRATION_OKVIS_ESTIMATOR_HPP_
#define INTEGRATION_OKVIS_ESTIMATOR_HPP_

#include <svin_ba.h>

#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

#include <okvis/FrameTypedefs.hpp>
#include <okvis/IdProvider.hpp>
#include <okvis/Measurements.hpp>
#include <okvis/MultiFrame.hpp>
#include <okvis/Parameters.hpp>
#include <okvis/Variables.hpp>
#include <okvis/VioBackendInterface.hpp>
#include <okvis/assert_macros.hpp>
#include <okvis/ceres/Map.hpp>
#include <okvis/kinematics/Transformation.hpp>

namespace okvis {

class Estimator : public VioBackendInterface {
 public:
  OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error)

  /// Estimator::Estimator() (Estimator.cpp:67-73).  `device` = HIP device of this estimator (one process per GPU).
  explicit Estimator(int device = 0) : h_(svin_ba_create(device)), mapPtr_(new okvis::ceres::Map()) {
    if (!h_) OKVIS_THROW(Exception, std::string("svin_ba_create: ") + svin_ba_last_error());
    mapPtr_->attach(h_);
    // ONE id space: frames (FrameSynchronizer.cpp:97), landmarks (Frontend.cpp:599) and the estimator's own extrinsics /
    // speed-bias blocks (Estimator.cpp:217,234) all draw from the process-wide provider
    svin_ba_set_id_provider(h_, &Estimator::idTrampoline, nullptr);
  }
  /// Estimator(std::shared_ptr<Map>) (Estimator.cpp:57-64): the map becomes a view of this estimator's handle
  explicit Estimator(std::shared_ptr<okvis::ceres::Map> mapPtr) : Estimator(0) { setMap(mapPtr); }
  Estimator(const Estimator&) = delete;
  Estimator& operator=(const Estimator&) = delete;
  virtual ~Estimator() { svin_ba_destroy(h_); }

  /// @name Sensor configuration related (Estimator.cpp:77-96)
  ///@{
  int addCamera(const okvis::ExtrinsicsEstimationParameters& e) override {
    const double intr[4] = {0, 0, 0, 0};
    const double sig[4] = {e.sigma_absolute_translation, e.sigma_absolute_orientation, e.sigma_c_relative_translation,
                           e.sigma_c_relative_orientation};
    // the geometry arrives with the first multi-frame (implementation/Estimator.hpp:62-66): see registerGeometry()
    const int idx = svin_ba_add_camera(h_, SVIN_DIST_NONE, intr, nullptr, 0, 0, 0, sig);
    if (idx < 0) OKVIS_THROW(Exception, std::string("svin_ba_add_camera: ") + svin_ba_last_error());
    extrinsicsEstimationParametersVec_.push_back(e);
    geometryKnown_.push_back(false);
    return idx;
  }
  int addImu(const okvis::ImuParameters& p) override {
    svin_imu_params q;
    q.a_max = p.a_max; q.g_max = p.g_max; q.sigma_g_c = p.sigma_g_c; q.sigma_a_c = p.sigma_a_c;
    q.sigma_bg = p.sigma_bg; q.sigma_ba = p.sigma_ba; q.sigma_gw_c = p.sigma_gw_c; q.sigma_aw_c = p.sigma_aw_c;
    q.tau = p.tau; q.g = p.g;
    for (int k = 0; k < 3; ++k) q.a0[k] = p.a0[k];
    const int idx = svin_ba_add_imu(h_, &q);
    if (idx >= 0) imuParametersVec_.push_back(p);
    return idx;   // -1 for a second IMU, like the reference (:84-87)
  }
  /// declared by the reference (Estimator.hpp:115) but neither defined nor called there: sonarParameters_ keeps its
  /// default (identity T_SSo) in the reference.  Here the call works and hands T_SSo to the core.
  int addSonar(const okvis::SonarParameters& sonarParameters) {
    sonarParameters_ = sonarParameters;
    double T[7];
    toArray(sonarParameters.T_SSo, T);
    svin_ba_set_sonar_extrinsics(h_, T);
    return 0;
  }
  void clearCameras() override { extrinsicsEstimationParametersVec_.clear(); geometryKnown_.clear(); svin_ba_clear_cameras(h_); }
  void clearImus() override { imuParametersVec_.clear(); svin_ba_clear_imus(h_); }
  ///@}

  /// Estimator::addStates (Estimator.cpp:98-411)
  bool addStates(okvis::MultiFramePtr multiFrame, const okvis::ImuMeasurementDeque& imuMeasurements, bool asKeyframe,
                 const okvis::SonarMeasurementDeque& sonarMeasurements = {},
                 const okvis::DepthMeasurementDeque& depthMeasurements = {}, double firstDepth = 0.0) override {
    const size_t nCam = multiFrame->numFrames();
    for (size_t i = 0; i < nCam && i < geometryKnown_.size(); ++i)
      if (!geometryKnown_[i]) registerGeometry(*multiFrame, i);
    std::vector<svin_imu_sample> imu(imuMeasurements.size());
    size_t n = 0;
    for (const auto& m : imuMeasurements) {
      svin_imu_sample& s = imu[n++];
      s.sec = m.timeStamp.sec; s.nsec = m.timeStamp.nsec;
      for (int k = 0; k < 3; ++k) { s.gyr[k] = m.measurement.gyroscopes[k]; s.acc[k] = m.measurement.accelerometers[k]; }
    }
    std::vector<double> T_SC(7 * nCam), sonar, depth;
    for (size_t i = 0; i < nCam; ++i) toArray(*multiFrame->T_SC(i), &T_SC[7 * i]);
    for (const auto& m : sonarMeasurements) { sonar.push_back(m.measurement.range); sonar.push_back(m.measurement.heading); }
    for (const auto& m : depthMeasurements) depth.push_back(m.measurement.depth);
    const int r = svin_ba_add_states(h_, multiFrame->id(), multiFrame->timestamp().sec, multiFrame->timestamp().nsec,
                                     multiFrame->numKeypoints(), T_SC.data(), (int)nCam, imu.empty() ? nullptr : imu.data(),
                                     (int)imu.size(), asKeyframe ? 1 : 0, sonar.empty() ? nullptr : sonar.data(),
                                     (int)(sonar.size() / 2), depth.empty() ? nullptr : depth.data(), (int)depth.size(), firstDepth);
    stateCount_ = svin_ba_state_count(h_);   // :171 counts every call that got past the prediction
    if (r < 0) OKVIS_THROW(Exception, std::string("svin_ba_add_states: ") + svin_ba_last_error());
    double adi[3], ai[3], dt;
    if (svin_ba_get_imu_preintegral(h_, multiFrame->id(), adi, ai, &dt) == 1)   // :165 setImuPreIntegral
      imuIntegralsMap_.insert(std::make_pair(multiFrame->id(), imu_integrals(Eigen::Vector3d(adi[0], adi[1], adi[2]),
                                                                              Eigen::Vector3d(ai[0], ai[1], ai[2]), dt)));
    if (r != 1) return false;
    multiFramePtrMap_.insert(std::make_pair(multiFrame->id(), multiFrame));   // :197
    return true;
  }

  /// Estimator::printStates (Estimator.cpp:817-845)
  void printStates(uint64_t poseId, std::ostream& buffer) const {
    double T[7], sb[9];
    buffer << "GLOBAL: ";
    if (svin_ba_get_T_WS(h_, poseId, T) == 1) buffer << "id=" << poseId << " T_WS";
    buffer << ", SENSOR: ";
    for (size_t c = 0; c < extrinsicsEstimationParametersVec_.size(); ++c)
      if (svin_ba_get_camera_sensor_states(h_, poseId, c, T) == 1) buffer << "cam" << c << ":T_SC ";
    if (svin_ba_get_speed_and_bias(h_, poseId, 0, sb) == 1) buffer << "imu0:speedAndBias";
    buffer << std::endl;
  }

  bool addLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) override {   // :414-429
    const double hp[4] = {landmark[0], landmark[1], landmark[2], landmark[3]};
    return check(svin_ba_add_landmark(h_, landmarkId, hp), "svin_ba_add_landmark");
  }

  /// Estimator::addObservation<GEOMETRY_TYPE> (implementation/Estimator.hpp:47-87): measurement and size from the
  /// multi-frame, information 64 / size^2, Cauchy(1); returns NULL for a duplicate
  template <class GEOMETRY_TYPE>
  ::ceres::ResidualBlockId addObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) {
    auto it = multiFramePtrMap_.find(poseId);
    OKVIS_ASSERT_TRUE(Exception, it != multiFramePtrMap_.end(), "pose ID " << poseId << " has no multi-frame");
    Eigen::Vector2d measurement;
    it->second->getKeypoint(camIdx, keypointIdx, measurement);
    double size = 1.0;
    it->second->getKeypointSize(camIdx, keypointIdx, size);
    const double uv[2] = {measurement[0], measurement[1]};
    return reinterpret_cast< ::ceres::ResidualBlockId>(svin_ba_add_observation(h_, landmarkId, poseId, camIdx, keypointIdx, uv, size));
  }
  /// implementation/Estimator.hpp:91-157: no caller anywhere in the reference (SURVEY 8(a) "dead"); kept for source compatibility
  template <class GEOMETRY_TYPE>
  ::ceres::ResidualBlockId addRelocObservation(uint64_t, uint64_t, size_t, size_t) {
    OKVIS_THROW(Exception, "addRelocObservation: dead code in the reference, not provided by the svin_ba backend");
    return nullptr;
  }
  bool removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) override {   // :452-474
    return check(svin_ba_remove_observation(h_, landmarkId, poseId, camIdx, keypointIdx), "svin_ba_remove_observation");
  }
  /// NOT a member of the reference's Estimator: the route by which a caller puts an okvis::ceres::HomogeneousPointError on a
  /// landmark of the window (the reference does it with mapPtr_->addResidualBlock(std::make_shared<HomogeneousPointError>(..)),
  /// HomogeneousPointError.cpp:48-117).  information: 3x3 row-major.  Returns the residual id, 0 on failure.
  uint64_t addHomogeneousPointError(uint64_t landmarkId, const Eigen::Vector4d& measurement, const double information[9]) {
    return svin_ba_add_homogeneous_point_error(h_, landmarkId, measurement.data(), information);
  }

  /// Estimator::applyMarginalizationStrategy (Estimator.cpp:495-814)
  bool applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, okvis::MapPointVector& removedLandmarks) {
    // the MapPoints of the landmarks that are about to go: fetched before, because afterwards they are gone
    std::vector<uint64_t> ids(std::max<size_t>(numLandmarks(), 1));
    int nRemoved = 0;
    okvis::PointMap before;
    getLandmarks(before);
    const int r = svin_ba_apply_marginalization_strategy(h_, numKeyframes, numImuFrames, ids.data(), (int)ids.size(), &nRemoved);
    if (r < 0) OKVIS_THROW(Exception, std::string("svin_ba_apply_marginalization_strategy: ") + svin_ba_last_error());
    for (int i = 0; i < nRemoved && i < (int)ids.size(); ++i) {
      auto it = before.find(ids[i]);
      if (it != before.end()) removedLandmarks.push_back(it->second);   // :709,:753
    }
    // multi-frames of frames that left the window (:769)
    std::vector<uint64_t> frames(std::max<size_t>(numFrames(), 1));
    const int nf = svin_ba_frame_ids(h_, frames.data(), (int)frames.size());
    for (auto it = multiFramePtrMap_.begin(); it != multiFramePtrMap_.end();) {
      bool alive = false;
      for (int k = 0; k < nf; ++k) alive = alive || frames[k] == it->first;
      if (alive) ++it; else it = multiFramePtrMap_.erase(it);
    }
    return r == 1;
  }

  /// static Estimator::initPoseFromImu (Estimator.cpp:848-873)
  static bool initPoseFromImu(const okvis::ImuMeasurementDeque& imuMeasurements, okvis::kinematics::Transformation& T_WS) {
    std::vector<svin_imu_sample> imu(imuMeasurements.size());
    size_t n = 0;
    for (const auto& m : imuMeasurements) {
      svin_imu_sample& s = imu[n++];
      s.sec = m.timeStamp.sec; s.nsec = m.timeStamp.nsec;
      for (int k = 0; k < 3; ++k) { s.gyr[k] = m.measurement.gyroscopes[k]; s.acc[k] = m.measurement.accelerometers[k]; }
    }
    double T[7];
    const int r = svin_ba_init_pose_from_imu(imu.empty() ? nullptr : imu.data(), (int)imu.size(), T);
    T_WS = fromArray(T);
    return r == 1;
  }

  /// Estimator::optimize (Estimator.cpp:876-929): options as the reference sets them, Map::solve, landmark qualities
  void optimize(size_t numIter, size_t numThreads = 1, bool verbose = false) override {
    mapPtr_->options.linear_solver_type = ::ceres::SPARSE_SCHUR;
    mapPtr_->options.trust_region_strategy_type = ::ceres::DOGLEG;
    mapPtr_->options.num_threads = (int)numThreads;
    mapPtr_->options.max_num_iterations = (int)numIter;
    mapPtr_->options.minimizer_progress_to_stdout = verbose;
    try {
      mapPtr_->solve();
    } catch (const std::runtime_error& e) {
      OKVIS_THROW(Exception, e.what());
    }
  }
  bool setOptimizationTimeLimit(double timeLimit, int minIterations) override {   // :932-951
    return svin_ba_set_optimization_time_limit(h_, timeLimit, minIterations) == 1;
  }

  /// @name Getters (Estimator.cpp:955-1077)
  ///@{
  bool isLandmarkAdded(uint64_t landmarkId) const override { return svin_ba_is_landmark_added(h_, landmarkId) == 1; }
  bool isLandmarkInitialized(uint64_t landmarkId) const override {
    const int r = svin_ba_is_landmark_initialized(h_, landmarkId);
    OKVIS_ASSERT_TRUE(Exception, r >= 0, "landmark not added");
    return r == 1;
  }
  bool getLandmark(uint64_t landmarkId, okvis::MapPoint& mapPoint) const override {
    std::lock_guard<std::mutex> l(statesMutex_);
    svin_landmark_info li;
    if (svin_ba_get_landmark(h_, landmarkId, &li) != 1) return false;
    fillMapPoint(landmarkId, li, mapPoint);
    return true;
  }
  size_t getLandmarks(okvis::PointMap& landmarks) const override {
    std::lock_guard<std::mutex> l(statesMutex_);
    landmarks.clear();
    forEachLandmark([&](const okvis::MapPoint& mp) { landmarks.insert(std::make_pair(mp.id, mp)); });
    return landmarks.size();
  }
  size_t getLandmarks(okvis::MapPointVector& landmarks) const {
    std::lock_guard<std::mutex> l(statesMutex_);
    landmarks.clear();
    forEachLandmark([&](const okvis::MapPoint& mp) { landmarks.push_back(mp); });
    return landmarks.size();
  }
  okvis::MultiFramePtr multiFrame(uint64_t frameId) const override {
    auto it = multiFramePtrMap_.find(frameId);
    OKVIS_ASSERT_TRUE(Exception, it != multiFramePtrMap_.end(), "Requested multi-frame does not exist in estimator.");
    return it->second;
  }
  bool get_T_WS(uint64_t poseId, okvis::kinematics::Transformation& T_WS) const override {
    double T[7];
    if (svin_ba_get_T_WS(h_, poseId, T) != 1) return false;
    T_WS = fromArray(T);
    return true;
  }
  bool getImuPreIntegral(uint64_t poseId, Eigen::Vector3d& acc_doubleintegral, Eigen::Vector3d& acc_integral, double& Delta_t) const {
    double a[3], b[3];
    if (svin_ba_get_imu_preintegral(h_, poseId, a, b, &Delta_t) != 1) return false;
    for (int k = 0; k < 3; ++k) { acc_doubleintegral[k] = a[k]; acc_integral[k] = b[k]; }
    return true;
  }
  bool getSpeedAndBias(uint64_t poseId, uint64_t imuIdx, okvis::SpeedAndBias& speedAndBias) const override {
    double sb[9];
    if (svin_ba_get_speed_and_bias(h_, poseId, imuIdx, sb) != 1) return false;
    for (int k = 0; k < 9; ++k) speedAndBias[k] = sb[k];
    return true;
  }
  bool getCameraSensorStates(uint64_t poseId, size_t cameraIdx, okvis::kinematics::Transformation& T_SCi) const override {
    double T[7];
    if (svin_ba_get_camera_sensor_states(h_, poseId, cameraIdx, T) != 1) return false;
    T_SCi = fromArray(T);
    return true;
  }
  size_t numFrames() const override { return (size_t)svin_ba_num_frames(h_); }
  size_t numLandmarks() const override { return (size_t)svin_ba_num_landmarks(h_); }
  uint64_t currentKeyframeId() const { return svin_ba_current_keyframe_id(h_); }
  uint64_t frameIdByAge(size_t age) const { return svin_ba_frame_id_by_age(h_, age); }
  uint64_t currentFrameId() const override { return svin_ba_current_frame_id(h_); }
  bool isKeyframe(uint64_t frameId) const override {
    const int r = svin_ba_is_keyframe(h_, frameId);
    OKVIS_ASSERT_TRUE(Exception, r >= 0, "unknown frame " << frameId);   // statesMap_.at() throws in the reference
    return r == 1;
  }
  bool isInImuWindow(uint64_t frameId) const { return svin_ba_is_in_imu_window(h_, frameId) == 1; }
  okvis::Time timestamp(uint64_t frameId) const override {
    uint32_t sec = 0, nsec = 0;
    const int r = svin_ba_timestamp(h_, frameId, &sec, &nsec);
    OKVIS_ASSERT_TRUE(Exception, r == 1, "unknown frame " << frameId);
    return okvis::Time(sec, nsec);
  }
  ///@}

  /// @name Setters (Estimator.cpp:1079-1130)
  ///@{
  bool set_T_WS(uint64_t poseId, const okvis::kinematics::Transformation& T_WS) override {
    double T[7];
    toArray(T_WS, T);
    return svin_ba_set_T_WS(h_, poseId, T) == 1;
  }
  void setImuPreIntegral(uint64_t poseId, Eigen::Vector3d& acc_doubleintegral, Eigen::Vector3d& acc_integral, double& Delta_t) {
    const double a[3] = {acc_doubleintegral[0], acc_doubleintegral[1], acc_doubleintegral[2]};
    const double b[3] = {acc_integral[0], acc_integral[1], acc_integral[2]};
    svin_ba_set_imu_preintegral(h_, poseId, a, b, Delta_t);
    imuIntegralsMap_.insert(std::make_pair(poseId, imu_integrals(acc_doubleintegral, acc_integral, Delta_t)));
  }
  struct imu_integrals {   // Estimator.hpp:394-402
    imu_integrals(Eigen::Vector3d acc_doubleintegral, Eigen::Vector3d acc_integral, double Delta_t)
        : acc_doubleintegral(acc_doubleintegral), acc_integral(acc_integral), Delta_t(Delta_t) {}
    Eigen::Vector3d acc_doubleintegral;
    Eigen::Vector3d acc_integral;
    double Delta_t;
  };
  std::map<uint64_t, imu_integrals> imuIntegralsMap_;   // public in the reference (:404-408)

  bool setSpeedAndBias(uint64_t poseId, size_t imuIdx, const okvis::SpeedAndBias& speedAndBias) override {
    double sb[9];
    for (int k = 0; k < 9; ++k) sb[k] = speedAndBias[k];
    return svin_ba_set_speed_and_bias(h_, poseId, imuIdx, sb) == 1;
  }
  bool setCameraSensorStates(uint64_t poseId, size_t cameraIdx, const okvis::kinematics::Transformation& T_SCi) override {
    double T[7];
    toArray(T_SCi, T);
    return svin_ba_set_camera_sensor_states(h_, poseId, cameraIdx, T) == 1;
  }
  bool setLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) override {
    const double hp[4] = {landmark[0], landmark[1], landmark[2], landmark[3]};
    return svin_ba_set_landmark(h_, landmarkId, hp) == 1;
  }
  void setLandmarkInitialized(uint64_t landmarkId, bool initialized) override {
    const int r = svin_ba_set_landmark_initialized(h_, landmarkId, initialized ? 1 : 0);
    OKVIS_ASSERT_TRUE(Exception, r == 1, "landmark not added");
  }
  void setKeyframe(uint64_t frameId, bool isKeyframe) override {
    const int r = svin_ba_set_keyframe(h_, frameId, isKeyframe ? 1 : 0);
    OKVIS_ASSERT_TRUE(Exception, r == 1, "unknown frame " << frameId);
  }
  /// Estimator::setMap (:448): the caller's Map object becomes the view of this estimator's graph
  void setMap(std::shared_ptr<okvis::ceres::Map> mapPtr) override {
    mapPtr_ = mapPtr;
    mapPtr_->attach(h_);
  }
  int stateCount_ = 0;   // public data member of the reference (Estimator.hpp:450), read by Frontend.cpp:269
  ///@}

  /// not part of the reference: the underlying handle and map (sharded multi-GPU set-up, inspection hooks)
  svin_ba* handle() const { return h_; }
  std::shared_ptr<okvis::ceres::Map> map() const { return mapPtr_; }

 private:
  static uint64_t idTrampoline(void*) { return okvis::IdProvider::instance().newId(); }
  bool check(int r, const char* what) const {
    if (r < 0) OKVIS_THROW(Exception, std::string(what) + ": " + svin_ba_last_error());
    return r == 1;
  }
  static void toArray(const okvis::kinematics::Transformation& T, double* out) {
    const Eigen::Vector3d r = T.r();
    const Eigen::Quaterniond q = T.q();
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
    out[3] = q.x(); out[4] = q.y(); out[5] = q.z(); out[6] = q.w();
  }
  static okvis::kinematics::Transformation fromArray(const double* T) {
    return okvis::kinematics::Transformation(Eigen::Vector3d(T[0], T[1], T[2]), Eigen::Quaterniond(T[6], T[3], T[4], T[5]));
  }
  /// camera geometry of camera i from the multi-frame: what implementation/Estimator.hpp:62-66 obtains per observation
  void registerGeometry(const okvis::MultiFrame& mf, size_t i) {
    auto g = mf.geometry(i);
    Eigen::VectorXd intr;
    g->getIntrinsics(intr);   // fu fv cu cv + distortion coefficients
    const std::string d = g->distortionType();
    int model = -1, nd = 0;
    if (d == "NoDistortion") { model = SVIN_DIST_NONE; nd = 0; }
    else if (d == "RadialTangentialDistortion") { model = SVIN_DIST_RADTAN; nd = 4; }
    else if (d == "EquidistantDistortion") { model = SVIN_DIST_EQUIDISTANT; nd = 4; }
    else if (d == "RadialTangentialDistortion8") { model = SVIN_DIST_RADTAN8; nd = 8; }
    OKVIS_ASSERT_TRUE(Exception, model >= 0, "unsupported distortion model " << d);
    OKVIS_ASSERT_TRUE(Exception, (int)intr.size() >= 4 + nd, "intrinsics vector too short for " << d);
    double in4[4], dist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) in4[k] = intr[k];
    for (int k = 0; k < nd; ++k) dist[k] = intr[4 + k];
    const int r = svin_ba_set_camera_geometry(h_, i, model, in4, nd ? dist : nullptr, nd, (int)g->imageWidth(), (int)g->imageHeight());
    OKVIS_ASSERT_TRUE(Exception, r == 1, "svin_ba_set_camera_geometry failed");
    geometryKnown_[i] = true;
  }
  void fillMapPoint(uint64_t id, const svin_landmark_info& li, okvis::MapPoint& mp) const {
    mp.id = id;
    mp.point = Eigen::Vector4d(li.point[0], li.point[1], li.point[2], li.point[3]);
    mp.quality = li.quality;
    mp.distance = li.distance;
    mp.observations.clear();
    const int n = li.num_observations;
    if (n <= 0) return;
    std::vector<uint64_t> f((size_t)n), c((size_t)n), k((size_t)n), r((size_t)n);
    const int m = svin_ba_get_landmark_observations(h_, id, f.data(), c.data(), k.data(), r.data(), n);
    for (int i = 0; i < m && i < n; ++i)
      mp.observations.insert(std::make_pair(okvis::KeypointIdentifier(f[i], (size_t)c[i], (size_t)k[i]), r[i]));
  }
  /// every landmark with its observation map from TWO boundary crossings (size query + fill), whatever the window size
  template <class F>
  void forEachLandmark(F&& fn) const {
    int32_t nObs = 0;
    const int n = svin_ba_get_all_landmark_observations(h_, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, &nObs);
    if (n <= 0) return;
    std::vector<uint64_t> ids((size_t)n), f((size_t)nObs + 1), c((size_t)nObs + 1), k((size_t)nObs + 1), r((size_t)nObs + 1);
    std::vector<svin_landmark_info> infos((size_t)n);
    std::vector<int32_t> ptr((size_t)n + 1);
    const int m = svin_ba_get_all_landmark_observations(h_, n, ids.data(), infos.data(), ptr.data(), nObs, f.data(), c.data(), k.data(), r.data(), &nObs);
    okvis::MapPoint mp;
    for (int i = 0; i < m && i < n; ++i) {
      const svin_landmark_info& li = infos[i];
      mp.id = ids[i];
      mp.point = Eigen::Vector4d(li.point[0], li.point[1], li.point[2], li.point[3]);
      mp.quality = li.quality;
      mp.distance = li.distance;
      mp.observations.clear();
      for (int32_t o = ptr[i]; o < ptr[i + 1]; ++o)
        mp.observations.insert(std::make_pair(okvis::KeypointIdentifier(f[o], (size_t)c[o], (size_t)k[o]), r[o]));
      fn(mp);
    }
  }

  svin_ba* h_;
  std::shared_ptr<okvis::ceres::Map> mapPtr_;                                        // :596
  std::map<uint64_t, okvis::MultiFramePtr> multiFramePtrMap_;                        // :600
  std::vector<okvis::ExtrinsicsEstimationParameters> extrinsicsEstimationParametersVec_;   // :611
  std::vector<bool> geometryKnown_;
  std::vector<okvis::ImuParameters> imuParametersVec_;                               // :614
  okvis::SonarParameters sonarParameters_;                                           // :617
  mutable std::mutex statesMutex_;                                                   // :630
};

}  // namespace okvis

#endif  // INTEGRATION_OKVIS_ESTIMATOR_HPP_
