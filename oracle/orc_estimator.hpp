// ORACLE -- test infrastructure only (never linked into the product library).
// Restatement of okvis::Estimator (okvis_ceres/include/okvis/Estimator.hpp:81,
// okvis_ceres/src/Estimator.cpp, include/okvis/implementation/Estimator.hpp).
#pragma once
#include <map>
#include <tuple>
#include "orc_marg.hpp"

namespace orc {

struct ExtrinsicsEstimationParameters {  // okvis_common Parameters.hpp
  double sigma_absolute_translation = 0, sigma_absolute_orientation = 0;
  double sigma_c_relative_translation = 0, sigma_c_relative_orientation = 0;
};

struct MapPoint {  // okvis_common FrameTypedefs.hpp
  uint64_t id = 0;
  double point[4] = {0, 0, 0, 1};
  double quality = 0, distance = 0;
  std::map<std::tuple<uint64_t, size_t, size_t>, uint64_t> observations;  // (frame, cam, keypoint) -> residual id
};

struct SonarMeasurement { double range, heading; };

class Estimator {
 public:
  struct StateInfo { uint64_t id = 0; bool exists = false; };
  struct States {
    bool isKeyframe = false;
    uint64_t id = 0;
    Time timestamp;
    StateInfo T_WS;
    std::vector<StateInfo> T_SC;  // per camera
    std::vector<StateInfo> speedAndBias;  // per imu
  };

  Estimator() : map_(new Map()) {}
  uint64_t newId() { return ++idCounter_; }  // IdProvider.cpp

  int addCamera(const ExtrinsicsEstimationParameters& p, const Camera& geometry);  // Estimator.cpp:77-80
  int addImu(const ImuParameters& p);                                              // :83-90
  void setSonarExtrinsics(const Transformation& T_SSo) { T_SSo_ = T_SSo; }

  // E1 Estimator.cpp:98-411.  T_SC: per-camera extrinsics of the multi-frame (7 doubles each)
  bool addStates(uint64_t frameId, Time stamp, size_t numKeypoints, const std::vector<Transformation>& T_SC,
                 const std::vector<ImuSample>& imu, bool asKeyframe, const std::vector<SonarMeasurement>& sonar,
                 const std::vector<double>& depth, double firstDepth);
  bool addLandmark(uint64_t landmarkId, const double* hp);                         // E2 :414-429
  // E3 implementation/Estimator.hpp:47-87; returns residual id or 0 for a duplicate
  uint64_t addObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx, const double* uv,
                          double size);
  bool removeObservation(uint64_t residualId);                                     // E4 :432-449
  bool removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx);  // :452-474
  bool applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, std::vector<MapPoint>& removed);  // E6
  void optimize(size_t numIter, size_t numThreads, bool verbose);                  // E5 :876-929
  bool setOptimizationTimeLimit(double timeLimit, int minIterations);              // E7 :932-951
  static bool initPoseFromImu(const std::vector<ImuSample>& imu, Transformation& T_WS);  // :848-873

  // E8 getters / setters
  bool get_T_WS(uint64_t poseId, double* T) const;
  bool getSpeedAndBias(uint64_t poseId, size_t imuIdx, double* sb) const;
  bool getCameraSensorStates(uint64_t poseId, size_t camIdx, double* T) const;
  bool getLandmark(uint64_t id, MapPoint& mp) const;
  bool isLandmarkAdded(uint64_t id) const { return landmarksMap_.count(id) != 0; }
  bool set_T_WS(uint64_t poseId, const double* T);
  bool setSpeedAndBias(uint64_t poseId, size_t imuIdx, const double* sb);
  bool setCameraSensorStates(uint64_t poseId, size_t camIdx, const double* T);
  bool setLandmark(uint64_t id, const double* hp);
  size_t numFrames() const { return statesMap_.size(); }
  size_t numLandmarks() const { return landmarksMap_.size(); }
  uint64_t currentKeyframeId() const;
  uint64_t frameIdByAge(size_t age) const;
  uint64_t currentFrameId() const { return statesMap_.empty() ? 0 : statesMap_.rbegin()->first; }
  bool isKeyframe(uint64_t id) const { return statesMap_.at(id).isKeyframe; }
  bool isInImuWindow(uint64_t id) const;

  Map& map() { return *map_; }
  const std::map<uint64_t, States>& states() const { return statesMap_; }
  const std::map<uint64_t, MapPoint>& landmarks() const { return landmarksMap_; }
  std::shared_ptr<MarginalizationError> marginalizationError() const { return margPtr_; }

 private:
  std::unique_ptr<Map> map_;
  std::map<uint64_t, States> statesMap_;
  std::map<uint64_t, MapPoint> landmarksMap_;
  std::vector<ExtrinsicsEstimationParameters> extrinsicsVec_;
  std::vector<Camera> cameras_;
  std::vector<ImuParameters> imuVec_;
  Transformation T_SSo_;
  std::shared_ptr<MarginalizationError> margPtr_;
  uint64_t margResidualId_ = 0;
  uint64_t referencePoseId_ = 0;
  uint64_t idCounter_ = 0;
  bool hasCallback_ = false;
};

}  // namespace orc
