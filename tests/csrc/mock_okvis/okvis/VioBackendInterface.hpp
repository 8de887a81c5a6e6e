// stand-in for okvis_common/include/okvis/VioBackendInterface.hpp:66-335: the pure virtuals, same signatures
#pragma once
#include <memory>
#include "mock_eigen.hpp"
#include <okvis/FrameTypedefs.hpp>
#include <okvis/Measurements.hpp>
#include <okvis/MultiFrame.hpp>
#include <okvis/Parameters.hpp>
#include <okvis/Variables.hpp>
#include <okvis/assert_macros.hpp>
#include <okvis/kinematics/Transformation.hpp>
namespace okvis {
namespace ceres { class Map; }
class VioBackendInterface {
 public:
  OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error)
  VioBackendInterface() {}
  virtual ~VioBackendInterface() {}
  virtual int addCamera(const ExtrinsicsEstimationParameters& extrinsicsEstimationParameters) = 0;
  virtual int addImu(const ImuParameters& imuParameters) = 0;
  virtual void clearCameras() = 0;
  virtual void clearImus() = 0;
  virtual bool addStates(okvis::MultiFramePtr multiFrame, const okvis::ImuMeasurementDeque& imuMeasurements, bool asKeyframe,
                         const okvis::SonarMeasurementDeque& sonarMeasurements = {}, const okvis::DepthMeasurementDeque& depthMeasurements = {},
                         double firstDepth = 0.0) = 0;
  virtual bool addLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) = 0;
  virtual bool removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) = 0;
  virtual void optimize(size_t numIter, size_t numThreads = 1, bool verbose = false) = 0;
  virtual bool setOptimizationTimeLimit(double timeLimit, int minIterations) = 0;
  virtual bool isLandmarkAdded(uint64_t landmarkId) const = 0;
  virtual bool isLandmarkInitialized(uint64_t landmarkId) const = 0;
  virtual bool getLandmark(uint64_t landmarkId, MapPoint& mapPoint) const = 0;
  virtual size_t getLandmarks(PointMap& landmarks) const = 0;
  virtual okvis::MultiFramePtr multiFrame(uint64_t frameId) const = 0;
  virtual bool get_T_WS(uint64_t poseId, okvis::kinematics::Transformation& T_WS) const = 0;
  virtual bool getSpeedAndBias(uint64_t poseId, uint64_t imuIdx, okvis::SpeedAndBias& speedAndBias) const = 0;
  virtual bool getCameraSensorStates(uint64_t poseId, size_t cameraIdx, okvis::kinematics::Transformation& T_SCi) const = 0;
  virtual size_t numFrames() const = 0;
  virtual size_t numLandmarks() const = 0;
  virtual uint64_t currentFrameId() const = 0;
  virtual bool isKeyframe(uint64_t frameId) const = 0;
  virtual okvis::Time timestamp(uint64_t frameId) const = 0;
  virtual bool set_T_WS(uint64_t poseId, const okvis::kinematics::Transformation& T_WS) = 0;
  virtual bool setSpeedAndBias(uint64_t poseId, size_t imuIdx, const okvis::SpeedAndBias& speedAndBias) = 0;
  virtual bool setCameraSensorStates(uint64_t poseId, size_t cameraIdx, const okvis::kinematics::Transformation& T_SCi) = 0;
  virtual bool setLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) = 0;
  virtual void setLandmarkInitialized(uint64_t landmarkId, bool initialized) = 0;
  virtual void setKeyframe(uint64_t frameId, bool isKeyframe) = 0;
  virtual void setMap(std::shared_ptr<okvis::ceres::Map> mapPtr) = 0;
};
}  // namespace okvis
