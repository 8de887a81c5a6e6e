// integration/okvis/ceres/ReprojectionError.hpp -- okvis::ceres::ReprojectionError<GEOMETRY_TYPE> as a stand-alone
// single-residual evaluator (okvis_ceres/include/okvis/ceres/ReprojectionError.hpp:60-170, implementation
// .../implementation/ReprojectionError.hpp:85-229): what ProbabilisticStereoTriangulator.cpp:266-300 and
// VioKeyframeWindowMatchingAlgorithm.cpp:453 construct on the stack.  No ::ceres::SizedCostFunction base -- the window's
// reprojection residuals are created by okvis::Estimator::addObservation and evaluated on the GPU; this class evaluates
// ONE residual on the CPU through svin_host_reprojection_error (the device function compiled for the host).
#ifndef INTEGRATION_OKVIS_CERES_REPROJECTIONERROR_HPP_
#define INTEGRATION_OKVIS_CERES_REPROJECTIONERROR_HPP_

#include <svin_ba.h>

#include <Eigen/Core>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>

namespace okvis {
namespace ceres {

template <class GEOMETRY_TYPE>
class ReprojectionError {
 public:
  typedef GEOMETRY_TYPE camera_geometry_t;
  typedef Eigen::Vector2d measurement_t;          // ReprojectionErrorBase.hpp:70
  typedef Eigen::Matrix<double, 2, 2> covariance_t;   // :73
  static const int kNumResiduals = 2;

  ReprojectionError() : cameraId_(0) { for (int k = 0; k < 4; ++k) info_[k] = (k % 3 == 0) ? 1.0 : 0.0; }
  ReprojectionError(std::shared_ptr<const camera_geometry_t> cameraGeometry, uint64_t cameraId, const measurement_t& measurement,
                    const covariance_t& information)
      : cameraId_(cameraId) {
    setCameraGeometry(cameraGeometry);
    setMeasurement(measurement);
    setInformation(information);
  }
  void setMeasurement(const measurement_t& measurement) { measurement_ = measurement; }
  void setCameraGeometry(std::shared_ptr<const camera_geometry_t> cameraGeometry) {
    cameraGeometry_ = cameraGeometry;
    Eigen::VectorXd intr;
    cameraGeometry->getIntrinsics(intr);
    const std::string d = cameraGeometry->distortionType();
    nDist_ = 0;
    if (d == "NoDistortion") { model_ = SVIN_DIST_NONE; }
    else if (d == "RadialTangentialDistortion") { model_ = SVIN_DIST_RADTAN; nDist_ = 4; }
    else if (d == "EquidistantDistortion") { model_ = SVIN_DIST_EQUIDISTANT; nDist_ = 4; }
    else if (d == "RadialTangentialDistortion8") { model_ = SVIN_DIST_RADTAN8; nDist_ = 8; }
    else throw std::runtime_error("ReprojectionError: unsupported distortion model " + d);
    for (int k = 0; k < 4; ++k) intr_[k] = intr[k];
    for (int k = 0; k < nDist_; ++k) dist_[k] = intr[4 + k];
  }
  void setInformation(const covariance_t& information) {
    information_ = information;
    info_[0] = information(0, 0); info_[1] = information(0, 1); info_[2] = information(1, 0); info_[3] = information(1, 1);
  }
  const measurement_t& measurement() const { return measurement_; }
  const covariance_t& information() const { return information_; }
  uint64_t cameraId() const { return cameraId_; }
  // what okvis::ceres::Map::addResidualBlock hands to the backend (svin_ba_add_camera / svin_ba_map_add_reprojection_error)
  std::shared_ptr<const camera_geometry_t> cameraGeometry() const { return cameraGeometry_; }
  int distortionModel() const { return model_; }
  int numDistortionCoefficients() const { return nDist_; }
  const double* intrinsicsArray() const { return intr_; }
  const double* distortionArray() const { return dist_; }
  const double* informationRowMajor() const { return info_; }
  void setCameraId(uint64_t cameraId) { cameraId_ = cameraId; }
  size_t residualDim() const { return kNumResiduals; }
  size_t parameterBlocks() const { return 3; }
  size_t parameterBlockDim(size_t i) const { return i == 1 ? 4 : 7; }
  std::string typeInfo() const { return "ReprojectionError"; }

  /// parameters: T_WS (7), hp_W (4), T_SC (7); jacobians: 2x7, 2x4, 2x7 row-major (any entry may be NULL)
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    return EvaluateWithMinimalJacobians(parameters, residuals, jacobians, nullptr);
  }
  /// jacobiansMinimal: 2x6, 2x3, 2x6 row-major
  bool EvaluateWithMinimalJacobians(double const* const* parameters, double* residuals, double** jacobians, double** jacobiansMinimal) const {
    const double uv[2] = {measurement_[0], measurement_[1]};
    auto ptr = [](double** a, int i) { return a ? a[i] : nullptr; };
    return svin_host_reprojection_error(model_, intr_, nDist_ ? dist_ : nullptr, nDist_, parameters[0], parameters[1], parameters[2], uv, info_,
                                        residuals, ptr(jacobiansMinimal, 0), ptr(jacobiansMinimal, 1), ptr(jacobiansMinimal, 2),
                                        ptr(jacobians, 0), ptr(jacobians, 1), ptr(jacobians, 2)) == 1;
  }

 private:
  std::shared_ptr<const camera_geometry_t> cameraGeometry_;
  uint64_t cameraId_;
  measurement_t measurement_;
  covariance_t information_;
  int model_ = SVIN_DIST_NONE, nDist_ = 0;
  double intr_[4] = {0, 0, 0, 0}, dist_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, info_[4];
};

}  // namespace ceres
}  // namespace okvis

#endif  // INTEGRATION_OKVIS_CERES_REPROJECTIONERROR_HPP_
