// integration/okvis/ceres/HomogeneousPointError.hpp -- okvis::ceres::HomogeneousPointError as a stand-alone evaluator
// (okvis_ceres/include/okvis/ceres/HomogeneousPointError.hpp:57-150, src/HomogeneousPointError.cpp:48-117): the absolute
// error of a homogeneous point (landmark).  okvis::Estimator never adds one to its map; a caller that wants the prior
// INSIDE the optimisation adds it with svin_ba_add_homogeneous_point_error(handle, landmarkId, measurement, information)
// (okvis::Estimator::addHomogeneousPointError below forwards to it) -- the residual then takes part in the Schur
// elimination of its landmark on the GPU.  This class evaluates ONE residual on the CPU (svin_host_homogeneous_point_error).
#ifndef INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTERROR_HPP_
#define INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTERROR_HPP_

#include <svin_ba.h>

#include <Eigen/Core>

#include <cstddef>
#include <string>

namespace okvis {
namespace ceres {

class HomogeneousPointError {
 public:
  typedef Eigen::Matrix<double, 3, 3> information_t;   // HomogeneousPointError.hpp:69
  typedef Eigen::Matrix<double, 3, 3> covariance_t;    // :72
  static const int kNumResiduals = 3;

  HomogeneousPointError() { setInformation(identity()); }
  /// (measurement, variance): information = I / variance  (HomogeneousPointError.cpp:52-55)
  HomogeneousPointError(const Eigen::Vector4d& measurement, double variance) {
    setMeasurement(measurement);
    information_t info = identity();
    for (int k = 0; k < 3; ++k) info(k, k) = 1.0 / variance;
    setInformation(info);
  }
  /// (measurement, information)  (:58-62)
  HomogeneousPointError(const Eigen::Vector4d& measurement, const information_t& information) {
    setMeasurement(measurement);
    setInformation(information);
  }
  void setMeasurement(const Eigen::Vector4d& measurement) { measurement_ = measurement; }
  void setInformation(const information_t& information) {   // :66-75
    information_ = information;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) info_[a * 3 + b] = information(a, b);
  }
  const Eigen::Vector4d& measurement() const { return measurement_; }
  const information_t& information() const { return information_; }
  const double* informationRowMajor() const { return info_; }
  size_t residualDim() const { return kNumResiduals; }
  size_t parameterBlocks() const { return 1; }
  size_t parameterBlockDim(size_t) const { return 4; }
  std::string typeInfo() const { return "HomogeneousPointError"; }

  /// ::ceres::CostFunction::Evaluate (:78-81): jacobians[0] = 3x4 row-major or NULL
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    return EvaluateWithMinimalJacobians(parameters, residuals, jacobians, nullptr);
  }
  /// :85-117: jacobiansMinimal[0] = 3x3 row-major or NULL
  bool EvaluateWithMinimalJacobians(double const* const* parameters, double* residuals, double** jacobians,
                                    double** jacobiansMinimal) const {
    double* J = (jacobians != nullptr) ? jacobians[0] : nullptr;
    double* Jm = (jacobiansMinimal != nullptr) ? jacobiansMinimal[0] : nullptr;
    const double meas[4] = {measurement_[0], measurement_[1], measurement_[2], measurement_[3]};
    return svin_host_homogeneous_point_error(parameters[0], meas, info_, residuals, Jm, J) == 1;
  }

 private:
  static information_t identity() {
    information_t m;
    for (int k = 0; k < 3; ++k) m(k, k) = 1.0;
    return m;
  }
  Eigen::Vector4d measurement_;
  information_t information_;
  double info_[9];
};

}  // namespace ceres
}  // namespace okvis

#endif  // INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTERROR_HPP_
