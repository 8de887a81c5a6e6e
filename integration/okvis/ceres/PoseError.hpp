// integration/okvis/ceres/PoseError.hpp -- okvis::ceres::PoseError as a stand-alone evaluator
// (okvis_ceres/include/okvis/ceres/PoseError.hpp:56-165, src/PoseError.cpp:52-132): the absolute pose prior
// ProbabilisticStereoTriangulator.cpp:87-99 / :128-140 constructs on the stack and evaluates once to obtain
// H = J_min^T J_min.  No ::ceres::SizedCostFunction base (there is no Ceres): the window's own pose priors are created by
// okvis::Estimator::addStates and evaluated on the GPU; this class runs the same function (dmath.hpp poseErrorEval) on
// the CPU through svin_host_pose_error, Eigen::LLT's early exit on a semi-definite information matrix included.
#ifndef INTEGRATION_OKVIS_CERES_POSEERROR_HPP_
#define INTEGRATION_OKVIS_CERES_POSEERROR_HPP_

#include <svin_ba.h>

#include <stdexcept>
#include <string>

#include <Eigen/Core>

#include <okvis/assert_macros.hpp>
#include <okvis/ceres/ErrorInterface.hpp>
#include <okvis/kinematics/Transformation.hpp>

namespace okvis {
namespace ceres {

class PoseError : public ErrorInterface {
 public:
  OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error)
  static const int kNumResiduals = 6;
  typedef Eigen::Matrix<double, 6, 6> information_t;
  typedef Eigen::Matrix<double, 6, 6> covariance_t;

  PoseError() {}
  PoseError(const okvis::kinematics::Transformation& measurement, const Eigen::Matrix<double, 6, 6>& information) {
    setMeasurement(measurement);
    setInformation(information);
  }
  /// information = blockdiag(I / translationVariance, I / rotationVariance)  (PoseError.cpp:56-67)
  PoseError(const okvis::kinematics::Transformation& measurement, double translationVariance, double rotationVariance) {
    setMeasurement(measurement);
    information_t information;
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) information(a, b) = (a != b) ? 0.0 : (a < 3 ? 1.0 / translationVariance : 1.0 / rotationVariance);
    setInformation(information);
  }
  virtual ~PoseError() {}

  void setMeasurement(const okvis::kinematics::Transformation& measurement) {
    measurement_ = measurement;
    for (int k = 0; k < 3; ++k) meas_[k] = measurement.r()[k];
    meas_[3] = measurement.q().x(); meas_[4] = measurement.q().y(); meas_[5] = measurement.q().z(); meas_[6] = measurement.q().w();
  }
  void setInformation(const information_t& information) {
    information_ = information;
    double info[36], cov[36];
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) info[a * 6 + b] = information(a, b);
    if (svin_host_pose_information(info, sqrtInfo_, cov) != 1) OKVIS_THROW(Exception, "svin_host_pose_information failed");
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) { covariance_(a, b) = cov[a * 6 + b]; squareRootInformation_(a, b) = sqrtInfo_[a * 6 + b]; }
  }
  const okvis::kinematics::Transformation& measurement() const { return measurement_; }
  const information_t& information() const { return information_; }
  const information_t& covariance() const { return covariance_; }

  /// parameters[0] = T_WS (7); residuals 6; jacobians[0] 6x7 row-major (may be NULL)
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    return EvaluateWithMinimalJacobians(parameters, residuals, jacobians, nullptr);
  }
  /// jacobiansMinimal[0] 6x6 row-major.  Like the reference (:100-127) the minimal Jacobian is only written when the
  /// ambient one was requested too.
  virtual bool EvaluateWithMinimalJacobians(double const* const* parameters, double* residuals, double** jacobians,
                                            double** jacobiansMinimal) const {
    double* J = (jacobians && jacobians[0]) ? jacobians[0] : nullptr;
    double* Jmin = (J && jacobiansMinimal && jacobiansMinimal[0]) ? jacobiansMinimal[0] : nullptr;
    return svin_host_pose_error(meas_, sqrtInfo_, parameters[0], residuals, Jmin, J) == 1;
  }

  size_t residualDim() const { return kNumResiduals; }
  size_t parameterBlocks() const { return 1; }
  size_t parameterBlockDim(size_t parameterBlockId) const {
    if (parameterBlockId != 0) throw std::out_of_range("PoseError::parameterBlockDim");   // vector::at in the reference (:150)
    return 7;
  }
  virtual std::string typeInfo() const { return "PoseError"; }

 protected:
  okvis::kinematics::Transformation measurement_;
  information_t information_, squareRootInformation_;
  covariance_t covariance_;
  double meas_[7] = {0, 0, 0, 0, 0, 0, 1}, sqrtInfo_[36] = {0};
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_POSEERROR_HPP_
