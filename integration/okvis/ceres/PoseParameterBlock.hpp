// integration/okvis/ceres/PoseParameterBlock.hpp -- okvis::ceres::PoseParameterBlock
// (okvis_ceres/include/okvis/ceres/PoseParameterBlock.hpp:53-128, src/PoseParameterBlock.cpp:47-88): [r(3) | q xyzw]
// storage, okvis::kinematics::Transformation estimate, time stamp, the PoseManifold operations.
#ifndef INTEGRATION_OKVIS_CERES_POSEPARAMETERBLOCK_HPP_
#define INTEGRATION_OKVIS_CERES_POSEPARAMETERBLOCK_HPP_

#include <string>

#include <okvis/Time.hpp>
#include <okvis/ceres/ParameterBlockSized.hpp>
#include <okvis/ceres/PoseManifold.hpp>
#include <okvis/kinematics/Transformation.hpp>

namespace okvis {
namespace ceres {

class PoseParameterBlock : public ParameterBlockSized<7, 6, okvis::kinematics::Transformation> {
 public:
  typedef okvis::kinematics::Transformation estimate_t;
  typedef ParameterBlockSized<7, 6, estimate_t> base_t;

  PoseParameterBlock() : base_t() { setFixed(false); }
  PoseParameterBlock(const okvis::kinematics::Transformation& T_WS, uint64_t id, const okvis::Time& timestamp) {
    setEstimate(T_WS);
    setId(id);
    setTimestamp(timestamp);
    setFixed(false);
  }
  virtual ~PoseParameterBlock() {}

  virtual void setEstimate(const okvis::kinematics::Transformation& T_WS) {
    for (int k = 0; k < 3; ++k) parameters_[k] = T_WS.r()[k];
    parameters_[3] = T_WS.q().x(); parameters_[4] = T_WS.q().y(); parameters_[5] = T_WS.q().z(); parameters_[6] = T_WS.q().w();
  }
  virtual okvis::kinematics::Transformation estimate() const {
    return okvis::kinematics::Transformation(Eigen::Vector3d(parameters_[0], parameters_[1], parameters_[2]),
                                             Eigen::Quaterniond(parameters_[6], parameters_[3], parameters_[4], parameters_[5]));
  }
  void setTimestamp(const okvis::Time& timestamp) { timestamp_ = timestamp; }
  okvis::Time timestamp() const { return timestamp_; }

  virtual void plus(const double* x0, const double* Delta_Chi, double* x0_plus_Delta) const { PoseManifold::plus(x0, Delta_Chi, x0_plus_Delta); }
  virtual void plusJacobian(const double* x0, double* jacobian) const { PoseManifold::plusJacobian(x0, jacobian); }
  virtual void minus(const double* x0_plus_Delta, const double* x0, double* Delta_Chi) const { PoseManifold::minus(x0_plus_Delta, x0, Delta_Chi); }
  virtual void liftJacobian(const double* x0, double* jacobian) const { PoseManifold::liftJacobian(x0, jacobian); }

  virtual std::string typeInfo() const { return "PoseParameterBlock"; }

 private:
  okvis::Time timestamp_;
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_POSEPARAMETERBLOCK_HPP_
