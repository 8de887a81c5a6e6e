// integration/okvis/ceres/HomogeneousPointParameterBlock.hpp -- okvis::ceres::HomogeneousPointParameterBlock
// (okvis_ceres/include/okvis/ceres/HomogeneousPointParameterBlock.hpp:54-146, src/HomogeneousPointParameterBlock.cpp:47-88):
// what VioKeyframeWindowMatchingAlgorithm.cpp:453 builds to hand `point.estimate()` to Estimator::setLandmark.
#ifndef INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTPARAMETERBLOCK_HPP_
#define INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTPARAMETERBLOCK_HPP_

#include <string>

#include <Eigen/Core>

#include <okvis/ceres/HomogeneousPointManifold.hpp>
#include <okvis/ceres/ParameterBlockSized.hpp>

namespace okvis {
namespace ceres {

class HomogeneousPointParameterBlock : public ParameterBlockSized<4, 3, Eigen::Vector4d> {
 public:
  typedef Eigen::Vector4d estimate_t;
  typedef ParameterBlockSized<4, 3, estimate_t> base_t;

  HomogeneousPointParameterBlock() : base_t(), initialized_(false) { setFixed(false); }
  HomogeneousPointParameterBlock(const Eigen::Vector4d& point, uint64_t id, bool initialized = true) {
    setEstimate(point);
    setId(id);
    setInitialized(initialized);
    setFixed(false);
  }
  HomogeneousPointParameterBlock(const Eigen::Vector3d& point, uint64_t id, bool initialized = true) {
    setEstimate(Eigen::Vector4d(point[0], point[1], point[2], 1.0));
    setId(id);
    setInitialized(initialized);
    setFixed(false);
  }
  virtual ~HomogeneousPointParameterBlock() {}

  virtual void setEstimate(const Eigen::Vector4d& point) { for (int k = 0; k < 4; ++k) parameters_[k] = point[k]; }
  virtual Eigen::Vector4d estimate() const { return Eigen::Vector4d(parameters_[0], parameters_[1], parameters_[2], parameters_[3]); }
  void setInitialized(bool initialized) { initialized_ = initialized; }
  bool initialized() const { return initialized_; }

  virtual void plus(const double* x0, const double* Delta_Chi, double* x0_plus_Delta) const { HomogeneousPointManifold::plus(x0, Delta_Chi, x0_plus_Delta); }
  virtual void plusJacobian(const double* x0, double* jacobian) const { HomogeneousPointManifold::plusJacobian(x0, jacobian); }
  virtual void minus(const double* x0_plus_Delta, const double* x0, double* Delta_Chi) const { HomogeneousPointManifold::minus(x0_plus_Delta, x0, Delta_Chi); }
  virtual void liftJacobian(const double* x0, double* jacobian) const { HomogeneousPointManifold::liftJacobian(x0, jacobian); }

  virtual std::string typeInfo() const { return "HomogeneousPointParameterBlock"; }

 private:
  bool initialized_;
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTPARAMETERBLOCK_HPP_
