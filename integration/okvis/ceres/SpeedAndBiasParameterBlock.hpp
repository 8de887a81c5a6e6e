// integration/okvis/ceres/SpeedAndBiasParameterBlock.hpp -- okvis::ceres::SpeedAndBiasParameterBlock
// (okvis_ceres/include/okvis/ceres/SpeedAndBiasParameterBlock.hpp:56-144): 9 Euclidean parameters [v | b_g | b_a].
#ifndef INTEGRATION_OKVIS_CERES_SPEEDANDBIASPARAMETERBLOCK_HPP_
#define INTEGRATION_OKVIS_CERES_SPEEDANDBIASPARAMETERBLOCK_HPP_

#include <string>

#include <Eigen/Core>

#include <okvis/Time.hpp>
#include <okvis/Variables.hpp>
#include <okvis/ceres/ParameterBlockSized.hpp>

namespace okvis {
namespace ceres {

typedef Eigen::Matrix<double, 9, 1> SpeedAndBias;

class SpeedAndBiasParameterBlock : public ParameterBlockSized<9, 9, SpeedAndBias> {
 public:
  typedef ParameterBlockSized<9, 9, SpeedAndBias> base_t;
  typedef SpeedAndBias estimate_t;

  SpeedAndBiasParameterBlock() : base_t() { setFixed(false); }
  SpeedAndBiasParameterBlock(const SpeedAndBias& speedAndBias, uint64_t id, const okvis::Time& timestamp) {
    setEstimate(speedAndBias);
    setId(id);
    setTimestamp(timestamp);
    setFixed(false);
  }
  virtual ~SpeedAndBiasParameterBlock() {}

  virtual void setEstimate(const SpeedAndBias& speedAndBias) { for (int k = 0; k < 9; ++k) parameters_[k] = speedAndBias[k]; }
  virtual SpeedAndBias estimate() const {
    SpeedAndBias s;
    for (int k = 0; k < 9; ++k) s[k] = parameters_[k];
    return s;
  }
  void setTimestamp(const okvis::Time& timestamp) { timestamp_ = timestamp; }
  okvis::Time timestamp() const { return timestamp_; }

  virtual void plus(const double* x0, const double* Delta_Chi, double* x0_plus_Delta) const { for (int k = 0; k < 9; ++k) x0_plus_Delta[k] = x0[k] + Delta_Chi[k]; }
  virtual void minus(const double* x0_plus_Delta, const double* x0, double* Delta_Chi) const { for (int k = 0; k < 9; ++k) Delta_Chi[k] = x0_plus_Delta[k] - x0[k]; }
  virtual void plusJacobian(const double*, double* jacobian) const { identity(jacobian); }
  virtual void liftJacobian(const double*, double* jacobian) const { identity(jacobian); }

  virtual std::string typeInfo() const { return "SpeedAndBiasParameterBlock"; }

 private:
  static void identity(double* J) { for (int k = 0; k < 81; ++k) J[k] = (k % 10 == 0) ? 1.0 : 0.0; }
  okvis::Time timestamp_;
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_SPEEDANDBIASPARAMETERBLOCK_HPP_
