// integration/okvis/ceres/HomogeneousPointManifold.hpp -- okvis::ceres::HomogeneousPointManifold
// (okvis_ceres/include/okvis/ceres/HomogeneousPointManifold.hpp:55-133, src/HomogeneousPointManifold.cpp:50-135): the
// "Euclidean style" plus on the first three homogeneous components, without `ceres/ceres.h`.
#ifndef INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTMANIFOLD_HPP_
#define INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTMANIFOLD_HPP_

#include <svin_ba.h>

#include <okvis/ceres/CeresTypes.hpp>
#include <okvis/ceres/ManifoldAdditionalInterfaces.hpp>

namespace okvis {
namespace ceres {

class HomogeneousPointManifold : public ::ceres::Manifold, public ManifoldAdditionalInterfaces {
 public:
  virtual ~HomogeneousPointManifold() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const { return plus(x, delta, x_plus_delta); }
  virtual bool Minus(const double* x_plus_delta, const double* x, double* delta) const { return minus(x_plus_delta, x, delta); }
  virtual bool PlusJacobian(const double* x, double* jacobian) const { return plusJacobian(x, jacobian); }
  virtual bool MinusJacobian(const double* x, double* jacobian) const { return minusJacobian(x, jacobian); }
  virtual bool ComputeLiftJacobian(const double* x, double* jacobian) const { return liftJacobian(x, jacobian); }
  virtual int AmbientSize() const { return 4; }
  virtual int TangentSize() const { return 3; }

  static bool plus(const double* x, const double* delta, double* x_plus_delta) { return svin_host_manifold_plus(SVIN_MANIFOLD_HPOINT, x, delta, x_plus_delta) == 1; }
  static bool minus(const double* x_plus_delta, const double* x, double* delta) { return svin_host_manifold_minus(SVIN_MANIFOLD_HPOINT, x_plus_delta, x, delta) == 1; }
  static bool plusJacobian(const double* x, double* jacobian) { return svin_host_manifold_plus_jacobian(SVIN_MANIFOLD_HPOINT, x, jacobian) == 1; }
  static bool minusJacobian(const double* x, double* jacobian) { return svin_host_manifold_minus_jacobian(SVIN_MANIFOLD_HPOINT, x, jacobian) == 1; }
  static bool liftJacobian(const double* x, double* jacobian) { return svin_host_manifold_lift_jacobian(SVIN_MANIFOLD_HPOINT, x, jacobian) == 1; }
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_HOMOGENEOUSPOINTMANIFOLD_HPP_
