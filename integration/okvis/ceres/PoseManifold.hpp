// integration/okvis/ceres/PoseManifold.hpp -- okvis::ceres::PoseManifold / PoseManifold3d / 4d / 2d
// (okvis_ceres/include/okvis/ceres/PoseManifold.hpp:52-345, src/PoseManifold.cpp) without `ceres/ceres.h`: the same
// class names, virtual members and static helpers, the arithmetic in libsvin_ba.so (svin_host_manifold_*, host_eval.cpp;
// Pose6d's plus is the very retraction the device solver applies).  Used by PoseParameterBlock's plus / minus /
// liftJacobian and by anything outside okvis_ceres that holds a manifold object.
#ifndef INTEGRATION_OKVIS_CERES_POSEMANIFOLD_HPP_
#define INTEGRATION_OKVIS_CERES_POSEMANIFOLD_HPP_

#include <svin_ba.h>

#include <cmath>

#include <okvis/ceres/CeresTypes.hpp>
#include <okvis/ceres/ManifoldAdditionalInterfaces.hpp>

namespace okvis {
namespace ceres {

namespace shim_detail {
/// one pose manifold over the C ABI; KIND = SVIN_MANIFOLD_POSE*, TANGENT = its tangent dimension
template <int KIND, int TANGENT>
class PoseManifoldOver : public ::ceres::Manifold, public ManifoldAdditionalInterfaces {
 public:
  virtual ~PoseManifoldOver() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const { return plus(x, delta, x_plus_delta); }
  virtual bool Minus(const double* x_plus_delta, const double* x, double* delta) const { return minus(x_plus_delta, x, delta); }
  virtual bool PlusJacobian(const double* x, double* jacobian) const { return plusJacobian(x, jacobian); }
  virtual bool MinusJacobian(const double* x, double* jacobian) const { return minusJacobian(x, jacobian); }
  virtual bool ComputeLiftJacobian(const double* x, double* jacobian) const { return liftJacobian(x, jacobian); }
  virtual int AmbientSize() const { return 7; }
  virtual int TangentSize() const { return TANGENT; }

  static bool plus(const double* x, const double* delta, double* x_plus_delta) { return svin_host_manifold_plus(KIND, x, delta, x_plus_delta) == 1; }
  static bool minus(const double* x_plus_delta, const double* x, double* delta) { return svin_host_manifold_minus(KIND, x_plus_delta, x, delta) == 1; }
  static bool plusJacobian(const double* x, double* jacobian) { return svin_host_manifold_plus_jacobian(KIND, x, jacobian) == 1; }
  static bool minusJacobian(const double* x, double* jacobian) { return svin_host_manifold_minus_jacobian(KIND, x, jacobian) == 1; }
  static bool liftJacobian(const double* x, double* jacobian) { return svin_host_manifold_lift_jacobian(KIND, x, jacobian) == 1; }
};
}  // namespace shim_detail

class PoseManifold : public shim_detail::PoseManifoldOver<SVIN_MANIFOLD_POSE6D, 6> {
 public:
  /// PoseManifold.cpp:152-175: central differences (dx = 1e-9) of Plus into jacobianNumDiff (7x6 row-major), compared
  /// with plusJacobian in the Frobenius norm
  bool VerifyJacobianNumDiff(const double* x, double* jacobian, double* jacobianNumDiff) {
    plusJacobian(x, jacobian);
    const double dx = 1e-9;
    double err = 0;
    for (int i = 0; i < 6; ++i) {
      double d[6] = {0, 0, 0, 0, 0, 0}, xp[7], xm[7];
      d[i] = dx;
      plus(x, d, xp);
      d[i] = -dx;
      plus(x, d, xm);
      for (int r = 0; r < 7; ++r) {
        jacobianNumDiff[r * 6 + i] = (xp[r] - xm[r]) / (2 * dx);
        const double e = jacobian[r * 6 + i] - jacobianNumDiff[r * 6 + i];
        err += e * e;
      }
    }
    return std::sqrt(err) < 1e-6;
  }
};
/// orientation varying (PoseManifold.hpp:136), position and yaw varying (:205), roll / pitch varying (:274)
class PoseManifold3d : public shim_detail::PoseManifoldOver<SVIN_MANIFOLD_POSE3D, 3> {};
class PoseManifold4d : public shim_detail::PoseManifoldOver<SVIN_MANIFOLD_POSE4D, 4> {};
class PoseManifold2d : public shim_detail::PoseManifoldOver<SVIN_MANIFOLD_POSE2D, 2> {};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_POSEMANIFOLD_HPP_
