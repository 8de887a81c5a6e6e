// integration/okvis/ceres/CeresTypes.hpp -- the handful of ::ceres names that appear in SIGNATURES of the okvis headers
// other packages include (okvis_ceres/include/okvis/ceres/ParameterBlock.hpp:48,:134-139 `const ::ceres::Manifold*`,
// Map.hpp:71-86 `::ceres::ResidualBlockId`, Estimator.hpp `::ceres::ResidualBlockId addObservation`).  With this backend
// there is no Ceres in the build: the names are declared here as plain interface types, so that okvis_frontend and
// okvis_multisensor_processing compile without `ceres/ceres.h`.  If the real Ceres headers were included first, theirs win.
#ifndef INTEGRATION_OKVIS_CERES_CERESTYPES_HPP_
#define INTEGRATION_OKVIS_CERES_CERESTYPES_HPP_

#if !defined(CERES_PUBLIC_TYPES_H_) && !defined(CERES_PUBLIC_MANIFOLD_H_)
namespace ceres {
struct ResidualBlock;
typedef ResidualBlock* ResidualBlockId;   // opaque handle carrying the core's residual id
/// The loss objects a caller hands to Map::addResidualBlock.  They carry no arithmetic: the device solver applies
/// CauchyLoss(1) to reprojection residuals (Estimator.cpp:69) and no loss to everything else; Map checks that the object it is
/// given names exactly that.
class LossFunction {
 public:
  virtual ~LossFunction() {}
};
class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : a_(a) {}
  double a() const { return a_; }
 private:
  double a_;
};
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
/// the abstract interface of ceres 2.2's Manifold (what PoseManifold / HomogeneousPointManifold override)
class Manifold {
 public:
  virtual ~Manifold() {}
  virtual int AmbientSize() const = 0;
  virtual int TangentSize() const = 0;
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool PlusJacobian(const double* x, double* jacobian) const = 0;
  virtual bool Minus(const double* y, const double* x, double* y_minus_x) const = 0;
  virtual bool MinusJacobian(const double* x, double* jacobian) const = 0;
};
}  // namespace ceres
#endif

#endif  // INTEGRATION_OKVIS_CERES_CERESTYPES_HPP_
