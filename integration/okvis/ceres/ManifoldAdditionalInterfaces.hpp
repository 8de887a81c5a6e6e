// integration/okvis/ceres/ManifoldAdditionalInterfaces.hpp -- okvis::ceres::ManifoldAdditionalInterfaces
// (okvis_ceres/include/okvis/ceres/ManifoldAdditionalInterfaces.hpp:53-69, src/ManifoldAdditionalInterfaces.cpp:46-105):
// ComputeLiftJacobian + the numeric self-check `verify` (central differences of Plus against PlusJacobian, and
// J_lift * J_plus = I, both to 1e-6 in the Frobenius norm).
#ifndef INTEGRATION_OKVIS_CERES_MANIFOLDADDITIONALINTERFACES_HPP_
#define INTEGRATION_OKVIS_CERES_MANIFOLDADDITIONALINTERFACES_HPP_

#include <cmath>
#include <vector>

#include <okvis/ceres/CeresTypes.hpp>

namespace okvis {
namespace ceres {

class ManifoldAdditionalInterfaces {
 public:
  virtual ~ManifoldAdditionalInterfaces() {}
  virtual bool ComputeLiftJacobian(const double* x, double* jacobian) const = 0;

  virtual bool verify(const double* x_raw, double purturbation_magnitude = 1.0e-6) const {
    const ::ceres::Manifold* m = dynamic_cast<const ::ceres::Manifold*>(this);
    if (!m) return false;
    const int na = m->AmbientSize(), nt = m->TangentSize();
    const double dx = purturbation_magnitude;
    std::vector<double> Jnum((size_t)na * nt), Jp((size_t)na * nt), Jl((size_t)nt * na), d((size_t)nt), xp((size_t)na), xm((size_t)na);
    for (int i = 0; i < nt; ++i) {
      for (int k = 0; k < nt; ++k) d[k] = 0.0;
      d[i] = dx;
      m->Plus(x_raw, d.data(), xp.data());
      d[i] = -dx;
      m->Plus(x_raw, d.data(), xm.data());
      for (int r = 0; r < na; ++r) Jnum[(size_t)r * nt + i] = (xp[r] - xm[r]) / (2 * dx);
    }
    m->PlusJacobian(x_raw, Jp.data());
    ComputeLiftJacobian(x_raw, Jl.data());
    double e1 = 0, e2 = 0;
    for (int a = 0; a < nt; ++a)
      for (int b = 0; b < nt; ++b) {
        double s = (a == b) ? -1.0 : 0.0;
        for (int r = 0; r < na; ++r) s += Jl[(size_t)a * na + r] * Jp[(size_t)r * nt + b];
        e1 += s * s;
      }
    for (size_t k = 0; k < Jp.size(); ++k) e2 += (Jp[k] - Jnum[k]) * (Jp[k] - Jnum[k]);
    return !(std::sqrt(e1) > 1.0e-6) && !(std::sqrt(e2) > 1.0e-6);
  }
};

}  // namespace ceres
}  // namespace okvis
#endif  // INTEGRATION_OKVIS_CERES_MANIFOLDADDITIONALINTERFACES_HPP_
