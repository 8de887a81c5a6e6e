// ORACLE -- test infrastructure only (never linked into the product library).
//
// Small dense math used by the CPU restatement of the SVIn/OKVIS sliding-window
// backend.  Plain C++17, no Eigen.  All matrices are row-major `double` arrays.
// Reference files restated (paths relative to /root/reference/okvis_ros/okvis):
//   okvis_kinematics/include/okvis/kinematics/operators.hpp:63-135  (crossMx, plus, oplus)
//   okvis_kinematics/include/okvis/kinematics/implementation/Transformation.hpp:47-253
//   okvis_ceres/include/okvis/ceres/ode/ode.hpp:58-70 (sinc)
// Third-party arithmetic that the reference reaches through Eigen 3 (not vendored):
//   Quaternion::toRotationMatrix, quaternion product/inverse/normalized, LLT,
//   PartialPivLU inverse, SelfAdjointEigenSolver -- restated here from their
//   published algorithms.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>

namespace orc {

// ---------------------------------------------------------------- fixed-size helpers
template <int R, int K, int C>
inline void matmul(const double* A, const double* B, double* out) {  // out(RxC) = A(RxK) B(KxC)
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * C + j];
      out[i * C + j] = s;
    }
}
template <int R, int C>
inline void transpose(const double* A, double* At) {
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) At[j * R + i] = A[i * C + j];
}
inline void mat3_vec(const double* A, const double* v, double* out) {
  for (int i = 0; i < 3; ++i) out[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
inline void mat3T_vec(const double* A, const double* v, double* out) {
  for (int i = 0; i < 3; ++i) out[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
inline double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// operators.hpp:63-75
inline void crossMx(const double* v, double* C) {
  C[0] = 0; C[1] = -v[2]; C[2] = v[1];
  C[3] = v[2]; C[4] = 0; C[5] = -v[0];
  C[6] = -v[1]; C[7] = v[0]; C[8] = 0;
}

// ---------------------------------------------------------------- quaternions [x y z w]
// Eigen Quaternion product a*b.
inline void qmul(const double* a, const double* b, double* o) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}
// Eigen Quaternion::inverse(): conjugate / squaredNorm.
inline void qinv(const double* q, double* o) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 > 0) {
    o[0] = -q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = q[3] / n2;
  } else {
    o[0] = o[1] = o[2] = o[3] = 0;
  }
}
inline void qnormalize(double* q) {
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
// Eigen Quaternion::toRotationMatrix() (no normalisation).
inline void q2R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// operators.hpp:91-112: q_AB*q_BC = plus(q_AB) * q_BC.coeffs()
inline void qplusMat(const double* q, double* Q) {
  Q[0] = q[3]; Q[1] = -q[2]; Q[2] = q[1]; Q[3] = q[0];
  Q[4] = q[2]; Q[5] = q[3]; Q[6] = -q[0]; Q[7] = q[1];
  Q[8] = -q[1]; Q[9] = q[0]; Q[10] = q[3]; Q[11] = q[2];
  Q[12] = -q[0]; Q[13] = -q[1]; Q[14] = -q[2]; Q[15] = q[3];
}
// operators.hpp:114-135: q_AB*q_BC = oplus(q_BC) * q_AB.coeffs()
inline void qoplusMat(const double* q, double* Q) {
  Q[0] = q[3]; Q[1] = q[2]; Q[2] = -q[1]; Q[3] = q[0];
  Q[4] = -q[2]; Q[5] = q[3]; Q[6] = q[0]; Q[7] = q[1];
  Q[8] = q[1]; Q[9] = -q[0]; Q[10] = q[3]; Q[11] = q[2];
  Q[12] = -q[0]; Q[13] = -q[1]; Q[14] = -q[2]; Q[15] = q[3];
}

// Transformation.hpp:47-60 / ode.hpp:58-70
inline double sinc(double x) {
  if (std::fabs(x) > 1e-6) return std::sin(x) / x;
  const double x2 = x * x, x4 = x2 * x2, x6 = x2 * x2 * x2;
  return 1.0 - (1.0 / 6.0) * x2 + (1.0 / 120.0) * x4 - (1.0 / 5040.0) * x6;
}
// Transformation.hpp:62-68
inline void deltaQ(const double* dAlpha, double* dq) {
  const double halfnorm = 0.5 * norm3(dAlpha);
  const double s = sinc(halfnorm) * 0.5;
  dq[0] = s * dAlpha[0]; dq[1] = s * dAlpha[1]; dq[2] = s * dAlpha[2];
  dq[3] = std::cos(halfnorm);
}
// Transformation.hpp:71-85
inline void rightJacobian(const double* phi, double* J) {
  const double Phi = norm3(phi);
  double X[9], X2[9];
  crossMx(phi, X);
  matmul<3, 3, 3>(X, X, X2);
  double a, b;
  if (Phi < 1.0e-4) {
    a = -0.5; b = 1.0 / 6.0;
  } else {
    const double Phi2 = Phi * Phi, Phi3 = Phi2 * Phi;
    a = -(1.0 - std::cos(Phi)) / Phi2;
    b = (Phi - std::sin(Phi)) / Phi3;
  }
  for (int i = 0; i < 9; ++i) J[i] = a * X[i] + b * X2[i];
  J[0] += 1; J[4] += 1; J[8] += 1;
}

// ---------------------------------------------------------------- Transformation
// kinematics::Transformation restated: 7 parameters [r(3) q(4)] + cached C.
struct Transformation {
  double p[7];
  double C[9];
  Transformation() { setIdentity(); }
  Transformation(const double* r, const double* q) { set(r, q); }
  explicit Transformation(const double* pose7) { set(pose7, pose7 + 3); }
  void setIdentity() {
    p[0] = p[1] = p[2] = 0; p[3] = p[4] = p[5] = 0; p[6] = 1;
    std::memset(C, 0, sizeof(C)); C[0] = C[4] = C[8] = 1;
  }
  // Transformation.hpp:167-171 (q normalised on set)
  void set(const double* r, const double* q) {
    p[0] = r[0]; p[1] = r[1]; p[2] = r[2];
    p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
    qnormalize(p + 3);
    q2R(p + 3, C);
  }
  const double* r() const { return p; }
  const double* q() const { return p + 3; }
  // Transformation.hpp:146
  Transformation inverse() const {
    double ri[3], qi[4];
    mat3T_vec(C, p, ri);
    ri[0] = -ri[0]; ri[1] = -ri[1]; ri[2] = -ri[2];
    qinv(p + 3, qi);
    return Transformation(ri, qi);
  }
  // Transformation.hpp:181-183
  Transformation operator*(const Transformation& rhs) const {
    double rr[3], qq[4];
    mat3_vec(C, rhs.p, rr);
    rr[0] += p[0]; rr[1] += p[1]; rr[2] += p[2];
    qmul(p + 3, rhs.p + 3, qq);
    return Transformation(rr, qq);
  }
  void mulh(const double* hp, double* out) const {  // :185-191
    const double s = hp[3];
    mat3_vec(C, hp, out);
    out[0] += p[0] * s; out[1] += p[1] * s; out[2] += p[2] * s;
    out[3] = s;
  }
  // Transformation.hpp:206-217
  void oplus(const double* delta) {
    p[0] += delta[0]; p[1] += delta[1]; p[2] += delta[2];
    double dq[4], qn[4];
    deltaQ(delta + 3, dq);
    qmul(dq, p + 3, qn);
    std::memcpy(p + 3, qn, sizeof(qn));
    qnormalize(p + 3);
    q2R(p + 3, C);
  }
};

// ---------------------------------------------------------------- dynamic dense helpers
// Cholesky (lower) mirroring Eigen's unblocked LLT: returns index of first
// non-positive pivot or -1.  On early exit the remainder of `A` is left in place
// ("singular information" quirk).
inline int llt_inplace(double* A, int n) {
  for (int k = 0; k < n; ++k) {
    double x = A[k * n + k];
    for (int j = 0; j < k; ++j) x -= A[k * n + j] * A[k * n + j];
    if (x <= 0) return k;
    x = std::sqrt(x);
    A[k * n + k] = x;
    for (int i = k + 1; i < n; ++i) {
      double s = A[i * n + k];
      for (int j = 0; j < k; ++j) s -= A[i * n + j] * A[k * n + j];
      A[i * n + k] = s / x;
    }
  }
  return -1;
}
// squareRootInformation = L^T with L from LLT(information) (e.g. PoseError.cpp:70-76).
// Mirrors `lltOfInformation.matrixL().transpose()`: lower triangle of the (possibly
// partially factorised) working matrix, transposed; strict upper part of L is zero.
inline void sqrt_information_upper(const double* info, int n, double* sqrtInfo) {
  std::vector<double> A(info, info + n * n);
  llt_inplace(A.data(), n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) sqrtInfo[i * n + j] = (j >= i) ? A[j * n + i] : 0.0;
}
// General inverse by LU with partial pivoting (Eigen PartialPivLU::inverse()).
inline bool lu_inverse(const double* Ain, int n, double* inv) {
  std::vector<double> A(Ain, Ain + n * n);
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[i * n + k]) > best) { best = std::fabs(A[i * n + k]); piv = i; }
    if (best == 0) return false;
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[piv * n + j]);
      std::swap(perm[k], perm[piv]);
    }
    for (int i = k + 1; i < n; ++i) {
      A[i * n + k] /= A[k * n + k];
      const double f = A[i * n + k];
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
    }
  }
  for (int c = 0; c < n; ++c) {
    std::vector<double> y(n);
    for (int i = 0; i < n; ++i) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= A[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * inv[j * n + c];
      inv[i * n + c] = s / A[i * n + i];
    }
  }
  return true;
}

// Symmetric eigendecomposition (ascending eigenvalues, eigenvectors in columns of V),
// Householder tridiagonalisation + implicit QL -- the published EISPACK tred2/tql2
// algorithm; stands in for Eigen::SelfAdjointEigenSolver (third-party, not vendored).
void sym_eig(const double* A, int n, double* evals, double* V);

}  // namespace orc
