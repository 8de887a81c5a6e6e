// svin_amd device math: SE(3)/quaternion helpers, pinhole + distortion models and the
// reprojection residual with analytic minimal Jacobians, written for CDNA4 lanes
// (straight-line FP64, no local arrays indexed dynamically, no heap).
//
// Reference arithmetic this mirrors (paths under /root/reference/okvis_ros/okvis/):
//   okvis_kinematics/include/okvis/kinematics/operators.hpp:63-135
//   okvis_kinematics/include/okvis/kinematics/implementation/Transformation.hpp:47-253
//   okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:143-212,:332-348 (+ distortions)
//   okvis_ceres/include/okvis/ceres/implementation/ReprojectionError.hpp:85-229
//
// SVIN_HD expands to __host__ __device__ under hipcc and to nothing under g++ so that the
// host-side unit tests in tests/ can exercise the exact device functions without a GPU.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SVIN_HD __host__ __device__ __forceinline__
#else
#define SVIN_HD inline
#endif

namespace svin {

enum : int { DIST_NONE = 0, DIST_RADTAN = 1, DIST_EQUIDISTANT = 2, DIST_RADTAN8 = 3 };

struct CameraModel {  // 16 doubles
  double fu, fv, cu, cv;
  double k[8];
  int model;
  int width, height, pad;
  double pad2;
};

struct Vec3 { double x, y, z; };
struct Quat { double x, y, z, w; };
struct Mat3 { double m[9]; };

SVIN_HD Mat3 quatToR(const Quat& q) {  // Eigen toRotationMatrix, no normalisation
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3 R;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz; R.m[2] = txz + twy;
  R.m[3] = txy + twz; R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy; R.m[7] = tyz + twx; R.m[8] = 1 - (txx + tyy);
  return R;
}
SVIN_HD Quat qmul(const Quat& a, const Quat& b) {
  Quat o;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  o.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return o;
}
SVIN_HD Quat qinv(const Quat& q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  Quat o;
  if (n2 > 0) { o.x = -q.x / n2; o.y = -q.y / n2; o.z = -q.z / n2; o.w = q.w / n2; }
  else { o.x = o.y = o.z = o.w = 0; }
  return o;
}
SVIN_HD Quat qnormalized(const Quat& q) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Quat o = {q.x / n, q.y / n, q.z / n, q.w / n};
  return o;
}
SVIN_HD double sinc(double x) {
  if (fabs(x) > 1e-6) return sin(x) / x;
  const double x2 = x * x, x4 = x2 * x2, x6 = x2 * x2 * x2;
  return 1.0 - (1.0 / 6.0) * x2 + (1.0 / 120.0) * x4 - (1.0 / 5040.0) * x6;
}
SVIN_HD Quat deltaQ(double ax, double ay, double az) {
  const double halfnorm = 0.5 * sqrt(ax * ax + ay * ay + az * az);
  const double s = sinc(halfnorm) * 0.5;
  Quat q = {s * ax, s * ay, s * az, cos(halfnorm)};
  return q;
}
SVIN_HD Vec3 rotate(const Mat3& R, const Vec3& v) {
  Vec3 o = {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
  return o;
}
SVIN_HD Vec3 rotateT(const Mat3& R, const Vec3& v) {
  Vec3 o = {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
            R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
  return o;
}

// pose retraction (Transformation::oplus): x = [r(3) q(4)], delta(6)
SVIN_HD void poseOplus(const double* x, const double* delta, double* xo) {
  xo[0] = x[0] + delta[0]; xo[1] = x[1] + delta[1]; xo[2] = x[2] + delta[2];
  // Transformation(r,q) constructor normalises first (PoseManifold::plus builds a Transformation)
  Quat q = qnormalized(Quat{x[3], x[4], x[5], x[6]});
  Quat dq = deltaQ(delta[3], delta[4], delta[5]);
  Quat qn = qnormalized(qmul(dq, q));
  xo[3] = qn.x; xo[4] = qn.y; xo[5] = qn.z; xo[6] = qn.w;
}
// PoseManifold::minus: delta = [xp.r - x.r ; 2 vec(xp.q * x.q^-1)]
SVIN_HD void poseMinus(const double* xp, const double* x, double* d) {
  d[0] = xp[0] - x[0]; d[1] = xp[1] - x[1]; d[2] = xp[2] - x[2];
  Quat dq = qmul(Quat{xp[3], xp[4], xp[5], xp[6]}, qinv(Quat{x[3], x[4], x[5], x[6]}));
  d[3] = 2 * dq.x; d[4] = 2 * dq.y; d[5] = 2 * dq.z;
}

// top-left 3x3 blocks of the quaternion product matrices (operators.hpp:91-133): plus(q) p = q * p, oplus(q) p = p * q
SVIN_HD void quatPlusMat3(const Quat& q, double* Q) {
  Q[0] = q.w; Q[1] = -q.z; Q[2] = q.y; Q[3] = q.z; Q[4] = q.w; Q[5] = -q.x; Q[6] = -q.y; Q[7] = q.x; Q[8] = q.w;
}
SVIN_HD void quatOplusMat3(const Quat& q, double* Q) {
  Q[0] = q.w; Q[1] = q.z; Q[2] = -q.y; Q[3] = -q.z; Q[4] = q.w; Q[5] = q.x; Q[6] = q.y; Q[7] = -q.x; Q[8] = q.w;
}
// PoseError (okvis_ceres/src/PoseError.cpp:87-132) before weighting: e = [meas.r - x.r ; 2 vec(meas.q * x.q^-1)] with both
// quaternions normalised the way Transformation's constructor does, F = de/d(delta) = -[I 0; 0 plus(dq)(0:3, 0:3)]
// (6x6 row-major, written completely).  Shared by the factor kernel (kind F_POSE_PRIOR) and svin_host_pose_error.
SVIN_HD void poseErrorEval(const double* meas, const double* x, double* e, double* F) {
  const Quat qm = qnormalized(Quat{meas[3], meas[4], meas[5], meas[6]});
  const Quat qx = qnormalized(Quat{x[3], x[4], x[5], x[6]});
  const Quat dq = qnormalized(qmul(qm, qnormalized(qinv(qx))));
  e[0] = meas[0] - x[0]; e[1] = meas[1] - x[1]; e[2] = meas[2] - x[2];
  e[3] = 2 * dq.x; e[4] = 2 * dq.y; e[5] = 2 * dq.z;
  double Q[9];
  quatPlusMat3(dq, Q);
  for (int k = 0; k < 36; ++k) F[k] = 0.0;
  for (int a = 0; a < 3; ++a) {
    F[a * 6 + a] = -1.0;
    for (int b = 0; b < 3; ++b) F[(3 + a) * 6 + 3 + b] = -Q[a * 3 + b];
  }
}

// Host-side data preparation (factor construction, not per-iteration arithmetic): the upper square-root information
// L^T of information = L L^T with Eigen::LLT's behaviour on a non-positive pivot -- it returns at that pivot and what
// matrixL() then reads is the partially factorised matrix (PoseError.cpp:70-76 on diag(1e4,1e4,1e4,0,0,1e8); SURVEY.md
// section 7).  n <= 15, out = n x n row-major.
inline void sqrtInformationUpper(const double* info, int n, double* out) {
  double A[225];
  for (int k = 0; k < n * n; ++k) A[k] = info[k];
  for (int k = 0; k < n; ++k) {
    double x = A[k * n + k];
    for (int j = 0; j < k; ++j) x -= A[k * n + j] * A[k * n + j];
    if (x <= 0) break;
    x = sqrt(x);
    A[k * n + k] = x;
    for (int i = k + 1; i < n; ++i) {
      double s = A[i * n + k];
      for (int j = 0; j < k; ++j) s -= A[i * n + j] * A[k * n + j];
      A[i * n + k] = s / x;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) out[i * n + j] = (j >= i) ? A[j * n + i] : 0.0;
}

// ---------------------------------------------------------------- distortion + projection
// d = distort(u), Jd = dd/du (row-major 2x2).  Returns false when the model rejects the point.
SVIN_HD bool distortPoint(const CameraModel& c, double u0, double u1, double& d0, double& d1, double* Jd) {
  if (c.model == DIST_RADTAN) {
    const double k1 = c.k[0], k2 = c.k[1], p1 = c.k[2], p2 = c.k[3];
    const double mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
    const double rad = k1 * rho + k2 * rho * rho;
    d0 = u0 + u0 * rad + 2.0 * p1 * mxy + p2 * (rho + 2.0 * mx);
    d1 = u1 + u1 * rad + 2.0 * p2 * mxy + p1 * (rho + 2.0 * my);
    Jd[0] = 1 + rad + k1 * 2.0 * mx + k2 * rho * 4 * mx + 2.0 * p1 * u1 + 6 * p2 * u0;
    Jd[2] = k1 * 2.0 * u0 * u1 + k2 * 4 * rho * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
    Jd[1] = Jd[2];
    Jd[3] = 1 + rad + k1 * 2.0 * my + k2 * rho * 4 * my + 6 * p1 * u1 + 2.0 * p2 * u0;
    return true;
  }
  if (c.model == DIST_EQUIDISTANT) {
    const double k1 = c.k[0], k2 = c.k[1], k3 = c.k[2], k4 = c.k[3];
    const double r = sqrt(u0 * u0 + u1 * u1);
    const double th = atan(r);
    const double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    const double thd = th * (1 + k1 * th2 + k2 * th4 + k3 * th6 + k4 * th8);
    if (r > 1e-8) {
      const double s = thd / r;
      d0 = s * u0; d1 = s * u1;
      const double dthd = 1 + 3 * k1 * th2 + 5 * k2 * th4 + 7 * k3 * th6 + 9 * k4 * th8;
      const double ds_over_r = (dthd * (1.0 / (1.0 + r * r)) * r - thd) / (r * r * r);
      Jd[0] = s + ds_over_r * u0 * u0;
      Jd[1] = ds_over_r * u0 * u1;
      Jd[2] = Jd[1];
      Jd[3] = s + ds_over_r * u1 * u1;
    } else {
      d0 = u0; d1 = u1;
      Jd[0] = 1; Jd[1] = 0; Jd[2] = 0; Jd[3] = 1;
    }
    return true;
  }
  if (c.model == DIST_RADTAN8) {
    const double k1 = c.k[0], k2 = c.k[1], p1 = c.k[2], p2 = c.k[3];
    const double k3 = c.k[4], k4 = c.k[5], k5 = c.k[6], k6 = c.k[7];
    const double mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
    if (rho > 9.0) { d0 = 0; d1 = 0; Jd[0] = Jd[1] = Jd[2] = Jd[3] = 0; return false; }
    const double num = 1.0 + ((k3 * rho + k2) * rho + k1) * rho;
    const double den = 1.0 + ((k6 * rho + k5) * rho + k4) * rho;
    const double rad = num / den;
    d0 = u0 * rad + 2.0 * p1 * mxy + p2 * (rho + 2.0 * mx);
    d1 = u1 * rad + 2.0 * p2 * mxy + p1 * (rho + 2.0 * my);
    const double dnum = k1 + rho * (2 * k2 + 3 * k3 * rho);
    const double dden = k4 + rho * (2 * k5 + 3 * k6 * rho);
    const double drad = (dnum * den - num * dden) / (den * den);
    Jd[0] = rad + 2 * drad * mx + 2.0 * p1 * u1 + 6.0 * p2 * u0;
    Jd[1] = 2 * drad * mxy + 2.0 * p1 * u0 + 2.0 * p2 * u1;
    Jd[2] = Jd[1];
    Jd[3] = rad + 2 * drad * my + 6.0 * p1 * u1 + 2.0 * p2 * u0;
    return true;
  }
  d0 = u0; d1 = u1;
  Jd[0] = 1; Jd[1] = 0; Jd[2] = 0; Jd[3] = 1;
  return true;
}

// projectHomogeneous (PinholeCamera.hpp:332-348): point (hx,hy,hz,hw); J3 = 2x3 w.r.t. the
// (possibly sign-flipped) head, returned un-negated exactly like the reference.
SVIN_HD void projectHomogeneous(const CameraModel& c, double hx, double hy, double hz, double hw, double& kx,
                                double& ky, double* J3) {
  if (hw < 0) { hx = -hx; hy = -hy; hz = -hz; }
  if (fabs(hz) < 1.0e-12) {  // reference: ProjectionStatus::Invalid, outputs untouched -> defined as zero here
    kx = 0; ky = 0;
    J3[0] = J3[1] = J3[2] = J3[3] = J3[4] = J3[5] = 0;
    return;
  }
  const double rz = 1.0 / hz, rz2 = rz * rz;
  double d0, d1, Jd[4];
  distortPoint(c, hx * rz, hy * rz, d0, d1, Jd);
  J3[0] = c.fu * Jd[0] * rz;
  J3[1] = c.fu * Jd[1] * rz;
  J3[2] = -c.fu * (hx * Jd[0] + hy * Jd[1]) * rz2;
  J3[3] = c.fv * Jd[2] * rz;
  J3[4] = c.fv * Jd[3] * rz;
  J3[5] = -c.fv * (hx * Jd[2] + hy * Jd[3]) * rz2;
  kx = c.fu * d0 + c.cu;
  ky = c.fv * d1 + c.cv;
}

// ---------------------------------------------------------------- reprojection residual
// Inputs: pose T_WS (7), landmark hp_W (4), extrinsics T_SC (7), measurement uv, isotropic
// square-root information w (= sqrt(64/size^2), Estimator.hpp:64-67).
// Outputs (all already multiplied by w): r(2), Jp 2x6, Jl 2x3, Je 2x6 (row-major).  Invalid
// points (|w_C|>1e-8 and z_C<0.2) keep their residual but get zero Jacobians (:140-147).
SVIN_HD void reprojEval(const CameraModel& cam, const double* T_WS, const double* hpW, const double* T_SC, double u,
                        double v, double w, double* r, double* Jp, double* Jl, double* Je) {
  const Mat3 C_WS = quatToR(Quat{T_WS[3], T_WS[4], T_WS[5], T_WS[6]});
  const Mat3 C_SC = quatToR(Quat{T_SC[3], T_SC[4], T_SC[5], T_SC[6]});
  const double hw = hpW[3];
  // hp_S = T_SW hp_W ;  hp_C = T_CS hp_S   (both keep the homogeneous scale hw)
  const Vec3 dW = {hpW[0] - T_WS[0] * hw, hpW[1] - T_WS[1] * hw, hpW[2] - T_WS[2] * hw};
  const Vec3 pS = rotateT(C_WS, dW);
  const Vec3 dS = {pS.x - T_SC[0] * hw, pS.y - T_SC[1] * hw, pS.z - T_SC[2] * hw};
  const Vec3 pC = rotateT(C_SC, dS);
  double kx, ky, J3[6];
  projectHomogeneous(cam, pC.x, pC.y, pC.z, hw, kx, ky, J3);
  r[0] = w * (u - kx);
  r[1] = w * (v - ky);
  bool valid = true;
  if (fabs(hw) > 1.0e-8) {
    if (pC.z / hw < 0.2) valid = false;
  }
  if (!valid) {
    for (int i = 0; i < 12; ++i) { Jp[i] = 0; Je[i] = 0; }
    for (int i = 0; i < 6; ++i) Jl[i] = 0;
    return;
  }
  // weighted projection Jacobian (2x3), the homogeneous column of Jh is zero
  double Jw[6];
  for (int i = 0; i < 6; ++i) Jw[i] = w * J3[i];
  // A = Jw * C_CS  (2x3),  C_CS = C_SC^T  ->  A[i][j] = sum_k Jw[i][k] C_SC[j][k]
  double A[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      A[i * 3 + j] = Jw[i * 3] * C_SC.m[j * 3] + Jw[i * 3 + 1] * C_SC.m[j * 3 + 1] + Jw[i * 3 + 2] * C_SC.m[j * 3 + 2];
  // B = A * C_SW (2x3), C_SW = C_WS^T  ->  B[i][j] = sum_k A[i][k] C_WS[j][k]
  double B[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      B[i * 3 + j] = A[i * 3] * C_WS.m[j * 3] + A[i * 3 + 1] * C_WS.m[j * 3 + 1] + A[i * 3 + 2] * C_WS.m[j * 3 + 2];
  // pose: J = Jw T_CS [C_SW*hw | -C_SW [p]x],  p = hp_W.head - t_WS*hw = dW   (:153-162)
  for (int i = 0; i < 2; ++i) {
    const double b0 = B[i * 3], b1 = B[i * 3 + 1], b2 = B[i * 3 + 2];
    Jp[i * 6 + 0] = b0 * hw; Jp[i * 6 + 1] = b1 * hw; Jp[i * 6 + 2] = b2 * hw;
    // -(b^T [p]x) = (p x b)^T ... row * crossMx(p): [b1*pz - b2*py, b2*px - b0*pz, b0*py - b1*px]; negate
    Jp[i * 6 + 3] = -(b1 * dW.z - b2 * dW.y);
    Jp[i * 6 + 4] = -(b2 * dW.x - b0 * dW.z);
    Jp[i * 6 + 5] = -(b0 * dW.y - b1 * dW.x);
    // landmark: J = -Jw * T_CW, minimal = first three columns (:181-195)
    Jl[i * 3 + 0] = -b0; Jl[i * 3 + 1] = -b1; Jl[i * 3 + 2] = -b2;
    // extrinsics: J = Jw [C_CS*hw_S | -C_CS [p_S]x], p_S = hp_S.head - t_SC*hw = dS   (:199-208)
    const double a0 = A[i * 3], a1 = A[i * 3 + 1], a2 = A[i * 3 + 2];
    Je[i * 6 + 0] = a0 * hw; Je[i * 6 + 1] = a1 * hw; Je[i * 6 + 2] = a2 * hw;
    Je[i * 6 + 3] = -(a1 * dS.z - a2 * dS.y);
    Je[i * 6 + 4] = -(a2 * dS.x - a0 * dS.z);
    Je[i * 6 + 5] = -(a0 * dS.y - a1 * dS.x);
  }
}

// Cauchy(1) loss (ceres loss_function.cc) on s = |r|^2: rho, rho', rho''
SVIN_HD void cauchyLoss(double s, double& rho0, double& rho1, double& rho2) {
  const double sum = 1.0 + s, inv = 1.0 / sum;
  rho0 = log(sum);
  rho1 = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308;
  rho2 = -(inv * inv);
}

}  // namespace svin
