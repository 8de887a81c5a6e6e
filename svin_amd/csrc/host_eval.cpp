// the CPU consumers of the backend's math that stay on the CPU.
//
// Two callers of okvis_ceres live OUTSIDE the window solve and run at sensor rate on the host:
//   * ImuError::propagation (okvis_ceres/src/ImuError.cpp:266-476, :479-697) -- called per IMU sample (100-200 Hz) by
//     ThreadedKFVio::imuConsumerLoop (okvis_multisensor_processing/src/ThreadedKFVio.cpp:808-819), per frame by
//     :599 and by Frontend.cpp:258.  A kernel launch + read-back per sample (~60 us) is the wrong tool there.
//   * ReprojectionError::EvaluateWithMinimalJacobians for ONE residual -- ProbabilisticStereoTriangulator.cpp:266-300
//     linearises two observations per candidate match to propagate the keypoint covariance.
// These are NOT a fall-back of the hot path (the window solve has none): they are the sequential, allocation-free
// twins the shim classes okvis::ceres::ImuError / ReprojectionError forward to.  The reprojection twin IS the device
// function (dmath.hpp reprojEval, compiled for the host); the propagation twin is the sequential form of the recurrences
// the device evaluates with wave-wide scans (kernels.hip imuIntegrate) and is held against it by
// tests/test_gpu_abi.py::test_host_evaluators_match_device.
#include "../../include/svin_ba.h"

#include <cmath>
#include <cstring>
#include <utility>

#include "dmath.hpp"

namespace {

using svin::Mat3;
using svin::Quat;
using svin::Vec3;

struct M3 {
  double m[9];
  static M3 zero() { M3 r; for (double& v : r.m) v = 0; return r; }
  static M3 from(const Mat3& a) { M3 r; std::memcpy(r.m, a.m, sizeof(r.m)); return r; }
};
inline M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return r;
}
inline M3 add(const M3& a, const M3& b, double sb = 1.0) {
  M3 r;
  for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + sb * b.m[i];
  return r;
}
inline M3 scaled(const M3& a, double s) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i]; return r; }
inline void mulv(const M3& a, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = a.m[i * 3] * v[0] + a.m[i * 3 + 1] * v[1] + a.m[i * 3 + 2] * v[2];
}
inline M3 crossMx(const double* v) {   // operators.hpp:63-67
  M3 r = M3::zero();
  r.m[1] = -v[2]; r.m[2] = v[1]; r.m[3] = v[2]; r.m[5] = -v[0]; r.m[6] = -v[1]; r.m[7] = v[0];
  return r;
}
inline M3 rightJacobian(const double* phi) {   // okvis_kinematics Transformation.hpp rightJacobian: series below 1e-4
  const double Phi = std::sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const M3 X = crossMx(phi), X2 = mul(X, X);
  double a = -0.5, b = 1.0 / 6.0;
  if (!(Phi < 1.0e-4)) {
    const double Phi2 = Phi * Phi, Phi3 = Phi2 * Phi;
    a = -(1.0 - std::cos(Phi)) / Phi2;
    b = (Phi - std::sin(Phi)) / Phi3;
  }
  M3 r;
  for (int i = 0; i < 9; ++i) r.m[i] = a * X.m[i] + b * X2.m[i];
  r.m[0] += 1; r.m[4] += 1; r.m[8] += 1;
  return r;
}
// okvis::Time arithmetic (okvis_time/include/okvis/Time.hpp:146): normalised (sec, nsec) difference -> seconds
inline double dtSec(uint32_t s1, uint32_t n1, uint32_t s0, uint32_t n0) {
  long long s = (long long)s1 - (long long)s0, ns = (long long)n1 - (long long)n0;
  while (ns < 0) { ns += 1000000000LL; s -= 1; }
  while (ns >= 1000000000LL) { ns -= 1000000000LL; s += 1; }
  return (double)s + 1e-9 * (double)ns;
}
inline bool timeLess(uint32_t s0, uint32_t n0, uint32_t s1, uint32_t n1) { return s0 < s1 || (s0 == s1 && n0 < n1); }

}  // namespace

extern "C" {

int svin_host_imu_propagation(const svin_imu_sample* imu, int n, const svin_imu_params* par, double T[7], double sb[9], uint32_t sec0,
                              uint32_t nsec0, uint32_t sec1, uint32_t nsec1, double* cov, double* jac, double* integrals) {
  if (!imu || n <= 0 || !par || !T || !sb) return SVIN_ERR_INVALID_ARG;
  // :279 -- the deque has to reach the end of the interval
  if (timeLess(imu[n - 1].sec, imu[n - 1].nsec, sec1, nsec1)) return -1;
  uint32_t ts = sec0, tn = nsec0;   // `time`
  const Quat q0 = svin::qnormalized(Quat{T[3], T[4], T[5], T[6]});
  const M3 C0 = M3::from(svin::quatToR(q0));
  Quat Dq = {0, 0, 0, 1};
  M3 Ci = M3::zero(), Cdi = M3::zero(), cross = M3::zero(), dal = M3::zero(), dv = M3::zero(), dp = M3::zero();
  double ai[3] = {0, 0, 0}, adi[3] = {0, 0, 0};
  double P[225];
  for (double& v : P) v = 0;
  double Delta_t = 0;
  bool started = false;
  int used = 0;
  for (int k = 0; k < n; ++k) {
    double w0[3], a0[3], w1[3], a1[3];
    const svin_imu_sample& m0 = imu[k];
    const bool last = (k + 1 == n);
    const svin_imu_sample& m1 = imu[last ? k : k + 1];   // (it + 1) past the end is never used: the loop leaves at t_end before
    for (int c = 0; c < 3; ++c) { w0[c] = m0.gyr[c]; a0[c] = m0.acc[c]; w1[c] = m1.gyr[c]; a1[c] = m1.acc[c]; }
    uint32_t ns_ = last ? sec1 : m1.sec, nn_ = last ? nsec1 : m1.nsec;   // nexttime
    double dt = dtSec(ns_, nn_, ts, tn);
    if (timeLess(sec1, nsec1, ns_, nn_)) {   // the sample interval straddles t_end: interpolate the right end (:319-326)
      const double interval = dtSec(ns_, nn_, m0.sec, m0.nsec);
      ns_ = sec1; nn_ = nsec1;
      dt = dtSec(ns_, nn_, ts, tn);
      const double r = dt / interval;
      for (int c = 0; c < 3; ++c) { w1[c] = (1.0 - r) * w0[c] + r * w1[c]; a1[c] = (1.0 - r) * a0[c] + r * a1[c]; }
    }
    if (dt <= 0.0) continue;
    Delta_t += dt;
    if (!started) {   // first used interval: interpolate the left end (:333-338)
      started = true;
      const double r = dt / dtSec(ns_, nn_, m0.sec, m0.nsec);
      for (int c = 0; c < 3; ++c) { w0[c] = r * w0[c] + (1.0 - r) * w1[c]; a0[c] = r * a0[c] + (1.0 - r) * a1[c]; }
    }
    double sigma_g_c = par->sigma_g_c, sigma_a_c = par->sigma_a_c;   // saturation (:341-356)
    bool gs = false, as = false;
    for (int c = 0; c < 3; ++c) {
      gs = gs || std::fabs(w0[c]) > par->g_max || std::fabs(w1[c]) > par->g_max;
      as = as || std::fabs(a0[c]) > par->a_max || std::fabs(a1[c]) > par->a_max;
    }
    if (gs) sigma_g_c *= 100;
    if (as) sigma_a_c *= 100;
    double wt[3], at[3];
    for (int c = 0; c < 3; ++c) { wt[c] = 0.5 * (w0[c] + w1[c]) - sb[3 + c]; at[c] = 0.5 * (a0[c] + a1[c]) - sb[6 + c]; }
    const double th = std::sqrt(wt[0] * wt[0] + wt[1] * wt[1] + wt[2] * wt[2]) * 0.5 * dt;
    const double sc = svin::sinc(th) * 0.5 * dt;
    const Quat dq = {sc * wt[0], sc * wt[1], sc * wt[2], std::cos(th)};
    const Quat Dq1 = svin::qmul(Dq, dq);
    const M3 C = M3::from(svin::quatToR(Dq)), C1 = M3::from(svin::quatToR(Dq1)), Cs = add(C, C1);
    double Csa[3];
    mulv(Cs, at, Csa);
    const M3 Ci1 = add(Ci, Cs, 0.5 * dt);
    double ai1[3];
    for (int c = 0; c < 3; ++c) ai1[c] = ai[c] + 0.5 * Csa[c] * dt;
    Cdi = add(add(Cdi, Ci, dt), Cs, 0.25 * dt * dt);
    double adiStep[3];
    for (int c = 0; c < 3; ++c) { adiStep[c] = ai[c] * dt + 0.25 * Csa[c] * dt * dt; adi[c] += adiStep[c]; }
    dal = add(dal, C1, dt);                                   // :384 (propagation flavour: dt * C_1)
    const double wdt[3] = {wt[0] * dt, wt[1] * dt, wt[2] * dt};
    const M3 cross1 = add(mul(M3::from(svin::quatToR(svin::qinv(dq))), cross), rightJacobian(wdt), dt);
    const M3 ax = crossMx(at);
    const M3 mix = add(mul(mul(C, ax), cross), mul(mul(C1, ax), cross1));   // C a_x cross + C_1 a_x cross_1
    const M3 dv1 = add(dv, mix, 0.5 * dt);
    const M3 dpStep = add(scaled(dv, dt), mix, 0.25 * dt * dt);
    dp = add(dp, dpStep);
    if (cov) {   // :394-435
      double F[225];
      for (int i = 0; i < 225; ++i) F[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
      auto setB = [&](int r0, int c0, const M3& B, double s) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) F[(r0 + a) * 15 + c0 + b] = s * B.m[a * 3 + b];
      };
      setB(0, 3, crossMx(adiStep), -1.0);
      M3 I3 = M3::zero(); I3.m[0] = I3.m[4] = I3.m[8] = 1.0;
      setB(0, 6, I3, dt);
      setB(0, 9, dpStep, 1.0);
      setB(0, 12, add(scaled(Ci, -dt), Cs, 0.25 * dt * dt), 1.0);   // the reference's sign: -C_integral*dt + 0.25 (C + C_1) dt^2
      setB(3, 9, C1, -dt);
      double h[3] = {0.5 * Csa[0] * dt, 0.5 * Csa[1] * dt, 0.5 * Csa[2] * dt};
      setB(6, 3, crossMx(h), -1.0);
      setB(6, 9, mix, 0.5 * dt);
      setB(6, 12, Cs, -0.5 * dt);
      double FP[225];
      for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) {
          double s = 0;
          for (int q = 0; q < 15; ++q) s += F[i * 15 + q] * P[q * 15 + j];
          FP[i * 15 + j] = s;
        }
      for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) {
          double s = 0;
          for (int q = 0; q < 15; ++q) s += FP[i * 15 + q] * F[j * 15 + q];
          P[i * 15 + j] = s;
        }
      const double s2a = dt * sigma_g_c * sigma_g_c, s2v = dt * sigma_a_c * par->sigma_a_c, s2p = 0.5 * dt * dt * s2v;
      const double s2bg = dt * par->sigma_gw_c * par->sigma_gw_c, s2ba = dt * par->sigma_aw_c * par->sigma_aw_c;
      for (int c = 0; c < 3; ++c) {
        P[(3 + c) * 16] += s2a; P[(6 + c) * 16] += s2v; P[c * 16] += s2p; P[(9 + c) * 16] += s2bg; P[(12 + c) * 16] += s2ba;
      }
    }
    Dq = Dq1; Ci = Ci1; cross = cross1; dv = dv1;
    for (int c = 0; c < 3; ++c) ai[c] = ai1[c];
    ts = ns_; tn = nn_;
    ++used;
    if (ns_ == sec1 && nn_ == nsec1) break;
  }
  // :452-458
  const double gz = par->g * (6371009.0 / std::sqrt(6371009.0 * 6371009.0));
  const double gW[3] = {par->g * 0.0, par->g * 0.0, gz};
  double c2[3], c1[3];
  mulv(C0, adi, c2);
  mulv(C0, ai, c1);
  for (int c = 0; c < 3; ++c) T[c] = T[c] + sb[c] * Delta_t + c2[c] - 0.5 * gW[c] * Delta_t * Delta_t;
  const Quat qn = svin::qnormalized(svin::qmul(q0, Dq));
  T[3] = qn.x; T[4] = qn.y; T[5] = qn.z; T[6] = qn.w;
  for (int c = 0; c < 3; ++c) sb[c] = sb[c] + c1[c] - gW[c] * Delta_t;
  if (integrals) {   // second overload (:664-667)
    for (int c = 0; c < 3; ++c) { integrals[c] = adi[c]; integrals[3 + c] = ai[c]; }
    integrals[6] = Delta_t;
  }
  if (jac) {   // :461-472
    for (int i = 0; i < 225; ++i) jac[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
    auto setB = [&](int r0, int c0v, const M3& B, double s) {
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) jac[(r0 + a) * 15 + c0v + b] = s * B.m[a * 3 + b];
    };
    M3 I3 = M3::zero(); I3.m[0] = I3.m[4] = I3.m[8] = 1.0;
    setB(0, 3, crossMx(c2), -1.0);
    setB(0, 6, I3, Delta_t);
    setB(0, 9, mul(C0, dp), 1.0);
    setB(0, 12, mul(C0, Cdi), -1.0);
    setB(3, 9, mul(C0, dal), -1.0);
    setB(6, 3, crossMx(c1), -1.0);
    setB(6, 9, mul(C0, dv), 1.0);
    setB(6, 12, mul(C0, Ci), -1.0);
  }
  if (cov) {   // :475-483: P = T P_delta T^T, T = blockdiag(C, C, C, I, I)
    double Tm[225], TP[225];
    for (int i = 0; i < 225; ++i) Tm[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
    for (int blk = 0; blk < 3; ++blk)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Tm[(3 * blk + a) * 15 + 3 * blk + b] = C0.m[a * 3 + b];
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0;
        for (int q = 0; q < 15; ++q) s += Tm[i * 15 + q] * P[q * 15 + j];
        TP[i * 15 + j] = s;
      }
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0;
        for (int q = 0; q < 15; ++q) s += TP[i * 15 + q] * Tm[j * 15 + q];
        cov[i * 15 + j] = s;
      }
  }
  return used;
}

int svin_host_reprojection_error(int model, const double intr[4], const double* dist, int n_dist, const double T_WS[7],
                                 const double hp_W[4], const double T_SC[7], const double uv[2], const double information[4],
                                 double residual[2], double* J_pose_min, double* J_lm_min, double* J_ext_min, double* J_pose,
                                 double* J_lm, double* J_ext) {
  if (!intr || !T_WS || !hp_W || !T_SC || !uv || !information || !residual || n_dist < 0 || n_dist > 8 || (n_dist > 0 && !dist))
    return SVIN_ERR_INVALID_ARG;
  svin::CameraModel cam;
  std::memset(&cam, 0, sizeof(cam));
  cam.fu = intr[0]; cam.fv = intr[1]; cam.cu = intr[2]; cam.cv = intr[3];
  for (int i = 0; i < n_dist; ++i) cam.k[i] = dist[i];
  cam.model = model;
  // unweighted residual and Jacobians, then the 2x2 square-root information L^T of information = L L^T
  // (ReprojectionErrorBase::setInformation: Eigen LLT, upper factor)
  double r[2], Jp[12], Jl[6], Je[12];
  svin::reprojEval(cam, T_WS, hp_W, T_SC, uv[0], uv[1], 1.0, r, Jp, Jl, Je);
  const double i00 = information[0], i10 = 0.5 * (information[1] + information[2]), i11 = information[3];
  if (!(i00 > 0)) return SVIN_ERR_INVALID_ARG;
  const double l00 = std::sqrt(i00), l10 = i10 / l00, t = i11 - l10 * l10;
  if (!(t > 0)) return SVIN_ERR_INVALID_ARG;
  const double l11 = std::sqrt(t);
  // W = L^T = [l00 l10; 0 l11]
  auto weight = [&](double* M, int cols) {
    for (int c = 0; c < cols; ++c) {
      const double a = M[c], b = M[cols + c];
      M[c] = l00 * a + l10 * b;
      M[cols + c] = l11 * b;
    }
  };
  weight(r, 1); weight(Jp, 6); weight(Jl, 3); weight(Je, 6);
  residual[0] = r[0]; residual[1] = r[1];
  if (J_pose_min) std::memcpy(J_pose_min, Jp, sizeof(Jp));
  if (J_lm_min) std::memcpy(J_lm_min, Jl, sizeof(Jl));
  if (J_ext_min) std::memcpy(J_ext_min, Je, sizeof(Je));
  // ambient Jacobians: J = J_min * lift(x), lift = [I3 0; 0 2 oplus(q^-1)[0:3, :]] (PoseManifold.cpp:128-140);
  // homogeneous point: [J_min | 0] (HomogeneousPointManifold.cpp:117-135)
  auto liftPose = [&](const double* Jm, const double* x, double* J) {
    const Quat qi = Quat{-x[3], -x[4], -x[5], x[6]};   // the reference takes the conjugate (PoseManifold.cpp:131)
    // oplus(q) rows 0..2 (operators.hpp:112-133)
    const double O[12] = {qi.w, qi.z, -qi.y, qi.x, -qi.z, qi.w, qi.x, qi.y, qi.y, -qi.x, qi.w, qi.z};
    for (int a = 0; a < 2; ++a) {
      for (int c = 0; c < 3; ++c) J[a * 7 + c] = Jm[a * 6 + c];
      for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int q = 0; q < 3; ++q) s += Jm[a * 6 + 3 + q] * 2.0 * O[q * 4 + c];
        J[a * 7 + 3 + c] = s;
      }
    }
  };
  if (J_pose) liftPose(Jp, T_WS, J_pose);
  if (J_ext) liftPose(Je, T_SC, J_ext);
  if (J_lm) {
    // ambient 2x4: J = -Jh_w T_CS T_SW (ReprojectionError.hpp:181-195).  Its first three columns are the minimal ones; the
    // fourth is the derivative with respect to the homogeneous scale, -Jh_w[:, 0:3] (T_CW)[0:3, 3] = B t_WS + A t_SC with
    // A = Jh_w C_CS, B = A C_SW (the blocks reprojEval builds internally; rebuilt here from the same projection)
    const Mat3 C_WS = svin::quatToR(Quat{T_WS[3], T_WS[4], T_WS[5], T_WS[6]}), C_SC = svin::quatToR(Quat{T_SC[3], T_SC[4], T_SC[5], T_SC[6]});
    const double hw = hp_W[3];
    const Vec3 pS = svin::rotateT(C_WS, Vec3{hp_W[0] - T_WS[0] * hw, hp_W[1] - T_WS[1] * hw, hp_W[2] - T_WS[2] * hw});
    const Vec3 pC = svin::rotateT(C_SC, Vec3{pS.x - T_SC[0] * hw, pS.y - T_SC[1] * hw, pS.z - T_SC[2] * hw});
    double kx, ky, J3[6];
    svin::projectHomogeneous(cam, pC.x, pC.y, pC.z, hw, kx, ky, J3);
    const bool valid = !(std::fabs(hw) > 1.0e-8 && pC.z / hw < 0.2);
    double A[6], B[6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) A[i * 3 + j] = J3[i * 3] * C_SC.m[j * 3] + J3[i * 3 + 1] * C_SC.m[j * 3 + 1] + J3[i * 3 + 2] * C_SC.m[j * 3 + 2];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) B[i * 3 + j] = A[i * 3] * C_WS.m[j * 3] + A[i * 3 + 1] * C_WS.m[j * 3 + 1] + A[i * 3 + 2] * C_WS.m[j * 3 + 2];
    double col[2];
    for (int i = 0; i < 2; ++i)
      col[i] = valid ? (B[i * 3] * T_WS[0] + B[i * 3 + 1] * T_WS[1] + B[i * 3 + 2] * T_WS[2] + A[i * 3] * T_SC[0] + A[i * 3 + 1] * T_SC[1] + A[i * 3 + 2] * T_SC[2]) : 0.0;
    weight(col, 1);
    for (int a = 0; a < 2; ++a) { for (int c = 0; c < 3; ++c) J_lm[a * 4 + c] = Jl[a * 3 + c]; J_lm[a * 4 + 3] = col[a]; }
  }
  return 1;
}

int svin_host_homogeneous_point_error(const double hp[4], const double meas[4], const double info[9], double residual[3],
                                      double* J_min, double* J) {
  if (!hp || !meas || !info || !residual) return -1;
  // squareRootInformation_ = L^T, information = L L^T (HomogeneousPointError.cpp:66-75)
  double L[9] = {0};
  for (int j = 0; j < 3; ++j) {
    double dsum = info[j * 3 + j];
    for (int k = 0; k < j; ++k) dsum -= L[j * 3 + k] * L[j * 3 + k];
    if (!(dsum > 0)) return 0;
    L[j * 3 + j] = std::sqrt(dsum);
    for (int i = j + 1; i < 3; ++i) {
      double v = info[i * 3 + j];
      for (int k = 0; k < j; ++k) v -= L[i * 3 + k] * L[j * 3 + k];
      L[i * 3 + j] = v / L[j * 3 + j];
    }
  }
  const double e[3] = {hp[0] - meas[0], hp[1] - meas[1], hp[2] - meas[2]};   // HomogeneousPointManifold::minus
  for (int a = 0; a < 3; ++a) {
    residual[a] = L[0 * 3 + a] * e[0] + L[1 * 3 + a] * e[1] + L[2 * 3 + a] * e[2];
    for (int b = 0; b < 3; ++b) {
      if (J_min) J_min[a * 3 + b] = L[b * 3 + a];
      if (J) J[a * 4 + b] = L[b * 3 + a];
    }
    if (J) J[a * 4 + 3] = 0.0;
  }
  return 1;
}


// ---- PoseError, the parameter-block manifolds: what okvis_frontend's ProbabilisticStereoTriangulator (:87-99,
// :128-140) and the Ceres-free shim classes under integration/okvis/ceres/ evaluate on the CPU.
int svin_host_pose_information(const double information[36], double* sqrt_information, double* covariance) {
  if (!information) return SVIN_ERR_INVALID_ARG;
  // squareRootInformation_ = LLT(information).matrixL().transpose() (PoseError.cpp:70-76) with Eigen's behaviour on a
  // non-positive pivot: the factorisation stops there and the untouched remainder of the matrix is read as the factor
  if (sqrt_information) svin::sqrtInformationUpper(information, 6, sqrt_information);
  if (covariance) {
    // covariance_ = information.inverse() (PoseError.cpp:72): Eigen's fixed-size 6x6 inverse is PartialPivLU; a singular
    // information matrix (the reference passes diag(1e4,1e4,1e4,0,0,1e8), Estimator.cpp:189-192) gives inf / nan there
    // and here alike -- nothing reads covariance() in that case
    double A[36], B[36];
    for (int k = 0; k < 36; ++k) { A[k] = information[k]; B[k] = (k % 7 == 0) ? 1.0 : 0.0; }
    for (int c = 0; c < 6; ++c) {
      int piv = c;
      for (int r = c + 1; r < 6; ++r)
        if (std::fabs(A[r * 6 + c]) > std::fabs(A[piv * 6 + c])) piv = r;
      if (piv != c)
        for (int k = 0; k < 6; ++k) { std::swap(A[c * 6 + k], A[piv * 6 + k]); std::swap(B[c * 6 + k], B[piv * 6 + k]); }
      const double d = A[c * 6 + c];
      for (int r = c + 1; r < 6; ++r) {
        const double f = A[r * 6 + c] / d;
        for (int k = c; k < 6; ++k) A[r * 6 + k] -= f * A[c * 6 + k];
        for (int k = 0; k < 6; ++k) B[r * 6 + k] -= f * B[c * 6 + k];
      }
    }
    for (int k = 0; k < 6; ++k)
      for (int r = 5; r >= 0; --r) {
        double v = B[r * 6 + k];
        for (int c = r + 1; c < 6; ++c) v -= A[r * 6 + c] * covariance[c * 6 + k];
        covariance[r * 6 + k] = v / A[r * 6 + r];
      }
  }
  return 1;
}

static void poseLift(const double* x, double* L) {   // PoseManifold::liftJacobian (PoseManifold.cpp:128-140), 6x7 row-major
  for (int k = 0; k < 42; ++k) L[k] = 0.0;
  L[0] = L[8] = L[16] = 1.0;
  const double qx = -x[3], qy = -x[4], qz = -x[5], qw = x[6];   // the conjugate, not normalised (:131)
  const double O[12] = {qw, qz, -qy, qx, -qz, qw, qx, qy, qy, -qx, qw, qz};   // rows 0..2 of oplus(q^-1)
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 4; ++c) L[(3 + a) * 7 + 3 + c] = 2.0 * O[a * 4 + c];
}

int svin_host_pose_error(const double measurement[7], const double sqrt_information[36], const double T_WS[7], double residual[6],
                         double* J_min, double* J) {
  if (!measurement || !sqrt_information || !T_WS || !residual) return SVIN_ERR_INVALID_ARG;
  double e[6], F[36], WF[36];
  svin::poseErrorEval(measurement, T_WS, e, F);
  for (int a = 0; a < 6; ++a) {
    double s = 0;
    for (int k = 0; k < 6; ++k) s += sqrt_information[a * 6 + k] * e[k];
    residual[a] = s;
  }
  if (!J_min && !J) return 1;
  for (int a = 0; a < 6; ++a)
    for (int c = 0; c < 6; ++c) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += sqrt_information[a * 6 + k] * F[k * 6 + c];
      WF[a * 6 + c] = s;
    }
  if (J_min) std::memcpy(J_min, WF, sizeof(WF));
  if (J) {   // J0 = J0_minimal * J_lift (PoseError.cpp:116-120)
    double L[42];
    poseLift(T_WS, L);
    for (int a = 0; a < 6; ++a)
      for (int c = 0; c < 7; ++c) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += WF[a * 6 + k] * L[k * 7 + c];
        J[a * 7 + c] = s;
      }
  }
  return 1;
}

// the tangent directions of the pose manifolds inside the 6-vector [dr ; dalpha] (PoseManifold.cpp:59-82, :172-196 3d,
// :262-288 4d, :350-374 2d)
static int poseTangent(int kind, int idx[6]) {
  switch (kind) {
    case SVIN_MANIFOLD_POSE6D: for (int k = 0; k < 6; ++k) idx[k] = k; return 6;
    case SVIN_MANIFOLD_POSE3D: idx[0] = 3; idx[1] = 4; idx[2] = 5; return 3;
    case SVIN_MANIFOLD_POSE4D: idx[0] = 0; idx[1] = 1; idx[2] = 2; idx[3] = 5; return 4;
    case SVIN_MANIFOLD_POSE2D: idx[0] = 3; idx[1] = 4; return 2;
  }
  return 0;
}

int svin_host_manifold_dims(int kind, int* ambient, int* tangent) {
  int idx[6];
  int a = 7, t = poseTangent(kind, idx);
  if (kind == SVIN_MANIFOLD_HPOINT) { a = 4; t = 3; }
  else if (t == 0) return SVIN_ERR_INVALID_ARG;
  if (ambient) *ambient = a;
  if (tangent) *tangent = t;
  return 1;
}

int svin_host_manifold_plus(int kind, const double* x, const double* delta, double* x_plus_delta) {
  if (!x || !delta || !x_plus_delta) return SVIN_ERR_INVALID_ARG;
  if (kind == SVIN_MANIFOLD_HPOINT) {   // HomogeneousPointManifold.cpp:57-67, Euclidean
    for (int k = 0; k < 3; ++k) x_plus_delta[k] = x[k] + delta[k];
    x_plus_delta[3] = x[3] + 0.0;
    return 1;
  }
  int idx[6];
  const int t = poseTangent(kind, idx);
  if (!t) return SVIN_ERR_INVALID_ARG;
  double d6[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < t; ++k) d6[idx[k]] = delta[k];
  double out[7];
  svin::poseOplus(x, d6, out);   // Transformation::oplus, the retraction the device applies (dmath.hpp)
  std::memcpy(x_plus_delta, out, sizeof(out));
  return 1;
}

int svin_host_manifold_minus(int kind, const double* x_plus_delta, const double* x, double* delta) {
  if (!x || !delta || !x_plus_delta) return SVIN_ERR_INVALID_ARG;
  if (kind == SVIN_MANIFOLD_HPOINT) {   // HomogeneousPointManifold.cpp:80-91
    for (int k = 0; k < 3; ++k) delta[k] = x_plus_delta[k] - x[k];
    return 1;
  }
  int idx[6];
  const int t = poseTangent(kind, idx);
  if (!t) return SVIN_ERR_INVALID_ARG;
  double d6[6];
  svin::poseMinus(x_plus_delta, x, d6);   // PoseManifold.cpp:93-102 (the 3d / 4d / 2d forms keep a subset: :199-211, :295-305, :381-392)
  for (int k = 0; k < t; ++k) delta[k] = d6[idx[k]];
  return 1;
}

int svin_host_manifold_plus_jacobian(int kind, const double* x, double* J) {
  if (!x || !J) return SVIN_ERR_INVALID_ARG;
  if (kind == SVIN_MANIFOLD_HPOINT) {   // 4x3 (HomogeneousPointManifold.cpp:104-113)
    for (int k = 0; k < 12; ++k) J[k] = 0.0;
    J[0] = J[4] = J[8] = 1.0;
    return 1;
  }
  int idx[6];
  const int t = poseTangent(kind, idx);
  if (!t) return SVIN_ERR_INVALID_ARG;
  // full 7x6: [I 0; 0 oplus(q) * 0.5 (first three columns)].  6d and 4d go through Transformation::oplusJacobian, whose
  // q_ was normalised by the constructor (PoseManifold.cpp:105-111, :310-318); 3d and 2d use the raw quaternion (:216-227, :397-408)
  Quat q = Quat{x[3], x[4], x[5], x[6]};
  if (kind == SVIN_MANIFOLD_POSE6D || kind == SVIN_MANIFOLD_POSE4D) q = svin::qnormalized(q);
  const double O[16] = {q.w, q.z, -q.y, q.x, -q.z, q.w, q.x, q.y, q.y, -q.x, q.w, q.z, -q.x, -q.y, -q.z, q.w};
  double full[42];
  for (int k = 0; k < 42; ++k) full[k] = 0.0;
  full[0] = full[7] = full[14] = 1.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) full[(3 + i) * 6 + 3 + j] = 0.5 * O[i * 4 + j];
  for (int r = 0; r < 7; ++r)
    for (int k = 0; k < t; ++k) J[r * t + k] = full[r * 6 + idx[k]];
  return 1;
}

int svin_host_manifold_lift_jacobian(int kind, const double* x, double* J) {
  if (!x || !J) return SVIN_ERR_INVALID_ARG;
  if (kind == SVIN_MANIFOLD_HPOINT) {   // 3x4 (HomogeneousPointManifold.cpp:126-135)
    for (int k = 0; k < 12; ++k) J[k] = 0.0;
    J[0] = J[5] = J[10] = 1.0;
    return 1;
  }
  int idx[6];
  const int t = poseTangent(kind, idx);
  if (!t) return SVIN_ERR_INVALID_ARG;
  double L[42];
  poseLift(x, L);   // the rows of the 6d lift the manifold keeps (PoseManifold.cpp:128-140, :241-254, :333-343, :423-436)
  for (int k = 0; k < t; ++k)
    for (int c = 0; c < 7; ++c) J[k * 7 + c] = L[idx[k] * 7 + c];
  return 1;
}

int svin_host_manifold_minus_jacobian(int kind, const double* x, double* J) {
  if (!x || !J) return SVIN_ERR_INVALID_ARG;
  if (kind == SVIN_MANIFOLD_HPOINT) {   // 3x4 (HomogeneousPointManifold.cpp:115-124)
    for (int k = 0; k < 12; ++k) J[k] = 0.0;
    J[0] = J[5] = J[10] = 1.0;
    return 1;
  }
  int idx[6];
  const int t = poseTangent(kind, idx);
  if (!t) return SVIN_ERR_INVALID_ARG;
  // 6x7: [I 0; 0 2 plus(q)(0:3, :)] with the last column negated (PoseManifold.cpp:114-125 -- the reference's own
  // "not sure why the last column is coming negative"; kept as it is)
  const double qx = x[3], qy = x[4], qz = x[5], qw = x[6];
  const double P[12] = {qw, -qz, qy, qx, qz, qw, -qx, qy, -qy, qx, qw, qz};
  double full[42];
  for (int k = 0; k < 42; ++k) full[k] = 0.0;
  full[0] = full[8] = full[16] = 1.0;
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 4; ++c) full[(3 + a) * 7 + 3 + c] = 2.0 * P[a * 4 + c];
  for (int a = 0; a < 6; ++a) full[a * 7 + 6] = -full[a * 7 + 6];
  for (int k = 0; k < t; ++k)
    for (int c = 0; c < 7; ++c) J[k * 7 + c] = full[idx[k] * 7 + c];
  return 1;
}

}  // extern "C"
