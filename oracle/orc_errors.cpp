// ORACLE -- test infrastructure only (never linked into the product library).
// CPU restatement of the OKVIS/SVIn error terms; each function cites the reference lines
// (relative to /root/reference/okvis_ros/okvis/okvis_ceres/).
#include "orc_errors.hpp"

namespace orc {

// ================================================================ manifolds
// src/PoseManifold.cpp:59-82 (plus via Transformation::oplus), HomogeneousPointManifold.cpp:57-67
void manifoldPlus(int type, const double* x, const double* delta, double* xp) {
  if (type == BLOCK_POSE) {
    Transformation T(x, x + 3);
    T.oplus(delta);
    std::memcpy(xp, T.p, 7 * sizeof(double));
  } else if (type == BLOCK_SPEEDBIAS) {
    for (int i = 0; i < 9; ++i) xp[i] = x[i] + delta[i];
  } else {
    xp[0] = x[0] + delta[0]; xp[1] = x[1] + delta[1]; xp[2] = x[2] + delta[2]; xp[3] = x[3] + 0.0;
  }
}
// PoseManifold.cpp:93-102, HomogeneousPointManifold.cpp:80-91
void manifoldMinus(int type, const double* xp, const double* x, double* delta) {
  if (type == BLOCK_POSE) {
    delta[0] = xp[0] - x[0]; delta[1] = xp[1] - x[1]; delta[2] = xp[2] - x[2];
    double qi[4], dq[4];
    qinv(x + 3, qi);
    qmul(xp + 3, qi, dq);
    delta[3] = 2 * dq[0]; delta[4] = 2 * dq[1]; delta[5] = 2 * dq[2];
  } else if (type == BLOCK_SPEEDBIAS) {
    for (int i = 0; i < 9; ++i) delta[i] = xp[i] - x[i];
  } else {
    delta[0] = xp[0] - x[0]; delta[1] = xp[1] - x[1]; delta[2] = xp[2] - x[2];
  }
}
// PoseManifold.cpp:105-111 -> Transformation::oplusJacobian (Transformation.hpp:231-241)
void manifoldPlusJacobian(int type, const double* x, double* J) {
  if (type == BLOCK_POSE) {
    std::memset(J, 0, 42 * sizeof(double));
    J[0] = J[7] = J[14] = 1.0;
    double qn[4] = {x[3], x[4], x[5], x[6]};
    qnormalize(qn);  // Transformation ctor normalises
    double Q[16];
    qoplusMat(qn, Q);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) J[(3 + i) * 6 + 3 + j] = Q[i * 4 + j] * 0.5;
  } else if (type == BLOCK_SPEEDBIAS) {
    std::memset(J, 0, 81 * sizeof(double));
    for (int i = 0; i < 9; ++i) J[i * 10] = 1.0;
  } else {
    std::memset(J, 0, 12 * sizeof(double));
    J[0] = J[4] = J[8] = 1.0;
  }
}
// PoseManifold.cpp:128-140
void manifoldLiftJacobian(int type, const double* x, double* J) {
  if (type == BLOCK_POSE) {
    std::memset(J, 0, 42 * sizeof(double));
    J[0] = J[8] = J[16] = 1.0;
    const double qinv_[4] = {-x[3], -x[4], -x[5], x[6]};
    double Q[16];
    qoplusMat(qinv_, Q);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) J[(3 + i) * 7 + 3 + j] = 2.0 * Q[i * 4 + j];
  } else if (type == BLOCK_SPEEDBIAS) {
    std::memset(J, 0, 81 * sizeof(double));
    for (int i = 0; i < 9; ++i) J[i * 10] = 1.0;
  } else {
    std::memset(J, 0, 12 * sizeof(double));
    J[0] = J[5] = J[10] = 1.0;
  }
}
// PoseManifold.cpp:114-125
void poseMinusJacobian(const double* x, double* J) {
  std::memset(J, 0, 42 * sizeof(double));
  J[0] = J[8] = J[16] = 1.0;
  double Q[16];
  qplusMat(x + 3, Q);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) J[(3 + i) * 7 + 3 + j] = 2.0 * Q[i * 4 + j];
  for (int i = 0; i < 6; ++i) J[i * 7 + 6] = -1.0 * J[i * 7 + 6];
}

// helper: J(m x 7) = Jmin(m x 6) * lift(x)
static void liftPose(int m, const double* Jmin, const double* x, double* J) {
  double L[42];
  manifoldLiftJacobian(BLOCK_POSE, x, L);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < 7; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += Jmin[i * 6 + k] * L[k * 7 + j];
      J[i * 7 + j] = s;
    }
}

// ================================================================ ReprojectionError
// implementation/ReprojectionError.hpp:85-229
bool ReprojectionError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  const double* t_WS = P[0];
  const double* q_WS = P[0] + 3;
  const double* hp_W = P[1];
  const double* t_SC = P[2];
  const double* q_SC = P[2] + 3;
  // :103-117 (raw quaternions, no normalisation)
  double C_SC[9], C_CS[9], C_WS[9], C_SW[9];
  q2R(q_SC, C_SC); transpose<3, 3>(C_SC, C_CS);
  q2R(q_WS, C_WS); transpose<3, 3>(C_WS, C_SW);
  double T_CS[16] = {0}, T_SW[16] = {0};
  double tmp[3];
  mat3_vec(C_CS, t_SC, tmp);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T_CS[i * 4 + j] = C_CS[i * 3 + j]; T_CS[i * 4 + 3] = -tmp[i]; }
  T_CS[15] = 1;
  mat3_vec(C_SW, t_WS, tmp);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T_SW[i * 4 + j] = C_SW[i * 3 + j]; T_SW[i * 4 + 3] = -tmp[i]; }
  T_SW[15] = 1;
  double hp_S[4], hp_C[4];
  matmul<4, 4, 1>(T_SW, hp_W, hp_S);
  matmul<4, 4, 1>(T_CS, hp_S, hp_C);
  // :120-137
  double kp[2], Jh[8], Jhw[8];
  const bool wantJ = (J != nullptr);
  projectHomogeneous(cam, hp_C, kp, wantJ ? Jh : nullptr);
  if (wantJ) matmul<2, 2, 4>(sqrtInfo, Jh, Jhw);
  const double e[2] = {z[0] - kp[0], z[1] - kp[1]};
  res[0] = sqrtInfo[0] * e[0] + sqrtInfo[1] * e[1];
  res[1] = sqrtInfo[2] * e[0] + sqrtInfo[3] * e[1];
  // :140-147
  bool valid = true;
  if (std::fabs(hp_C[3]) > 1.0e-8) {
    if (hp_C[2] / hp_C[3] < 0.2) valid = false;
  }
  if (!wantJ) return true;
  // :152-176 pose
  if (J[0] != nullptr || (Jmin && Jmin[0])) {
    const double p[3] = {hp_W[0] - t_WS[0] * hp_W[3], hp_W[1] - t_WS[1] * hp_W[3], hp_W[2] - t_WS[2] * hp_W[3]};
    double px[9], Cpx[9];
    crossMx(p, px);
    matmul<3, 3, 3>(C_SW, px, Cpx);
    double Jb[24] = {0};  // 4x6
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { Jb[i * 6 + j] = C_SW[i * 3 + j] * hp_W[3]; Jb[i * 6 + 3 + j] = -Cpx[i * 3 + j]; }
    double JT[8], J0m[12];
    matmul<2, 4, 4>(Jhw, T_CS, JT);
    matmul<2, 4, 6>(JT, Jb, J0m);
    if (!valid) std::memset(J0m, 0, sizeof(J0m));
    if (J[0]) liftPose(2, J0m, P[0], J[0]);
    if (Jmin && Jmin[0]) std::memcpy(Jmin[0], J0m, sizeof(J0m));
  }
  // :177-196 landmark
  if (J[1] != nullptr || (Jmin && Jmin[1])) {
    double T_CW[16], J1[8];
    matmul<4, 4, 4>(T_CS, T_SW, T_CW);
    matmul<2, 4, 4>(Jhw, T_CW, J1);
    for (int i = 0; i < 8; ++i) J1[i] = -J1[i];
    if (!valid) std::memset(J1, 0, sizeof(J1));
    if (J[1]) std::memcpy(J[1], J1, sizeof(J1));
    if (Jmin && Jmin[1])
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) Jmin[1][i * 3 + j] = J1[i * 4 + j];
  }
  // :197-226 extrinsics
  if (J[2] != nullptr || (Jmin && Jmin[2])) {
    const double p[3] = {hp_S[0] - t_SC[0] * hp_S[3], hp_S[1] - t_SC[1] * hp_S[3], hp_S[2] - t_SC[2] * hp_S[3]};
    double px[9], Cpx[9];
    crossMx(p, px);
    matmul<3, 3, 3>(C_CS, px, Cpx);
    double Jb[24] = {0};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { Jb[i * 6 + j] = C_CS[i * 3 + j] * hp_S[3]; Jb[i * 6 + 3 + j] = -Cpx[i * 3 + j]; }
    double J2m[12];
    matmul<2, 4, 6>(Jhw, Jb, J2m);
    if (!valid) std::memset(J2m, 0, sizeof(J2m));
    if (J[2]) liftPose(2, J2m, P[2], J[2]);
    if (Jmin && Jmin[2]) std::memcpy(Jmin[2], J2m, sizeof(J2m));
  }
  return true;
}

// ================================================================ ImuError
namespace {
struct PreintState {
  double Delta_q[4], C_integral[9], C_doubleintegral[9], acc_integral[3], acc_doubleintegral[3];
  double cross[9], dalpha_db_g[9], dv_db_g[9], dp_db_g[9], P_delta[225], Delta_t;
};
inline void add3(double* a, const double* b, double s) { for (int i = 0; i < 9; ++i) a[i] += s * b[i]; }

// Shared integration loop of ImuError.cpp:118-243 (redoPreintegration) and :309-452
// (propagation).  The two differ in exactly two places, both reproduced:
//   * dalpha_db_g: redo accumulates C_1*rightJacobian(omega*dt)*dt (:189), propagation dt*C_1 (:384)
//   * sigma2_v: redo uses sigma_a_c*sigma_a_c (:215), propagation sigma_a_c*par.sigma_a_c (:412)
// `withCov` = covariance requested (always true for redo).
int integrate(const std::vector<ImuSample>& meas, const ImuParameters& par, const double* sb, const Time& t_start,
              const Time& t_end, bool redoFlavour, bool withCov, PreintState& S) {
  Time time = t_start;
  const Time end = t_end;
  if (meas.empty()) return -1;
  if (!(meas.back().t >= end)) return -1;
  S.Delta_q[0] = S.Delta_q[1] = S.Delta_q[2] = 0; S.Delta_q[3] = 1;
  std::memset(S.C_integral, 0, sizeof(S.C_integral));
  std::memset(S.C_doubleintegral, 0, sizeof(S.C_doubleintegral));
  std::memset(S.acc_integral, 0, sizeof(S.acc_integral));
  std::memset(S.acc_doubleintegral, 0, sizeof(S.acc_doubleintegral));
  std::memset(S.cross, 0, sizeof(S.cross));
  std::memset(S.dalpha_db_g, 0, sizeof(S.dalpha_db_g));
  std::memset(S.dv_db_g, 0, sizeof(S.dv_db_g));
  std::memset(S.dp_db_g, 0, sizeof(S.dp_db_g));
  std::memset(S.P_delta, 0, sizeof(S.P_delta));
  S.Delta_t = 0;
  bool hasStarted = false;
  int i = 0;
  const size_t n = meas.size();
  for (size_t it = 0; it < n; ++it) {
    double omega_S_0[3], acc_S_0[3], omega_S_1[3], acc_S_1[3];
    std::memcpy(omega_S_0, meas[it].gyr, 24);
    std::memcpy(acc_S_0, meas[it].acc, 24);
    const bool last = (it + 1 == n);
    // the reference dereferences (it+1) unconditionally; for the last element that read is
    // out of range and its value is never used on a valid path -> we substitute the sample itself.
    const ImuSample& nx = last ? meas[it] : meas[it + 1];
    std::memcpy(omega_S_1, nx.gyr, 24);
    std::memcpy(acc_S_1, nx.acc, 24);
    Time nexttime = last ? t_end : nx.t;
    double dt = dtSec(nexttime, time);
    if (end < nexttime) {
      const double interval = dtSec(nexttime, meas[it].t);
      nexttime = t_end;
      dt = dtSec(nexttime, time);
      const double r = dt / interval;
      for (int k = 0; k < 3; ++k) {
        omega_S_1[k] = (1.0 - r) * omega_S_0[k] + r * omega_S_1[k];
        acc_S_1[k] = (1.0 - r) * acc_S_0[k] + r * acc_S_1[k];
      }
    }
    if (dt <= 0.0) continue;
    S.Delta_t += dt;
    if (!hasStarted) {
      hasStarted = true;
      const double r = dt / dtSec(nexttime, meas[it].t);
      for (int k = 0; k < 3; ++k) {
        omega_S_0[k] = r * omega_S_0[k] + (1.0 - r) * omega_S_1[k];
        acc_S_0[k] = r * acc_S_0[k] + (1.0 - r) * acc_S_1[k];
      }
    }
    double sigma_g_c = par.sigma_g_c, sigma_a_c = par.sigma_a_c;
    bool gsat = false, asat = false;
    for (int k = 0; k < 3; ++k) {
      if (std::fabs(omega_S_0[k]) > par.g_max || std::fabs(omega_S_1[k]) > par.g_max) gsat = true;
      if (std::fabs(acc_S_0[k]) > par.a_max || std::fabs(acc_S_1[k]) > par.a_max) asat = true;
    }
    if (gsat) sigma_g_c *= 100;
    if (asat) sigma_a_c *= 100;

    // orientation (:169-177)
    double omega_true[3], acc_true[3];
    for (int k = 0; k < 3; ++k) {
      omega_true[k] = 0.5 * (omega_S_0[k] + omega_S_1[k]) - sb[3 + k];
      acc_true[k] = 0.5 * (acc_S_0[k] + acc_S_1[k]) - sb[6 + k];
    }
    const double theta_half = norm3(omega_true) * 0.5 * dt;
    const double sinc_th = sinc(theta_half);
    double dq[4] = {sinc_th * omega_true[0] * 0.5 * dt, sinc_th * omega_true[1] * 0.5 * dt,
                    sinc_th * omega_true[2] * 0.5 * dt, std::cos(theta_half)};
    double Delta_q_1[4];
    qmul(S.Delta_q, dq, Delta_q_1);
    double C[9], C_1[9], Csum[9];
    q2R(S.Delta_q, C);
    q2R(Delta_q_1, C_1);
    for (int k = 0; k < 9; ++k) Csum[k] = C[k] + C_1[k];
    double Csum_acc[3];
    mat3_vec(Csum, acc_true, Csum_acc);
    double C_integral_1[9], acc_integral_1[3];
    for (int k = 0; k < 9; ++k) C_integral_1[k] = S.C_integral[k] + 0.5 * Csum[k] * dt;
    for (int k = 0; k < 3; ++k) acc_integral_1[k] = S.acc_integral[k] + 0.5 * Csum_acc[k] * dt;
    for (int k = 0; k < 9; ++k) S.C_doubleintegral[k] += S.C_integral[k] * dt + 0.25 * Csum[k] * dt * dt;
    double pterm[3];  // acc_integral*dt + 0.25*(C+C_1)*acc*dt*dt (also used in F_delta(0,3))
    for (int k = 0; k < 3; ++k) pterm[k] = S.acc_integral[k] * dt + 0.25 * Csum_acc[k] * dt * dt;
    for (int k = 0; k < 3; ++k) S.acc_doubleintegral[k] += pterm[k];

    // Jacobian parts (:188-194)
    double wdt[3] = {omega_true[0] * dt, omega_true[1] * dt, omega_true[2] * dt};
    double RJ[9];
    rightJacobian(wdt, RJ);
    if (redoFlavour) {
      double t9[9];
      matmul<3, 3, 3>(C_1, RJ, t9);
      add3(S.dalpha_db_g, t9, dt);
    } else {
      add3(S.dalpha_db_g, C_1, dt);
    }
    double dqi[4], Rdqi[9], cross_1[9];
    qinv(dq, dqi);
    q2R(dqi, Rdqi);
    matmul<3, 3, 3>(Rdqi, S.cross, cross_1);
    add3(cross_1, RJ, dt);
    double acc_x[9], t1[9], t2[9], M[9];
    crossMx(acc_true, acc_x);
    matmul<3, 3, 3>(C, acc_x, t1);
    matmul<3, 3, 3>(t1, S.cross, M);
    matmul<3, 3, 3>(C_1, acc_x, t1);
    matmul<3, 3, 3>(t1, cross_1, t2);
    for (int k = 0; k < 9; ++k) M[k] += t2[k];  // C*acc_x*cross + C_1*acc_x*cross_1
    double dv_db_g_1[9];
    for (int k = 0; k < 9; ++k) dv_db_g_1[k] = S.dv_db_g[k] + 0.5 * dt * M[k];
    double F09[9];
    for (int k = 0; k < 9; ++k) F09[k] = dt * S.dv_db_g[k] + 0.25 * dt * dt * M[k];
    for (int k = 0; k < 9; ++k) S.dp_db_g[k] += F09[k];

    if (withCov) {
      // covariance propagation (:197-230)
      double F[225] = {0};
      for (int k = 0; k < 15; ++k) F[k * 16] = 1.0;
      auto setBlock = [&](int r0, int c0, const double* B, double s) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) F[(r0 + a) * 15 + c0 + b] = s * B[a * 3 + b];
      };
      double X[9];
      crossMx(pterm, X);
      setBlock(0, 3, X, -1.0);
      const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      setBlock(0, 6, I3, dt);
      setBlock(0, 9, F09, 1.0);
      double B012[9];
      for (int k = 0; k < 9; ++k) B012[k] = -S.C_integral[k] * dt + 0.25 * Csum[k] * dt * dt;
      setBlock(0, 12, B012, 1.0);
      setBlock(3, 9, C_1, -dt);
      double hv[3] = {0.5 * Csum_acc[0] * dt, 0.5 * Csum_acc[1] * dt, 0.5 * Csum_acc[2] * dt};
      crossMx(hv, X);
      setBlock(6, 3, X, -1.0);
      setBlock(6, 9, M, 0.5 * dt);
      setBlock(6, 12, Csum, -0.5 * dt);
      double FP[225], FPFt[225];
      matmul<15, 15, 15>(F, S.P_delta, FP);
      for (int a = 0; a < 15; ++a)
        for (int b = 0; b < 15; ++b) {
          double s = 0;
          for (int k = 0; k < 15; ++k) s += FP[a * 15 + k] * F[b * 15 + k];
          FPFt[a * 15 + b] = s;
        }
      std::memcpy(S.P_delta, FPFt, sizeof(FPFt));
      const double sigma2_dalpha = dt * sigma_g_c * sigma_g_c;
      const double sigma2_v = redoFlavour ? dt * sigma_a_c * sigma_a_c : dt * sigma_a_c * par.sigma_a_c;
      const double sigma2_p = 0.5 * dt * dt * sigma2_v;
      const double sigma2_b_g = dt * par.sigma_gw_c * par.sigma_gw_c;
      const double sigma2_b_a = dt * par.sigma_aw_c * par.sigma_aw_c;
      for (int k = 0; k < 3; ++k) {
        S.P_delta[(3 + k) * 16] += sigma2_dalpha;
        S.P_delta[(6 + k) * 16] += sigma2_v;
        S.P_delta[(0 + k) * 16] += sigma2_p;
        S.P_delta[(9 + k) * 16] += sigma2_b_g;
        S.P_delta[(12 + k) * 16] += sigma2_b_a;
      }
    }
    // memory shift (:233-239)
    std::memcpy(S.Delta_q, Delta_q_1, sizeof(Delta_q_1));
    std::memcpy(S.C_integral, C_integral_1, sizeof(C_integral_1));
    std::memcpy(S.acc_integral, acc_integral_1, sizeof(acc_integral_1));
    std::memcpy(S.cross, cross_1, sizeof(cross_1));
    std::memcpy(S.dv_db_g, dv_db_g_1, sizeof(dv_db_g_1));
    time = nexttime;
    ++i;
    if (nexttime == t_end) break;
  }
  return i;
}
}  // namespace

// ImuError.cpp:76-263
int ImuError::redoPreintegration(const double* sb) const {
  PreintState S;
  const int i = integrate(meas, par, sb, t0, t1, /*redoFlavour=*/true, /*withCov=*/true, S);
  if (i < 0) return i;
  std::memcpy(Delta_q, S.Delta_q, sizeof(Delta_q));
  std::memcpy(C_integral, S.C_integral, sizeof(C_integral));
  std::memcpy(C_doubleintegral, S.C_doubleintegral, sizeof(C_doubleintegral));
  std::memcpy(acc_integral, S.acc_integral, sizeof(acc_integral));
  std::memcpy(acc_doubleintegral, S.acc_doubleintegral, sizeof(acc_doubleintegral));
  std::memcpy(cross, S.cross, sizeof(cross));
  std::memcpy(dalpha_db_g, S.dalpha_db_g, sizeof(dalpha_db_g));
  std::memcpy(dv_db_g, S.dv_db_g, sizeof(dv_db_g));
  std::memcpy(dp_db_g, S.dp_db_g, sizeof(dp_db_g));
  std::memcpy(sb_ref, sb, sizeof(sb_ref));
  // :246-258 symmetrise, invert, symmetrise, LLT
  for (int a = 0; a < 15; ++a)
    for (int b = 0; b < 15; ++b) P_delta[a * 15 + b] = 0.5 * S.P_delta[a * 15 + b] + 0.5 * S.P_delta[b * 15 + a];
  double inv[225];
  lu_inverse(P_delta, 15, inv);
  for (int a = 0; a < 15; ++a)
    for (int b = 0; b < 15; ++b) information[a * 15 + b] = 0.5 * inv[a * 15 + b] + 0.5 * inv[b * 15 + a];
  sqrt_information_upper(information, 15, sqrtInformation);
  return i;
}

// ImuError.cpp:266-476 and :479-697
int ImuError::propagation(const std::vector<ImuSample>& meas, const ImuParameters& par, Transformation& T_WS,
                          double* sb, const Time& t_start, const Time& t_end, double* covariance, double* jacobian,
                          double* acc_doubleinteg, double* acc_integ, double* Del_t) {
  PreintState S;
  const int i = integrate(meas, par, sb, t_start, t_end, /*redoFlavour=*/false, covariance != nullptr, S);
  if (i < 0) return i;
  double r_0[3] = {T_WS.p[0], T_WS.p[1], T_WS.p[2]};
  double q_WS_0[4] = {T_WS.p[3], T_WS.p[4], T_WS.p[5], T_WS.p[6]};
  double C_WS_0[9];
  std::memcpy(C_WS_0, T_WS.C, sizeof(C_WS_0));
  // g_W = g * (0,0,6371009).normalized()
  const double gz = par.g * (6371009.0 / std::sqrt(6371009.0 * 6371009.0));
  const double g_W[3] = {par.g * 0.0, par.g * 0.0, gz};
  const double Dt = S.Delta_t;
  double Cacc2[3], Cacc1[3], rn[3], qn[4];
  mat3_vec(C_WS_0, S.acc_doubleintegral, Cacc2);
  mat3_vec(C_WS_0, S.acc_integral, Cacc1);
  for (int k = 0; k < 3; ++k) rn[k] = r_0[k] + sb[k] * Dt + Cacc2[k] - 0.5 * g_W[k] * Dt * Dt;
  qmul(q_WS_0, S.Delta_q, qn);
  T_WS.set(rn, qn);
  for (int k = 0; k < 3; ++k) sb[k] += Cacc1[k] - g_W[k] * Dt;
  if (jacobian) {
    double* F = jacobian;
    std::memset(F, 0, 225 * sizeof(double));
    for (int k = 0; k < 15; ++k) F[k * 16] = 1.0;
    auto setBlock = [&](int r0, int c0, const double* B, double s) {
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) F[(r0 + a) * 15 + c0 + b] = s * B[a * 3 + b];
    };
    double X[9], T9[9];
    crossMx(Cacc2, X); setBlock(0, 3, X, -1.0);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    setBlock(0, 6, I3, Dt);
    matmul<3, 3, 3>(C_WS_0, S.dp_db_g, T9); setBlock(0, 9, T9, 1.0);
    matmul<3, 3, 3>(C_WS_0, S.C_doubleintegral, T9); setBlock(0, 12, T9, -1.0);
    matmul<3, 3, 3>(C_WS_0, S.dalpha_db_g, T9); setBlock(3, 9, T9, -1.0);
    crossMx(Cacc1, X); setBlock(6, 3, X, -1.0);
    matmul<3, 3, 3>(C_WS_0, S.dv_db_g, T9); setBlock(6, 9, T9, 1.0);
    matmul<3, 3, 3>(C_WS_0, S.C_integral, T9); setBlock(6, 12, T9, -1.0);
  }
  if (covariance) {
    double T[225] = {0};
    for (int k = 0; k < 15; ++k) T[k * 16] = 1.0;
    for (int blk = 0; blk < 3; ++blk)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) T[(3 * blk + a) * 15 + 3 * blk + b] = C_WS_0[a * 3 + b];
    double TP[225];
    matmul<15, 15, 15>(T, S.P_delta, TP);
    for (int a = 0; a < 15; ++a)
      for (int b = 0; b < 15; ++b) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += TP[a * 15 + k] * T[b * 15 + k];
        covariance[a * 15 + b] = s;
      }
  }
  if (acc_doubleinteg) std::memcpy(acc_doubleinteg, S.acc_doubleintegral, 24);
  if (acc_integ) std::memcpy(acc_integ, S.acc_integral, 24);
  if (Del_t) *Del_t = Dt;
  return i;
}

// ImuError.cpp:706-866
bool ImuError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  const Transformation T_WS_0(P[0], P[0] + 3);
  const Transformation T_WS_1(P[2], P[2] + 3);
  const double* sb0 = P[1];
  const double* sb1 = P[3];
  double C_S0_W[9];
  transpose<3, 3>(T_WS_0.C, C_S0_W);
  const double Delta_t = dtSec(t1, t0);
  double Delta_b[6];
  for (int k = 0; k < 6; ++k) Delta_b[k] = sb0[3 + k] - sb_ref[3 + k];
  redo = redo || (norm3(Delta_b) * Delta_t > 0.0001);
  if (redo) {
    redoPreintegration(sb0);
    redoCounter++;
    for (int k = 0; k < 6; ++k) Delta_b[k] = 0;
    redo = false;
  }
  const double gz = par.g * (6371009.0 / std::sqrt(6371009.0 * 6371009.0));
  const double g_W[3] = {par.g * 0.0, par.g * 0.0, gz};
  double F0[225] = {0}, F1[225] = {0};
  for (int k = 0; k < 15; ++k) { F0[k * 16] = 1.0; F1[k * 16] = -1.0; }
  double dp[3], dv[3];
  for (int k = 0; k < 3; ++k) {
    dp[k] = T_WS_0.p[k] - T_WS_1.p[k] + sb0[k] * Delta_t - 0.5 * g_W[k] * Delta_t * Delta_t;
    dv[k] = sb0[k] - sb1[k] - g_W[k] * Delta_t;
  }
  // Dq = deltaQ(-dalpha_db_g*Delta_b.head<3>()) * Delta_q
  double a3[3], dqa[4], Dq[4];
  mat3_vec(dalpha_db_g, Delta_b, a3);
  a3[0] = -a3[0]; a3[1] = -a3[1]; a3[2] = -a3[2];
  deltaQ(a3, dqa);
  qmul(dqa, Delta_q, Dq);
  auto setBlock = [](double* F, int r0, int c0, const double* B, double s) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) F[(r0 + a) * 15 + c0 + b] = s * B[a * 3 + b];
  };
  double X[9], T9[9];
  setBlock(F0, 0, 0, C_S0_W, 1.0);
  crossMx(dp, X); matmul<3, 3, 3>(C_S0_W, X, T9); setBlock(F0, 0, 3, T9, 1.0);
  setBlock(F0, 0, 6, C_S0_W, Delta_t);
  setBlock(F0, 0, 9, dp_db_g, 1.0);
  setBlock(F0, 0, 12, C_doubleintegral, -1.0);
  double q1inv[4], qa[4], Qp[16], Qo[16], Q44[16];
  qinv(T_WS_1.q(), q1inv);
  // F0(3,3) = (plus(Dq*q1^-1) * oplus(q0)).topLeft<3,3>
  qmul(Dq, q1inv, qa);
  qplusMat(qa, Qp);
  qoplusMat(T_WS_0.q(), Qo);
  matmul<4, 4, 4>(Qp, Qo, Q44);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) F0[(3 + a) * 15 + 3 + b] = Q44[a * 4 + b];
  // F0(3,9) = (oplus(q1^-1*q0) * oplus(Dq)).topLeft<3,3> * (-dalpha_db_g)
  double qb[4], Qo1[16], Qo2[16];
  qmul(q1inv, T_WS_0.q(), qb);
  qoplusMat(qb, Qo1);
  qoplusMat(Dq, Qo2);
  matmul<4, 4, 4>(Qo1, Qo2, Q44);
  double TL[9], ndal[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) { TL[a * 3 + b] = Q44[a * 4 + b]; ndal[a * 3 + b] = -dalpha_db_g[a * 3 + b]; }
  matmul<3, 3, 3>(TL, ndal, T9);
  setBlock(F0, 3, 9, T9, 1.0);
  crossMx(dv, X); matmul<3, 3, 3>(C_S0_W, X, T9); setBlock(F0, 6, 3, T9, 1.0);
  setBlock(F0, 6, 6, C_S0_W, 1.0);
  setBlock(F0, 6, 9, dv_db_g, 1.0);
  setBlock(F0, 6, 12, C_integral, -1.0);
  // F1
  setBlock(F1, 0, 0, C_S0_W, -1.0);
  double Qp2[16], Qp3[16], Qt[16];
  qplusMat(Dq, Qp2);
  qplusMat(q1inv, Qp3);
  matmul<4, 4, 4>(Qp2, Qo, Qt);
  matmul<4, 4, 4>(Qt, Qp3, Q44);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) F1[(3 + a) * 15 + 3 + b] = -Q44[a * 4 + b];
  setBlock(F1, 6, 6, C_S0_W, -1.0);
  // error (:786-791)
  double err[15], v3[3];
  mat3_vec(C_S0_W, dp, v3);
  for (int a = 0; a < 3; ++a) {
    double s = 0;
    for (int k = 0; k < 6; ++k) s += F0[a * 15 + 9 + k] * Delta_b[k];
    err[a] = v3[a] + acc_doubleintegral[a] + s;
  }
  double qc[4], qd[4];
  qmul(q1inv, T_WS_0.q(), qc);
  qmul(Dq, qc, qd);
  err[3] = 2 * qd[0]; err[4] = 2 * qd[1]; err[5] = 2 * qd[2];
  mat3_vec(C_S0_W, dv, v3);
  for (int a = 0; a < 3; ++a) {
    double s = 0;
    for (int k = 0; k < 6; ++k) s += F0[(6 + a) * 15 + 9 + k] * Delta_b[k];
    err[6 + a] = v3[a] + acc_integral[a] + s;
  }
  for (int k = 0; k < 6; ++k) err[9 + k] = sb0[3 + k] - sb1[3 + k];
  for (int a = 0; a < 15; ++a) {
    double s = 0;
    for (int k = 0; k < 15; ++k) s += sqrtInformation[a * 15 + k] * err[k];
    res[a] = s;
  }
  if (J == nullptr) return true;
  // Jacobians (:796-863)
  auto weighted = [&](const double* F, int c0, int nc, double* out) {  // out(15 x nc) = sqrtInfo * F[:, c0:c0+nc]
    for (int a = 0; a < 15; ++a)
      for (int b = 0; b < nc; ++b) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += sqrtInformation[a * 15 + k] * F[k * 15 + c0 + b];
        out[a * nc + b] = s;
      }
  };
  double Jm6[90], Jm9[135];
  if (J[0] || (Jmin && Jmin[0])) {
    weighted(F0, 0, 6, Jm6);
    if (J[0]) liftPose(15, Jm6, P[0], J[0]);
    if (Jmin && Jmin[0]) std::memcpy(Jmin[0], Jm6, sizeof(Jm6));
  }
  if (J[1] || (Jmin && Jmin[1])) {
    weighted(F0, 6, 9, Jm9);
    if (J[1]) std::memcpy(J[1], Jm9, sizeof(Jm9));
    if (Jmin && Jmin[1]) std::memcpy(Jmin[1], Jm9, sizeof(Jm9));
  }
  if (J[2] || (Jmin && Jmin[2])) {
    weighted(F1, 0, 6, Jm6);
    if (J[2]) liftPose(15, Jm6, P[2], J[2]);
    if (Jmin && Jmin[2]) std::memcpy(Jmin[2], Jm6, sizeof(Jm6));
  }
  if (J[3] || (Jmin && Jmin[3])) {
    weighted(F1, 6, 9, Jm9);
    if (J[3]) std::memcpy(J[3], Jm9, sizeof(Jm9));
    if (Jmin && Jmin[3]) std::memcpy(Jmin[3], Jm9, sizeof(Jm9));
  }
  return true;
}

// ================================================================ PoseError  (PoseError.cpp:87-132)
bool PoseError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  const Transformation T_WS(P[0], P[0] + 3);
  const Transformation dp = meas * T_WS.inverse();
  double e[6];
  for (int k = 0; k < 3; ++k) { e[k] = meas.p[k] - T_WS.p[k]; e[3 + k] = 2 * dp.p[3 + k]; }
  matmul<6, 6, 1>(sqrtInfo, e, res);
  if (J == nullptr) return true;
  if (J[0] || (Jmin && Jmin[0])) {
    double J0[36] = {0};
    for (int k = 0; k < 6; ++k) J0[k * 7] = -1.0;
    double Q[16];
    qplusMat(dp.q(), Q);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) J0[(3 + a) * 6 + 3 + b] = -Q[a * 4 + b];
    double J0w[36];
    matmul<6, 6, 6>(sqrtInfo, J0, J0w);
    if (J[0]) liftPose(6, J0w, P[0], J[0]);
    if (Jmin && Jmin[0]) std::memcpy(Jmin[0], J0w, sizeof(J0w));
  }
  return true;
}

// ================================================================ SpeedAndBiasError (:83-113)
bool SpeedAndBiasError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  double e[9];
  for (int k = 0; k < 9; ++k) e[k] = meas[k] - P[0][k];
  matmul<9, 9, 1>(sqrtInfo, e, res);
  if (J && J[0]) for (int k = 0; k < 81; ++k) J[0][k] = -sqrtInfo[k];
  if (Jmin && Jmin[0]) for (int k = 0; k < 81; ++k) Jmin[0][k] = -sqrtInfo[k];
  return true;
}

// ================================================================ RelativePoseError (:79-147)
bool RelativePoseError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  const Transformation T0(P[0], P[0] + 3), T1(P[1], P[1] + 3);
  const Transformation dp = T1 * T0.inverse();
  double e[6];
  for (int k = 0; k < 3; ++k) { e[k] = T1.p[k] - T0.p[k]; e[3 + k] = 2 * dp.p[3 + k]; }
  matmul<6, 6, 1>(sqrtInfo, e, res);
  if (J == nullptr) return true;
  if (J[0] || (Jmin && Jmin[0])) {
    double J0[36] = {0};
    for (int k = 0; k < 6; ++k) J0[k * 7] = -1.0;
    double Q[16];
    qplusMat(dp.q(), Q);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) J0[(3 + a) * 6 + 3 + b] = -Q[a * 4 + b];
    double J0w[36];
    matmul<6, 6, 6>(sqrtInfo, J0, J0w);
    if (J[0]) liftPose(6, J0w, P[0], J[0]);
    if (Jmin && Jmin[0]) std::memcpy(Jmin[0], J0w, sizeof(J0w));
  }
  if (J[1] || (Jmin && Jmin[1])) {
    double J1[36] = {0};
    for (int k = 0; k < 6; ++k) J1[k * 7] = 1.0;
    double Q[16];
    qoplusMat(dp.q(), Q);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) J1[(3 + a) * 6 + 3 + b] = Q[a * 4 + b];
    double J1w[36];
    matmul<6, 6, 6>(sqrtInfo, J1, J1w);
    if (J[1]) liftPose(6, J1w, P[1], J[1]);
    if (Jmin && Jmin[1]) std::memcpy(Jmin[1], J1w, sizeof(J1w));
  }
  return true;
}

// ================================================================ SonarError (:118-183)
bool SonarError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  double mean[3] = {0, 0, 0};
  const size_t k = patch.size() / 3;
  for (size_t i = 0; i < k; ++i) { mean[0] += patch[3 * i]; mean[1] += patch[3 * i + 1]; mean[2] += patch[3 * i + 2]; }
  mean[0] = mean[0] / k; mean[1] = mean[1] / k; mean[2] = mean[2] / k;
  const Transformation T_WS(P[0], P[0] + 3);
  const double d[3] = {T_WS.p[0] - mean[0], T_WS.p[1] - mean[1], T_WS.p[2] - mean[2]};
  const double range_corrected = norm3(d);
  res[0] = sqrtInfo * (range - range_corrected);
  if (J == nullptr) return true;
  if (J[0] || (Jmin && Jmin[0])) {
    const Transformation T_WSo = T_WS * T_SSo;
    const double rp[3] = {range * std::cos(heading), range * std::sin(heading), 0.0};
    const double qid[4] = {0, 0, 0, 1};
    const Transformation sonar_point(rp, qid);
    const Transformation T_WSo_point = T_WSo * sonar_point;
    double J7[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 3; ++a) J7[a] = sqrtInfo * ((T_WS.p[a] - T_WSo_point.p[a]) / range);
    if (J[0]) std::memcpy(J[0], J7, sizeof(J7));
    if (Jmin && Jmin[0]) std::memcpy(Jmin[0], J7, 6 * sizeof(double));  // reference writes 1x7 into a 1x6 buffer
  }
  return true;
}

// ================================================================ DepthError (:75-139)
bool DepthError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  const double error = P[0][2] - (-1 * depth + firstDepth);
  res[0] = sqrtInfo * error;
  if (J == nullptr) return true;
  double J7[7] = {0, 0, sqrtInfo * 1.0, 0, 0, 0, 0};
  if (J[0]) std::memcpy(J[0], J7, sizeof(J7));
  if (Jmin && Jmin[0]) std::memcpy(Jmin[0], J7, 6 * sizeof(double));
  return true;
}

// ================================================================ HomogeneousPointError (:77-117)
bool HomogeneousPointError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  double e[3];
  manifoldMinus(BLOCK_HPOINT, P[0], meas, e);
  matmul<3, 3, 1>(sqrtInfo, e, res);
  if (J == nullptr) return true;
  if (J[0])
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) J[0][a * 4 + b] = sqrtInfo[a * 3 + b]; J[0][a * 4 + 3] = 0; }
  if (Jmin && Jmin[0]) std::memcpy(Jmin[0], sqrtInfo, sizeof(sqrtInfo));
  return true;
}

}  // namespace orc
