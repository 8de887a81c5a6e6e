// ORACLE -- test infrastructure only (never linked into the product library).
//
// CPU restatement of the OKVIS/SVIn error terms and manifolds.  Each class cites the
// reference file:line it follows (paths relative to
// /root/reference/okvis_ros/okvis/okvis_ceres/).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
#include "orc_math.hpp"
#include "orc_camera.hpp"

namespace orc {

// ---------------------------------------------------------------- parameter blocks / manifolds
enum BlockType { BLOCK_POSE = 0, BLOCK_SPEEDBIAS = 1, BLOCK_HPOINT = 2 };
inline int blockDim(int type) { return type == BLOCK_POSE ? 7 : (type == BLOCK_SPEEDBIAS ? 9 : 4); }
inline int blockMinDim(int type) { return type == BLOCK_POSE ? 6 : (type == BLOCK_SPEEDBIAS ? 9 : 3); }

// src/PoseManifold.cpp:59-140, src/HomogeneousPointManifold.cpp:57-135
void manifoldPlus(int type, const double* x, const double* delta, double* xPlus);
void manifoldMinus(int type, const double* xPlus, const double* x, double* delta);
void manifoldPlusJacobian(int type, const double* x, double* J);   // dim x mdim, row-major
void manifoldLiftJacobian(int type, const double* x, double* J);   // mdim x dim, row-major
void poseMinusJacobian(const double* x, double* J);                // 6x7 (PoseManifold.cpp:114-125)

// ---------------------------------------------------------------- error-term interface
// include/okvis/ceres/ErrorInterface.hpp:54
struct ErrorTerm {
  enum Kind { REPROJECTION, IMU, POSE, SPEEDBIAS, RELPOSE, SONAR, DEPTH, HPOINT, MARGINALIZATION };
  virtual ~ErrorTerm() {}
  virtual Kind kind() const = 0;
  virtual int residualDim() const = 0;
  virtual int numBlocks() const = 0;
  virtual int blockType(int i) const = 0;
  // J[i]: residualDim x dim(i), Jmin[i]: residualDim x mindim(i), row-major, any may be null.
  virtual bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const = 0;
};

// implementation/ReprojectionError.hpp:85-229, SizedCostFunction<2,7,4,7>
struct ReprojectionError : ErrorTerm {
  Camera cam;
  double z[2];
  double sqrtInfo[4];  // upper-triangular L^T of the 2x2 information (setInformation :66-73)
  uint64_t camIdx = 0;
  ReprojectionError(const Camera& c, const double* uv, const double* information2x2) : cam(c) {
    z[0] = uv[0]; z[1] = uv[1];
    sqrt_information_upper(information2x2, 2, sqrtInfo);
  }
  Kind kind() const override { return REPROJECTION; }
  int residualDim() const override { return 2; }
  int numBlocks() const override { return 3; }
  int blockType(int i) const override { return i == 1 ? BLOCK_HPOINT : BLOCK_POSE; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// okvis::ImuParameters (okvis_common Parameters.hpp)
struct ImuParameters {
  double a_max = 176, g_max = 7.8, sigma_g_c = 12e-4, sigma_a_c = 8e-3, sigma_bg = 0.03, sigma_ba = 0.1;
  double sigma_gw_c = 4e-6, sigma_aw_c = 4e-5, tau = 3600, g = 9.81007;
  double a0[3] = {0, 0, 0};
  int rate = 200;
};
struct Time {  // okvis_time Time.hpp: u32 sec + u32 nsec
  uint32_t sec = 0, nsec = 0;
  bool operator==(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
  bool operator<(const Time& o) const { return sec < o.sec || (sec == o.sec && nsec < o.nsec); }
  bool operator<=(const Time& o) const { return !(o < *this); }
  bool operator>=(const Time& o) const { return !(*this < o); }
};
// (a - b).toSec() with Duration normalisation (Time.hpp:146, Duration.hpp)
inline double dtSec(const Time& a, const Time& b) {
  int64_t s = int64_t(a.sec) - int64_t(b.sec);
  int64_t ns = int64_t(a.nsec) - int64_t(b.nsec);
  while (ns < 0) { ns += 1000000000LL; s -= 1; }
  while (ns >= 1000000000LL) { ns -= 1000000000LL; s += 1; }
  return static_cast<double>(s) + 1e-9 * static_cast<double>(ns);
}
struct ImuSample { Time t; double gyr[3]; double acc[3]; };

// src/ImuError.cpp, SizedCostFunction<15,7,9,7,9>
struct ImuError : ErrorTerm {
  std::vector<ImuSample> meas;
  ImuParameters par;
  Time t0, t1;
  // mutable pre-integration state (ImuError.hpp:239-270)
  mutable double Delta_q[4] = {0, 0, 0, 1};
  mutable double C_integral[9] = {0}, C_doubleintegral[9] = {0};
  mutable double acc_integral[3] = {0}, acc_doubleintegral[3] = {0};
  mutable double cross[9] = {0};
  mutable double dalpha_db_g[9] = {0}, dv_db_g[9] = {0}, dp_db_g[9] = {0};
  mutable double P_delta[225] = {0};
  mutable double sb_ref[9] = {0};
  mutable bool redo = true;
  mutable int redoCounter = 0;
  mutable double information[225] = {0}, sqrtInformation[225] = {0};

  ImuError(const std::vector<ImuSample>& m, const ImuParameters& p, Time a, Time b) : meas(m), par(p), t0(a), t1(b) {}
  Kind kind() const override { return IMU; }
  int residualDim() const override { return 15; }
  int numBlocks() const override { return 4; }
  int blockType(int i) const override { return (i % 2) ? BLOCK_SPEEDBIAS : BLOCK_POSE; }
  int redoPreintegration(const double* sb) const;  // ImuError.cpp:76-263
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
  // ImuError.cpp:266-476 / :479-697 (static propagation).  covariance / jacobian 15x15 or null.
  static int propagation(const std::vector<ImuSample>& meas, const ImuParameters& par, Transformation& T_WS,
                         double* speedAndBias, const Time& t_start, const Time& t_end, double* covariance,
                         double* jacobian, double* acc_doubleinteg = nullptr, double* acc_integ = nullptr,
                         double* Del_t = nullptr);
};

// src/PoseError.cpp:49-132 <6,7>
struct PoseError : ErrorTerm {
  Transformation meas;
  double sqrtInfo[36];
  PoseError(const Transformation& m, const double* information6x6) : meas(m) {
    sqrt_information_upper(information6x6, 6, sqrtInfo);
  }
  PoseError(const Transformation& m, double translationVariance, double rotationVariance) : meas(m) {
    double info[36] = {0};
    for (int i = 0; i < 3; ++i) { info[i * 7] = 1.0 * 1.0 / translationVariance; info[(i + 3) * 7] = 1.0 * 1.0 / rotationVariance; }
    sqrt_information_upper(info, 6, sqrtInfo);
  }
  Kind kind() const override { return POSE; }
  int residualDim() const override { return 6; }
  int numBlocks() const override { return 1; }
  int blockType(int) const override { return BLOCK_POSE; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// src/SpeedAndBiasError.cpp:47-113 <9,9>
struct SpeedAndBiasError : ErrorTerm {
  double meas[9];
  double sqrtInfo[81];
  SpeedAndBiasError(const double* m, double speedVariance, double gyrBiasVariance, double accBiasVariance) {
    std::memcpy(meas, m, sizeof(meas));
    double info[81] = {0};
    for (int i = 0; i < 3; ++i) {
      info[i * 10] = 1.0 * 1.0 / speedVariance;
      info[(i + 3) * 10] = 1.0 * 1.0 / gyrBiasVariance;
      info[(i + 6) * 10] = 1.0 * 1.0 / accBiasVariance;
    }
    sqrt_information_upper(info, 9, sqrtInfo);
  }
  Kind kind() const override { return SPEEDBIAS; }
  int residualDim() const override { return 9; }
  int numBlocks() const override { return 1; }
  int blockType(int) const override { return BLOCK_SPEEDBIAS; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// src/RelativePoseError.cpp:48-147 <6,7,7>
struct RelativePoseError : ErrorTerm {
  double sqrtInfo[36];
  RelativePoseError(double translationVariance, double rotationVariance) {
    double info[36] = {0};
    for (int i = 0; i < 3; ++i) { info[i * 7] = 1.0 * 1.0 / translationVariance; info[(i + 3) * 7] = 1.0 * 1.0 / rotationVariance; }
    sqrt_information_upper(info, 6, sqrtInfo);
  }
  Kind kind() const override { return RELPOSE; }
  int residualDim() const override { return 6; }
  int numBlocks() const override { return 2; }
  int blockType(int) const override { return BLOCK_POSE; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// src/SonarError.cpp:56-183 <1,7>.  NOTE the reference quirks reproduced on purpose
// (SURVEY.md section 7): residual uses the mean of the landmark patch, the Jacobian uses the
// sonar point and /range with the opposite sign of d(residual)/d(r); the "minimal"
// Jacobian is 1x7 (we expose its first 6 entries as the 1x6 minimal block).
struct SonarError : ErrorTerm {
  Transformation T_SSo;
  double range, heading, sqrtInfo;
  std::vector<double> patch;  // k x 3
  SonarError(const Transformation& tsso, double r, double h, double information, const std::vector<double>& p)
      : T_SSo(tsso), range(r), heading(h), sqrtInfo(std::sqrt(information)), patch(p) {}
  Kind kind() const override { return SONAR; }
  int residualDim() const override { return 1; }
  int numBlocks() const override { return 1; }
  int blockType(int) const override { return BLOCK_POSE; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// src/DepthError.cpp:48-139 <1,7>
struct DepthError : ErrorTerm {
  double depth, firstDepth, sqrtInfo;
  DepthError(double d, double information, double fd) : depth(d), firstDepth(fd), sqrtInfo(std::sqrt(information)) {}
  Kind kind() const override { return DEPTH; }
  int residualDim() const override { return 1; }
  int numBlocks() const override { return 1; }
  int blockType(int) const override { return BLOCK_POSE; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// src/HomogeneousPointError.cpp:48-117 <3,4>
struct HomogeneousPointError : ErrorTerm {
  double meas[4];
  double sqrtInfo[9];
  HomogeneousPointError(const double* m, double variance) {
    std::memcpy(meas, m, sizeof(meas));
    double info[9] = {0};
    info[0] = info[4] = info[8] = 1.0 / variance;
    sqrt_information_upper(info, 3, sqrtInfo);
  }
  Kind kind() const override { return HPOINT; }
  int residualDim() const override { return 3; }
  int numBlocks() const override { return 1; }
  int blockType(int) const override { return BLOCK_HPOINT; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;
};

// ceres::CauchyLoss(a) / HuberLoss(a) (Ceres loss_function.cc), rho[0..2] = rho, rho', rho''
enum LossType { LOSS_NONE = 0, LOSS_CAUCHY = 1, LOSS_HUBER = 2 };
inline void lossEvaluate(int loss, double a, double s, double rho[3]) {
  if (loss == LOSS_CAUCHY) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c, inv = 1.0 / sum;
    rho[0] = b * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c * (inv * inv);
  } else if (loss == LOSS_HUBER) {
    const double b = a * a;
    if (s > b) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

}  // namespace orc
