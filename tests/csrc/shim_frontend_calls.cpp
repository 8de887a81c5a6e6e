// The calls okvis_frontend makes into okvis_ceres OUTSIDE the window solve, made against the Ceres-free shim headers
// (integration/okvis/ceres/) exactly as the frontend makes them:
//   * ProbabilisticStereoTriangulator.cpp:87-99 / :128-140 -- PoseError on the stack, three PoseParameterBlock members,
//     EvaluateWithMinimalJacobians with Eigen-row-major output buffers, H = J_min^T J_min;
//   * ProbabilisticStereoTriangulator.cpp:266-300 -- two ReprojectionError evaluations whose parameter pointers come from
//     PoseParameterBlock::parameters();
//   * VioKeyframeWindowMatchingAlgorithm.cpp:453 -- HomogeneousPointParameterBlock(point, 0).estimate();
// plus the manifold classes (plus / minus / Jacobians / verify()) and the parameter-block operations.  Prints numbers
// for tests/test_shim_compile.py to hold against the oracle.  Runs without a GPU and includes no ceres header.
#include <okvis/MultiFrame.hpp>
#include <okvis/ceres/HomogeneousPointParameterBlock.hpp>
#include <okvis/ceres/PoseError.hpp>
#include <okvis/ceres/PoseParameterBlock.hpp>
#include <okvis/ceres/ReprojectionError.hpp>
#include <okvis/ceres/SpeedAndBiasParameterBlock.hpp>

#include <cstdio>
#include <fstream>
#include <vector>

#ifdef CERES_PUBLIC_CERES_H_
#error "the shim set must not pull in ceres/ceres.h"
#endif

static void printv(const char* tag, const double* v, int n) {
  std::printf("%s", tag);
  for (int i = 0; i < n; ++i) std::printf(" %.17g", v[i]);
  std::printf("\n");
}

template <class M>
static void runManifold(const char* tag, const double* x, const double* delta) {
  M m;
  const int na = m.AmbientSize(), nt = m.TangentSize();
  std::vector<double> xp(na), d(nt), Jp(na * nt), Jl(nt * na), Jm(nt * na);
  m.Plus(x, delta, xp.data());
  m.Minus(xp.data(), x, d.data());
  m.PlusJacobian(x, Jp.data());
  m.ComputeLiftJacobian(x, Jl.data());
  m.MinusJacobian(x, Jm.data());
  std::printf("%s dims %d %d verify %d\n", tag, na, nt, (int)m.verify(x));
  printv((std::string(tag) + " plus").c_str(), xp.data(), na);
  printv((std::string(tag) + " minus").c_str(), d.data(), nt);
  printv((std::string(tag) + " Jplus").c_str(), Jp.data(), na * nt);
  printv((std::string(tag) + " Jlift").c_str(), Jl.data(), nt * na);
  printv((std::string(tag) + " Jminus").c_str(), Jm.data(), nt * na);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  double Tab[7], info[36], Tx[7], delta[6];
  for (double& v : Tab) in >> v;
  for (double& v : info) in >> v;
  for (double& v : Tx) in >> v;
  for (double& v : delta) in >> v;
  const okvis::kinematics::Transformation T_AB_(Eigen::Vector3d(Tab[0], Tab[1], Tab[2]), Eigen::Quaterniond(Tab[6], Tab[3], Tab[4], Tab[5]));
  Eigen::Matrix<double, 6, 6> information;
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) information(a, b) = info[a * 6 + b];

  // ---- ProbabilisticStereoTriangulator.cpp:87-99
  ::okvis::ceres::PoseError poseError(T_AB_, information);
  Eigen::Matrix<double, 6, 6, Eigen::RowMajor> J_minimal;
  Eigen::Matrix<double, 6, 7, Eigen::RowMajor> J;
  ::okvis::ceres::PoseParameterBlock poseA_, poseB_, extrinsics_;
  poseA_ = ::okvis::ceres::PoseParameterBlock(okvis::kinematics::Transformation(), 0, okvis::Time(0, 0));
  poseB_ = ::okvis::ceres::PoseParameterBlock(T_AB_, 0, okvis::Time(0, 0));
  extrinsics_ = ::okvis::ceres::PoseParameterBlock(okvis::kinematics::Transformation(), 0, okvis::Time(0, 0));
  double residuals[6];
  double* parameters = poseB_.parameters();
  double* jacobians = J.data();
  double* jacobians_minimal = J_minimal.data();
  bool ok = poseError.EvaluateWithMinimalJacobians(&parameters, &residuals[0], &jacobians, &jacobians_minimal);
  std::printf("pose_at_measurement %d\n", (int)ok);
  printv("r", residuals, 6);
  printv("Jmin", J_minimal.data(), 36);
  printv("J", J.data(), 42);
  // the same error term away from its measurement
  okvis::ceres::PoseParameterBlock other(okvis::kinematics::Transformation(Eigen::Vector3d(Tx[0], Tx[1], Tx[2]), Eigen::Quaterniond(Tx[6], Tx[3], Tx[4], Tx[5])), 7,
                                         okvis::Time(3, 4));
  parameters = other.parameters();
  ok = poseError.EvaluateWithMinimalJacobians(&parameters, &residuals[0], &jacobians, &jacobians_minimal);
  std::printf("pose_away %d dim %zu blocks %zu bdim %zu type %s id %llu fixed %d t %u %u\n", (int)ok, poseError.residualDim(), poseError.parameterBlocks(),
              poseError.parameterBlockDim(0), poseError.typeInfo().c_str(), (unsigned long long)other.id(), (int)other.fixed(), other.timestamp().sec,
              other.timestamp().nsec);
  printv("r", residuals, 6);
  printv("Jmin", J_minimal.data(), 36);
  printv("J", J.data(), 42);
  printv("cov", poseError.covariance().data(), 36);
  // Evaluate() without Jacobians, and through the ErrorInterface base
  const okvis::ceres::ErrorInterface& base = poseError;
  double r2[6];
  ok = poseError.Evaluate(&parameters, r2, nullptr) && base.EvaluateWithMinimalJacobians(&parameters, r2, nullptr, nullptr);
  std::printf("pose_nojac %d %.17g\n", (int)ok, r2[5]);
  // variance constructor
  okvis::ceres::PoseError pe2(T_AB_, 0.04, 0.0009);
  ok = pe2.EvaluateWithMinimalJacobians(&parameters, &residuals[0], &jacobians, &jacobians_minimal);
  printv("r_var", residuals, 6);

  // ---- parameter-block operations (PoseParameterBlock.hpp:96-120)
  double xp[7], dm[6], Jp[42], Jl[42];
  other.plus(other.parameters(), delta, xp);
  other.minus(xp, other.parameters(), dm);
  other.plusJacobian(other.parameters(), Jp);
  other.liftJacobian(other.parameters(), Jl);
  printv("pb_plus", xp, 7);
  printv("pb_minus", dm, 6);
  printv("pb_Jplus", Jp, 42);
  printv("pb_Jlift", Jl, 42);
  const okvis::kinematics::Transformation est = other.estimate();
  std::printf("pb_estimate %.17g %.17g %.17g %.17g dim %zu min %zu type %s\n", est.r()[0], est.q().x(), est.q().w(), other.parameters()[6], other.dimension(),
              other.minimalDimension(), other.typeInfo().c_str());

  // ---- manifolds
  runManifold<okvis::ceres::PoseManifold>("m6", Tx, delta);
  runManifold<okvis::ceres::PoseManifold3d>("m3", Tx, delta);
  runManifold<okvis::ceres::PoseManifold4d>("m4", Tx, delta);
  runManifold<okvis::ceres::PoseManifold2d>("m2", Tx, delta);
  const double hp[4] = {0.3, -1.2, 4.0, 1.0};
  runManifold<okvis::ceres::HomogeneousPointManifold>("mh", hp, delta);
  okvis::ceres::PoseManifold pm;
  double Ja[42], Jn[42];
  std::printf("numdiff %d\n", (int)pm.VerifyJacobianNumDiff(Tx, Ja, Jn));

  // ---- VioKeyframeWindowMatchingAlgorithm.cpp:453
  okvis::ceres::HomogeneousPointParameterBlock point(Eigen::Vector4d(hp[0], hp[1], hp[2], hp[3]), 0);
  const Eigen::Vector4d e = point.estimate();
  okvis::ceres::HomogeneousPointParameterBlock p3(Eigen::Vector3d(1.0, 2.0, 3.0), 9, false);
  std::printf("hpoint %.17g %.17g %.17g %.17g init %d | %.17g %.17g init %d dim %zu min %zu %s\n", e[0], e[1], e[2], e[3], (int)point.initialized(), p3.estimate()[2],
              p3.estimate()[3], (int)p3.initialized(), p3.dimension(), p3.minimalDimension(), p3.typeInfo().c_str());
  okvis::SpeedAndBias sb0;
  for (int k = 0; k < 9; ++k) sb0[k] = 0.1 * k;
  okvis::ceres::SpeedAndBiasParameterBlock sbb(sb0, 11, okvis::Time(1, 2));
  double sbp[9], dd[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
  sbb.plus(sbb.parameters(), dd, sbp);
  std::printf("sb %.17g %.17g %s\n", sbp[8], sbb.estimate()[3], sbb.typeInfo().c_str());

  // ---- ProbabilisticStereoTriangulator.cpp:266-300: reprojection residuals fed from the parameter blocks
  std::string dist;
  int w, h, nIntr;
  in >> dist >> w >> h >> nIntr;
  std::vector<double> intr(nIntr);
  for (double& v : intr) in >> v;
  double hPA_[4], uv[2];
  for (double& v : hPA_) in >> v;
  for (double& v : uv) in >> v;
  auto geo = std::make_shared<okvis::cameras::CameraBase>(w, h, dist, intr);
  Eigen::Matrix<double, 2, 2> inverseMeasurementCovariance;
  inverseMeasurementCovariance(0, 0) = inverseMeasurementCovariance(1, 1) = 1.0 / (0.53 * 0.53);
  inverseMeasurementCovariance(0, 1) = inverseMeasurementCovariance(1, 0) = 0.0;
  Eigen::Vector4d hPA(hPA_[0], hPA_[1], hPA_[2], hPA_[3]);
  ::okvis::ceres::ReprojectionError<okvis::cameras::CameraBase> reprojectionErrorB(geo, 0, Eigen::Vector2d(uv[0], uv[1]), inverseMeasurementCovariance);
  Eigen::Matrix<double, 2, 1> residualB;
  Eigen::Matrix<double, 2, 7, Eigen::RowMajor> J_TB;
  Eigen::Matrix<double, 2, 6, Eigen::RowMajor> J_TB_min;
  Eigen::Matrix<double, 2, 4, Eigen::RowMajor> J_hpB;
  Eigen::Matrix<double, 2, 3, Eigen::RowMajor> J_hpB_min;
  double* jacobiansB[3] = {J_TB.data(), J_hpB.data(), 0};
  double* jacobiansB_min[3] = {J_TB_min.data(), J_hpB_min.data(), 0};
  const double* parametersB[3] = {poseB_.parameters(), hPA.data(), extrinsics_.parameters()};
  ok = reprojectionErrorB.EvaluateWithMinimalJacobians(parametersB, residualB.data(), jacobiansB, jacobiansB_min);
  std::printf("reprojB %d %.17g %.17g\n", (int)ok, residualB[0], residualB[1]);
  printv("J_TB_min", J_TB_min.data(), 12);
  printv("J_hpB_min", J_hpB_min.data(), 6);
  return 0;
}
