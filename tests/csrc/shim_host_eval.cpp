// The CPU-side shim classes (integration/okvis/ceres/ImuError.hpp, ReprojectionError.hpp) driven the way
// ThreadedKFVio.cpp:599 and ProbabilisticStereoTriangulator.cpp:266-300 drive the reference classes; prints what they
// return for tests/test_shim_compile.py to compare with the C ABI called directly.  Runs without a GPU.
#include <okvis/MultiFrame.hpp>
#include <okvis/ceres/ImuError.hpp>
#include <okvis/ceres/ReprojectionError.hpp>

#include <cstdio>
#include <fstream>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  okvis::ImuParameters ip;
  in >> ip.a_max >> ip.g_max >> ip.sigma_g_c >> ip.sigma_a_c >> ip.sigma_bg >> ip.sigma_ba >> ip.sigma_gw_c >> ip.sigma_aw_c >> ip.tau >> ip.g;
  in >> ip.a0[0] >> ip.a0[1] >> ip.a0[2];
  int n;
  in >> n;
  okvis::ImuMeasurementDeque imu;
  for (int i = 0; i < n; ++i) {
    okvis::ImuMeasurement m;
    in >> m.timeStamp.sec >> m.timeStamp.nsec;
    for (int a = 0; a < 3; ++a) in >> m.measurement.gyroscopes[a];
    for (int a = 0; a < 3; ++a) in >> m.measurement.accelerometers[a];
    imu.push_back(m);
  }
  double T[7], s9[9];
  for (double& v : T) in >> v;
  for (double& v : s9) in >> v;
  okvis::Time t0, t1;
  in >> t0.sec >> t0.nsec >> t1.sec >> t1.nsec;
  okvis::kinematics::Transformation T_WS(Eigen::Vector3d(T[0], T[1], T[2]), Eigen::Quaterniond(T[6], T[3], T[4], T[5]));
  okvis::SpeedAndBias sb;
  for (int k = 0; k < 9; ++k) sb[k] = s9[k];
  okvis::ceres::ImuError::covariance_t P;
  okvis::ceres::ImuError::jacobian_t F;
  Eigen::Vector3d adi, ai;
  double dt = 0;
  const int used = okvis::ceres::ImuError::propagation(imu, ip, T_WS, sb, t0, t1, &P, &F, adi, ai, dt);
  std::printf("imu %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g | %.17g %.17g %.17g | %.17g %.17g %.17g %.17g | P00 %.17g P_3_9 %.17g F_0_9 %.17g F_9_0 %.17g\n", used,
              T_WS.r()[0], T_WS.r()[1], T_WS.r()[2], T_WS.q().x(), T_WS.q().y(), T_WS.q().z(), T_WS.q().w(), sb[0], sb[1], sb[2], adi[0], ai[1],
              dt, sb[4], P(0, 0), P(3, 9), F(0, 9), F(9, 0));
  // first overload, no covariance
  okvis::kinematics::Transformation T2(Eigen::Vector3d(T[0], T[1], T[2]), Eigen::Quaterniond(T[6], T[3], T[4], T[5]));
  okvis::SpeedAndBias sb2;
  for (int k = 0; k < 9; ++k) sb2[k] = s9[k];
  const int used2 = okvis::ceres::ImuError::propagation(imu, ip, T2, sb2, t0, t1);
  std::printf("imu1 %d %.17g %.17g\n", used2, T2.r()[0], sb2[2]);
  // one reprojection residual
  std::string dist;
  int w, h, nIntr;
  in >> dist >> w >> h >> nIntr;
  std::vector<double> intr(nIntr);
  for (double& v : intr) in >> v;
  double Tws[7], hp[4], Tsc[7], uv[2], info[4];
  for (double& v : Tws) in >> v;
  for (double& v : hp) in >> v;
  for (double& v : Tsc) in >> v;
  for (double& v : uv) in >> v;
  for (double& v : info) in >> v;
  auto geo = std::make_shared<okvis::cameras::CameraBase>(w, h, dist, intr);
  Eigen::Matrix<double, 2, 2> information;
  information(0, 0) = info[0]; information(0, 1) = info[1]; information(1, 0) = info[2]; information(1, 1) = info[3];
  okvis::ceres::ReprojectionError<okvis::cameras::CameraBase> err(geo, 0, Eigen::Vector2d(uv[0], uv[1]), information);
  double r[2], J0[14], J1[8], J2[14], M0[12], M1[6], M2[12];
  double* J[3] = {J0, J1, J2};
  double* M[3] = {M0, M1, M2};
  const double* params[3] = {Tws, hp, Tsc};
  const bool ok = err.EvaluateWithMinimalJacobians(params, r, J, M);
  std::printf("reproj %d %.17g %.17g | %.17g %.17g %.17g | %.17g %.17g | %.17g\n", (int)ok, r[0], r[1], M0[0], M0[11], M1[5], J0[6], J1[3], J2[13]);
  // the triangulator's usage: pose Jacobian not requested
  double* J_b[3] = {nullptr, J1, nullptr};
  double* M_b[3] = {nullptr, M1, nullptr};
  const bool ok2 = err.EvaluateWithMinimalJacobians(params, r, J_b, M_b) && err.Evaluate(params, r, nullptr);
  std::printf("reproj2 %d %.17g\n", (int)ok2, M1[0]);
  return 0;
}
