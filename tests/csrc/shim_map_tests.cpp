// Two programs of the reference's test suite (and a third-party error term), re-created against integration/okvis/ceres/Map.hpp (the graph is built block by
// block through okvis::ceres::Map, solved on the GPU, the estimates are read back from the caller's parameter-block objects):
//   part 1  okvis_ceres/test/TestHomogeneousPointError.cpp:57-99 -- 100 points, one HomogeneousPointError (variance 0.1)
//           each, points disturbed, isJacobianCorrect per residual, solve, final_cost < 1e-10;
//   part 2  okvis_ceres/test/TestMap.cpp:60-156 (the Pose2d re-solve of :146-156 included): pose + constant extrinsics + N CONSTANT points ("no point optimization", :93) with
//           Cauchy-robustified ReprojectionError<equidistant pinhole>, some residuals / blocks removed again, 10 iterations,
//           the pose must come back to the truth (quaternion 1e-2, translation 1e-1: the reference's thresholds).
//   part 3  Map::addResidualBlock with an error term of the caller's own (ErrorInterface only: no kernel exists for it), evaluated by the host.
// Prints one line per part for tests/test_gpu_shim.py.
#include <okvis/MultiFrame.hpp>
#include <okvis/ceres/Map.hpp>

#include <cmath>
#include <cstdio>
#include <memory>

namespace {
struct Rng {   // deterministic uniform numbers in [-1, 1)
  uint64_t s = 0x2545F4914F6CDD1Dull;
  double next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  }
};
void quatRotate(const double q[4], const double v[3], double out[3]) {   // q = (x, y, z, w)
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                       2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  for (int i = 0; i < 3; ++i) out[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
void quatMul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// An error term the library has no kernel for (part 3): the distance between the positions of two poses held to a value.  Written
// the way a third party writes one against okvis::ceres: ErrorInterface, analytic minimal Jacobians (the position part of a pose's
// minimal coordinates is the position itself, PoseManifold.cpp:59-82).
class DistanceError : public okvis::ceres::ErrorInterface {
 public:
  DistanceError(double distance, double weight) : d_(distance), w_(weight) {}
  size_t residualDim() const override { return 1; }
  size_t parameterBlocks() const override { return 2; }
  size_t parameterBlockDim(size_t) const override { return 7; }
  std::string typeInfo() const override { return "DistanceError"; }
  bool EvaluateWithMinimalJacobians(double const* const* p, double* r, double** J, double** Jm) const override {
    const double dx = p[1][0] - p[0][0], dy = p[1][1] - p[0][1], dz = p[1][2] - p[0][2];
    const double n = std::sqrt(dx * dx + dy * dy + dz * dz);
    r[0] = w_ * (n - d_);
    const double g[3] = {w_ * dx / n, w_ * dy / n, w_ * dz / n};
    for (int b = 0; b < 2; ++b) {
      const double sgn = b == 0 ? -1.0 : 1.0;
      if (J && J[b]) { for (int k = 0; k < 7; ++k) J[b][k] = k < 3 ? sgn * g[k] : 0.0; }
      if (Jm && Jm[b]) { for (int k = 0; k < 6; ++k) Jm[b][k] = k < 3 ? sgn * g[k] : 0.0; }
    }
    return true;
  }
 private:
  double d_, w_;
};
okvis::kinematics::Transformation makeT(const double r[3], const double q[4]) {
  return okvis::kinematics::Transformation(Eigen::Vector3d(r[0], r[1], r[2]), Eigen::Quaterniond(q[3], q[0], q[1], q[2]));
}
}  // namespace

int main() {
  Rng rng;
  {  // ---------------------------------------------------------------- part 1
    okvis::ceres::Map map;
    int jacOk = 0;
    std::vector<std::shared_ptr<okvis::ceres::HomogeneousPointParameterBlock> > blocks;
    std::vector<Eigen::Vector4d> truth;
    for (size_t i = 0; i < 100; ++i) {
      Eigen::Vector4d point(100.0 * rng.next(), 100.0 * rng.next(), 100.0 * rng.next(), 1.0);
      std::shared_ptr<okvis::ceres::HomogeneousPointParameterBlock> block(new okvis::ceres::HomogeneousPointParameterBlock(point, i + 1));
      if (!map.addParameterBlock(block, okvis::ceres::Map::HomogeneousPoint)) return 3;
      map.setParameterBlockVariable(i + 1);
      std::shared_ptr<okvis::ceres::HomogeneousPointError> err(new okvis::ceres::HomogeneousPointError(block->estimate(), 0.1));
      ::ceres::ResidualBlockId id = map.addResidualBlock(err, NULL, block);
      if (!id) return 4;
      Eigen::Vector4d disturbed(point[0] + 0.2 * rng.next(), point[1] + 0.2 * rng.next(), point[2] + 0.2 * rng.next(), 1.0);
      block->setEstimate(disturbed);
      jacOk += map.isJacobianCorrect(id) ? 1 : 0;
      blocks.push_back(block);
      truth.push_back(point);
    }
    map.options.minimizer_progress_to_stdout = false;
    map.solve();
    double worst = 0;
    for (size_t i = 0; i < blocks.size(); ++i)
      for (int k = 0; k < 3; ++k) worst = std::max(worst, std::fabs(blocks[i]->estimate()[k] - truth[i][k]));
    std::printf("hpe final_cost %.6e initial_cost %.6e jac_ok %d worst %.3e iterations %d\n", map.summary.final_cost, map.summary.initial_cost,
                jacOk, worst, (int)map.summary.iterations.size() - 1);
  }
  {  // ---------------------------------------------------------------- part 2
    double rWS[3] = {10.0 * rng.next(), 10.0 * rng.next(), 10.0 * rng.next()};
    double qWS[4] = {rng.next(), rng.next(), rng.next(), rng.next()};
    double n = std::sqrt(qWS[0] * qWS[0] + qWS[1] * qWS[1] + qWS[2] * qWS[2] + qWS[3] * qWS[3]);
    for (double& v : qWS) v /= n;
    double dq[4] = {0.005 * rng.next(), 0.005 * rng.next(), 0.005 * rng.next(), 1.0};
    n = std::sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    for (double& v : dq) v /= n;
    double qInit[4], rInit[3] = {rWS[0] + 0.3 * rng.next(), rWS[1] + 0.3 * rng.next(), rWS[2] + 0.3 * rng.next()};
    quatMul(qWS, dq, qInit);
    const double rSC[3] = {0.1, -0.05, 0.02};
    double qSC[4] = {0.02, -0.01, 0.03, 1.0};
    n = std::sqrt(qSC[0] * qSC[0] + qSC[1] * qSC[1] + qSC[2] * qSC[2] + qSC[3] * qSC[3]);
    for (double& v : qSC) v /= n;
    std::shared_ptr<okvis::ceres::PoseParameterBlock> pose(new okvis::ceres::PoseParameterBlock(makeT(rInit, qInit), 1, okvis::Time(0, 0)));
    std::shared_ptr<okvis::ceres::PoseParameterBlock> extr(new okvis::ceres::PoseParameterBlock(makeT(rSC, qSC), 2, okvis::Time(0, 0)));
    okvis::ceres::Map map;
    if (!map.addParameterBlock(pose, okvis::ceres::Map::Pose6d) || !map.addParameterBlock(extr, okvis::ceres::Map::Pose6d)) return 5;
    map.setParameterBlockConstant(extr);
    // PinholeCamera<EquidistantDistortion>::createTestObject (PinholeCamera.hpp:276-280, EquidistantDistortion test coefficients)
    typedef okvis::cameras::CameraBase Geometry;
    std::shared_ptr<const Geometry> geometry(new Geometry(752, 480, "EquidistantDistortion", {350.0, 360.0, 378.0, 238.0, -0.21, 0.14, 0.0006, 0.0003}));
    ::ceres::CauchyLoss loss(1);
    const size_t N = 300;
    int jacOk = 0, removedBlocks = 0, removedResiduals = 0;
    const double Tws[7] = {rWS[0], rWS[1], rWS[2], qWS[0], qWS[1], qWS[2], qWS[3]};
    const double Tsc[7] = {rSC[0], rSC[1], rSC[2], qSC[0], qSC[1], qSC[2], qSC[3]};
    const double intr[4] = {350.0, 360.0, 378.0, 238.0}, dist[4] = {-0.21, 0.14, 0.0006, 0.0003}, zero2[2] = {0, 0}, eye2[4] = {1, 0, 0, 1};
    for (size_t i = 0; i < N; ++i) {
      const double depth = (double)(i % 10) * 3 + 2.0;
      const double pc[3] = {0.6 * rng.next() * depth, 0.4 * rng.next() * depth, depth};
      double ps[3], pw[3];
      quatRotate(qSC, pc, ps);
      for (int k = 0; k < 3; ++k) ps[k] += rSC[k];
      quatRotate(qWS, ps, pw);
      for (int k = 0; k < 3; ++k) pw[k] += rWS[k];
      const double hp[4] = {pw[0], pw[1], pw[2], 1.0};
      double r[2];
      if (svin_host_reprojection_error(SVIN_DIST_EQUIDISTANT, intr, dist, 4, Tws, hp, Tsc, zero2, eye2, r, nullptr, nullptr, nullptr, nullptr,
                                       nullptr, nullptr) != 1) return 6;
      Eigen::Vector2d kp(-r[0] + rng.next(), -r[1] + rng.next());   // the projection (residual = measurement - projection) + noise
      Eigen::Vector4d start(pw[0], pw[1], pw[2], 1.0);
      std::shared_ptr<okvis::ceres::HomogeneousPointParameterBlock> point(new okvis::ceres::HomogeneousPointParameterBlock(start, i + 3));
      if (!map.addParameterBlock(point, okvis::ceres::Map::HomogeneousPoint)) return 7;
      if (!map.setParameterBlockConstant(point)) return 8;   // no point optimization (TestMap.cpp:93)
      okvis::ceres::ReprojectionError<Geometry>::covariance_t information;
      information(0, 0) = 1.0; information(1, 1) = 1.0; information(0, 1) = 0.0; information(1, 0) = 0.0;
      std::shared_ptr<okvis::ceres::ReprojectionError<Geometry> > cost(new okvis::ceres::ReprojectionError<Geometry>(geometry, 1, kp, information));
      ::ceres::ResidualBlockId id = map.addResidualBlock(cost, &loss, pose, point, extr);
      if (!id) return 9;
      jacOk += map.isJacobianCorrect(id) ? 1 : 0;
      if (i % 10 == 0) {   // "randomly delete some just for fun to test" (TestMap.cpp:117-122)
        if (i % 20 == 0) removedBlocks += map.removeParameterBlock(point) ? 1 : 0;
        else removedResiduals += map.removeResidualBlock(id) ? 1 : 0;
      }
    }
    map.options.max_num_iterations = 10;
    map.solve();
    const okvis::kinematics::Transformation est = pose->estimate();
    const double qe[4] = {est.q().x(), est.q().y(), est.q().z(), est.q().w()};
    const double qinv[4] = {-qe[0], -qe[1], -qe[2], qe[3]};
    double qd[4];
    quatMul(qWS, qinv, qd);
    const double dRot = 2.0 * std::sqrt(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2]);
    const double dTr = std::sqrt((est.r()[0] - rWS[0]) * (est.r()[0] - rWS[0]) + (est.r()[1] - rWS[1]) * (est.r()[1] - rWS[1]) +
                                 (est.r()[2] - rWS[2]) * (est.r()[2] - rWS[2]));
    std::printf("map final_cost %.6e initial_cost %.6e jac_ok %d of %d removed_blocks %d removed_residuals %d d_rot %.3e d_trans %.3e iterations %d exists3 %d\n",
                map.summary.final_cost, map.summary.initial_cost, jacOk, (int)N, removedBlocks, removedResiduals, dRot, dTr,
                (int)map.summary.iterations.size() - 1, map.parameterBlockExists(3) ? 1 : 0);
    // "also try out the resetting of parameterization" (TestMap.cpp:146-156): Pose2d, roll / pitch disturbed by 0.01, solve again.
    // Beyond the reference's test (which only runs it): the position must not move at all and the cost must come back down.
    if (!map.resetParameterization(pose->id(), okvis::ceres::Map::Pose2d)) return 10;
    if (map.resetParameterization(777777, okvis::ceres::Map::Pose2d)) return 11;   // unknown block: false (Map.cpp:514)
    const double cost6 = map.summary.final_cost;
    double dq2[4] = {0.5 * 0.01 * rng.next(), 0.5 * 0.01 * rng.next(), 0.0, 1.0};   // oplus: q <- exp(dalpha) * q, dalpha = (a, b, 0)
    n = std::sqrt(dq2[0] * dq2[0] + dq2[1] * dq2[1] + dq2[2] * dq2[2] + dq2[3] * dq2[3]);
    for (double& v : dq2) v /= n;
    double qStart[4];
    quatMul(dq2, qe, qStart);
    const double rStart[3] = {est.r()[0], est.r()[1], est.r()[2]};
    pose->setEstimate(makeT(rStart, qStart));
    map.solve();
    const okvis::kinematics::Transformation est2 = pose->estimate();
    const double dPos = std::fabs(est2.r()[0] - rStart[0]) + std::fabs(est2.r()[1] - rStart[1]) + std::fabs(est2.r()[2] - rStart[2]);
    const double qe2[4] = {est2.q().x(), est2.q().y(), est2.q().z(), est2.q().w()};
    const double qinv2[4] = {-qe2[0], -qe2[1], -qe2[2], qe2[3]};
    double qd2[4];
    quatMul(qWS, qinv2, qd2);
    std::printf("map2d final_cost %.9e initial_cost %.9e cost6 %.9e d_pos %.3e d_rot %.3e iterations %d\n", map.summary.final_cost,
                map.summary.initial_cost, cost6, dPos, 2.0 * std::sqrt(qd2[0] * qd2[0] + qd2[1] * qd2[1] + qd2[2] * qd2[2]),
                (int)map.summary.iterations.size() - 1);
  }
  {  // ---------------------------------------------------------------- part 3
    // Map::addResidualBlock with an error term of the caller's own (Map.cpp:341-376 takes any cost function): two poses, a PoseError on
    // each (the device's kernels), and a DistanceError between them that the HOST evaluates between the launches.  The priors want the
    // poses 1 m apart, the distance term (weight 100 against information 1) wants 2 m: the solve must end near 2 m, and
    // isJacobianCorrect must accept the term's Jacobians.
    okvis::ceres::Map map;
    const double q0[4] = {0, 0, 0, 1}, rA[3] = {0, 0, 0}, rB[3] = {1, 0.2, -0.1};
    std::shared_ptr<okvis::ceres::PoseParameterBlock> A(new okvis::ceres::PoseParameterBlock(makeT(rA, q0), 1, okvis::Time(0, 0)));
    std::shared_ptr<okvis::ceres::PoseParameterBlock> B(new okvis::ceres::PoseParameterBlock(makeT(rB, q0), 2, okvis::Time(0, 0)));
    if (!map.addParameterBlock(A, okvis::ceres::Map::Pose6d) || !map.addParameterBlock(B, okvis::ceres::Map::Pose6d)) return 12;
    Eigen::Matrix<double, 6, 6> info;   // (the stand-in Eigen of the test tree has no Identity())
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) info(a, b) = a == b ? 1.0 : 0.0;
    std::shared_ptr<okvis::ceres::PoseError> pa(new okvis::ceres::PoseError(makeT(rA, q0), info)), pb(new okvis::ceres::PoseError(makeT(rB, q0), info));
    if (!map.addResidualBlock(pa, NULL, A) || !map.addResidualBlock(pb, NULL, B)) return 13;
    std::shared_ptr<DistanceError> dist(new DistanceError(2.0, 100.0));
    ::ceres::ResidualBlockId id = map.addResidualBlock(dist, NULL, A, B);
    if (!id) return 14;
    const int jac = map.isJacobianCorrect(id) ? 1 : 0;
    map.options.max_num_iterations = 20;
    map.solve();
    const okvis::kinematics::Transformation ea = A->estimate(), eb = B->estimate();
    const double dd = std::sqrt((eb.r()[0] - ea.r()[0]) * (eb.r()[0] - ea.r()[0]) + (eb.r()[1] - ea.r()[1]) * (eb.r()[1] - ea.r()[1]) +
                                (eb.r()[2] - ea.r()[2]) * (eb.r()[2] - ea.r()[2]));
    const bool removed = map.removeResidualBlock(id);
    map.solve();   // without the term the priors pull the poses back
    const okvis::kinematics::Transformation fa = A->estimate(), fb = B->estimate();
    const double d2 = std::sqrt((fb.r()[0] - fa.r()[0]) * (fb.r()[0] - fa.r()[0]) + (fb.r()[1] - fa.r()[1]) * (fb.r()[1] - fa.r()[1]) +
                                (fb.r()[2] - fa.r()[2]) * (fb.r()[2] - fa.r()[2]));
    std::printf("host jac_ok %d distance %.6f final_cost %.6e initial_cost %.6e removed %d distance_after_removal %.6f\n", jac, dd, map.summary.final_cost,
                map.summary.initial_cost, removed ? 1 : 0, d2);
  }
  return 0;
}
