// ORACLE -- test infrastructure only (never linked into the product library).
//
// CPU restatement of SVIn's global pose-graph optimisation (BASELINE config #5):
//   /root/reference/pose_graph/src/pose_graph/PoseGraph.cpp:226-385  optimize4DoFPoseGraph
//   /root/reference/pose_graph/src/pose_graph/PoseGraph.cpp:387-543  optimize6DoFPoseGraph
//   /root/reference/pose_graph/include/pose_graph/PoseGraph.h:85-231  NormalizeAngle, YawAngleFunctor,
//        YawPitchRollToRotationMatrix, FourDOFError, FourDOFWeightError
//   /root/reference/pose_graph/include/pose_graph/Pose3DError.h:103-147  PoseGraph3dErrorTerm
//   /root/reference/pose_graph/include/utils/Utils.h:71-103  R2ypr, ypr2R
//   /root/reference/pose_graph/src/pose_graph/Keyframe.cpp:495-500, :576-582  loop-edge measurements
// PARITY UNPINNED against a running reference: pose_graph has no unit tests, golden vectors or fixtures in the
// reference, and neither pose_graph nor Ceres can be built here (Ceres / Eigen / OpenCV / ROS absent).  What pins this
// file: central differences of the error terms through the same manifold Plus, an independent numpy restatement of
// the problem construction + error terms + Huber (costs equal to 1e-10), scipy least_squares landing on the same
// minimum, convergence on loop-closure graphs, dense vs envelope solver agreement (tests/test_oracle_posegraph.py).
// The reference differentiates its error functors with ceres::AutoDiffCostFunction (exact derivatives); the
// Jacobians below are the analytic equivalents (checked against central differences in the tests).
// Third-party arithmetic restated from its published algorithm (Ceres Solver 2.2.0, not vendored):
//   TrustRegionMinimizer + LevenbergMarquardtStrategy with default options (trust_region_minimizer.cc,
//   levenberg_marquardt_strategy.cc), HuberLoss + Corrector (loss_function.cc, corrector.cc),
//   EigenQuaternionManifold::Plus/PlusJacobian (manifold.cc), SPARSE_NORMAL_CHOLESKY as an exact solve of the
//   damped normal equations (dense Cholesky for small graphs, envelope Cholesky in natural order otherwise).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "orc_math.hpp"

namespace orc {
namespace pg {

constexpr double kPi = 3.14159265358979323846;

// PoseGraph.h:85-93
inline double normalizeAngle(double deg) {
  if (deg > 180.0) return deg - 360.0;
  if (deg < -180.0) return deg + 360.0;
  return deg;
}
// Utils.h:71-86 (degrees)
inline void r2ypr(const double* R, double* ypr) {
  const double n0 = R[0], n1 = R[3], n2 = R[6];  // column 0
  const double o0 = R[1], o1 = R[4];             // column 1
  const double a0 = R[2], a1 = R[5];             // column 2
  const double y = std::atan2(n1, n0);
  const double p = std::atan2(-n2, n0 * std::cos(y) + n1 * std::sin(y));
  const double r = std::atan2(a0 * std::sin(y) - a1 * std::cos(y), -o0 * std::sin(y) + o1 * std::cos(y));
  ypr[0] = y / kPi * 180.0; ypr[1] = p / kPi * 180.0; ypr[2] = r / kPi * 180.0;
}
// PoseGraph.h:110-127 == Utils.h:88-103 (Rz Ry Rx, degrees); dR = d/d(yaw degree)
inline void ypr2R(double yaw, double pitch, double roll, double* R, double* dR_dyaw = nullptr) {
  const double y = yaw / 180.0 * kPi, p = pitch / 180.0 * kPi, r = roll / 180.0 * kPi;
  const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
  R[0] = cy * cp; R[1] = -sy * cr + cy * sp * sr; R[2] = sy * sr + cy * sp * cr;
  R[3] = sy * cp; R[4] = cy * cr + sy * sp * sr;  R[5] = -cy * sr + sy * sp * cr;
  R[6] = -sp;     R[7] = cp * sr;                 R[8] = cp * cr;
  if (dR_dyaw) {
    const double k = kPi / 180.0;
    dR_dyaw[0] = -sy * cp * k; dR_dyaw[1] = (-cy * cr - sy * sp * sr) * k; dR_dyaw[2] = (cy * sr - sy * sp * cr) * k;
    dR_dyaw[3] = cy * cp * k;  dR_dyaw[4] = (-sy * cr + cy * sp * sr) * k; dR_dyaw[5] = (sy * sr + cy * sp * cr) * k;
    dR_dyaw[6] = 0; dR_dyaw[7] = 0; dR_dyaw[8] = 0;
  }
}

struct Keyframe {
  int index = 0, sequence = 0;
  double t[3] = {0, 0, 0};
  double q[4] = {0, 0, 0, 1};        // [x y z w]   (t, q) = Keyframe::getSVInPose, never modified by the optimisation
  double P[3] = {0, 0, 0};           // Keyframe::getPose: the drift-corrected / optimised pose (updatePose)
  double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  bool hasLoop = false;
  int loopIndex = -1;
  double loopT[3] = {0, 0, 0}, loopQ[4] = {0, 0, 0, 1}, loopYaw = 0;
};

// ---------------------------------------------------------------- edges
struct Edge {
  int a = 0, b = 0;           // local indices: a = connected / earlier keyframe ("i"), b = current ("j")
  bool loop = false;
  // 4-DoF (FourDOFError / FourDOFWeightError): t_meas, relative_yaw, pitch_i, roll_i
  // 6-DoF (PoseGraph3dErrorTerm): t_meas, q_meas, sqrt information diagonal
  double t[3] = {0, 0, 0}, relYaw = 0, pitch = 0, roll = 0;
  double q[4] = {0, 0, 0, 1}, sqrtInfo[6] = {1, 1, 1, 1, 1, 1};
};

struct Summary {
  double initial_cost = 0, final_cost = 0;
  int iterations = 0, termination = 1, num_successful_steps = 0;  // 0 convergence, 1 no convergence
};

class Graph {
 public:
  bool sixDof = false;
  int maxIterations = 10;
  std::vector<Keyframe> kfs;
  // local problem
  std::vector<double> yaw, pitch, roll;   // 4-DoF: degrees (yaw variable)
  std::vector<double> t;                  // 3 per node
  std::vector<double> q;                  // 6-DoF: 4 per node [x y z w]
  std::vector<char> fixed;
  std::vector<Edge> edges;
  std::vector<int> off;                   // tangent offset per node (-1 fixed)
  int n = 0, m = 0;                       // tangent size, residual count
  Summary summary;
  bool useEnvelope = false;

  int dofPerNode() const { return sixDof ? 6 : 4; }
  int resPerEdge() const { return sixDof ? 6 : 4; }

  // PoseGraph.cpp:262-332 / :436-489
  void build(int earliestLoopIndex, int curIndex) {
    yaw.clear(); pitch.clear(); roll.clear(); t.clear(); q.clear(); fixed.clear(); edges.clear();
    std::vector<int> local(kfs.size(), -1), seq;
    int i = 0;
    for (size_t k = 0; k < kfs.size(); ++k) {
      const Keyframe& kf = kfs[k];
      if (kf.index < earliestLoopIndex) continue;
      local[k] = i;
      double R[9], ypr[3];
      q2R(kf.q, R);
      r2ypr(R, ypr);
      yaw.push_back(ypr[0]); pitch.push_back(ypr[1]); roll.push_back(ypr[2]);
      t.insert(t.end(), kf.t, kf.t + 3);
      q.insert(q.end(), kf.q, kf.q + 4);
      seq.push_back(kf.sequence);
      fixed.push_back(sixDof ? (kf.index == earliestLoopIndex || kf.sequence == 0) : (kf.index <= earliestLoopIndex));
      const int nSeq = sixDof ? 4 : 2;
      for (int j = 1; j <= nSeq; ++j) {
        if (i - j >= 0 && seq[i] == seq[i - j]) {
          Edge e;
          e.a = i - j; e.b = i; e.loop = false;
          const double* qa = &q[4 * (i - j)];
          double Ra[9], d[3] = {t[3 * i] - t[3 * (i - j)], t[3 * i + 1] - t[3 * (i - j) + 1], t[3 * i + 2] - t[3 * (i - j) + 2]};
          q2R(qa, Ra);
          mat3T_vec(Ra, d, e.t);  // q_a^-1 * (t_i - t_a)
          if (!sixDof) {
            e.relYaw = yaw[i] - yaw[i - j];
            e.pitch = pitch[i - j]; e.roll = roll[i - j];
          } else {
            double qai[4];
            qinv(qa, qai);
            qmul(qai, &q[4 * i], e.q);
            const double si[6] = {20, 20, 20, 100, 100, 57.3};
            std::memcpy(e.sqrtInfo, si, sizeof(si));
          }
          edges.push_back(e);
        }
      }
      if (kf.hasLoop) {
        int ci = -1;
        for (size_t kk = 0; kk < kfs.size(); ++kk)
          if (kfs[kk].index == kf.loopIndex) ci = local[kk];
        if (ci >= 0) {
          Edge e;
          e.a = ci; e.b = i; e.loop = true;
          std::memcpy(e.t, kf.loopT, sizeof(e.t));
          if (!sixDof) {
            e.relYaw = kf.loopYaw;
            e.pitch = pitch[ci]; e.roll = roll[ci];
          } else {
            std::memcpy(e.q, kf.loopQ, sizeof(e.q));
            const double si[6] = {20, 20, 20, 100, 100, 100};
            std::memcpy(e.sqrtInfo, si, sizeof(si));
          }
          edges.push_back(e);
        }
      }
      if (kf.index == curIndex) { ++i; break; }
      ++i;
    }
    const int nNodes = (int)fixed.size();
    off.assign(nNodes, -1);
    n = 0;
    for (int k = 0; k < nNodes; ++k)
      if (!fixed[k]) { off[k] = n; n += dofPerNode(); }
    m = (int)edges.size() * resPerEdge();
  }

  // residual (4 or 6) and minimal Jacobians (res x dof) w.r.t. node a and node b, before the loss
  // 4-DoF tangent order: [yaw, tx, ty, tz]; 6-DoF: [tx, ty, tz, dq(3)] with q <- [sin|d| d/|d|, cos|d|] * q
  void evalEdge(const Edge& e, double* r, double* Ja, double* Jb) const {
    if (!sixDof) {
      double R[9], dR[9];
      ypr2R(yaw[e.a], e.pitch, e.roll, R, dR);
      const double d[3] = {t[3 * e.b] - t[3 * e.a], t[3 * e.b + 1] - t[3 * e.a + 1], t[3 * e.b + 2] - t[3 * e.a + 2]};
      double ti[3], dti[3];
      mat3T_vec(R, d, ti);
      mat3T_vec(dR, d, dti);
      const double w = 1.0;                       // FourDOFWeightError::weight (PoseGraph.h:182)
      const double wy = e.loop ? w / 10.0 : 1.0;  // PoseGraph.h:206
      const double wt = e.loop ? w : 1.0;
      for (int k = 0; k < 3; ++k) r[k] = (ti[k] - e.t[k]) * wt;
      r[3] = normalizeAngle(yaw[e.b] - yaw[e.a] - e.relYaw) * wy;
      std::memset(Ja, 0, sizeof(double) * 16);
      std::memset(Jb, 0, sizeof(double) * 16);
      for (int k = 0; k < 3; ++k) {
        Ja[k * 4 + 0] = dti[k] * wt;
        for (int c = 0; c < 3; ++c) {
          Ja[k * 4 + 1 + c] = -R[c * 3 + k] * wt;  // -(R^T)[k][c]
          Jb[k * 4 + 1 + c] = R[c * 3 + k] * wt;
        }
      }
      Ja[3 * 4 + 0] = -wy;
      Jb[3 * 4 + 0] = wy;
    } else {
      const double* pa = &t[3 * e.a];
      const double* pb = &t[3 * e.b];
      const double* qa = &q[4 * e.a];
      const double* qb = &q[4 * e.b];
      double Ra[9], d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]}, pab[3];
      // Eigen's q.conjugate() * v for a (near-)unit quaternion: R(q)^T v
      q2R(qa, Ra);
      mat3T_vec(Ra, d, pab);
      double qac[4] = {-qa[0], -qa[1], -qa[2], qa[3]}, qab[4], qabc[4], dq[4];
      qmul(qac, qb, qab);
      qabc[0] = -qab[0]; qabc[1] = -qab[1]; qabc[2] = -qab[2]; qabc[3] = qab[3];
      qmul(e.q, qabc, dq);
      double u[6] = {pab[0] - e.t[0], pab[1] - e.t[1], pab[2] - e.t[2], 2 * dq[0], 2 * dq[1], 2 * dq[2]};
      for (int k = 0; k < 6; ++k) r[k] = e.sqrtInfo[k] * u[k];
      std::memset(Ja, 0, sizeof(double) * 36);
      std::memset(Jb, 0, sizeof(double) * 36);
      // position rows
      double dx[9], RtX[9], Rt[9];
      crossMx(d, dx);
      transpose<3, 3>(Ra, Rt);
      matmul<3, 3, 3>(Rt, dx, RtX);
      for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
          Ja[k * 6 + c] = -Rt[k * 3 + c] * e.sqrtInfo[k];
          Jb[k * 6 + c] = Rt[k * 3 + c] * e.sqrtInfo[k];
          Ja[k * 6 + 3 + c] = 2.0 * RtX[k * 3 + c] * e.sqrtInfo[k];
        }
      // orientation rows: e = 2 vec(q_m q_b^-1 [d_a,1] q_a) -> de/dd_a = 2 [plus(A) oplus(q_a)]_{3x3}, A = q_m q_b^-1;
      // a perturbation of q_b enters as q_b^-1 [-d_b,1]  -> de/dd_b = -de/dd_a
      double qbc[4] = {-qb[0], -qb[1], -qb[2], qb[3]}, A[4], PA[16], OB[16], Mm[16];
      qmul(e.q, qbc, A);
      qplusMat(A, PA);
      qoplusMat(qa, OB);
      matmul<4, 4, 4>(PA, OB, Mm);
      for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) {
          Ja[(3 + k) * 6 + 3 + c] = 2.0 * Mm[k * 4 + c] * e.sqrtInfo[3 + k];
          Jb[(3 + k) * 6 + 3 + c] = -2.0 * Mm[k * 4 + c] * e.sqrtInfo[3 + k];
        }
    }
  }

  // Plus on the local problem (YawAngleFunctor PoseGraph.h:95-108; EigenQuaternionManifold::Plus)
  void plus(const std::vector<double>& delta) {
    for (size_t k = 0; k < fixed.size(); ++k) {
      if (off[k] < 0) continue;
      const double* d = &delta[off[k]];
      if (!sixDof) {
        yaw[k] = normalizeAngle(yaw[k] + d[0]);
        for (int c = 0; c < 3; ++c) t[3 * k + c] += d[1 + c];
      } else {
        for (int c = 0; c < 3; ++c) t[3 * k + c] += d[c];
        const double nd = norm3(d + 3);
        if (nd > 0.0) {
          const double s = std::sin(nd) / nd;
          const double dq[4] = {s * d[3], s * d[4], s * d[5], std::cos(nd)};
          double out[4];
          qmul(dq, &q[4 * k], out);
          std::memcpy(&q[4 * k], out, sizeof(out));
        }
      }
    }
  }

  // HuberLoss(0.1) (loss_function.cc): rho(s), rho'(s), rho''(s)
  static void huber(double s, double a, double* rho) {
    const double b = a * a;
    if (s > b) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }

  // cost = sum 0.5 rho(|r|^2); optionally the robustified residual vector and the dense Jacobian (m x n, row-major)
  double evaluate(std::vector<double>* res, std::vector<double>* J) const {
    const int R = resPerEdge(), D = dofPerNode();
    double cost = 0;
    if (res) res->assign(m, 0.0);
    if (J) J->assign((size_t)m * n, 0.0);
    for (size_t ei = 0; ei < edges.size(); ++ei) {
      const Edge& e = edges[ei];
      double r[6], Ja[36], Jb[36];
      evalEdge(e, r, Ja, Jb);
      double s = 0;
      for (int k = 0; k < R; ++k) s += r[k] * r[k];
      double scale = 1.0;
      if (e.loop) {
        double rho[3];
        huber(s, 0.1, rho);
        cost += 0.5 * rho[0];
        scale = std::sqrt(rho[1]);  // corrector.cc: rho'' <= 0 -> plain sqrt(rho') scaling of residual and Jacobian
      } else {
        cost += 0.5 * s;
      }
      if (res)
        for (int k = 0; k < R; ++k) (*res)[ei * R + k] = scale * r[k];
      if (J) {
        for (int k = 0; k < R; ++k) {
          double* row = &(*J)[(ei * R + k) * (size_t)n];
          if (off[e.a] >= 0)
            for (int c = 0; c < D; ++c) row[off[e.a] + c] += scale * Ja[k * D + c];
          if (off[e.b] >= 0)
            for (int c = 0; c < D; ++c) row[off[e.b] + c] += scale * Jb[k * D + c];
        }
      }
    }
    return cost;
  }

  // exact solve of (A + diag(dd)) x = b, A = Js^T Js symmetric (dense lower Cholesky)
  static bool solveSpd(std::vector<double>& A, int n, std::vector<double>& b) {
    for (int j = 0; j < n; ++j) {
      double s = A[(size_t)j * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(s > 0)) return false;
      const double ljj = std::sqrt(s);
      A[(size_t)j * n + j] = ljj;
      for (int i = j + 1; i < n; ++i) {
        double v = A[(size_t)i * n + j];
        for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = v / ljj;
      }
    }
    for (int i = 0; i < n; ++i) {
      double v = b[i];
      for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * b[k];
      b[i] = v / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = b[i];
      for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * b[k];
      b[i] = v / A[(size_t)i * n + i];
    }
    return true;
  }

  // Envelope (profile) Cholesky in natural order for large graphs: row i of the lower triangle is stored from its
  // first non-zero column first[i]; fill stays inside the envelope.  Same arithmetic as the dense factorisation
  // restricted to the envelope.
  struct Envelope {
    int n = 0;
    std::vector<int> first;
    std::vector<size_t> ptr;   // row i occupies val[ptr[i] .. ptr[i] + i - first[i]]
    std::vector<double> val;
    double& at(int i, int j) { return val[ptr[i] + (j - first[i])]; }
  };
  bool solveEnvelope(Envelope& E, std::vector<double>& b) const {
    const int nn = E.n;
    for (int i = 0; i < nn; ++i) {
      for (int j = E.first[i]; j <= i; ++j) {
        double s = E.at(i, j);
        const int k0 = std::max(E.first[i], E.first[j]);
        for (int k = k0; k < j; ++k) s -= E.at(i, k) * E.at(j, k);
        if (j < i) E.at(i, j) = s / E.at(j, j);
        else { if (!(s > 0)) return false; E.at(i, i) = std::sqrt(s); }
      }
    }
    for (int i = 0; i < nn; ++i) {
      double v = b[i];
      for (int k = E.first[i]; k < i; ++k) v -= E.at(i, k) * b[k];
      b[i] = v / E.at(i, i);
    }
    for (int i = nn - 1; i >= 0; --i) {
      const double v = b[i] / E.at(i, i);
      b[i] = v;
      for (int k = E.first[i]; k < i; ++k) b[k] -= E.at(i, k) * v;
    }
    return true;
  }

  // Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy, default options (see file header)
  void solve() {
    summary = Summary();
    if (n == 0) { summary.termination = 0; return; }
    const int R = resPerEdge(), D = dofPerNode();
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    // sparse storage of the Jacobian: per edge two D-wide blocks
    std::vector<double> res(m), JaAll((size_t)edges.size() * R * D), JbAll((size_t)edges.size() * R * D);
    auto evalAll = [&](bool withJac) {
      double cost = 0;
      for (size_t ei = 0; ei < edges.size(); ++ei) {
        const Edge& e = edges[ei];
        double r[6], Ja[36], Jb[36];
        evalEdge(e, r, Ja, Jb);
        double s = 0;
        for (int k = 0; k < R; ++k) s += r[k] * r[k];
        double sc = 1.0;
        if (e.loop) { double rho[3]; huber(s, 0.1, rho); cost += 0.5 * rho[0]; sc = std::sqrt(rho[1]); }
        else cost += 0.5 * s;
        if (withJac) {
          for (int k = 0; k < R; ++k) res[ei * R + k] = sc * r[k];
          for (int k = 0; k < R * D; ++k) { JaAll[ei * R * D + k] = sc * Ja[k]; JbAll[ei * R * D + k] = sc * Jb[k]; }
        }
      }
      return cost;
    };
    double x_cost = evalAll(true);
    summary.initial_cost = x_cost;
    std::vector<double> scale(n, 1.0), g(n), colsq(n), diagonal(n), step(n), delta(n);
    auto gradientAndNorms = [&]() {  // unscaled gradient J^T r and squared column norms
      std::fill(g.begin(), g.end(), 0.0);
      std::fill(colsq.begin(), colsq.end(), 0.0);
      for (size_t ei = 0; ei < edges.size(); ++ei) {
        const Edge& e = edges[ei];
        for (int side = 0; side < 2; ++side) {
          const int o = off[side ? e.b : e.a];
          if (o < 0) continue;
          const double* Jx = (side ? JbAll : JaAll).data() + ei * R * D;
          for (int k = 0; k < R; ++k)
            for (int c = 0; c < D; ++c) { g[o + c] += Jx[k * D + c] * res[ei * R + k]; colsq[o + c] += Jx[k * D + c] * Jx[k * D + c]; }
        }
      }
    };
    gradientAndNorms();
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(colsq[i]));  // jacobi_scaling, fixed at iteration 0
    auto gradMax = [&]() { double mx = 0; for (double v : g) mx = std::max(mx, std::fabs(v)); return mx; };
    auto xNorm = [&]() {
      double s = 0;
      for (size_t k = 0; k < fixed.size(); ++k) {
        if (off[k] < 0) continue;
        if (!sixDof) s += yaw[k] * yaw[k];
        else for (int c = 0; c < 4; ++c) s += q[4 * k + c] * q[4 * k + c];
        for (int c = 0; c < 3; ++c) s += t[3 * k + c] * t[3 * k + c];
      }
      return std::sqrt(s);
    };
    // envelope structure (first non-zero column per row) when requested
    Envelope env;
    if (useEnvelope) {
      env.n = n;
      env.first.resize(n);
      for (int i = 0; i < n; ++i) env.first[i] = i;
      for (const Edge& e : edges) {
        const int oa = off[e.a], ob = off[e.b];
        if (oa < 0 || ob < 0) continue;
        const int lo = std::min(oa, ob), hi = std::max(oa, ob);
        for (int c = 0; c < D; ++c) env.first[hi + c] = std::min(env.first[hi + c], lo);
      }
      for (int k = 0; k < (int)fixed.size(); ++k)
        if (off[k] >= 0)
          for (int c = 0; c < D; ++c) env.first[off[k] + c] = std::min(env.first[off[k] + c], off[k]);
      env.ptr.resize(n + 1);
      size_t tot = 0;
      for (int i = 0; i < n; ++i) { env.ptr[i] = tot; tot += (size_t)(i - env.first[i] + 1); }
      env.ptr[n] = tot;
      env.val.resize(tot);
    }
    int iteration = 0, invalid = 0;
    bool lastSuccessful = false;
    summary.termination = 1;
    auto finish = [&](int term) { summary.termination = term; summary.final_cost = x_cost; summary.iterations = iteration; };
    std::vector<double> A;
    while (true) {
      if (lastSuccessful) summary.num_successful_steps++;
      if (iteration >= maxIterations) { finish(1); return; }
      if (gradMax() <= gradient_tolerance) { finish(0); return; }
      if (radius <= min_radius) { finish(0); return; }
      ++iteration;
      lastSuccessful = false;
      // --- LevenbergMarquardtStrategy::ComputeStep on the scaled Jacobian Js = J diag(scale)
      if (!reuse_diagonal)
        for (int i = 0; i < n; ++i) diagonal[i] = std::min(std::max(colsq[i] * scale[i] * scale[i], min_diag), max_diag);
      // normal equations (Js^T Js + diag(diagonal / radius)) y = Js^T r ; step = -y
      std::vector<double> rhs(n);
      for (int i = 0; i < n; ++i) rhs[i] = g[i] * scale[i];
      bool ok;
      auto addBlocks = [&](auto&& put) {
        for (size_t ei = 0; ei < edges.size(); ++ei) {
          const Edge& e = edges[ei];
          const int oa = off[e.a], ob = off[e.b];
          const double* Ja = JaAll.data() + ei * R * D;
          const double* Jb = JbAll.data() + ei * R * D;
          for (int c1 = 0; c1 < D; ++c1)
            for (int c2 = 0; c2 < D; ++c2) {
              double saa = 0, sbb = 0, sba = 0;
              for (int k = 0; k < R; ++k) {
                saa += Ja[k * D + c1] * Ja[k * D + c2];
                sbb += Jb[k * D + c1] * Jb[k * D + c2];
                sba += Jb[k * D + c1] * Ja[k * D + c2];
              }
              if (oa >= 0) put(oa + c1, oa + c2, saa * scale[oa + c1] * scale[oa + c2]);
              if (ob >= 0) put(ob + c1, ob + c2, sbb * scale[ob + c1] * scale[ob + c2]);
              if (oa >= 0 && ob >= 0) {
                put(ob + c1, oa + c2, sba * scale[ob + c1] * scale[oa + c2]);
                put(oa + c2, ob + c1, sba * scale[ob + c1] * scale[oa + c2]);
              }
            }
        }
      };
      if (!useEnvelope) {
        A.assign((size_t)n * n, 0.0);
        addBlocks([&](int i, int j, double v) { A[(size_t)i * n + j] += v; });
        for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += diagonal[i] / radius;
        ok = solveSpd(A, n, rhs);
      } else {
        std::fill(env.val.begin(), env.val.end(), 0.0);
        addBlocks([&](int i, int j, double v) { if (j <= i) env.at(i, j) += v; });
        for (int i = 0; i < n; ++i) env.at(i, i) += diagonal[i] / radius;
        ok = solveEnvelope(env, rhs);
      }
      reuse_diagonal = true;
      double model_cost_change = 0;
      if (ok) {
        for (int i = 0; i < n; ++i) { step[i] = -rhs[i]; if (!std::isfinite(step[i])) ok = false; }
      }
      if (ok) {
        // model_cost_change = -(Js step) . (r + Js step / 2)
        double acc = 0;
        for (size_t ei = 0; ei < edges.size(); ++ei) {
          const Edge& e = edges[ei];
          const int oa = off[e.a], ob = off[e.b];
          for (int k = 0; k < R; ++k) {
            double mr = 0;
            if (oa >= 0) for (int c = 0; c < D; ++c) mr += JaAll[(ei * R + k) * D + c] * scale[oa + c] * step[oa + c];
            if (ob >= 0) for (int c = 0; c < D; ++c) mr += JbAll[(ei * R + k) * D + c] * scale[ob + c] * step[ob + c];
            acc += mr * (res[ei * R + k] + 0.5 * mr);
          }
        }
        model_cost_change = -acc;
        if (!(model_cost_change > 0.0)) ok = false;
      }
      if (!ok) {  // HandleInvalidStep -> strategy->StepIsInvalid()
        if (++invalid >= 5) { finish(3); return; }
        radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
        continue;
      }
      invalid = 0;
      for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
      // candidate
      const std::vector<double> yaw0 = yaw, t0 = t, q0 = q;
      const double x_norm = xNorm();
      plus(delta);
      double step_norm = 0;
      for (size_t k = 0; k < fixed.size(); ++k) {
        if (off[k] < 0) continue;
        if (!sixDof) step_norm += (yaw[k] - yaw0[k]) * (yaw[k] - yaw0[k]);
        else for (int c = 0; c < 4; ++c) step_norm += (q[4 * k + c] - q0[4 * k + c]) * (q[4 * k + c] - q0[4 * k + c]);
        for (int c = 0; c < 3; ++c) step_norm += (t[3 * k + c] - t0[3 * k + c]) * (t[3 * k + c] - t0[3 * k + c]);
      }
      step_norm = std::sqrt(step_norm);
      const double cand_cost = evalAll(false);
      auto restore = [&]() { yaw = yaw0; t = t0; q = q0; };
      if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { restore(); finish(0); return; }
      const double cost_change = x_cost - cand_cost;
      if (std::fabs(cost_change) <= function_tolerance * x_cost) { restore(); finish(0); return; }
      const double rel = cost_change / model_cost_change;
      if (rel > min_relative_decrease) {
        x_cost = evalAll(true);
        gradientAndNorms();
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
        radius = std::min(max_radius, radius);
        decrease_factor = 2.0;
        reuse_diagonal = false;
        lastSuccessful = true;
      } else {
        restore();
        radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      }
    }
  }

  // drift of the odometry frame against the optimised map (PoseGraph.cpp:356-363 / :521-526)
  double yawDrift = 0, rDrift[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tDrift[3] = {0, 0, 0};
  // addKeyframe's pose update (PoseGraph.cpp:127-132): pose = drift * SVIn pose
  void applyDrift(Keyframe& kf) const {
    double R[9];
    q2R(kf.q, R);
    mat3_vec(rDrift, kf.t, kf.P);
    for (int c = 0; c < 3; ++c) kf.P[c] += tDrift[c];
    matmul<3, 3, 3>(rDrift, R, kf.Rp);
  }
  // PoseGraph.cpp:340-375 / :504-534: write the optimised poses back (4-DoF: ypr2R(yaw, pitch, roll)), update the
  // drift from the current keyframe and move the keyframes after it
  void writeBack(int earliestLoopIndex, int curIndex) {
    int i = 0;
    size_t k = 0;
    Keyframe* cur = nullptr;
    for (; k < kfs.size(); ++k) {
      Keyframe& kf = kfs[k];
      if (kf.index < earliestLoopIndex) continue;
      for (int c = 0; c < 3; ++c) kf.P[c] = t[3 * i + c];
      if (!sixDof) {
        double R[9], qq[4];
        ypr2R(yaw[i], pitch[i], roll[i], R);
        r2q(R, qq);      // tmp_q = ypr2R(...); tmp_r = tmp_q.toRotationMatrix()
        q2R(qq, kf.Rp);
      } else {
        q2R(&q[4 * i], kf.Rp);
      }
      if (kf.index == curIndex) { cur = &kf; break; }
      ++i;
    }
    if (!cur) return;
    double Rs[9];
    q2R(cur->q, Rs);
    if (!sixDof) {
      double a[3], b[3];
      r2ypr(cur->Rp, a);
      r2ypr(Rs, b);
      yawDrift = a[0] - b[0];
      ypr2R(yawDrift, 0, 0, rDrift);
    } else {
      double Rt[9], a[3];
      transpose<3, 3>(cur->Rp, Rt);
      matmul<3, 3, 3>(Rt, Rs, rDrift);   // r_drift = cur_r.transpose() * svin_r (PoseGraph.cpp:523)
      r2ypr(rDrift, a);
      yawDrift = a[0];
    }
    double rs[3];
    mat3_vec(rDrift, cur->t, rs);
    for (int c = 0; c < 3; ++c) tDrift[c] = cur->P[c] - rs[c];
    for (++k; k < kfs.size(); ++k) applyDrift(kfs[k]);
  }
  // Eigen Quaterniond(Matrix3d) (Shepperd)
  static void r2q(const double* R, double* qo) {
    const double tr = R[0] + R[4] + R[8];
    double x, y, z, w;
    if (tr > 0) {
      double s = std::sqrt(tr + 1.0);
      w = 0.5 * s; s = 0.5 / s;
      x = (R[7] - R[5]) * s; y = (R[2] - R[6]) * s; z = (R[3] - R[1]) * s;
    } else {
      int i = 0;
      if (R[4] > R[0]) i = 1;
      if (R[8] > R[i * 4]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
      double qv[3];
      qv[i] = 0.5 * s; s = 0.5 / s;
      w = (R[k * 3 + j] - R[j * 3 + k]) * s;
      qv[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
      qv[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
      x = qv[0]; y = qv[1]; z = qv[2];
    }
    qo[0] = x; qo[1] = y; qo[2] = z; qo[3] = w;
  }
};

}  // namespace pg
}  // namespace orc

// ---------------------------------------------------------------- C API (tests, bench cpu_baseline)
using orc::pg::Graph;
using orc::pg::Keyframe;
extern "C" {
void* orc_pg_create(int six_dof, int max_iterations) {
  Graph* g = new Graph();
  g->sixDof = six_dof != 0;
  g->maxIterations = max_iterations > 0 ? max_iterations : (six_dof ? 5 : 10);
  return g;
}
void orc_pg_destroy(void* h) { delete static_cast<Graph*>(h); }
void orc_pg_set_envelope(void* h, int on) { static_cast<Graph*>(h)->useEnvelope = on != 0; }
// pose: t[3], q[4] (x y z w); loop: loop_index (< 0: none), rel_t[3], rel_q[4], rel_yaw (degrees)
void orc_pg_add_keyframe(void* h, int index, int sequence, const double* t, const double* q, int loop_index,
                         const double* rel_t, const double* rel_q, double rel_yaw) {
  Keyframe kf;
  kf.index = index; kf.sequence = sequence;
  std::memcpy(kf.t, t, sizeof(kf.t));
  std::memcpy(kf.q, q, sizeof(kf.q));
  if (loop_index >= 0) {
    kf.hasLoop = true; kf.loopIndex = loop_index;
    std::memcpy(kf.loopT, rel_t, sizeof(kf.loopT));
    std::memcpy(kf.loopQ, rel_q, sizeof(kf.loopQ));
    kf.loopYaw = rel_yaw;
  }
  static_cast<Graph*>(h)->applyDrift(kf);
  static_cast<Graph*>(h)->kfs.push_back(kf);
}
// returns the number of iterations; summary5 = initial_cost, final_cost, iterations, termination, successful steps
int orc_pg_optimize(void* h, int earliest_loop_index, int cur_index, double* summary5) {
  Graph* g = static_cast<Graph*>(h);
  g->build(earliest_loop_index, cur_index);
  g->solve();
  g->writeBack(earliest_loop_index, cur_index);
  if (summary5) {
    summary5[0] = g->summary.initial_cost; summary5[1] = g->summary.final_cost; summary5[2] = g->summary.iterations;
    summary5[3] = g->summary.termination; summary5[4] = g->summary.num_successful_steps;
  }
  return g->summary.iterations;
}
int orc_pg_num_keyframes(void* h) { return (int)static_cast<Graph*>(h)->kfs.size(); }
void orc_pg_get_pose(void* h, int k, double* t, double* q) {
  const Keyframe& kf = static_cast<Graph*>(h)->kfs.at(k);
  std::memcpy(t, kf.P, sizeof(kf.P));
  Graph::r2q(kf.Rp, q);
}
// yaw_drift (degrees), r_drift (3x3 row-major), t_drift
void orc_pg_get_drift(void* h, double* yaw, double* r, double* t) {
  const Graph* g = static_cast<Graph*>(h);
  if (yaw) *yaw = g->yawDrift;
  if (r) std::memcpy(r, g->rDrift, sizeof(g->rDrift));
  if (t) std::memcpy(t, g->tDrift, sizeof(g->tDrift));
}
// inspection: builds the local problem and returns sizes; then edge-level evaluation for the Jacobian checks
int orc_pg_build(void* h, int earliest_loop_index, int cur_index, int* n_tangent, int* n_edges) {
  Graph* g = static_cast<Graph*>(h);
  g->build(earliest_loop_index, cur_index);
  if (n_tangent) *n_tangent = g->n;
  if (n_edges) *n_edges = (int)g->edges.size();
  return (int)g->fixed.size();
}
void orc_pg_eval_edge(void* h, int e, int* a, int* b, int* is_loop, double* r, double* Ja, double* Jb) {
  Graph* g = static_cast<Graph*>(h);
  const orc::pg::Edge& ed = g->edges.at(e);
  if (a) *a = ed.a;
  if (b) *b = ed.b;
  if (is_loop) *is_loop = ed.loop ? 1 : 0;
  g->evalEdge(ed, r, Ja, Jb);
}
// perturb node k of the local problem by a tangent vector (Plus) -- numeric differentiation in the tests
void orc_pg_perturb_node(void* h, int k, const double* d) {
  Graph* g = static_cast<Graph*>(h);
  std::vector<double> delta(g->n, 0.0);
  if (g->off.at(k) < 0) {  // temporarily treat as free
    const int D = g->dofPerNode();
    std::vector<int> off = g->off;
    std::vector<double> dl(D);
    g->off.assign(off.size(), -1);
    g->off[k] = 0;
    for (int c = 0; c < D; ++c) dl[c] = d[c];
    g->plus(dl);
    g->off = off;
    return;
  }
  for (int c = 0; c < g->dofPerNode(); ++c) delta[g->off[k] + c] = d[c];
  g->plus(delta);
}
double orc_pg_cost(void* h) { return static_cast<Graph*>(h)->evaluate(nullptr, nullptr); }
void orc_pg_linearize(void* h, double* res, double* J) {
  Graph* g = static_cast<Graph*>(h);
  std::vector<double> r, Jm;
  g->evaluate(&r, &Jm);
  if (res) std::memcpy(res, r.data(), sizeof(double) * r.size());
  if (J) std::memcpy(J, Jm.data(), sizeof(double) * Jm.size());
}
void orc_pg_ypr(const double* q, double* ypr) {
  double R[9];
  orc::q2R(q, R);
  orc::pg::r2ypr(R, ypr);
}
}
