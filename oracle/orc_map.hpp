// ORACLE -- test infrastructure only (never linked into the product library).
//
// Restatement of okvis::ceres::Map (okvis_ceres/include/okvis/ceres/Map.hpp:65-420,
// okvis_ceres/src/Map.cpp) plus a CPU restatement of the part of Ceres Solver 2.2.0
// (pinned tag f3356504f954d1fbc8b5daba0aeef2f5de5fa196, NOT vendored in the reference)
// that `Map::solve()` (Map.hpp:347) reaches with the options Estimator::optimize sets
// (Estimator.cpp:878-890): trust-region minimiser, TRADITIONAL_DOGLEG, SPARSE_SCHUR
// (landmarks eliminated first), Jacobi scaling, loss-function corrector.  The Ceres
// algorithm is restated from its published documentation/source semantics; solver
// *trajectory* parity with the real Ceres is UNPINNED (it cannot be built here).
#pragma once
#include <map>
#include <unordered_map>
#include <string>
#include "orc_errors.hpp"

namespace orc {

struct ParamBlock {
  uint64_t id = 0;
  int type = BLOCK_POSE;
  bool fixed = false;
  double x[9] = {0};
  int dim() const { return blockDim(type); }
  int mdim() const { return blockMinDim(type); }
  // Map::resetParameterization (Map.cpp:513-543) on a pose block: 6 = PoseManifold, 3 / 4 / 2 = PoseManifold3d / 4d / 2d
  // (PoseManifold.cpp:173-466).  Only the SOLVER sees it (Ceres takes the tangent size from the manifold); the error terms' own
  // minimal Jacobians, getLhs and isJacobianCorrect keep six columns, as in the reference (ParameterBlock::minimalDimension()).
  int manifold = 6;
  int sdim() const { return type == BLOCK_POSE ? manifold : mdim(); }
};
// component of the 6-vector (dr, dalpha) that minimal coordinate c of a reduced pose manifold drives:
// PoseManifold3d::Plus delta_[3..5] (PoseManifold.cpp:176-178), 4d: delta_[0..2], delta_[5] (:279-282), 2d: delta_[3..4] (:375-376)
inline int poseTangentIndex(int manifold, int c) {
  if (manifold == 3) return 3 + c;
  if (manifold == 4) return c < 3 ? c : 5;
  if (manifold == 2) return 3 + c;
  return c;
}

struct ResidualBlock {
  uint64_t id = 0;
  std::shared_ptr<ErrorTerm> err;
  int loss = LOSS_NONE;
  double lossParam = 1.0;
  std::vector<uint64_t> params;
};

struct SolverOptions {
  int max_num_iterations = 50;
  int min_iterations = 0;       // CeresIterationCallback (CeresIterationCallback.hpp:57-99)
  double time_limit = -1.0;     // seconds, <0: none
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  bool jacobi_scaling = true;
  int num_threads = 1;          // Estimator::optimize numThreads (Estimator.cpp:889): residual evaluation + Schur elimination
  bool verbose = false;
};
struct SolverSummary {
  double initial_cost = 0, final_cost = 0;
  int iterations = 0;            // total (successful + unsuccessful), excluding iteration 0
  int num_successful_steps = 0;
  int termination = 0;           // 0 convergence, 1 no convergence (max iter), 2 user (time), 3 failure
  double total_time = 0;
  std::vector<double> cost_history;
};

class MarginalizationError;

class Map {
 public:
  // --- graph (Map.cpp:255-620)
  bool addParameterBlock(uint64_t id, int type, const double* x);
  bool removeParameterBlock(uint64_t id);  // cascades residual removal (Map.cpp:322-333)
  bool parameterBlockExists(uint64_t id) const { return params_.count(id) != 0; }
  ParamBlock& param(uint64_t id) { return params_.at(id); }
  const ParamBlock& param(uint64_t id) const { return params_.at(id); }
  bool setParameterBlockConstant(uint64_t id) { if (!params_.count(id)) return false; params_[id].fixed = true; return true; }
  bool setParameterBlockVariable(uint64_t id) { if (!params_.count(id)) return false; params_[id].fixed = false; return true; }
  bool resetParameterization(uint64_t id, int manifold);   // Map.cpp:513-543; manifold = 6 / 3 / 4 / 2 on a pose block
  uint64_t addResidualBlock(std::shared_ptr<ErrorTerm> err, int loss, const std::vector<uint64_t>& paramIds);
  bool removeResidualBlock(uint64_t resId);
  std::vector<uint64_t> residuals(uint64_t paramId) const;   // copy, insertion order
  const ResidualBlock& residual(uint64_t resId) const { return residuals_.at(resId); }
  bool residualExists(uint64_t resId) const { return residuals_.count(resId) != 0; }
  size_t numResiduals() const { return residuals_.size(); }
  size_t numParams() const { return params_.size(); }
  const std::map<uint64_t, ParamBlock>& params() const { return params_; }
  const std::map<uint64_t, ResidualBlock>& residualMap() const { return residuals_; }

  // --- Map.cpp:105-150 / :153-252
  void getLhs(uint64_t paramId, double* H) const;  // mdim x mdim
  bool isJacobianCorrect(uint64_t resId, double relTol, double* worstRel = nullptr) const;

  // --- solve (Map.hpp:341-347)
  SolverOptions options;
  SolverSummary summary;
  void solve();

  // debug / parity helpers: linearise at the current point (loss-corrected local Jacobians);
  // returns the Schur-reduced system of the *undamped* normal equations when mu == 0.
  struct Linearization {
    std::vector<uint64_t> camIds;   // order of the reduced system
    std::vector<int> camOffsets;
    int d = 0;
    std::vector<double> S, g;       // reduced d x d, d
    std::vector<double> A, b;       // un-reduced camera part
    std::vector<uint64_t> lmIds;
    std::vector<double> V, bl;      // per landmark 9 / 3
    double cost = 0;
  };
  void linearize(Linearization& out, double mu);

 private:
  std::map<uint64_t, ParamBlock> params_;
  std::map<uint64_t, ResidualBlock> residuals_;
  std::unordered_map<uint64_t, std::vector<uint64_t>> param2res_;
  uint64_t nextResId_ = 1;
};

}  // namespace orc
