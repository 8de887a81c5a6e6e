// integration/okvis/ceres/Map.hpp -- the part of okvis::ceres::Map that survives on the MI355X backend.
//
// Drop-in for okvis_ceres/include/okvis/ceres/Map.hpp:65-420 (implementation src/Map.cpp) as far as callers OUTSIDE
// okvis_ceres see it: the public `options` / `summary` members Estimator::optimize and the tests write and read
// (Map.hpp:341-347), `solve()`, and the graph queries.  The graph itself lives in libsvin_ba.so (the handle the owning
// okvis::Estimator created); this class is a view onto it.  What does NOT survive, and why:
//   * addParameterBlock / addResidualBlock with caller-supplied ::ceres::CostFunction objects (Map.cpp:255-376): a
//     device solver cannot call virtual CPU cost functions.  The window is built through okvis::Estimator
//     (addStates / addLandmark / addObservation), which is how the pipeline builds it anyway.
//   * ::ceres::Problem / Manifold pointers (DO_NOT_TAKE_OWNERSHIP plumbing, Map.cpp:59-64): there is no Ceres.
//   * computeCovariance (Map.hpp:352-372): debug code of the reference, never called.
#ifndef INTEGRATION_OKVIS_CERES_MAP_HPP_
#define INTEGRATION_OKVIS_CERES_MAP_HPP_

#include <svin_ba.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

// The frontend stores residual ids as ::ceres::ResidualBlockId (implementation/Estimator.hpp:84 casts them to uint64_t).
// Without Ceres in the build the name is provided here: an opaque pointer-sized handle carrying the core's id.
#ifndef CERES_PUBLIC_TYPES_H_
namespace ceres {
struct ResidualBlock;
typedef ResidualBlock* ResidualBlockId;
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
}  // namespace ceres
#endif

namespace okvis {
namespace ceres {

class Map {
 public:
  /// The fields of ::ceres::Solver::Options the reference sets (Estimator.cpp:878-890) or tests touch.  The device
  /// solver IS Ceres 2.2's trust-region minimiser with DOGLEG + Schur elimination restated, so
  /// linear_solver_type / trust_region_strategy_type are accepted and recorded, not dispatched on.
  struct Options {
    ::ceres::LinearSolverType linear_solver_type = ::ceres::SPARSE_SCHUR;
    ::ceres::TrustRegionStrategyType trust_region_strategy_type = ::ceres::DOGLEG;
    int num_threads = 1;               ///< accepted and ignored: the solve runs on the GPU
    int max_num_iterations = 50;
    bool minimizer_progress_to_stdout = false;
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    std::vector<void*> callbacks;      ///< Estimator::setOptimizationTimeLimit registers its callback here in the reference;
                                       ///< the time limit is a property of the handle now (svin_ba_set_optimization_time_limit)
  };
  /// ::ceres::Solver::Summary subset (Map.hpp:344)
  struct Summary {
    double initial_cost = 0, final_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    std::vector<int> iterations;       ///< one entry per iteration incl. iteration 0, like ceres (size() - 1 = iterations done)
    ::ceres::TerminationType termination_type = ::ceres::NO_CONVERGENCE;
    double total_time_in_seconds = 0;
    std::string BriefReport() const {
      char b[256];
      std::snprintf(b, sizeof(b), "svin_ba: iterations %d, initial cost %.6e, final cost %.6e, termination %d", (int)iterations.size() - 1,
                    initial_cost, final_cost, (int)termination_type);
      return b;
    }
    std::string FullReport() const { return BriefReport(); }
  };

  enum Parameterization { HomogeneousPoint, Pose6d, Pose3d, Pose4d, Pose2d, Trivial };   // Map.hpp:97-105

  typedef std::pair<uint64_t, int> ResidualBlockSpec;                    ///< residual id, kind (svin_ba_parameters_of)
  typedef std::vector<uint64_t> ResidualBlockCollection;                 ///< residual ids (Map::residuals)
  typedef std::vector<uint64_t> ParameterBlockCollection;                ///< parameter block ids (Map::parameters)

  explicit Map(svin_ba* handle = nullptr) : h_(handle) {}
  void attach(svin_ba* handle) { h_ = handle; }
  svin_ba* handle() const { return h_; }

  Options options;   ///< public like the reference's (Map.hpp:341)
  Summary summary;   ///< public like the reference's (Map.hpp:344)

  /// Map::solve (Map.hpp:347): ::ceres::Solve(options, problem, &summary)
  void solve() {
    need();
    svin_ba_set_solver_tolerances(h_, options.function_tolerance, options.gradient_tolerance, options.parameter_tolerance);
    if (svin_ba_optimize(h_, (uint64_t)options.max_num_iterations, (uint64_t)options.num_threads,
                         options.minimizer_progress_to_stdout ? 1 : 0) < 0)
      throw std::runtime_error(std::string("svin_ba_optimize: ") + svin_ba_last_error());
    svin_summary s;
    svin_ba_get_summary(h_, &s);
    summary.initial_cost = s.initial_cost;
    summary.final_cost = s.final_cost;
    summary.num_successful_steps = s.num_successful_steps;
    summary.num_unsuccessful_steps = s.iterations - s.num_successful_steps;
    summary.iterations.assign((size_t)s.iterations + 1, 0);
    summary.total_time_in_seconds = s.total_time_s;
    summary.termination_type = s.termination == 0 ? ::ceres::CONVERGENCE
                               : s.termination == 1 ? ::ceres::NO_CONVERGENCE
                               : s.termination == 2 ? ::ceres::USER_SUCCESS : ::ceres::FAILURE;
  }

  bool parameterBlockExists(uint64_t id) const { need(); return svin_ba_parameter_block_exists(h_, id) == 1; }   // Map.cpp:77-80
  bool setParameterBlockConstant(uint64_t id) { need(); return svin_ba_set_parameter_block_constant(h_, id, 1) == 1; }  // :495-501
  bool setParameterBlockVariable(uint64_t id) { need(); return svin_ba_set_parameter_block_constant(h_, id, 0) == 1; }  // :504-510
  bool isParameterBlockConstant(uint64_t id) const { need(); return svin_ba_is_parameter_block_constant(h_, id) == 1; } // ParameterBlock::fixed()
  /// Map::resetParameterization (Map.cpp:513-543).  The estimator only ever uses Pose6d for poses / extrinsics, the
  /// homogeneous-point manifold for landmarks and none for speed/bias (Estimator.cpp:186-238, :417); the other pose
  /// manifolds appear in commented-out code only (:801) and are not available on the device.
  bool resetParameterization(uint64_t id, int parameterization) const {
    need();
    if (svin_ba_parameter_block_exists(h_, id) != 1) return false;
    return parameterization == Pose6d || parameterization == HomogeneousPoint || parameterization == Trivial;
  }
  /// Map::residuals (Map.cpp:576-587): every residual touching the block, in insertion order
  ResidualBlockCollection residuals(uint64_t id) const {
    need();
    ResidualBlockCollection out;
    const int n = svin_ba_residuals_of(h_, id, nullptr, 0);
    if (n <= 0) return out;
    out.resize((size_t)n);
    svin_ba_residuals_of(h_, id, out.data(), n);
    return out;
  }
  /// Map::parameters (Map.cpp:602-620): the blocks of a residual in the cost function's parameter order
  ParameterBlockCollection parameters(::ceres::ResidualBlockId residual) const { return parameters(reinterpret_cast<uint64_t>(residual)); }
  ParameterBlockCollection parameters(uint64_t residualId, int* kind = nullptr) const {
    need();
    uint64_t ids[64];
    int32_t k = -1;
    const int n = svin_ba_parameters_of(h_, residualId, ids, 64, &k);
    if (kind) *kind = k;
    return n > 0 ? ParameterBlockCollection(ids, ids + n) : ParameterBlockCollection();
  }
  /// Map::removeResidualBlock (Map.cpp:467-492) for reprojection residuals (what Estimator::removeObservation does)
  bool removeResidualBlock(::ceres::ResidualBlockId residual) {
    need();
    return svin_ba_remove_observation_by_id(h_, reinterpret_cast<uint64_t>(residual)) == 1;
  }

 private:
  void need() const {
    if (!h_) throw std::runtime_error("okvis::ceres::Map (svin_ba shim): not attached to an estimator handle");
  }
  svin_ba* h_;
};

}  // namespace ceres
}  // namespace okvis

#endif  // INTEGRATION_OKVIS_CERES_MAP_HPP_
