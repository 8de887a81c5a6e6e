// Host side of the trust-region iteration: every DECISION Window::solve takes, as a HIP-free state machine.
//
// Ceres 2.2 TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG) semantics with the options Estimator::optimize sets
// (okvis_ceres/src/Estimator.cpp:878-890): monotonic steps, min_relative_decrease 1e-3, initial radius 1e4, the mu retry
// ladder of the dogleg strategy (min_mu 1e-8, max_mu 1, x10), five consecutive invalid steps = failure.  All numbers the
// decisions read come from ONE record per evaluation (SolverScalars); in the landmark-sharded mode every field read here is
// either all-reduced or computed redundantly from all-reduced data, so the ranks take identical decisions
// (tests/test_trust_region_host.py feeds two instances the same reduced fields and different rank-local ones).
// Included by window.cpp (the driver: launches + read-back) and by tests/csrc/trust_region_shim.cpp (g++, no GPU).
#pragma once
#include <algorithm>
#include <cmath>

namespace svin {

// the fields of SolverScalars the host reads (kernels.hpp); rank-invariant in sharded mode
struct TrScalars {
  double cost = 0;            // group A (summed over ranks)
  double stepNormSq = 0, xNormSq = 0;
  double gradMax = 0;         // max over the ranks' gathered values
  double failMax = 0;         // != 0: the reduced system or a landmark block was not positive definite, on some rank
  double jdSq = 0, jdDotR = 0, doglegStepNorm = 0;   // derived on every rank from all-reduced sums and the radius
};

struct TrustRegionHost {
  // options
  double fTol = 1e-6, gTol = 1e-10, pTol = 1e-8;
  int maxIterations = 10;
  // state
  double radius = 1e4, mu = 1e-8, x_cost = 0;
  bool reuse = false, initScale = true;
  int invalid = 0, iteration = 0, successful = 0;
  int termination = 1;   // 0 convergence, 1 max iterations, 2 time limit (callback), 3 failure
  bool stepOk = true;
  static constexpr double kMinMu = 1e-8, kMaxMu = 1.0, kMuIncrease = 10.0;

  void start(double initialCost) { x_cost = initialCost; }
  // top of an iteration: false = the loop ends here with `termination` set (stopRequested: the time-limit callback)
  bool beginIteration(bool stopRequested) {
    if (stopRequested) { termination = 2; return false; }
    if (iteration >= maxIterations) { termination = 1; return false; }
    if (radius <= 1e-32) { termination = 0; return false; }
    ++iteration;
    stepOk = true;
    return true;
  }
  // damping an accepted step would leave behind (what the speculative build of the next iteration assumes)
  double muAfterAccept() const { return std::max(kMinMu, 2.0 * mu / kMuIncrease); }
  // after the candidate evaluation of a FRESH linearisation: true = the factorisation failed and the same iteration is
  // retried with a larger mu (DoglegStrategy::ComputeStep's retry ladder); false = go on to endIteration()
  bool retryFactorisation(const TrScalars& sc) {
    if (reuse || sc.failMax == 0.0) return false;
    mu *= kMuIncrease;
    if (mu < kMaxMu) return true;
    stepOk = false;
    return false;
  }
  enum Outcome { kAccepted, kRejected, kInvalid, kTerminated };
  // the decision of the iteration; kTerminated: `termination` is set
  Outcome endIteration(const TrScalars& sc) {
    if (!reuse) { initScale = false; reuse = true; }
    if (sc.gradMax <= gTol) { --iteration; termination = 0; return kTerminated; }
    const double model_cost_change = -(sc.jdDotR + 0.5 * sc.jdSq);
    if (!stepOk || !(model_cost_change > 0.0)) {
      if (++invalid >= 5) { termination = 3; return kTerminated; }
      mu *= kMuIncrease;
      reuse = false;
      return kInvalid;
    }
    invalid = 0;
    const double step_norm = std::sqrt(sc.stepNormSq), x_norm = std::sqrt(sc.xNormSq);
    if (step_norm <= pTol * (x_norm + pTol)) { termination = 0; return kTerminated; }
    const double candidate_cost = sc.cost;
    const double cost_change = x_cost - candidate_cost;
    if (std::fabs(cost_change) <= fTol * x_cost) { termination = 0; return kTerminated; }
    relative_decrease = cost_change / model_cost_change;
    last_step_norm = step_norm;
    if (relative_decrease > 1e-3) {
      x_cost = candidate_cost;
      ++successful;
      if (relative_decrease < 0.25) radius *= 0.5;
      if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * sc.doglegStepNorm);
      radius = std::min(radius, 1e16);
      mu = muAfterAccept();
      reuse = false;
      return kAccepted;
    }
    radius *= 0.5;
    reuse = true;
    return kRejected;
  }
  double relative_decrease = 0, last_step_norm = 0;   // of the last accepted / rejected step (progress output)
};

}  // namespace svin
