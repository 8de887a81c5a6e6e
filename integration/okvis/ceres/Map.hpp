// integration/okvis/ceres/Map.hpp -- the part of okvis::ceres::Map that survives on the MI355X backend.
//
// Drop-in for okvis_ceres/include/okvis/ceres/Map.hpp:65-420 (implementation src/Map.cpp) as far as callers OUTSIDE
// okvis_ceres see it: the public `options` / `summary` members Estimator::optimize and the tests write and read
// (Map.hpp:341-347), `solve()`, and the graph queries.  The graph itself lives in libsvin_ba.so (the handle the owning
// okvis::Estimator created); this class is a view onto it.  What does NOT survive, and why:
//   * addResidualBlock with an ARBITRARY caller-supplied ::ceres::CostFunction (Map.cpp:341-376): a device solver cannot
//     call virtual CPU cost functions.  Since round 6 any error term that implements ErrorInterface is accepted and evaluated by
//     the host between the launches (a slow path, pose / speed-bias blocks, no loss); a bare ::ceres::CostFunction is not (there
//     is no Ceres).  The fast path: addParameterBlock, and addResidualBlock for the error-term
//     classes of this directory (PoseError, HomogeneousPointError, ReprojectionError<GEOMETRY> under CauchyLoss(1)) -- each
//     maps onto a factor kind of the device solver (svin_ba_map_*), so that a program shaped like the reference's own tests
//     (okvis_ceres/test/TestHomogeneousPointError.cpp:57-99, TestMap.cpp:60-150) builds its graph block by block, checks
//     Jacobians with isJacobianCorrect, solves and reads the estimates back from its parameter-block objects.
//   * ::ceres::Problem / Manifold pointers (DO_NOT_TAKE_OWNERSHIP plumbing, Map.cpp:59-64): there is no Ceres.
//   * computeCovariance (Map.hpp:352-372): debug code of the reference, never called.
#ifndef INTEGRATION_OKVIS_CERES_MAP_HPP_
#define INTEGRATION_OKVIS_CERES_MAP_HPP_

#include <svin_ba.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <okvis/Time.hpp>
#include <okvis/ceres/CeresTypes.hpp>   // ::ceres::ResidualBlockId & co. without Ceres
#include <okvis/ceres/ErrorInterface.hpp>
#include <okvis/ceres/HomogeneousPointError.hpp>
#include <okvis/ceres/HomogeneousPointParameterBlock.hpp>
#include <okvis/ceres/ParameterBlock.hpp>
#include <okvis/ceres/PoseError.hpp>
#include <okvis/ceres/PoseParameterBlock.hpp>
#include <okvis/ceres/ReprojectionError.hpp>
#include <okvis/ceres/SpeedAndBiasParameterBlock.hpp>

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

  /// Map.hpp:71-86.  lossFunctionPtr is always NULL here: the Cauchy loss of the reprojection residuals
  /// (Estimator.cpp:69) is applied inside the device solver, there is no ::ceres::LossFunction object to point to.
  struct ResidualBlockSpec {
    ResidualBlockSpec() : residualBlockId(0), lossFunctionPtr(0) {}
    ResidualBlockSpec(::ceres::ResidualBlockId id, ::ceres::LossFunction* loss, std::shared_ptr<ErrorInterface> e)
        : residualBlockId(id), lossFunctionPtr(loss), errorInterfacePtr(e) {}
    ::ceres::ResidualBlockId residualBlockId;
    ::ceres::LossFunction* lossFunctionPtr;
    std::shared_ptr<ErrorInterface> errorInterfacePtr;
  };
  typedef std::pair<uint64_t, std::shared_ptr<okvis::ceres::ParameterBlock> > ParameterBlockSpec;   // Map.hpp:87
  typedef std::vector<ResidualBlockSpec> ResidualBlockCollection;                                    // :90
  typedef std::vector<ParameterBlockSpec> ParameterBlockCollection;                                  // :93
  typedef std::unordered_map<uint64_t, std::shared_ptr<okvis::ceres::ParameterBlock> > Id2ParameterBlock_Map;   // :183

  /// What errorInterfacePtr returns for a residual of the device window: the sizes and the type of the error term.
  /// Evaluating it at caller-supplied parameters is not available on this object (the residual lives on the GPU:
  /// svin_ba_eval_reprojection / svin_ba_eval_factors evaluate the window's residuals in bulk) and throws.
  class ResidualView : public ErrorInterface {
   public:
    ResidualView(int kind, size_t residualDim, std::vector<size_t> dims) : kind_(kind), m_(residualDim), dims_(std::move(dims)) {}
    size_t residualDim() const override { return m_; }
    size_t parameterBlocks() const override { return dims_.size(); }
    size_t parameterBlockDim(size_t i) const override { return dims_.at(i); }
    bool EvaluateWithMinimalJacobians(double const* const*, double*, double**, double**) const override {
      throw Exception("okvis::ceres::Map (svin_ba shim): residuals of the device window are evaluated on the GPU "
                      "(svin_ba_eval_reprojection / svin_ba_eval_factors), not through ErrorInterface");
    }
    std::string typeInfo() const override {
      switch (kind_) {
        case 100: return "ReprojectionError";
        case 101: return "MarginalizationError";
        case 102: return "HomogeneousPointError";
        case 0: return "ImuError";
        case 1: return "PoseError";
        case 2: return "SpeedAndBiasError";
        case 3: return "RelativePoseError";
        case 4: return "SonarError";
        case 5: return "DepthError";
      }
      return "unknown";
    }
    int kind() const { return kind_; }   ///< svin_ba_parameters_of's kind
   private:
    int kind_;
    size_t m_;
    std::vector<size_t> dims_;
  };

  /// A Map of its own (the reference's tests construct one directly, TestMap.cpp:74) creates its backend handle with the
  /// first block; the Map inside an okvis::Estimator is attached to the estimator's.
  explicit Map(svin_ba* handle = nullptr) : h_(handle), owned_(false) {}
  ~Map() { if (owned_ && h_) svin_ba_destroy(h_); }
  Map(const Map&) = delete;
  Map& operator=(const Map&) = delete;
  void attach(svin_ba* handle) {
    if (owned_ && h_) svin_ba_destroy(h_);
    h_ = handle; owned_ = false;
  }
  svin_ba* handle() const { return h_; }

  // ---- graph building (Map.cpp:255-376)
  /// Map::addParameterBlock (Map.cpp:255-319).  The block object stays with the caller and the Map (shared_ptr, like the
  /// reference: ceres never owns it); solve() sends its current parameters to the device and writes the result back into it.
  bool addParameterBlock(std::shared_ptr<okvis::ceres::ParameterBlock> parameterBlock, int parameterization = Trivial, const int /*group*/ = -1) {
    ensureHandle();
    int type = -1;
    if (parameterBlock->dimension() == 7 && (parameterization == Pose6d || parameterization == Trivial)) type = 0;
    else if (parameterBlock->dimension() == 9 && parameterization == Trivial) type = 2;
    else if (parameterBlock->dimension() == 4 && (parameterization == HomogeneousPoint || parameterization == Trivial)) type = 3;
    if (type < 0) throw std::runtime_error("okvis::ceres::Map (svin_ba shim): this block / parameterisation pair has no device counterpart");
    if (svin_ba_map_add_parameter_block(h_, parameterBlock->id(), type, parameterBlock->parameters()) != 1) return false;   // id in use
    blocks_[parameterBlock->id()] = parameterBlock;
    if (parameterBlock->fixed()) svin_ba_set_parameter_block_constant(h_, parameterBlock->id(), 1);
    return true;
  }
  bool removeParameterBlock(uint64_t id) {   // Map.cpp:322-333
    need();
    if (svin_ba_map_remove_parameter_block(h_, id) != 1) return false;
    blocks_.erase(id);
    for (auto it = built_.begin(); it != built_.end();) {
      bool touches = false;
      for (const auto& b : it->second.blocks) touches |= b->id() == id;
      it = touches ? built_.erase(it) : std::next(it);
    }
    return true;
  }
  bool removeParameterBlock(std::shared_ptr<okvis::ceres::ParameterBlock> parameterBlock) { return removeParameterBlock(parameterBlock->id()); }
  bool setParameterBlockConstant(std::shared_ptr<okvis::ceres::ParameterBlock> b) { b->setFixed(true); return setParameterBlockConstant(b->id()); }
  bool setParameterBlockVariable(std::shared_ptr<okvis::ceres::ParameterBlock> b) { b->setFixed(false); return setParameterBlockVariable(b->id()); }

  /// Map::addResidualBlock (Map.cpp:341-376) for the error terms of this directory.  PoseError: no loss.
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<PoseError> e, ::ceres::LossFunction* loss, std::shared_ptr<okvis::ceres::ParameterBlock> x0) {
    need(); noLoss(loss);
    double meas[7], info[36];
    const okvis::kinematics::Transformation& T = e->measurement();
    for (int k = 0; k < 3; ++k) meas[k] = T.r()[k];
    meas[3] = T.q().x(); meas[4] = T.q().y(); meas[5] = T.q().z(); meas[6] = T.q().w();
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) info[a * 6 + b] = e->information()(a, b);
    return record(svin_ba_map_add_pose_error(h_, x0->id(), meas, info), e, {x0});
  }
  /// HomogeneousPointError(measurement, information) on a landmark block: no loss (TestHomogeneousPointError.cpp:74-78)
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<HomogeneousPointError> e, ::ceres::LossFunction* loss,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> x0) {
    need(); noLoss(loss);
    const double meas[4] = {e->measurement()[0], e->measurement()[1], e->measurement()[2], e->measurement()[3]};
    return record(svin_ba_add_homogeneous_point_error(h_, x0->id(), meas, e->informationRowMajor()), e, {x0});
  }
  /// ReprojectionError<GEOMETRY>(geometry, cameraId, measurement, information) under CauchyLoss(1) on (T_WS, hp_W, T_SC)
  /// (TestMap.cpp:104-108, Estimator::addObservation).  information: a multiple of the identity.
  template <class GEOMETRY_T>
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<ReprojectionError<GEOMETRY_T> > e, ::ceres::LossFunction* loss,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> pose, std::shared_ptr<okvis::ceres::ParameterBlock> point,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> extrinsics) {
    need();
    const ::ceres::CauchyLoss* cauchy = dynamic_cast<const ::ceres::CauchyLoss*>(loss);
    if (!cauchy || cauchy->a() != 1.0)
      throw std::runtime_error("okvis::ceres::Map (svin_ba shim): reprojection residuals are solved under CauchyLoss(1) (Estimator.cpp:69)");
    const void* key = e->cameraGeometry().get();
    auto it = cams_.find(key);
    if (it == cams_.end()) {
      const double sig[4] = {0, 0, 0, 0};
      const int cam = svin_ba_add_camera(h_, e->distortionModel(), e->intrinsicsArray(), e->distortionArray(), e->numDistortionCoefficients(),
                                         (int)e->cameraGeometry()->imageWidth(), (int)e->cameraGeometry()->imageHeight(), sig);
      if (cam < 0) throw std::runtime_error(std::string("svin_ba_add_camera: ") + svin_ba_last_error());
      it = cams_.emplace(key, cam).first;
    }
    const double uv[2] = {e->measurement()[0], e->measurement()[1]};
    return record(svin_ba_map_add_reprojection_error(h_, pose->id(), point->id(), extrinsics->id(), (uint64_t)it->second, uv, e->informationRowMajor()),
                  e, {pose, point, extrinsics});
  }
  /// Map::addResidualBlock (Map.cpp:341-376) for ANY OTHER error term that implements ErrorInterface (the reference hands any
  /// ::ceres::CostFunction to Ceres): the object is evaluated by the HOST before every evaluation launch of the solve
  /// (svin_ba_map_add_host_residual: EvaluateWithMinimalJacobians at the current / candidate blocks, residual and minimal Jacobians to
  /// the device).  Pose / extrinsics and speed / bias blocks only, no loss, residual dimension <= 15, at most four blocks; a slow
  /// path by construction.  Returns NULL when the backend refuses the residual (Map.cpp:349-351).
  template <class ERROR_T>
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<ERROR_T> e, ::ceres::LossFunction* loss,
                                            std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> > blocks) {
    need(); noLoss(loss);
    if (blocks.empty() || blocks.size() > 4 || blocks.size() != e->parameterBlocks()) return nullptr;
    std::unique_ptr<HostTerm> term(new HostTerm);
    uint64_t ids[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < blocks.size(); ++i) {
      ids[i] = blocks[i]->id();
      term->dims.push_back(blocks[i]->dimension());
    }
    term->m = e->residualDim();
    term->eval = [e](double const* const* p, double* r, double** J, double** Jm) { return e->EvaluateWithMinimalJacobians(p, r, J, Jm); };
    const uint64_t rid = svin_ba_map_add_host_residual(h_, ids, (int)blocks.size(), (int)term->m, &Map::hostTrampoline, term.get());
    if (rid == 0) return nullptr;
    hostTerms_[rid] = std::move(term);
    return record(rid, e, std::move(blocks));
  }
  template <class ERROR_T>
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<ERROR_T> e, ::ceres::LossFunction* loss, std::shared_ptr<okvis::ceres::ParameterBlock> x0) {
    return addResidualBlock(e, loss, std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> >{x0});
  }
  template <class ERROR_T>
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<ERROR_T> e, ::ceres::LossFunction* loss, std::shared_ptr<okvis::ceres::ParameterBlock> x0,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> x1) {
    return addResidualBlock(e, loss, std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> >{x0, x1});
  }
  template <class ERROR_T>
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<ERROR_T> e, ::ceres::LossFunction* loss, std::shared_ptr<okvis::ceres::ParameterBlock> x0,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> x1, std::shared_ptr<okvis::ceres::ParameterBlock> x2) {
    return addResidualBlock(e, loss, std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> >{x0, x1, x2});
  }
  template <class ERROR_T>
  ::ceres::ResidualBlockId addResidualBlock(std::shared_ptr<ERROR_T> e, ::ceres::LossFunction* loss, std::shared_ptr<okvis::ceres::ParameterBlock> x0,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> x1, std::shared_ptr<okvis::ceres::ParameterBlock> x2,
                                            std::shared_ptr<okvis::ceres::ParameterBlock> x3) {
    return addResidualBlock(e, loss, std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> >{x0, x1, x2, x3});
  }
  /// Map::removeResidualBlock (Map.cpp:467-492), any residual of the graph
  bool removeResidualBlock(::ceres::ResidualBlockId residual) {
    need();
    const uint64_t rid = reinterpret_cast<uint64_t>(residual);
    built_.erase(rid);
    const bool ok = svin_ba_map_remove_residual_block(h_, rid) == 1;
    hostTerms_.erase(rid);   // (after the backend has forgotten the function pointer)
    return ok;
  }
  /// Map::isJacobianCorrect (Map.cpp:153-252): central differences (delta 1e-8) through the blocks' plus() against the
  /// analytic minimal Jacobians of the error term, max |difference| / ||numeric|| <= relTol per block.  For residuals added
  /// through addResidualBlock (the object is evaluated on the CPU by the library's host twins).
  bool isJacobianCorrect(::ceres::ResidualBlockId residual, double relTol = 1e-6) const {
    auto it = built_.find(reinterpret_cast<uint64_t>(residual));
    if (it == built_.end()) throw std::runtime_error("okvis::ceres::Map (svin_ba shim): isJacobianCorrect needs a residual added through addResidualBlock");
    const Built& r = it->second;
    const size_t nb = r.blocks.size(), m = r.m;
    std::vector<const double*> par(nb);
    std::vector<std::vector<double> > J(nb), Jmin(nb), Jnum(nb);
    std::vector<double*> Jp(nb), Jminp(nb);
    for (size_t i = 0; i < nb; ++i) {
      par[i] = r.blocks[i]->parameters();
      J[i].assign(m * r.blocks[i]->dimension(), 0.0); Jmin[i].assign(m * r.blocks[i]->minimalDimension(), 0.0);
      Jnum[i].assign(m * r.blocks[i]->minimalDimension(), 0.0);
      Jp[i] = J[i].data(); Jminp[i] = Jmin[i].data();
    }
    const double delta = 1e-8;
    std::vector<double> rp(m), rm(m), res(m);
    for (size_t i = 0; i < nb; ++i) {
      const size_t dim = r.blocks[i]->dimension(), md = r.blocks[i]->minimalDimension();
      std::vector<double> xp(dim), xm(dim), step(md);
      for (size_t j = 0; j < md; ++j) {
        std::fill(step.begin(), step.end(), 0.0);
        step[j] = delta;
        r.blocks[i]->plus(r.blocks[i]->parameters(), step.data(), xp.data());
        step[j] = -delta;
        r.blocks[i]->plus(r.blocks[i]->parameters(), step.data(), xm.data());
        par[i] = xp.data();
        r.eval(par.data(), rp.data(), nullptr, nullptr);
        par[i] = xm.data();
        r.eval(par.data(), rm.data(), nullptr, nullptr);
        par[i] = r.blocks[i]->parameters();
        for (size_t k = 0; k < m; ++k) Jnum[i][k * md + j] = (rp[k] - rm[k]) / (2.0 * delta);
      }
    }
    r.eval(par.data(), res.data(), Jp.data(), Jminp.data());
    bool ok = true;
    for (size_t i = 0; i < nb; ++i) {
      double norm = 0, maxDiff = 0;
      for (size_t k = 0; k < Jnum[i].size(); ++k) {
        norm += Jnum[i][k] * Jnum[i][k];
        maxDiff = std::max(maxDiff, std::fabs(Jnum[i][k] - Jmin[i][k]));
      }
      if (maxDiff / std::sqrt(norm) > relTol) ok = false;
    }
    return ok;
  }

  Options options;   ///< public like the reference's (Map.hpp:341)
  Summary summary;   ///< public like the reference's (Map.hpp:344)

  /// Map::solve (Map.hpp:347): ::ceres::Solve(options, problem, &summary)
  void solve() {
    need();
    // blocks added through addParameterBlock are optimised "in place" (ceres works on their memory in the reference): their
    // current parameters go to the device, the result comes back into the same objects
    for (const auto& kv : blocks_) svin_ba_set_parameter_block(h_, kv.first, kv.second->parameters());
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
    for (const auto& kv : blocks_) {
      double x[9];
      if (svin_ba_get_parameter_block(h_, kv.first, nullptr, x, nullptr, nullptr, nullptr, nullptr) > 0) kv.second->setParameters(x);
    }
  }

  bool parameterBlockExists(uint64_t id) const { need(); return svin_ba_parameter_block_exists(h_, id) == 1; }   // Map.cpp:77-80
  bool setParameterBlockConstant(uint64_t id) { need(); return svin_ba_set_parameter_block_constant(h_, id, 1) == 1; }  // :495-501
  bool setParameterBlockVariable(uint64_t id) { need(); return svin_ba_set_parameter_block_constant(h_, id, 0) == 1; }  // :504-510
  bool isParameterBlockConstant(uint64_t id) const { need(); return svin_ba_is_parameter_block_constant(h_, id) == 1; } // ParameterBlock::fixed()
  /// Map::resetParameterization (Map.cpp:513-543): Pose3d / Pose4d / Pose2d hold tangent directions of a pose or extrinsics
  /// block on the device (svin_ba_reset_parameterization); a manifold the block's type cannot take is refused.
  bool resetParameterization(uint64_t id, int parameterization) const {
    need();
    return svin_ba_reset_parameterization(h_, id, parameterization) == 1;
  }
  /// Map::parameterBlockPtr (Map.hpp:166-170): a snapshot of the block (values, id, fixed, time stamp / initialised flag)
  std::shared_ptr<okvis::ceres::ParameterBlock> parameterBlockPtr(uint64_t id) const {
    need();
    int32_t type = -1, fixed = 0, init = 1;
    uint32_t sec = 0, nsec = 0;
    double x[9];
    if (svin_ba_get_parameter_block(h_, id, &type, x, &sec, &nsec, &fixed, &init) < 0) return std::shared_ptr<okvis::ceres::ParameterBlock>();
    std::shared_ptr<okvis::ceres::ParameterBlock> out;
    if (type == 0 || type == 1) {
      auto b = std::make_shared<PoseParameterBlock>();
      b->setTimestamp(okvis::Time(sec, nsec));
      out = b;
    } else if (type == 2) {
      auto b = std::make_shared<SpeedAndBiasParameterBlock>();
      b->setTimestamp(okvis::Time(sec, nsec));
      out = b;
    } else {
      auto b = std::make_shared<HomogeneousPointParameterBlock>();
      b->setInitialized(init != 0);
      out = b;
    }
    out->setParameters(x);
    out->setId(id);
    out->setFixed(fixed != 0);
    return out;
  }
  /// Map::id2parameterBlockMap (Map.hpp:188): every block of the window, as snapshots (returned by value)
  Id2ParameterBlock_Map id2parameterBlockMap() const {
    need();
    Id2ParameterBlock_Map out;
    const int n = svin_ba_parameter_block_ids(h_, nullptr, 0);
    if (n <= 0) return out;
    std::vector<uint64_t> ids((size_t)n);
    svin_ba_parameter_block_ids(h_, ids.data(), n);
    for (uint64_t id : ids) out.emplace(id, parameterBlockPtr(id));
    return out;
  }
  /// Map::errorInterfacePtr (Map.hpp:173-180): sizes and type of the residual (see ResidualView)
  std::shared_ptr<okvis::ceres::ErrorInterface> errorInterfacePtr(::ceres::ResidualBlockId residual) const {
    return errorInterfacePtr(reinterpret_cast<uint64_t>(residual));
  }
  std::shared_ptr<okvis::ceres::ErrorInterface> errorInterfacePtr(uint64_t residualId) const {
    need();
    int32_t kind = -1;
    const std::vector<uint64_t> ids = parameterIds(residualId, &kind);
    if (ids.empty()) return std::shared_ptr<okvis::ceres::ErrorInterface>();
    std::vector<size_t> dims;
    size_t m = 0;
    for (uint64_t id : ids) {
      const int d = svin_ba_get_parameter_block(h_, id, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      dims.push_back(d > 0 ? (size_t)d : 0);
    }
    switch (kind) {
      case 100: m = 2; break;
      case 101: {   // the prior's dimension (svin_ba_get_prior reports -m when the capacity is too small)
        const int pm = svin_ba_get_prior(h_, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        m = (size_t)(pm < 0 ? -pm : pm);
        break;
      }
      case 102: m = 3; break;
      case 0: m = 15; break;
      case 1: case 3: m = 6; break;
      case 2: m = 9; break;
      default: m = 1;
    }
    return std::make_shared<ResidualView>(kind, m, dims);
  }
  /// Map::residuals (Map.cpp:576-587): every residual touching the block, in insertion order
  ResidualBlockCollection residuals(uint64_t id) const {
    need();
    ResidualBlockCollection out;
    const int n = svin_ba_residuals_of(h_, id, nullptr, 0);
    if (n <= 0) return out;
    std::vector<uint64_t> rids((size_t)n);
    svin_ba_residuals_of(h_, id, rids.data(), n);
    // sizes and types of all of them in ONE call (a pose block of a window is touched by a thousand reprojection residuals)
    std::vector<int32_t> kind((size_t)n), m((size_t)n), nb((size_t)n), dims((size_t)4 * n);
    svin_ba_residual_info(h_, n, rids.data(), kind.data(), m.data(), nb.data(), dims.data());
    for (int i = 0; i < n; ++i) {
      std::shared_ptr<ErrorInterface> e;
      if (kind[i] == 101) e = errorInterfacePtr(rids[i]);   // the prior: its block list comes from svin_ba_parameters_of
      else if (kind[i] >= 0) {
        std::vector<size_t> d;
        for (int b = 0; b < nb[i] && b < 4; ++b) d.push_back((size_t)dims[4 * i + b]);
        e = std::make_shared<ResidualView>(kind[i], (size_t)m[i], d);
      }
      out.push_back(ResidualBlockSpec(reinterpret_cast< ::ceres::ResidualBlockId>(rids[i]), nullptr, e));
    }
    return out;
  }
  /// Map::parameters (Map.cpp:602-620): the blocks of a residual in the cost function's parameter order
  ParameterBlockCollection parameters(::ceres::ResidualBlockId residual) const { return parameters(reinterpret_cast<uint64_t>(residual)); }
  ParameterBlockCollection parameters(uint64_t residualId, int* kind = nullptr) const {
    need();
    ParameterBlockCollection out;
    int32_t k = -1;
    for (uint64_t id : parameterIds(residualId, &k)) out.push_back(ParameterBlockSpec(id, parameterBlockPtr(id)));
    if (kind) *kind = k;
    return out;
  }
  /// the ids only (no snapshots): two calls, the prior of a wide window touches hundreds of blocks
  std::vector<uint64_t> parameterIds(uint64_t residualId, int32_t* kind = nullptr) const {
    need();
    std::vector<uint64_t> ids;
    const int n = svin_ba_parameters_of(h_, residualId, nullptr, 0, kind);
    if (n <= 0) return ids;
    ids.resize((size_t)n);
    svin_ba_parameters_of(h_, residualId, ids.data(), n, kind);
    return ids;
  }

 private:
  void need() const {
    if (!h_) throw std::runtime_error("okvis::ceres::Map (svin_ba shim): not attached to an estimator handle");
  }
  void ensureHandle() {
    if (h_) return;
    h_ = svin_ba_create(0);
    if (!h_) throw std::runtime_error(std::string("svin_ba_create: ") + svin_ba_last_error());
    owned_ = true;
  }
  static void noLoss(const ::ceres::LossFunction* loss) {
    if (loss) throw std::runtime_error("okvis::ceres::Map (svin_ba shim): only reprojection residuals take a loss function (CauchyLoss(1))");
  }
  /// a residual added through addResidualBlock: the error-term object (kept alive, evaluated by isJacobianCorrect) and its blocks
  struct Built {
    std::function<bool(double const* const*, double*, double**, double**)> eval;
    size_t m;
    std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> > blocks;
  };
  template <class ERROR_T>
  ::ceres::ResidualBlockId record(uint64_t rid, std::shared_ptr<ERROR_T> e, std::vector<std::shared_ptr<okvis::ceres::ParameterBlock> > blocks) {
    if (rid == 0) return nullptr;   // refused: unknown block, wrong block type (Map.cpp:349-351 returns NULL too)
    Built b;
    b.eval = [e](double const* const* p, double* r, double** J, double** Jm) { return e->EvaluateWithMinimalJacobians(p, r, J, Jm); };
    b.m = e->residualDim();
    b.blocks = std::move(blocks);
    built_[rid] = std::move(b);
    return reinterpret_cast< ::ceres::ResidualBlockId>(rid);
  }
  /// an error term the host evaluates for the backend (svin_cost_function's `user`)
  struct HostTerm {
    std::function<bool(double const* const*, double*, double**, double**)> eval;
    size_t m;
    std::vector<size_t> dims;   // ambient dimensions of the blocks (7 / 9)
  };
  static int hostTrampoline(void* user, const double* const* parameters, double* residuals, double** jacobiansMinimal) {
    const HostTerm* t = static_cast<const HostTerm*>(user);
    double ambient[4][15 * 9];   // the ambient Jacobians EvaluateWithMinimalJacobians also fills: not used by the backend
    double* J[4] = {ambient[0], ambient[1], ambient[2], ambient[3]};
    try {
      return t->eval(parameters, residuals, J, jacobiansMinimal) ? 1 : 0;
    } catch (...) {
      return 0;
    }
  }
  svin_ba* h_;
  bool owned_;
  std::map<uint64_t, std::unique_ptr<HostTerm> > hostTerms_;
  std::map<uint64_t, std::shared_ptr<okvis::ceres::ParameterBlock> > blocks_;   ///< blocks added through addParameterBlock
  std::unordered_map<uint64_t, Built> built_;
  std::map<const void*, int> cams_;                                             ///< camera geometry object -> backend camera index
};

}  // namespace ceres
}  // namespace okvis

#endif  // INTEGRATION_OKVIS_CERES_MAP_HPP_
