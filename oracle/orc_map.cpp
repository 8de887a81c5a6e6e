// ORACLE -- test infrastructure only (never linked into the product library).
// See orc_map.hpp for what is restated from where.
#include "orc_map.hpp"
#include <chrono>
#include <omp.h>
#include <cstdio>
#include <cassert>

namespace orc {

// ================================================================ graph bookkeeping
bool Map::addParameterBlock(uint64_t id, int type, const double* x) {  // Map.cpp:255-319
  if (params_.count(id)) return false;
  ParamBlock b;
  b.id = id; b.type = type; b.fixed = false;
  std::memcpy(b.x, x, sizeof(double) * blockDim(type));
  params_[id] = b;
  param2res_[id];
  return true;
}
bool Map::removeParameterBlock(uint64_t id) {  // Map.cpp:322-333
  if (!params_.count(id)) return false;
  const std::vector<uint64_t> res = residuals(id);
  for (uint64_t r : res) removeResidualBlock(r);
  params_.erase(id);
  param2res_.erase(id);
  return true;
}
uint64_t Map::addResidualBlock(std::shared_ptr<ErrorTerm> err, int loss, const std::vector<uint64_t>& paramIds) {
  for (uint64_t p : paramIds)
    if (!params_.count(p)) return 0;
  ResidualBlock rb;
  rb.id = nextResId_++;
  rb.err = err; rb.loss = loss; rb.params = paramIds;
  residuals_[rb.id] = rb;
  for (uint64_t p : paramIds) param2res_[p].push_back(rb.id);
  return rb.id;
}
bool Map::removeResidualBlock(uint64_t resId) {  // Map.cpp:467-492
  auto it = residuals_.find(resId);
  if (it == residuals_.end()) return false;
  for (uint64_t p : it->second.params) {
    auto& v = param2res_[p];
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i] == resId) { v.erase(v.begin() + i); break; }
  }
  residuals_.erase(it);
  return true;
}
bool Map::resetParameterization(uint64_t id, int manifold) {
  auto it = params_.find(id);
  if (it == params_.end()) return false;
  if (it->second.type != BLOCK_POSE || !(manifold == 6 || manifold == 3 || manifold == 4 || manifold == 2)) return false;
  it->second.manifold = manifold;
  return true;
}
std::vector<uint64_t> Map::residuals(uint64_t paramId) const {
  auto it = param2res_.find(paramId);
  if (it == param2res_.end()) return {};
  return it->second;
}

// ================================================================ getLhs (Map.cpp:105-150)
void Map::getLhs(uint64_t paramId, double* H) const {
  const ParamBlock& pb = params_.at(paramId);
  const int md = pb.mdim();
  std::memset(H, 0, sizeof(double) * md * md);
  for (uint64_t rid : residuals(paramId)) {
    const ResidualBlock& rb = residuals_.at(rid);
    const int m = rb.err->residualDim();
    const int nb = (int)rb.params.size();
    std::vector<const double*> P(nb);
    std::vector<std::vector<double>> Ja(nb), Jm(nb);
    std::vector<double*> Jap(nb), Jmp(nb);
    int Jsel = -1;
    for (int j = 0; j < nb; ++j) {
      const ParamBlock& b = params_.at(rb.params[j]);
      if (b.id == paramId) Jsel = j;
      P[j] = b.x;
      Ja[j].assign(m * b.dim(), 0.0);
      Jm[j].assign(m * b.mdim(), 0.0);
      Jap[j] = Ja[j].data(); Jmp[j] = Jm[j].data();
    }
    std::vector<double> r(m);
    rb.err->evaluate(P.data(), r.data(), Jap.data(), Jmp.data());
    const double* Jx = Jm[Jsel].data();
    for (int a = 0; a < md; ++a)
      for (int b = 0; b < md; ++b) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += Jx[k * md + a] * Jx[k * md + b];
        H[a * md + b] += s;
      }
  }
}

// ================================================================ isJacobianCorrect (Map.cpp:153-252)
bool Map::isJacobianCorrect(uint64_t resId, double relTol, double* worstRel) const {
  const ResidualBlock& rb = residuals_.at(resId);
  const int m = rb.err->residualDim();
  const int nb = (int)rb.params.size();
  std::vector<const double*> P(nb);
  std::vector<std::vector<double>> Ja(nb), Jm(nb), Jnum(nb);
  std::vector<double*> Jap(nb), Jmp(nb);
  for (int i = 0; i < nb; ++i) {
    const ParamBlock& b = params_.at(rb.params[i]);
    P[i] = b.x;
    Ja[i].assign(m * b.dim(), 0.0);
    Jm[i].assign(m * b.mdim(), 0.0);
    Jnum[i].assign(m * b.mdim(), 0.0);
    Jap[i] = Ja[i].data(); Jmp[i] = Jm[i].data();
  }
  const double delta = 1e-8;
  for (int i = 0; i < nb; ++i) {
    const ParamBlock& b = params_.at(rb.params[i]);
    const int md = b.mdim();
    for (int j = 0; j < md; ++j) {
      std::vector<double> rp(m), rm(m), plus(md, 0.0);
      double xp[9], xm[9];
      plus[j] = delta;
      manifoldPlus(b.type, b.x, plus.data(), xp);
      P[i] = xp;
      rb.err->evaluate(P.data(), rp.data(), nullptr, nullptr);
      plus[j] = -delta;
      manifoldPlus(b.type, b.x, plus.data(), xm);
      P[i] = xm;
      rb.err->evaluate(P.data(), rm.data(), nullptr, nullptr);
      P[i] = b.x;
      for (int k = 0; k < m; ++k) Jnum[i][k * md + j] = (rp[k] - rm[k]) * 1.0 / (2.0 * delta);
    }
  }
  std::vector<double> r(m);
  rb.err->evaluate(P.data(), r.data(), Jap.data(), Jmp.data());
  bool ok = true;
  double worst = 0;
  for (int i = 0; i < nb; ++i) {
    double norm = 0, maxDiff = -1e300;
    for (size_t k = 0; k < Jnum[i].size(); ++k) {
      norm += Jnum[i][k] * Jnum[i][k];
      const double dd = Jnum[i][k] - Jm[i][k];
      maxDiff = std::max(maxDiff, std::max(dd, -dd));
    }
    norm = std::sqrt(norm);
    const double rel = maxDiff / norm;
    if (rel > worst) worst = rel;
    if (rel > relTol) ok = false;
  }
  if (worstRel) *worstRel = worst;
  return ok;
}

// ================================================================ Ceres-like solve
namespace {

struct ResInfo {
  const ResidualBlock* rb;
  int m, nb;
  size_t roff;               // offset into r
  std::vector<size_t> joff;  // per block offset into Jloc
  std::vector<int> kind;     // -1 fixed, 0 camera-side, 1 landmark
  std::vector<int> idx;      // cam index or landmark index
  std::vector<int> mdim;
  std::vector<ParamBlock*> pb;
};

struct PhaseTimer {   // ORC_TIMING=1: where a solve spends its time (diagnostics for the CPU baseline)
  double* acc;
  std::chrono::steady_clock::time_point t0;
  explicit PhaseTimer(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
  ~PhaseTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Problem {
  Map& map;
  mutable double tEvalJ = 0, tEval = 0, tSchur = 0, tChol = 0, tJv = 0, tGrad = 0;
  int numThreads = 1;
  std::vector<ParamBlock*> cam;
  std::vector<int> camOff;
  int d = 0;
  std::vector<ParamBlock*> lm;
  std::vector<ResInfo> res;
  std::vector<std::vector<int>> lmRes;
  std::vector<int> camOnlyRes;
  std::vector<double> r, Jloc;
  double cost = 0;
  int nvar() const { return d + 3 * (int)lm.size(); }

  explicit Problem(Map& m) : map(m) {
    std::unordered_map<uint64_t, int> camIndex, lmIndex;
    // variable blocks referenced by at least one residual, in parameter-id order
    for (auto& kv : const_cast<std::map<uint64_t, ParamBlock>&>(map.params())) {
      ParamBlock& b = kv.second;
      if (b.fixed) continue;
      if (map.residuals(b.id).empty()) continue;
      if (b.type == BLOCK_HPOINT) {
        lmIndex[b.id] = (int)lm.size();
        lm.push_back(&b);
      } else {
        camIndex[b.id] = (int)cam.size();
        cam.push_back(&b);
        camOff.push_back(d);
        d += b.sdim();
      }
    }
    lmRes.resize(lm.size());
    size_t roff = 0, joff = 0;
    for (auto& kv : map.residualMap()) {
      ResInfo ri;
      ri.rb = &kv.second;
      ri.m = ri.rb->err->residualDim();
      ri.nb = (int)ri.rb->params.size();
      ri.roff = roff;
      roff += ri.m;
      int lmIdx = -1;
      for (int j = 0; j < ri.nb; ++j) {
        ParamBlock& b = map.param(ri.rb->params[j]);
        ri.pb.push_back(&b);
        ri.mdim.push_back(b.sdim());
        ri.joff.push_back(joff);
        joff += (size_t)ri.m * b.sdim();
        if (b.fixed) { ri.kind.push_back(-1); ri.idx.push_back(-1); }
        else if (b.type == BLOCK_HPOINT) { ri.kind.push_back(1); ri.idx.push_back(lmIndex.at(b.id)); lmIdx = lmIndex.at(b.id); }
        else { ri.kind.push_back(0); ri.idx.push_back(camIndex.at(b.id)); }
      }
      if (lmIdx >= 0) lmRes[lmIdx].push_back((int)res.size());
      else camOnlyRes.push_back((int)res.size());
      res.push_back(std::move(ri));
    }
    r.assign(roff, 0.0);
    Jloc.assign(joff, 0.0);
  }

  // Evaluate all residual blocks at the current parameter values.  With jac: local
  // (tangent-space) Jacobians = ambient Jacobian * PlusJacobian(x), then Ceres' Corrector.
  struct EvalScratch {
    std::vector<double> scratch;
    std::vector<const double*> P;
    std::vector<double*> Jp;
  };
  // one residual block: residual (+ local Jacobians and Ceres' Corrector with jac); returns its cost term
  double evalOne(ResInfo& ri, bool jac, EvalScratch& es) {
    std::vector<double>& scratch = es.scratch;
    std::vector<const double*>& P = es.P;
    std::vector<double*>& Jp = es.Jp;
    double cost = 0;
    {
      const int m = ri.m;
      P.resize(ri.nb);
      for (int j = 0; j < ri.nb; ++j) P[j] = ri.pb[j]->x;
      double* rr = &r[ri.roff];
      if (jac) {
        size_t tot = 0;
        for (int j = 0; j < ri.nb; ++j) tot += (size_t)m * ri.pb[j]->dim();
        scratch.assign(tot, 0.0);
        Jp.resize(ri.nb);
        size_t o = 0;
        for (int j = 0; j < ri.nb; ++j) { Jp[j] = &scratch[o]; o += (size_t)m * ri.pb[j]->dim(); }
        ri.rb->err->evaluate(P.data(), rr, Jp.data(), nullptr);
        for (int j = 0; j < ri.nb; ++j) {
          if (ri.kind[j] < 0) continue;
          const ParamBlock& b = *ri.pb[j];
          double* out = &Jloc[ri.joff[j]];
          const int dm = b.dim(), md = b.sdim();
          if (b.type == BLOCK_POSE) {
            // (reduced manifolds: PlusJacobian is the 6-DoF one with the held columns left out, PoseManifold.cpp:223-236 / :325-334 / :420-433)
            double Jplus[42];
            manifoldPlusJacobian(BLOCK_POSE, b.x, Jplus);
            for (int a = 0; a < m; ++a)
              for (int c = 0; c < md; ++c) {
                const int c6 = poseTangentIndex(b.manifold, c);
                double s = 0;
                for (int k = 0; k < 7; ++k) s += Jp[j][a * 7 + k] * Jplus[k * 6 + c6];
                out[a * md + c] = s;
              }
          } else if (b.type == BLOCK_HPOINT) {
            for (int a = 0; a < m; ++a)
              for (int c = 0; c < 3; ++c) out[a * 3 + c] = Jp[j][a * dm + c];
          } else {
            std::memcpy(out, Jp[j], sizeof(double) * m * md);
          }
        }
      } else {
        ri.rb->err->evaluate(P.data(), rr, nullptr, nullptr);
      }
      double sq = 0;
      for (int a = 0; a < m; ++a) sq += rr[a] * rr[a];
      if (ri.rb->loss == LOSS_NONE) {
        cost = 0.5 * sq;
      } else {
        double rho[3];
        lossEvaluate(ri.rb->loss, ri.rb->lossParam, sq, rho);
        cost = 0.5 * rho[0];
        if (jac) {
          // ceres/internal/ceres/corrector.cc
          const double sqrt_rho1 = std::sqrt(rho[1]);
          double residual_scaling, alpha_sq_norm;
          if (sq == 0.0 || rho[2] <= 0.0) {
            residual_scaling = sqrt_rho1;
            alpha_sq_norm = 0.0;
          } else {
            const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
            const double alpha = 1.0 - std::sqrt(D);
            residual_scaling = sqrt_rho1 / (1 - alpha);
            alpha_sq_norm = alpha / sq;
          }
          for (int j = 0; j < ri.nb; ++j) {
            if (ri.kind[j] < 0) continue;
            double* Jb = &Jloc[ri.joff[j]];
            const int md = ri.mdim[j];
            if (alpha_sq_norm == 0.0) {
              for (int k = 0; k < m * md; ++k) Jb[k] *= sqrt_rho1;
            } else {
              for (int c = 0; c < md; ++c) {
                double rtJ = 0;
                for (int a = 0; a < m; ++a) rtJ += rr[a] * Jb[a * md + c];
                for (int a = 0; a < m; ++a) Jb[a * md + c] = sqrt_rho1 * (Jb[a * md + c] - alpha_sq_norm * rr[a] * rtJ);
              }
            }
          }
          for (int a = 0; a < m; ++a) rr[a] *= residual_scaling;
        }
      }
    }
    return cost;
  }
  double evaluate(bool jac) {
    PhaseTimer pt(jac ? &tEvalJ : &tEval);
    if (numThreads > 1) return evaluateParallel(jac);
    double total = 0;
    EvalScratch es;
    for (ResInfo& ri : res) total += evalOne(ri, jac, es);
    return total;
  }
  // the same evaluation on numThreads threads (Ceres: options.num_threads parallelises the residual / Jacobian
  // evaluation).  The per-block cost terms are summed in block order afterwards, so the result does not depend on the
  // thread count.
  double evaluateParallel(bool jac) {
    const int n = (int)res.size();
    std::vector<double> costs(n);
    const int nt = std::max(1, std::min(numThreads, n / 512));   // a thread is worth waking for >= 512 residual blocks
#pragma omp parallel num_threads(nt)
    {
      EvalScratch es;
#pragma omp for schedule(static)
      for (int i = 0; i < n; ++i) costs[i] = evalOne(res[i], jac, es);
    }
    double total = 0;
    for (int i = 0; i < n; ++i) total += costs[i];
    return total;
  }

  // full gradient g = J^T r and squared column norms h of the local Jacobian. layout: [cam(d), lm(3L)]
  void gradientAndColumnNorms(std::vector<double>& g, std::vector<double>& h) const {
    PhaseTimer pt(&tGrad);
    g.assign(nvar(), 0.0);
    h.assign(nvar(), 0.0);
    for (const ResInfo& ri : res) {
      const double* rr = &r[ri.roff];
      for (int j = 0; j < ri.nb; ++j) {
        if (ri.kind[j] < 0) continue;
        const int md = ri.mdim[j];
        const int off = ri.kind[j] == 0 ? camOff[ri.idx[j]] : d + 3 * ri.idx[j];
        const double* Jb = &Jloc[ri.joff[j]];
        for (int a = 0; a < ri.m; ++a)
          for (int c = 0; c < md; ++c) {
            g[off + c] += Jb[a * md + c] * rr[a];
            h[off + c] += Jb[a * md + c] * Jb[a * md + c];
          }
      }
    }
  }
  // out = J v  (per-residual), returns sum |Jv|^2 and (Jv).r
  void Jtimes(const std::vector<double>& v, double& sqnorm, double& dot_r) const {
    PhaseTimer pt(&tJv);
    sqnorm = 0; dot_r = 0;
    std::vector<double> jv;
    for (const ResInfo& ri : res) {
      jv.assign(ri.m, 0.0);
      for (int j = 0; j < ri.nb; ++j) {
        if (ri.kind[j] < 0) continue;
        const int md = ri.mdim[j];
        const int off = ri.kind[j] == 0 ? camOff[ri.idx[j]] : d + 3 * ri.idx[j];
        const double* Jb = &Jloc[ri.joff[j]];
        for (int a = 0; a < ri.m; ++a)
          for (int c = 0; c < md; ++c) jv[a] += Jb[a * md + c] * v[off + c];
      }
      const double* rr = &r[ri.roff];
      for (int a = 0; a < ri.m; ++a) { sqnorm += jv[a] * jv[a]; dot_r += jv[a] * rr[a]; }
    }
  }

  // Schur-reduced normal equations with diagonal damping `damp` (size nvar, may be empty = 0):
  //   (H + diag(damp)) y = g,  landmarks eliminated first.  Returns false if a block is not PD.
  struct Schur {
    std::vector<double> S, gred;                 // d x d, d
    std::vector<double> A, b;                    // camera part before reduction
    std::vector<double> Vinv, bl, V;             // per landmark
    std::vector<int> wStart;                     // per landmark start into wCam/W
    std::vector<int> wCam;                       // cam index
    std::vector<double> W;                       // 18 or 27 (mdim x 3) per entry, row-major mdim x 3, padded to 27
  };
  bool buildSchur(const std::vector<double>& damp, Schur& s) const {
    PhaseTimer pt(&tSchur);
    if (numThreads > 1) return buildSchurParallel(damp, s);
    const int L = (int)lm.size();
    s.A.assign((size_t)d * d, 0.0);
    s.b.assign(d, 0.0);
    s.V.assign((size_t)L * 9, 0.0);
    s.Vinv.assign((size_t)L * 9, 0.0);
    s.bl.assign((size_t)L * 3, 0.0);
    s.wStart.assign(L + 1, 0);
    s.wCam.clear();
    s.W.clear();
    auto accumulateCam = [&](const ResInfo& ri) {
      const double* rr = &r[ri.roff];
      for (int j = 0; j < ri.nb; ++j) {
        if (ri.kind[j] != 0) continue;
        const int mj = ri.mdim[j], oj = camOff[ri.idx[j]];
        const double* Jj = &Jloc[ri.joff[j]];
        for (int a = 0; a < mj; ++a) {
          double sb = 0;
          for (int k = 0; k < ri.m; ++k) sb += Jj[k * mj + a] * rr[k];
          s.b[oj + a] += sb;
        }
        for (int i = 0; i < ri.nb; ++i) {
          if (ri.kind[i] != 0) continue;
          const int mi = ri.mdim[i], oi = camOff[ri.idx[i]];
          const double* Ji = &Jloc[ri.joff[i]];
          for (int a = 0; a < mj; ++a)
            for (int c = 0; c < mi; ++c) {
              double sa = 0;
              for (int k = 0; k < ri.m; ++k) sa += Jj[k * mj + a] * Ji[k * mi + c];
              s.A[(size_t)(oj + a) * d + oi + c] += sa;
            }
        }
      }
    };
    for (int ridx : camOnlyRes) accumulateCam(res[ridx]);
    if (!damp.empty())
      for (int i = 0; i < d; ++i) s.A[(size_t)i * d + i] += damp[i];
    s.S = s.A;
    s.gred = s.b;
    // the un-damped A/b are reported by linearize(); add landmark-residual camera parts next
    for (int l = 0; l < L; ++l) {
      double V[9] = {0}, bl[3] = {0};
      const int w0 = (int)s.wCam.size();
      s.wStart[l] = w0;
      for (int ridx : lmRes[l]) {
        const ResInfo& ri = res[ridx];
        const double* rr = &r[ri.roff];
        int jl = -1;
        for (int j = 0; j < ri.nb; ++j) if (ri.kind[j] == 1) jl = j;
        const double* Jl = &Jloc[ri.joff[jl]];
        for (int a = 0; a < 3; ++a) {
          for (int c = 0; c < 3; ++c) {
            double sv = 0;
            for (int k = 0; k < ri.m; ++k) sv += Jl[k * 3 + a] * Jl[k * 3 + c];
            V[a * 3 + c] += sv;
          }
          double sb = 0;
          for (int k = 0; k < ri.m; ++k) sb += Jl[k * 3 + a] * rr[k];
          bl[a] += sb;
        }
        // camera-side parts of this residual go to S/g directly (they are part of A/b)
        {
          // A/b contribution
          for (int j = 0; j < ri.nb; ++j) {
            if (ri.kind[j] != 0) continue;
            const int mj = ri.mdim[j], oj = camOff[ri.idx[j]];
            const double* Jj = &Jloc[ri.joff[j]];
            for (int a = 0; a < mj; ++a) {
              double sb = 0;
              for (int k = 0; k < ri.m; ++k) sb += Jj[k * mj + a] * rr[k];
              s.gred[oj + a] += sb;
              s.b[oj + a] += sb;
            }
            for (int i = 0; i < ri.nb; ++i) {
              if (ri.kind[i] != 0) continue;
              const int mi = ri.mdim[i], oi = camOff[ri.idx[i]];
              const double* Ji = &Jloc[ri.joff[i]];
              for (int a = 0; a < mj; ++a)
                for (int c = 0; c < mi; ++c) {
                  double sa = 0;
                  for (int k = 0; k < ri.m; ++k) sa += Jj[k * mj + a] * Ji[k * mi + c];
                  s.S[(size_t)(oj + a) * d + oi + c] += sa;
                  s.A[(size_t)(oj + a) * d + oi + c] += sa;
                }
            }
            // W_c += Jc^T Jl
            int wi = -1;
            for (int w = w0; w < (int)s.wCam.size(); ++w) if (s.wCam[w] == ri.idx[j]) { wi = w; break; }
            if (wi < 0) { wi = (int)s.wCam.size(); s.wCam.push_back(ri.idx[j]); s.W.resize(s.W.size() + 27, 0.0); }
            double* W = &s.W[(size_t)wi * 27];
            for (int a = 0; a < mj; ++a)
              for (int c = 0; c < 3; ++c) {
                double sw = 0;
                for (int k = 0; k < ri.m; ++k) sw += Jj[k * mj + a] * Jl[k * 3 + c];
                W[a * 3 + c] += sw;
              }
          }
        }
      }
      std::memcpy(&s.V[(size_t)l * 9], V, sizeof(V));
      if (!damp.empty()) { V[0] += damp[d + 3 * l]; V[4] += damp[d + 3 * l + 1]; V[8] += damp[d + 3 * l + 2]; }
      // 3x3 inverse by Cholesky (Ceres InvertPSDMatrix)
      double Lc[9];
      std::memcpy(Lc, V, sizeof(V));
      if (llt_inplace(Lc, 3) >= 0) return false;
      double Linv[9] = {0};
      Linv[0] = 1 / Lc[0]; Linv[4] = 1 / Lc[4]; Linv[8] = 1 / Lc[8];
      Linv[3] = -Lc[3] * Linv[0] / Lc[4];
      Linv[7] = -Lc[7] * Linv[4] / Lc[8];
      Linv[6] = -(Lc[6] * Linv[0] + Lc[7] * Linv[3]) / Lc[8];
      double Vi[9];
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
          double sv = 0;
          for (int k = 0; k < 3; ++k) sv += Linv[k * 3 + a] * Linv[k * 3 + c];
          Vi[a * 3 + c] = sv;
        }
      std::memcpy(&s.Vinv[(size_t)l * 9], Vi, sizeof(Vi));
      std::memcpy(&s.bl[(size_t)l * 3], bl, sizeof(bl));
      // S -= W Vinv W^T ; g -= W Vinv bl
      const int w1 = (int)s.wCam.size();
      double Vibl[3];
      mat3_vec(Vi, bl, Vibl);
      for (int wa = w0; wa < w1; ++wa) {
        const int ca = s.wCam[wa], ma = cam[ca]->sdim(), oa = camOff[ca];
        const double* Wa = &s.W[(size_t)wa * 27];
        double WVi[27];
        for (int a = 0; a < ma; ++a)
          for (int c = 0; c < 3; ++c) WVi[a * 3 + c] = Wa[a * 3] * Vi[c] + Wa[a * 3 + 1] * Vi[3 + c] + Wa[a * 3 + 2] * Vi[6 + c];
        for (int a = 0; a < ma; ++a) s.gred[oa + a] -= Wa[a * 3] * Vibl[0] + Wa[a * 3 + 1] * Vibl[1] + Wa[a * 3 + 2] * Vibl[2];
        for (int wb = w0; wb < w1; ++wb) {
          const int cb = s.wCam[wb], mb = cam[cb]->sdim(), ob = camOff[cb];
          const double* Wb = &s.W[(size_t)wb * 27];
          for (int a = 0; a < ma; ++a)
            for (int c = 0; c < mb; ++c)
              s.S[(size_t)(oa + a) * d + ob + c] -= WVi[a * 3] * Wb[c * 3] + WVi[a * 3 + 1] * Wb[c * 3 + 1] + WVi[a * 3 + 2] * Wb[c * 3 + 2];
        }
      }
    }
    s.wStart[L] = (int)s.wCam.size();
    return true;
  }
  // buildSchur on numThreads threads (Ceres' SchurEliminator is threaded by chunks of landmarks as well): every thread
  // eliminates a contiguous range of landmarks into private copies of the camera matrices, which are summed in thread
  // order afterwards.  Same arithmetic per landmark as buildSchur; the summation order across landmarks differs, so the
  // result matches the sequential one to rounding only -- used for the multi-thread CPU baseline, never by the parity tests.
  bool buildSchurParallel(const std::vector<double>& damp, Schur& s) const {
    const int L = (int)lm.size();
    s.A.assign((size_t)d * d, 0.0);
    s.b.assign(d, 0.0);
    s.V.assign((size_t)L * 9, 0.0);
    s.Vinv.assign((size_t)L * 9, 0.0);
    s.bl.assign((size_t)L * 3, 0.0);
    s.wStart.assign(L + 1, 0);
    s.wCam.clear();
    s.W.clear();
    auto camPart = [&](const ResInfo& ri, double* A, double* b) {   // J_c^T J_c and J_c^T r of one residual block
      const double* rr = &r[ri.roff];
      for (int j = 0; j < ri.nb; ++j) {
        if (ri.kind[j] != 0) continue;
        const int mj = ri.mdim[j], oj = camOff[ri.idx[j]];
        const double* Jj = &Jloc[ri.joff[j]];
        for (int a = 0; a < mj; ++a) {
          double sb = 0;
          for (int k = 0; k < ri.m; ++k) sb += Jj[k * mj + a] * rr[k];
          b[oj + a] += sb;
        }
        for (int i = 0; i < ri.nb; ++i) {
          if (ri.kind[i] != 0) continue;
          const int mi = ri.mdim[i], oi = camOff[ri.idx[i]];
          const double* Ji = &Jloc[ri.joff[i]];
          for (int a = 0; a < mj; ++a)
            for (int c = 0; c < mi; ++c) {
              double sa = 0;
              for (int k = 0; k < ri.m; ++k) sa += Jj[k * mj + a] * Ji[k * mi + c];
              A[(size_t)(oj + a) * d + oi + c] += sa;
            }
        }
      }
    };
    for (int ridx : camOnlyRes) camPart(res[ridx], s.A.data(), s.b.data());
    if (!damp.empty())
      for (int i = 0; i < d; ++i) s.A[(size_t)i * d + i] += damp[i];
    s.S = s.A;
    s.gred = s.b;
    const int nt = std::max(1, std::min(numThreads, L / 128));    // >= 128 landmarks per private copy of the camera system
    std::vector<std::vector<double>> tA(nt), tB(nt), tS(nt), tG(nt);
    std::vector<std::vector<int>> lmCam(L);
    std::vector<std::vector<double>> lmW(L);
    int failed = 0;
#pragma omp parallel num_threads(nt)
    {
      const int t = omp_get_thread_num();
      std::vector<double>& dA = tA[t]; std::vector<double>& dB = tB[t];
      std::vector<double>& dS = tS[t]; std::vector<double>& dG = tG[t];
      dA.assign((size_t)d * d, 0.0); dB.assign(d, 0.0); dS.assign((size_t)d * d, 0.0); dG.assign(d, 0.0);
#pragma omp for schedule(static)
      for (int l = 0; l < L; ++l) {
        double V[9] = {0}, bl[3] = {0};
        std::vector<int>& wc = lmCam[l];
        std::vector<double>& W = lmW[l];
        for (int ridx : lmRes[l]) {
          const ResInfo& ri = res[ridx];
          const double* rr = &r[ri.roff];
          int jl = -1;
          for (int j = 0; j < ri.nb; ++j) if (ri.kind[j] == 1) jl = j;
          const double* Jl = &Jloc[ri.joff[jl]];
          for (int a = 0; a < 3; ++a) {
            for (int c = 0; c < 3; ++c) {
              double sv = 0;
              for (int k = 0; k < ri.m; ++k) sv += Jl[k * 3 + a] * Jl[k * 3 + c];
              V[a * 3 + c] += sv;
            }
            double sb = 0;
            for (int k = 0; k < ri.m; ++k) sb += Jl[k * 3 + a] * rr[k];
            bl[a] += sb;
          }
          camPart(ri, dA.data(), dB.data());
          for (int j = 0; j < ri.nb; ++j) {
            if (ri.kind[j] != 0) continue;
            const int mj = ri.mdim[j];
            const double* Jj = &Jloc[ri.joff[j]];
            int wi = -1;
            for (int w = 0; w < (int)wc.size(); ++w) if (wc[w] == ri.idx[j]) { wi = w; break; }
            if (wi < 0) { wi = (int)wc.size(); wc.push_back(ri.idx[j]); W.resize(W.size() + 27, 0.0); }
            double* Ww = &W[(size_t)wi * 27];
            for (int a = 0; a < mj; ++a)
              for (int c = 0; c < 3; ++c) {
                double sw = 0;
                for (int k = 0; k < ri.m; ++k) sw += Jj[k * mj + a] * Jl[k * 3 + c];
                Ww[a * 3 + c] += sw;
              }
          }
        }
        std::memcpy(&s.V[(size_t)l * 9], V, sizeof(V));
        if (!damp.empty()) { V[0] += damp[d + 3 * l]; V[4] += damp[d + 3 * l + 1]; V[8] += damp[d + 3 * l + 2]; }
        double Lc[9];
        std::memcpy(Lc, V, sizeof(V));
        if (llt_inplace(Lc, 3) >= 0) {
#pragma omp atomic write
          failed = 1;
          continue;
        }
        double Linv[9] = {0};
        Linv[0] = 1 / Lc[0]; Linv[4] = 1 / Lc[4]; Linv[8] = 1 / Lc[8];
        Linv[3] = -Lc[3] * Linv[0] / Lc[4];
        Linv[7] = -Lc[7] * Linv[4] / Lc[8];
        Linv[6] = -(Lc[6] * Linv[0] + Lc[7] * Linv[3]) / Lc[8];
        double Vi[9];
        for (int a = 0; a < 3; ++a)
          for (int c = 0; c < 3; ++c) {
            double sv = 0;
            for (int k = 0; k < 3; ++k) sv += Linv[k * 3 + a] * Linv[k * 3 + c];
            Vi[a * 3 + c] = sv;
          }
        std::memcpy(&s.Vinv[(size_t)l * 9], Vi, sizeof(Vi));
        std::memcpy(&s.bl[(size_t)l * 3], bl, sizeof(bl));
        double Vibl[3];
        mat3_vec(Vi, bl, Vibl);
        const int nw = (int)wc.size();
        for (int wa = 0; wa < nw; ++wa) {
          const int ca = wc[wa], ma = cam[ca]->sdim(), oa = camOff[ca];
          const double* Wa = &W[(size_t)wa * 27];
          double WVi[27];
          for (int a = 0; a < ma; ++a)
            for (int c = 0; c < 3; ++c) WVi[a * 3 + c] = Wa[a * 3] * Vi[c] + Wa[a * 3 + 1] * Vi[3 + c] + Wa[a * 3 + 2] * Vi[6 + c];
          for (int a = 0; a < ma; ++a) dG[oa + a] -= Wa[a * 3] * Vibl[0] + Wa[a * 3 + 1] * Vibl[1] + Wa[a * 3 + 2] * Vibl[2];
          for (int wb = 0; wb < nw; ++wb) {
            const int cb = wc[wb], mb = cam[cb]->sdim(), ob = camOff[cb];
            const double* Wb = &W[(size_t)wb * 27];
            for (int a = 0; a < ma; ++a)
              for (int c = 0; c < mb; ++c)
                dS[(size_t)(oa + a) * d + ob + c] -= WVi[a * 3] * Wb[c * 3] + WVi[a * 3 + 1] * Wb[c * 3 + 1] + WVi[a * 3 + 2] * Wb[c * 3 + 2];
          }
        }
      }
      // merge the private copies: rows split over the threads, thread order inside
#pragma omp for schedule(static)
      for (int i = 0; i < d; ++i) {
        for (int tt = 0; tt < nt; ++tt) {
          const double* a = &tA[tt][(size_t)i * d];
          const double* sp = &tS[tt][(size_t)i * d];
          for (int j = 0; j < d; ++j) { s.A[(size_t)i * d + j] += a[j]; s.S[(size_t)i * d + j] += a[j] + sp[j]; }
          s.b[i] += tB[tt][i];
          s.gred[i] += tB[tt][i] + tG[tt][i];
        }
      }
    }
    if (failed) return false;
    for (int l = 0; l < L; ++l) {
      s.wStart[l] = (int)s.wCam.size();
      s.wCam.insert(s.wCam.end(), lmCam[l].begin(), lmCam[l].end());
      s.W.insert(s.W.end(), lmW[l].begin(), lmW[l].end());
    }
    s.wStart[L] = (int)s.wCam.size();
    return true;
  }
  // y = (H + damp)^-1 g  via the Schur complement; false on Cholesky failure
  bool solveNormal(const std::vector<double>& damp, std::vector<double>& y, Schur* keep = nullptr) const {
    Schur local;
    Schur& s = keep ? *keep : local;
    if (!buildSchur(damp, s)) return false;
    std::vector<double> Lm(s.S);
    {
      PhaseTimer pt(&tChol);
      if (d > 0 && llt_inplace(Lm.data(), d) >= 0) return false;
    }
    y.assign(nvar(), 0.0);
    // forward / backward
    for (int i = 0; i < d; ++i) {
      double v = s.gred[i];
      for (int k = 0; k < i; ++k) v -= Lm[(size_t)i * d + k] * y[k];
      y[i] = v / Lm[(size_t)i * d + i];
    }
    for (int i = d - 1; i >= 0; --i) {
      double v = y[i];
      for (int k = i + 1; k < d; ++k) v -= Lm[(size_t)k * d + i] * y[k];
      y[i] = v / Lm[(size_t)i * d + i];
    }
    // landmarks: y_l = Vinv (bl - W^T y_c)
    for (int l = 0; l < (int)lm.size(); ++l) {
      double t[3] = {s.bl[3 * l], s.bl[3 * l + 1], s.bl[3 * l + 2]};
      for (int w = s.wStart[l]; w < s.wStart[l + 1]; ++w) {
        const int c = s.wCam[w], mc = cam[c]->sdim(), oc = camOff[c];
        const double* W = &s.W[(size_t)w * 27];
        for (int a = 0; a < mc; ++a) { t[0] -= W[a * 3] * y[oc + a]; t[1] -= W[a * 3 + 1] * y[oc + a]; t[2] -= W[a * 3 + 2] * y[oc + a]; }
      }
      mat3_vec(&s.Vinv[(size_t)l * 9], t, &y[d + 3 * l]);
    }
    for (double v : y) if (!std::isfinite(v)) return false;
    return true;
  }

  // ambient state snapshot / restore / plus
  void snapshot(std::vector<double>& x) const {
    x.clear();
    for (auto* b : cam) x.insert(x.end(), b->x, b->x + b->dim());
    for (auto* b : lm) x.insert(x.end(), b->x, b->x + 4);
  }
  void restore(const std::vector<double>& x) {
    size_t o = 0;
    for (auto* b : cam) { std::memcpy(b->x, &x[o], sizeof(double) * b->dim()); o += b->dim(); }
    for (auto* b : lm) { std::memcpy(b->x, &x[o], sizeof(double) * 4); o += 4; }
  }
  void applyDelta(const std::vector<double>& delta) {
    for (size_t c = 0; c < cam.size(); ++c) {
      double xp[9];
      if (cam[c]->type == BLOCK_POSE && cam[c]->manifold != 6) {
        double d6[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < cam[c]->manifold; ++k) d6[poseTangentIndex(cam[c]->manifold, k)] = delta[camOff[c] + k];
        manifoldPlus(BLOCK_POSE, cam[c]->x, d6, xp);
      } else
      manifoldPlus(cam[c]->type, cam[c]->x, &delta[camOff[c]], xp);
      std::memcpy(cam[c]->x, xp, sizeof(double) * cam[c]->dim());
    }
    for (size_t l = 0; l < lm.size(); ++l) {
      double xp[4];
      manifoldPlus(BLOCK_HPOINT, lm[l]->x, &delta[d + 3 * l], xp);
      std::memcpy(lm[l]->x, xp, sizeof(xp));
    }
  }
};

double vnorm(const std::vector<double>& v) { double s = 0; for (double a : v) s += a * a; return std::sqrt(s); }

}  // namespace

void Map::linearize(Linearization& out, double mu) {
  Problem p(*this);
  out.cost = p.evaluate(true);
  std::vector<double> g, h, damp;
  p.gradientAndColumnNorms(g, h);
  if (mu > 0) { damp.resize(h.size()); for (size_t i = 0; i < h.size(); ++i) damp[i] = mu * h[i]; }
  Problem::Schur s;
  p.buildSchur(damp, s);
  out.d = p.d;
  out.camIds.clear(); out.camOffsets = p.camOff;
  for (auto* b : p.cam) out.camIds.push_back(b->id);
  out.lmIds.clear();
  for (auto* b : p.lm) out.lmIds.push_back(b->id);
  out.S = s.S; out.g = s.gred; out.A = s.A; out.b = s.b; out.V = s.V; out.bl = s.bl;
}

void Map::solve() {
  using clock = std::chrono::steady_clock;
  const auto t_start = clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(clock::now() - t_start).count(); };
  summary = SolverSummary();
  Problem p(*this);
  p.numThreads = std::max(1, options.num_threads);
  const int n = p.nvar();
  if (n == 0) { summary.termination = 0; return; }

  // --- iteration zero
  std::vector<double> x, xcand;
  p.snapshot(x);
  double x_norm = vnorm(x);
  double x_cost = p.evaluate(true);
  summary.initial_cost = x_cost;
  summary.cost_history.push_back(x_cost);
  std::vector<double> g, h;
  p.gradientAndColumnNorms(g, h);
  std::vector<double> scale(n, 1.0);  // jacobian_scaling_ (fixed after iteration 0)
  if (options.jacobi_scaling)
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(h[i]));
  auto gradMax = [&]() { double m = 0; for (double v : g) m = std::max(m, std::fabs(v)); return m; };

  // --- dogleg strategy state (ceres/internal/ceres/dogleg_strategy.cc)
  double radius = options.initial_trust_region_radius;
  const double min_mu = 1e-8, max_mu = 1.0, mu_increase_factor = 10.0;
  const double min_diagonal = 1e-6, max_diagonal = 1e32;
  double mu = min_mu;
  bool reuse = false;
  double dogleg_step_norm = 0, alpha = 0;
  std::vector<double> htil(n), ghat(n), gnhat(n), stephat(n), delta(n);
  int num_consecutive_invalid = 0;
  int iteration = 0;
  double last_iter_time = 0;
  bool lastSuccessful = false;
  summary.termination = 1;

  auto finish = [&](int term) {
    summary.termination = term;
    summary.final_cost = x_cost;
    summary.iterations = iteration;
    summary.total_time = elapsed();
    if (getenv("ORC_TIMING"))
      std::printf("[orc timing, %d threads] total %.4f s: evaluate+jacobians %.4f, evaluate %.4f, schur %.4f, cholesky %.4f, "
                  "J*v %.4f, gradient %.4f\n", p.numThreads, summary.total_time, p.tEvalJ, p.tEval, p.tSchur, p.tChol, p.tJv, p.tGrad);
  };

  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (lastSuccessful) summary.num_successful_steps++;
    // iteration callback (CeresIterationCallback.hpp:73-81)
    if (options.time_limit >= 0.0 && iteration >= options.min_iterations &&
        elapsed() + last_iter_time > options.time_limit) { finish(2); return; }
    if (iteration >= options.max_num_iterations) { finish(1); return; }
    if (gradMax() <= options.gradient_tolerance) { finish(0); return; }
    if (radius <= options.min_trust_region_radius) { finish(0); return; }
    const double t_iter0 = elapsed();
    ++iteration;
    lastSuccessful = false;

    // --- ComputeStep
    bool stepOk = true;
    if (!reuse) {
      reuse = true;
      for (int i = 0; i < n; ++i) {
        const double ds2 = std::min(std::max(h[i] * scale[i] * scale[i], min_diagonal), max_diagonal);
        htil[i] = ds2 / (scale[i] * scale[i]);
        ghat[i] = g[i] / std::sqrt(htil[i]);
      }
      // Cauchy point
      std::vector<double> v(n);
      for (int i = 0; i < n; ++i) v[i] = g[i] / htil[i];
      double Jg2, dummy;
      p.Jtimes(v, Jg2, dummy);
      double gh2 = 0;
      for (double a : ghat) gh2 += a * a;
      alpha = gh2 / Jg2;
      // Gauss-Newton step with increasing regularisation on failure
      bool solved = false;
      std::vector<double> y, damp(n);
      while (mu < max_mu) {
        for (int i = 0; i < n; ++i) damp[i] = mu * htil[i];
        if (p.solveNormal(damp, y)) { solved = true; break; }
        mu *= mu_increase_factor;
      }
      if (!solved) stepOk = false;
      else for (int i = 0; i < n; ++i) gnhat[i] = -std::sqrt(htil[i]) * y[i];
    }
    if (stepOk) {
      // traditional dogleg
      double gnorm = vnorm(ghat), gnnorm = vnorm(gnhat);
      if (gnnorm <= radius) {
        stephat = gnhat;
        dogleg_step_norm = gnnorm;
      } else if (gnorm * alpha >= radius) {
        for (int i = 0; i < n; ++i) stephat[i] = -(radius / gnorm) * ghat[i];
        dogleg_step_norm = radius;
      } else {
        double b_dot_a = 0;
        for (int i = 0; i < n; ++i) b_dot_a += -alpha * ghat[i] * gnhat[i];
        const double a_sq = (alpha * gnorm) * (alpha * gnorm);
        const double b_minus_a_sq = a_sq - 2 * b_dot_a + gnnorm * gnnorm;
        const double c = b_dot_a - a_sq;
        const double dd = std::sqrt(c * c + b_minus_a_sq * (radius * radius - a_sq));
        const double beta = (c <= 0) ? (dd - c) / b_minus_a_sq : (radius * radius - a_sq) / (dd + c);
        for (int i = 0; i < n; ++i) stephat[i] = (-alpha * (1.0 - beta)) * ghat[i] + beta * gnhat[i];
        dogleg_step_norm = vnorm(stephat);
      }
      for (int i = 0; i < n; ++i) delta[i] = stephat[i] / std::sqrt(htil[i]);
    }
    // --- model cost change
    double model_cost_change = 0;
    if (stepOk) {
      double Jd2, Jdr;
      p.Jtimes(delta, Jd2, Jdr);
      model_cost_change = -(Jdr + 0.5 * Jd2);
    }
    if (!stepOk || !(model_cost_change > 0.0)) {
      // HandleInvalidStep
      if (++num_consecutive_invalid >= 5) { finish(3); return; }
      mu *= mu_increase_factor;
      reuse = false;
      summary.cost_history.push_back(x_cost);
      last_iter_time = elapsed() - t_iter0;
      continue;
    }
    num_consecutive_invalid = 0;
    // --- candidate
    p.applyDelta(delta);
    p.snapshot(xcand);
    const double candidate_cost = p.evaluate(false);
    double step_norm = 0;
    for (size_t i = 0; i < x.size(); ++i) step_norm += (x[i] - xcand[i]) * (x[i] - xcand[i]);
    step_norm = std::sqrt(step_norm);
    if (step_norm <= options.parameter_tolerance * (x_norm + options.parameter_tolerance)) {
      p.restore(x);
      finish(0); return;
    }
    const double cost_change = x_cost - candidate_cost;
    if (std::fabs(cost_change) <= options.function_tolerance * x_cost) {
      p.restore(x);
      finish(0); return;
    }
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > options.min_relative_decrease) {
      // HandleSuccessfulStep
      x = xcand;
      x_norm = vnorm(x);
      x_cost = p.evaluate(true);
      p.gradientAndColumnNorms(g, h);
      lastSuccessful = true;
      if (relative_decrease < 0.25) radius *= 0.5;
      if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
      radius = std::min(radius, options.max_trust_region_radius);
      mu = std::max(min_mu, 2.0 * mu / mu_increase_factor);
      reuse = false;
    } else {
      p.restore(x);
      radius *= 0.5;
      reuse = true;
    }
    summary.cost_history.push_back(x_cost);
    if (options.verbose)
      std::printf("[orc] it %d cost %.9e rel_dec %.3e radius %.3e step %.3e\n", iteration, x_cost, relative_decrease,
                  radius, step_norm);
    last_iter_time = elapsed() - t_iter0;
  }
}

}  // namespace orc
