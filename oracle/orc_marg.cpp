// ORACLE -- test infrastructure only (never linked into the product library).
// Restatement of okvis_ceres/src/MarginalizationError.cpp (line numbers cited inline).
#include "orc_marg.hpp"

namespace orc {

void pseudoInverseSymmSqrt(const double* A, int n, double* result) {
  std::vector<double> ev(n), U((size_t)n * n);
  sym_eig(A, n, ev.data(), U.data());
  double mx = ev[0];
  for (int i = 1; i < n; ++i) mx = std::max(mx, ev[i]);
  const double tol = std::numeric_limits<double>::epsilon() * n * mx;
  for (int j = 0; j < n; ++j) {
    const double s = (ev[j] > tol) ? std::sqrt(1.0 / ev[j]) : 0.0;
    for (int i = 0; i < n; ++i) result[(size_t)i * n + j] = U[(size_t)i * n + j] * s;
  }
}

void MarginalizationError::insertZeros(int pos, int k) {
  const int n0 = n_, n1 = n_ + k;
  std::vector<double> Hn((size_t)n1 * n1, 0.0), bn(n1, 0.0);
  auto map = [&](int i) { return i < pos ? i : i + k; };
  for (int i = 0; i < n0; ++i) {
    bn[map(i)] = b0_[i];
    for (int j = 0; j < n0; ++j) Hn[(size_t)map(i) * n1 + map(j)] = H_[(size_t)i * n0 + j];
  }
  H_.swap(Hn);
  b0_.swap(bn);
  n_ = n1;
}

// :126-397
bool MarginalizationError::addResidualBlock(uint64_t resId, bool keep) {
  if (!map_->residualExists(resId)) return false;
  const ResidualBlock rb = map_->residual(resId);  // copy (we may remove it)
  valid_ = false;
  {   // test hook: what was linearised (definition only, see orc_marg.hpp M1Entry)
    M1Entry e;
    e.resId = resId; e.kind = (int)rb.err->kind(); e.loss = rb.loss; e.lossParam = rb.lossParam; e.ids = rb.params;
    std::vector<double>& d = e.def;
    if (auto* r = dynamic_cast<const ReprojectionError*>(rb.err.get())) {
      d = {r->z[0], r->z[1], r->sqrtInfo[0], r->sqrtInfo[1], r->sqrtInfo[2], r->sqrtInfo[3], (double)r->camIdx, (double)r->cam.model,
           r->cam.fu, r->cam.fv, r->cam.cu, r->cam.cv};
      d.insert(d.end(), r->cam.k, r->cam.k + 8);
    } else if (auto* im = dynamic_cast<const ImuError*>(rb.err.get())) {
      d = {(double)im->t0.sec, (double)im->t0.nsec, (double)im->t1.sec, (double)im->t1.nsec, im->redo ? 1.0 : 0.0};
      d.insert(d.end(), im->sb_ref, im->sb_ref + 9);
      const ImuParameters& q = im->par;
      const double pv[10] = {q.a_max, q.g_max, q.sigma_g_c, q.sigma_a_c, q.sigma_bg, q.sigma_ba, q.sigma_gw_c, q.sigma_aw_c, q.tau, q.g};
      d.insert(d.end(), pv, pv + 10);
      d.push_back((double)im->meas.size());
      for (const ImuSample& sm : im->meas) {
        d.push_back((double)sm.t.sec); d.push_back((double)sm.t.nsec);
        d.insert(d.end(), sm.gyr, sm.gyr + 3);
        d.insert(d.end(), sm.acc, sm.acc + 3);
      }
    } else if (auto* pe = dynamic_cast<const PoseError*>(rb.err.get())) {
      d.assign(pe->meas.p, pe->meas.p + 7);
      d.insert(d.end(), pe->sqrtInfo, pe->sqrtInfo + 36);
    } else if (auto* se = dynamic_cast<const SpeedAndBiasError*>(rb.err.get())) {
      d.assign(se->meas, se->meas + 9);
      d.insert(d.end(), se->sqrtInfo, se->sqrtInfo + 81);
    } else if (auto* re = dynamic_cast<const RelativePoseError*>(rb.err.get())) {
      d.assign(re->sqrtInfo, re->sqrtInfo + 36);
    }
    m1log_.push_back(std::move(e));
  }
  const int nb = (int)rb.params.size();
  // :139-228 book-keeping
  for (int i = 0; i < nb; ++i) {
    const ParamBlock& pb = map_->param(rb.params[i]);
    if (id2idx_.count(pb.id)) continue;
    const bool isLandmark = (pb.type == BLOCK_HPOINT);
    const int additional = pb.fixed ? 0 : pb.mdim();
    int denseSize = 0;
    if (denseIndices_ > 0) denseSize = infos_[denseIndices_ - 1].orderingIdx + infos_[denseIndices_ - 1].mdim;
    Info info;
    info.id = pb.id; info.type = pb.type; info.dim = pb.dim();
    info.mdim = pb.fixed ? 0 : pb.mdim();
    info.isLandmark = isLandmark;
    std::memcpy(info.lin, pb.x, sizeof(double) * pb.dim());
    if (!isLandmark) {
      if (additional > 0) insertZeros(denseSize, additional);
      info.orderingIdx = denseSize;
      infos_.insert(infos_.begin() + denseIndices_, info);
      denseIndices_++;
      for (size_t j = denseIndices_; j < infos_.size(); ++j) infos_[j].orderingIdx += additional;
    } else {
      if (additional > 0) insertZeros(n_, additional);
      info.orderingIdx = infos_.empty() ? 0 : infos_.back().orderingIdx + infos_.back().mdim;
      infos_.push_back(info);
    }
    id2idx_.clear();
    for (size_t j = 0; j < infos_.size(); ++j) id2idx_[infos_[j].id] = j;
  }
  // :233-270 evaluate at the stored linearisation points (first-estimate Jacobians)
  const int m = rb.err->residualDim();
  std::vector<const double*> P(nb);
  std::vector<std::vector<double>> Ja(nb), Jm(nb);
  std::vector<double*> Jap(nb), Jmp(nb);
  for (int i = 0; i < nb; ++i) {
    const Info& inf = infos_[id2idx_.at(rb.params[i])];
    P[i] = inf.lin;
    Ja[i].assign((size_t)m * blockDim(inf.type), 0.0);
    Jm[i].assign((size_t)m * blockMinDim(inf.type), 0.0);
    Jap[i] = Ja[i].data(); Jmp[i] = Jm[i].data();
  }
  std::vector<double> r(m);
  rb.err->evaluate(P.data(), r.data(), Jap.data(), Jmp.data());
  // :274-330 robust-loss corrector (Ceres corrector.cc)
  if (rb.loss != LOSS_NONE) {
    double sq = 0;
    for (double v : r) sq += v * v;
    double rho[3];
    lossEvaluate(rb.loss, rb.lossParam, sq, rho);
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
    for (int i = 0; i < nb; ++i) {
      const int md = blockMinDim(infos_[id2idx_.at(rb.params[i])].type);
      for (int c = 0; c < md; ++c) {
        double rtJ = 0;
        for (int a = 0; a < m; ++a) rtJ += r[a] * Jm[i][(size_t)a * md + c];
        for (int a = 0; a < m; ++a)
          Jm[i][(size_t)a * md + c] = sqrt_rho1 * (Jm[i][(size_t)a * md + c] - alpha_sq_norm * r[a] * rtJ);
      }
    }
    for (double& v : r) v *= residual_scaling;
  }
  // :333-382 accumulate H, b0
  for (int i = 0; i < nb; ++i) {
    const Info& ii = infos_[id2idx_.at(rb.params[i])];
    if (ii.mdim == 0) continue;
    const int mdi = blockMinDim(ii.type);
    for (int a = 0; a < ii.mdim; ++a) {
      double sb = 0;
      for (int k = 0; k < m; ++k) sb += Jm[i][(size_t)k * mdi + a] * r[k];
      b0_[ii.orderingIdx + a] -= sb;
    }
    for (int j = 0; j <= i; ++j) {
      const Info& ij = infos_[id2idx_.at(rb.params[j])];
      if (ij.mdim == 0) continue;
      const int mdj = blockMinDim(ij.type);
      for (int a = 0; a < ii.mdim; ++a)
        for (int c = 0; c < ij.mdim; ++c) {
          double s = 0;
          for (int k = 0; k < m; ++k) s += Jm[i][(size_t)k * mdi + a] * Jm[j][(size_t)k * mdj + c];
          H_[(size_t)(ii.orderingIdx + a) * n_ + ij.orderingIdx + c] += s;
          if (j != i) H_[(size_t)(ij.orderingIdx + c) * n_ + ii.orderingIdx + a] += s;
        }
    }
  }
  if (!keep) map_->removeResidualBlock(resId);
  return true;
}

namespace {
// splitSymmetricMatrix / splitVector (implementation/MarginalizationError.hpp:48-160) expressed
// through index lists: `keepIdx` = kept rows, `margIdx` = marginalised rows, both ascending.
void buildIndexLists(int n, const std::vector<std::pair<int, int>>& pairs, std::vector<int>& keepIdx,
                     std::vector<int>& margIdx) {
  std::vector<char> isMarg(n, 0);
  for (auto& p : pairs)
    for (int k = 0; k < p.second; ++k) isMarg[p.first + k] = 1;
  for (int i = 0; i < n; ++i) (isMarg[i] ? margIdx : keepIdx).push_back(i);
}
}  // namespace

// :463-721
bool MarginalizationError::marginalizeOut(const std::vector<uint64_t>& idsIn) {
  if (idsIn.empty()) return false;
  std::vector<uint64_t> ids = idsIn;
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  std::vector<std::pair<int, int>> pairsLm, pairsDense;
  int margLm = 0, margDense = 0;
  for (uint64_t id : ids) {
    auto it = id2idx_.find(id);
    if (it == id2idx_.end()) return false;  // OKVIS_ASSERT_TRUE :504-507 (throws in the reference)
    const Info& inf = infos_[it->second];
    if (inf.isLandmark) { pairsLm.emplace_back(inf.orderingIdx, inf.mdim); margLm += inf.mdim; }
    else { pairsDense.emplace_back(inf.orderingIdx, inf.mdim); margDense += inf.mdim; }
  }
  auto byFirst = [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; };
  std::sort(pairsLm.begin(), pairsLm.end(), byFirst);
  std::sort(pairsDense.begin(), pairsDense.end(), byFirst);
  valid_ = false;
  pre_.n = n_; pre_.H = H_; pre_.b0 = b0_; pre_.lm = pairsLm; pre_.dense = pairsDense;   // test hook, see orc_marg.hpp
  pre_.infos = infos_; pre_.log.swap(m1log_); m1log_.clear(); pre_.hadPrior = logStartedWithPrior_; logStartedWithPrior_ = true;

  // ---- landmark part (:557-619)
  if (!pairsLm.empty()) {
    const int n = n_;
    std::vector<double> p(n), pinv(n);
    for (int i = 0; i < n; ++i) {
      const double hd = H_[(size_t)i * n + i];
      p[i] = (hd > 1.0e-9) ? std::sqrt(hd) : 1.0e-3;
      pinv[i] = 1.0 / p[i];
    }
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) H_[(size_t)i * n + j] = pinv[i] * H_[(size_t)i * n + j] * pinv[j];
      b0_[i] = pinv[i] * b0_[i];
    }
    std::vector<int> keepIdx, margIdx;
    buildIndexLists(n, pairsLm, keepIdx, margIdx);
    const int na = (int)keepIdx.size(), nm = (int)margIdx.size();
    std::vector<double> U((size_t)na * na), W((size_t)na * nm), b_a(na), b_b(nm), p_a(na);
    for (int i = 0; i < na; ++i) {
      p_a[i] = p[keepIdx[i]];
      b_a[i] = b0_[keepIdx[i]];
      for (int j = 0; j < na; ++j) U[(size_t)i * na + j] = H_[(size_t)keepIdx[i] * n + keepIdx[j]];
      for (int j = 0; j < nm; ++j) W[(size_t)i * nm + j] = H_[(size_t)keepIdx[i] * n + margIdx[j]];
    }
    for (int j = 0; j < nm; ++j) b_b[j] = b0_[margIdx[j]];
    std::vector<double> dH((size_t)na * na, 0.0), db(na, 0.0);
    std::vector<double> M((size_t)na * 3), M1((size_t)na * 3);
    for (int i = 0; i < nm; i += 3) {
      double V1[9], Vis[9];
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) V1[a * 3 + c] = H_[(size_t)margIdx[i + a] * n + margIdx[i + c]];
      pseudoInverseSymmSqrt(V1, 3, Vis);
      double VV[9];  // V_inv_sqrt * V_inv_sqrt^T
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) VV[a * 3 + c] = Vis[a * 3] * Vis[c * 3] + Vis[a * 3 + 1] * Vis[c * 3 + 1] + Vis[a * 3 + 2] * Vis[c * 3 + 2];
      for (int r = 0; r < na; ++r) {
        const double* w = &W[(size_t)r * nm + i];
        for (int c = 0; c < 3; ++c) {
          M[(size_t)r * 3 + c] = w[0] * Vis[c] + w[1] * Vis[3 + c] + w[2] * Vis[6 + c];
          M1[(size_t)r * 3 + c] = w[0] * VV[c] + w[1] * VV[3 + c] + w[2] * VV[6 + c];
        }
      }
      for (int r = 0; r < na; ++r) {
        db[r] += M1[(size_t)r * 3] * b_b[i] + M1[(size_t)r * 3 + 1] * b_b[i + 1] + M1[(size_t)r * 3 + 2] * b_b[i + 2];
        for (int c = 0; c < na; ++c)
          dH[(size_t)r * na + c] += M[(size_t)r * 3] * M[(size_t)c * 3] + M[(size_t)r * 3 + 1] * M[(size_t)c * 3 + 1] + M[(size_t)r * 3 + 2] * M[(size_t)c * 3 + 2];
      }
    }
    H_.assign((size_t)na * na, 0.0);
    b0_.assign(na, 0.0);
    for (int i = 0; i < na; ++i) {
      b0_[i] = p_a[i] * (b_a[i] - db[i]);
      for (int j = 0; j < na; ++j) H_[(size_t)i * na + j] = p_a[i] * (U[(size_t)i * na + j] - dH[(size_t)i * na + j]) * p_a[j];
    }
    n_ = na;
    // the dense pairs were expressed in the old ordering: shift them
    for (auto& pr : pairsDense) {
      int shift = 0;
      for (auto& pl : pairsLm) if (pl.first < pr.first) shift += pl.second;
      pr.first -= shift;
    }
  }
  // ---- dense part (:622-667)
  if (!pairsDense.empty()) {
    const int n = n_;
    std::vector<double> p(n), pinv(n);
    for (int i = 0; i < n; ++i) {
      const double hd = H_[(size_t)i * n + i];
      p[i] = (hd > 1.0e-9) ? std::sqrt(hd) : 1.0e-3;
      pinv[i] = 1.0 / p[i];
    }
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) H_[(size_t)i * n + j] = pinv[i] * H_[(size_t)i * n + j] * pinv[j];
      b0_[i] = pinv[i] * b0_[i];
    }
    std::vector<int> keepIdx, margIdx;
    buildIndexLists(n, pairsDense, keepIdx, margIdx);
    const int na = (int)keepIdx.size(), nm = (int)margIdx.size();
    std::vector<double> U((size_t)na * na), W((size_t)na * nm), V((size_t)nm * nm), b_a(na), b_b(nm), p_a(na);
    for (int i = 0; i < na; ++i) {
      p_a[i] = p[keepIdx[i]];
      b_a[i] = b0_[keepIdx[i]];
      for (int j = 0; j < na; ++j) U[(size_t)i * na + j] = H_[(size_t)keepIdx[i] * n + keepIdx[j]];
      for (int j = 0; j < nm; ++j) W[(size_t)i * nm + j] = H_[(size_t)keepIdx[i] * n + margIdx[j]];
    }
    for (int i = 0; i < nm; ++i) {
      b_b[i] = b0_[margIdx[i]];
      for (int j = 0; j < nm; ++j) V[(size_t)i * nm + j] = H_[(size_t)margIdx[i] * n + margIdx[j]];
    }
    std::vector<double> V1((size_t)nm * nm), Vis((size_t)nm * nm);
    for (int i = 0; i < nm; ++i)
      for (int j = 0; j < nm; ++j) V1[(size_t)i * nm + j] = 0.5 * (V[(size_t)i * nm + j] + V[(size_t)j * nm + i]);
    pseudoInverseSymmSqrt(V1.data(), nm, Vis.data());
    std::vector<double> M((size_t)na * nm), t(nm);
    for (int i = 0; i < na; ++i)
      for (int j = 0; j < nm; ++j) {
        double s = 0;
        for (int k = 0; k < nm; ++k) s += W[(size_t)i * nm + k] * Vis[(size_t)k * nm + j];
        M[(size_t)i * nm + j] = s;
      }
    for (int j = 0; j < nm; ++j) {  // t = V_inverse_sqrt^T * b_b
      double s = 0;
      for (int k = 0; k < nm; ++k) s += Vis[(size_t)k * nm + j] * b_b[k];
      t[j] = s;
    }
    H_.assign((size_t)na * na, 0.0);
    b0_.assign(na, 0.0);
    for (int i = 0; i < na; ++i) {
      double s = 0;
      for (int k = 0; k < nm; ++k) s += M[(size_t)i * nm + k] * t[k];
      b0_[i] = p_a[i] * (b_a[i] - s);
      for (int j = 0; j < na; ++j) {
        double mm = 0;
        for (int k = 0; k < nm; ++k) mm += M[(size_t)i * nm + k] * M[(size_t)j * nm + k];
        H_[(size_t)i * na + j] = p_a[i] * (U[(size_t)i * na + j] - mm) * p_a[j];
      }
    }
    n_ = na;
  }
  // ---- book-keeping (:673-716)
  for (uint64_t id : ids) {
    const size_t idx = id2idx_.at(id);
    const int margSize = infos_[idx].mdim;
    infos_.erase(infos_.begin() + idx);
    for (size_t j = idx; j < infos_.size(); ++j) infos_[j].orderingIdx -= margSize;
    id2idx_.clear();
    for (size_t j = 0; j < infos_.size(); ++j) id2idx_[infos_[j].id] = j;
  }
  denseIndices_ = infos_.size();
  for (auto& inf : infos_) inf.isLandmark = false;
  for (uint64_t id : ids) map_->removeParameterBlock(id);
  return true;
}

// :725-758
void MarginalizationError::updateErrorComputation() {
  if (valid_) return;
  const int n = n_;
  std::vector<double> p(n), pinv(n);
  for (int i = 0; i < n; ++i) {
    const double hd = H_[(size_t)i * n + i];
    p[i] = (hd > 1.0e-9) ? std::sqrt(hd) : 1.0e-3;
    pinv[i] = 1.0 / p[i];
  }
  std::vector<double> Hs((size_t)n * n), ev(n), U((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Hs[(size_t)i * n + j] = 0.5 * pinv[i] * (H_[(size_t)i * n + j] + H_[(size_t)j * n + i]) * pinv[j];
  sym_eig(Hs.data(), n, ev.data(), U.data());
  double mx = n ? ev[0] : 0;
  for (int i = 1; i < n; ++i) mx = std::max(mx, ev[i]);
  const double tol = std::numeric_limits<double>::epsilon() * n * mx;
  std::vector<double> S_sqrt(n), S_pinv_sqrt(n);
  for (int i = 0; i < n; ++i) {
    const double s = ev[i] > tol ? ev[i] : 0.0;
    const double si = ev[i] > tol ? 1.0 / ev[i] : 0.0;
    S_sqrt[i] = std::sqrt(s);
    S_pinv_sqrt[i] = std::sqrt(si);
  }
  J_.assign((size_t)n * n, 0.0);
  e0_.assign(n, 0.0);
  for (int i = 0; i < n; ++i) {      // row i of J = i-th eigen-direction
    double e = 0;
    for (int j = 0; j < n; ++j) {
      J_[(size_t)i * n + j] = p[j] * U[(size_t)j * n + i] * S_sqrt[i];
      e += S_pinv_sqrt[i] * U[(size_t)j * n + i] * pinv[j] * b0_[j];
    }
    e0_[i] = -e;
  }
  valid_ = true;
}

// :798-844 (+ computeDeltaChi :776-789)
bool MarginalizationError::evaluate(double const* const* P, double* res, double** J, double** Jmin) const {
  const int n = n_;
  std::vector<double> dchi(n, 0.0);
  for (size_t i = 0; i < infos_.size(); ++i) {
    const Info& inf = infos_[i];
    if (inf.mdim == 0) continue;
    double d[9];
    manifoldMinus(inf.type, P[i], inf.lin, d);
    for (int k = 0; k < inf.mdim; ++k) dchi[inf.orderingIdx + k] = d[k];
  }
  for (size_t i = 0; i < infos_.size(); ++i) {
    const Info& inf = infos_[i];
    if (Jmin && Jmin[i])
      for (int a = 0; a < n; ++a)
        for (int c = 0; c < inf.mdim; ++c) Jmin[i][(size_t)a * inf.mdim + c] = J_[(size_t)a * n + inf.orderingIdx + c];
    if (J && J[i]) {
      if (inf.mdim == 0) { std::memset(J[i], 0, sizeof(double) * n * inf.dim); continue; }
      double L[81];
      manifoldLiftJacobian(inf.type, inf.lin, L);
      for (int a = 0; a < n; ++a)
        for (int c = 0; c < inf.dim; ++c) {
          double s = 0;
          for (int k = 0; k < inf.mdim; ++k) s += J_[(size_t)a * n + inf.orderingIdx + k] * L[k * inf.dim + c];
          J[i][(size_t)a * inf.dim + c] = s;
        }
    }
  }
  for (int a = 0; a < n; ++a) {
    double s = e0_[a];
    for (int k = 0; k < n; ++k) s += J_[(size_t)a * n + k] * dchi[k];
    res[a] = s;
  }
  return true;
}

}  // namespace orc
