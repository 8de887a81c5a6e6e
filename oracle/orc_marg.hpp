// ORACLE -- test infrastructure only (never linked into the product library).
// Restatement of okvis::ceres::MarginalizationError
// (okvis_ceres/src/MarginalizationError.cpp, include/okvis/ceres/implementation/MarginalizationError.hpp).
#pragma once
#include "orc_map.hpp"

namespace orc {

class MarginalizationError : public ErrorTerm {
 public:
  struct Info {  // MarginalizationError.hpp:296-345
    uint64_t id = 0;
    int type = 0;
    int orderingIdx = 0, dim = 0, mdim = 0;
    bool isLandmark = false;
    double lin[9] = {0};
  };
  explicit MarginalizationError(Map* map) : map_(map) {}
  // M1: MarginalizationError.cpp:126-397
  bool addResidualBlock(uint64_t resId, bool keep = false);
  // M2: :463-721
  bool marginalizeOut(const std::vector<uint64_t>& ids);
  // M3: :725-758
  void updateErrorComputation();
  bool isParameterBlockConnected(uint64_t id) const { return id2idx_.count(id) != 0; }
  std::vector<uint64_t> parameterBlockIds() const { std::vector<uint64_t> v; for (auto& i : infos_) v.push_back(i.id); return v; }

  // ErrorTerm (M4: :798-844)
  Kind kind() const override { return MARGINALIZATION; }
  int residualDim() const override { return n_; }
  int numBlocks() const override { return (int)infos_.size(); }
  int blockType(int i) const override { return infos_[i].type; }
  bool evaluate(double const* const* params, double* residuals, double** J, double** Jmin) const override;

  // inspection
  int size() const { return n_; }
  const std::vector<double>& H() const { return H_; }
  const std::vector<double>& b0() const { return b0_; }
  const std::vector<double>& Jmat() const { return J_; }
  const std::vector<double>& e0() const { return e0_; }
  const std::vector<Info>& infos() const { return infos_; }
  // test hook: the system as it stood when the last marginalizeOut() began (after M1, before M2), the marginalised index
  // ranges (first row, size) of the landmark and of the dense part, both in THAT ordering -- lets an independent
  // high-precision restatement of M2 / M3 arbitrate between two double-precision implementations
  // M1Entry: one addResidualBlock call since the previous marginalizeOut -- WHICH residual was linearised (kind, loss, its
  // parameter blocks) and its DEFINITION (measurement, weights, for an IMU factor the samples and the bias its
  // pre-integration currently refers to).  No number the oracle computed from them: tests/mp_m1.py evaluates residual,
  // Jacobians and corrector again from these definitions at 40 digits.
  struct M1Entry {
    uint64_t resId = 0;
    int kind = 0, loss = 0;
    double lossParam = 0;
    std::vector<uint64_t> ids;
    std::vector<double> def;
  };
  struct PreMarg {
    int n = 0;
    std::vector<double> H, b0;
    std::vector<std::pair<int, int>> lm, dense;
    std::vector<Info> infos;      // the blocks with their ordering and linearisation points, as M1 left them
    std::vector<M1Entry> log;     // the residuals M1 took in since the previous marginalizeOut
    bool hadPrior = false;        // a previous prior (H, b0) was already in the system when the log starts
  };
  const PreMarg& preMarg() const { return pre_; }

 private:
  void insertZeros(int pos, int k);  // grow H_/b0_ by k rows+cols at index pos
  Map* map_;
  int n_ = 0;
  std::vector<double> H_, b0_;
  std::vector<Info> infos_;
  std::map<uint64_t, size_t> id2idx_;
  size_t denseIndices_ = 0;
  bool valid_ = false;
  std::vector<double> J_, e0_;
  PreMarg pre_;
  std::vector<M1Entry> m1log_;
  bool logStartedWithPrior_ = false;
};

// helpers shared with tests: pseudo-inverse square root of a symmetric PSD matrix
// (implementation/MarginalizationError.hpp:187-220): result = U * diag(sqrt(1/lambda_i | 0))
void pseudoInverseSymmSqrt(const double* A, int n, double* result);

}  // namespace orc
