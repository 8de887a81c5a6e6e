// C entry points over svin_amd/csrc/trust_region.hpp (the host decisions of Window::solve) for tests/test_trust_region_host.py:
// the state machine is HIP-free, so its rules are checked on the CPU -- including that two ranks fed the same all-reduced
// numbers take the same decisions whatever their rank-local fields hold.
#include "../../svin_amd/csrc/trust_region.hpp"

extern "C" {
void* tr_create(double fTol, double gTol, double pTol, int maxIterations, double initialCost) {
  auto* t = new svin::TrustRegionHost();
  t->fTol = fTol; t->gTol = gTol; t->pTol = pTol; t->maxIterations = maxIterations;
  t->start(initialCost);
  return t;
}
void tr_destroy(void* h) { delete static_cast<svin::TrustRegionHost*>(h); }
int tr_begin(void* h, int stopRequested) { return static_cast<svin::TrustRegionHost*>(h)->beginIteration(stopRequested != 0) ? 1 : 0; }
static svin::TrScalars unpack(const double* v) {
  svin::TrScalars s;
  s.cost = v[0]; s.stepNormSq = v[1]; s.xNormSq = v[2]; s.gradMax = v[3]; s.failMax = v[4]; s.jdSq = v[5]; s.jdDotR = v[6]; s.doglegStepNorm = v[7];
  return s;
}
int tr_retry(void* h, const double* v) { return static_cast<svin::TrustRegionHost*>(h)->retryFactorisation(unpack(v)) ? 1 : 0; }
int tr_end(void* h, const double* v) { return (int)static_cast<svin::TrustRegionHost*>(h)->endIteration(unpack(v)); }
// radius, mu, x_cost, reuse, initScale, invalid, iteration, successful, termination, muAfterAccept
void tr_state(void* h, double* out) {
  auto* t = static_cast<svin::TrustRegionHost*>(h);
  out[0] = t->radius; out[1] = t->mu; out[2] = t->x_cost; out[3] = t->reuse; out[4] = t->initScale; out[5] = t->invalid;
  out[6] = t->iteration; out[7] = t->successful; out[8] = t->termination; out[9] = t->muAfterAccept();
}
}
