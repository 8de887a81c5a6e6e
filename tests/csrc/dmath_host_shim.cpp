// Test-only shim: compiles the product's device math header (svin_amd/csrc/dmath.hpp) for the HOST so
// that the exact device functions can be compared with the oracle without a GPU.  Not part of the product.
#include "../../svin_amd/csrc/dmath.hpp"
using namespace svin;
extern "C" {
void hd_reproj(const double* cam12, int model, const double* T_WS, const double* hp, const double* T_SC, double u, double v,
               double w, double* r, double* Jp, double* Jl, double* Je) {
  CameraModel c;
  c.fu = cam12[0]; c.fv = cam12[1]; c.cu = cam12[2]; c.cv = cam12[3];
  for (int i = 0; i < 8; ++i) c.k[i] = cam12[4 + i];
  c.model = model; c.width = 0; c.height = 0; c.pad = 0; c.pad2 = 0;
  reprojEval(c, T_WS, hp, T_SC, u, v, w, r, Jp, Jl, Je);
}
void hd_pose_oplus(const double* x, const double* d, double* xo) { poseOplus(x, d, xo); }
void hd_pose_minus(const double* xp, const double* x, double* d) { poseMinus(xp, x, d); }
}
