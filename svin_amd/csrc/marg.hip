// svin_amd marginalisation (K10): Estimator::applyMarginalizationStrategy policy on the host,
// MarginalizationError algebra on the device.
//
// Reference (relative to /root/reference/okvis_ros/okvis/okvis_ceres/):
//   policy  src/Estimator.cpp:495-814
//   M1      src/MarginalizationError.cpp:126-397  (linearise at first-estimate points, H += J^T J, b0 -= J^T r)
//   M2      :463-721 + include/okvis/ceres/implementation/MarginalizationError.hpp:48-220
//   M3      :725-758
// Device layout of one marginalisation job: dense part U (m x m) / ba (m), landmark coupling W (m x 3Lm),
// landmark blocks V (Lm x 9) / bb (3Lm).  The reference's Jacobi preconditioner only influences the
// rank-revealing thresholds of the eliminated blocks; it is applied to exactly those blocks here
// (U - W V^+ W^T with V^+ = D^-1 (D^-1 V D^-1)^+ D^-1), which is the same matrix in exact arithmetic.
#include "window.hpp"
#include "symeig.hpp"
#include "tile16.hpp"
#include <chrono>
#include <memory>
#include <type_traits>
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace svin {

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

__device__ __forceinline__ double waveSumM(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------- M1 accumulation
struct MargDev {
  int m, Lm, N, F;
  double *U, *ba, *W, *V, *bb;  // U m x m, W m x 3Lm (row-major), V Lm x 9, bb 3Lm
};

// M1 is accumulated in a FIXED order (no atomics): the prior feeds a sequence of optimisations that amplifies rounding
// differences, so two runs of the same window must produce the same bits.
//   k_marg_accum_lm    one thread per marginalised landmark: V_l, bb_l and the three columns of W it owns, over its
//                      observations in CSR order
//   k_marg_accum_cam   one wave per (camera-side block, part): lane i sums the observations i, i + 64, ... that touch
//                      the block, then a fixed butterfly over the lanes; part 0 = diagonal block + right-hand side,
//                      part 1 + c = the pose x extrinsics cross block of camera c (written with its transpose)
//   k_marg_accum_factors  one workgroup, the small factors one after the other
template <bool WITH_EXT>
__global__ void k_marg_accum_lm(DeviceProblem p, MargDev md) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= md.Lm) return;
  const size_t N = (size_t)p.N;
  const int wld = 3 * md.Lm;
  double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
  for (int o = p.lmPtr[l]; o < p.lmPtr[l + 1]; ++o) {
    const uint32_t idx = p.obsIdx[o];
    const int offP = p.poseOff[idx & 0xfff];
    const int offE = WITH_EXT ? p.extOff[(idx >> 12) & 0xfff] : -1;
    const double r0 = p.rCur[o], r1 = p.rCur[N + o];
    double jl[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) jl[k] = p.JlCur[k * N + o];
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) V[a * 3 + b] += jl[a] * jl[b] + jl[3 + a] * jl[3 + b];
      bb[a] += -(jl[a] * r0 + jl[3 + a] * r1);
    }
    auto cam = [&](const double* J, int off) {
      if (off < 0) return;
      for (int a = 0; a < 6; ++a) {
        const double j0 = J[(size_t)a * N + o], j1 = J[(size_t)(6 + a) * N + o];
        double* w = md.W + (size_t)(off + a) * wld + 3 * l;   // columns 3l .. 3l+2 belong to this thread
        for (int b = 0; b < 3; ++b) w[b] += j0 * jl[b] + j1 * jl[3 + b];
      }
    };
    cam(p.JpCur, offP);
    if (WITH_EXT) cam(p.JeCur, offE);
  }
  for (int k = 0; k < 9; ++k) md.V[9 * (size_t)l + k] = V[k];
  for (int a = 0; a < 3; ++a) md.bb[3 * l + a] = bb[a];
}

template <bool WITH_EXT>
__global__ __launch_bounds__(64) void k_marg_accum_cam(DeviceProblem p, MargDev md) {
  const int target = blockIdx.x, part = blockIdx.y, lane = threadIdx.x;
  const bool isPose = target < p.nPose;
  const int slot = isPose ? target : target - p.nPose;
  const int off = isPose ? p.poseOff[slot] : p.extOff[slot];
  if (off < 0) return;
  if (part > 0 && !isPose) return;                 // cross blocks are written by the pose side
  const size_t N = (size_t)p.N;
  const int m = md.m;
  double acc[42];
#pragma unroll
  for (int k = 0; k < 42; ++k) acc[k] = 0.0;
  int partner = -1;                                // extrinsics slot of the cross block (unique per pose and camera)
  for (int o = lane; o < p.N; o += 64) {
    const uint32_t idx = p.obsIdx[o];
    const int ps = idx & 0xfff, es = (idx >> 12) & 0xfff, cam = (idx >> 24) & 0xf;
    if (isPose ? ps != slot : es != slot) continue;
    const double* J = isPose ? p.JpCur : p.JeCur;
    double ja[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) ja[k] = J[(size_t)k * N + o];
    if (part == 0) {
      const double r0 = p.rCur[o], r1 = p.rCur[N + o];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a * 6 + b] += ja[a] * ja[b] + ja[6 + a] * ja[6 + b];
        acc[36 + a] += -(ja[a] * r0 + ja[6 + a] * r1);
      }
    } else if (WITH_EXT) {
      if (cam != part - 1 || p.extOff[es] < 0) continue;
      partner = es;
      double je[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) je[k] = p.JeCur[(size_t)k * N + o];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a * 6 + b] += ja[a] * je[b] + ja[6 + a] * je[6 + b];
    }
  }
#pragma unroll
  for (int k = 0; k < 42; ++k) acc[k] = waveSumM(acc[k]);   // xor butterfly: the same order on every run
  for (int o = 32; o > 0; o >>= 1) partner = max(partner, __shfl_xor(partner, o, 64));
  if (lane != 0) return;
  if (part == 0) {
    for (int a = 0; a < 6; ++a) {
      for (int b = 0; b < 6; ++b) md.U[(size_t)(off + a) * m + off + b] += acc[a * 6 + b];
      md.ba[off + a] += acc[36 + a];
    }
  } else if (partner >= 0) {
    const int offE = p.extOff[partner];
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        md.U[(size_t)(off + a) * m + offE + b] += acc[a * 6 + b];
        md.U[(size_t)(offE + b) * m + off + a] += acc[a * 6 + b];
      }
  }
}

__global__ __launch_bounds__(256) void k_marg_accum_factors(DeviceProblem p, MargDev md) {
  __shared__ int colRow[30];
  for (int f = 0; f < md.F; ++f) {
    const FactorLin& lin = p.linCur[f];
    const int mm = lin.m, nc = lin.ncols;
    __syncthreads();
    if (threadIdx.x < 30) {
      int c = threadIdx.x, row = -1, base = 0;
      for (int b = 0; b < 4; ++b) {
        if (c >= base && c < base + lin.dim[b]) row = lin.off[b] < 0 ? -1 : lin.off[b] + (c - base);
        base += lin.dim[b];
      }
      colRow[threadIdx.x] = (c < nc) ? row : -1;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nc * nc; idx += blockDim.x) {
      const int a = idx / nc, b = idx % nc;
      const int ra = colRow[a], rb = colRow[b];
      if (ra < 0 || rb < 0) continue;
      double s = 0;
      for (int k = 0; k < mm; ++k) s += lin.J[k * nc + a] * lin.J[k * nc + b];
      md.U[(size_t)ra * md.m + rb] += s;             // distinct columns of one factor are distinct rows of U
      if (a == b) {
        double g = 0;
        for (int k = 0; k < mm; ++k) g += lin.J[k * nc + a] * lin.r[k];
        md.ba[ra] -= g;
      }
    }
  }
}

// ---------------------------------------------------------------- small symmetric eigen-solvers
// 3x3: cyclic Jacobi with eigenvectors (columns of Q, row-major 3x3)
__device__ void symEig3(const double* A9, double* ev, double* Q) {
  double a00 = A9[0], a01 = 0.5 * (A9[1] + A9[3]), a02 = 0.5 * (A9[2] + A9[6]), a11 = A9[4], a12 = 0.5 * (A9[5] + A9[7]),
         a22 = A9[8];
  for (int k = 0; k < 9; ++k) Q[k] = (k % 4 == 0) ? 1.0 : 0.0;
  auto rotQ = [&](int p, int q, double c, double s) {
    for (int r = 0; r < 3; ++r) {
      const double qp = Q[r * 3 + p], qq = Q[r * 3 + q];
      Q[r * 3 + p] = c * qp - s * qq;
      Q[r * 3 + q] = s * qp + c * qq;
    }
  };
  for (int sweep = 0; sweep < 16; ++sweep) {
    if (fabs(a01) + fabs(a02) + fabs(a12) == 0.0) break;
    if (a01 != 0.0) {
      const double th = (a11 - a00) / (2.0 * a01);
      const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
      const double n00 = a00 - tt * a01, n11 = a11 + tt * a01, n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
      a00 = n00; a11 = n11; a01 = 0; a02 = n02; a12 = n12;
      rotQ(0, 1, c, s);
    }
    if (a02 != 0.0) {
      const double th = (a22 - a00) / (2.0 * a02);
      const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
      const double n00 = a00 - tt * a02, n22 = a22 + tt * a02, n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
      a00 = n00; a22 = n22; a02 = 0; a01 = n01; a12 = n12;
      rotQ(0, 2, c, s);
    }
    if (a12 != 0.0) {
      const double th = (a22 - a11) / (2.0 * a12);
      const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
      const double n11 = a11 - tt * a12, n22 = a22 + tt * a12, n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
      a11 = n11; a22 = n22; a12 = 0; a01 = n01; a02 = n02;
      rotQ(1, 2, c, s);
    }
  }
  ev[0] = a00; ev[1] = a11; ev[2] = a22;
}

// n x n symmetric eigendecomposition by one-sided (Hestenes) Jacobi, executed by one workgroup.
// G (row j = column j of A on entry) is overwritten by the columns of A*Q; Q (row j = eigenvector j)
// must hold the identity on entry.  Eigenvalue j = Q_j . G_j.  Rounds follow the round-robin
// tournament so that the n/2 rotations of one round touch disjoint columns.
// columns count as orthogonal below this relative inner product: a few times the rounding noise eps*sqrt(n) of the
// dot product itself (1e-15 kept the solver chasing that noise for 5+ extra sweeps)
constexpr double kJacobiOrthTol = 2.0e-14;
constexpr double kJacobiFinalCos = 3.0e-9;
constexpr int kJacobiRegLen = 144;   // columns up to this (padded) length are held in registers by their lane group

// one 32-bit half at a time through DPP; kCtrl: row_ror:N = 0x120 + N (lane i of a 16-lane row reads lane (i - N) & 15)
template <int kCtrl>
__device__ __forceinline__ double dppRowMov(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_mov_dpp(lo, kCtrl, 0xf, 0xf, false);   // (row rotations only: every lane receives a value)
  hi = __builtin_amdgcn_mov_dpp(hi, kCtrl, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
// sum over the 16 lanes of a DPP row, in every lane; all 16 lanes must be active.  (__shfl_xor(.., 16) compiles to
// ds_bpermute: 24 LDS round trips per pair for the three inner products.)
__device__ __forceinline__ double rowSum16(double v) {
  v += dppRowMov<0x128>(v);
  v += dppRowMov<0x124>(v);
  v += dppRowMov<0x122>(v);
  v += dppRowMov<0x121>(v);
  return v;
}

// Workgroup barrier that only orders LDS traffic (__syncthreads() also waits for every outstanding global access).
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// pair k of round `round` of the round-robin tournament over np players (np even)
__device__ __forceinline__ void jacobiPair(int np, int round, int k, int& a, int& b) {
  if (k == 0) { a = np - 1; b = round; return; }
  a = round + k; if (a >= np - 1) a -= np - 1;
  b = round - k; if (b < 0) b += np - 1;
}
// Jacobi rotation that orthogonalises two columns with |p|^2 = al, |q|^2 = be, p.q = ga: the small-angle solution of
// tan(2 theta) = 2 ga / (be - al).  The rotation sits on the critical path of every tournament round, so it is written
// with two reciprocal square roots and no division: with h = hypot(be - al, 2 ga), cos(2 theta) = |be - al| / h and
// x = (1 + cos(2 theta)) / 2 in [1/2, 1]:  c = sqrt(x) = x rsqrt(x),  s = sin(2 theta) / (2 c) = sin(2 theta) rsqrt(x) / 2.
__device__ __forceinline__ void jacobiRotation(double al, double be, double ga, double& c, double& s) {
  const double d = be - al, g2 = 2.0 * ga;
  const double rh = rsqrt(d * d + g2 * g2);
  const double x = 0.5 + 0.5 * fabs(d) * rh;
  const double rx = rsqrt(x);
  c = x * rx;
  s = copysign(0.5 * g2 * rh * rx, d * g2);
}
// row length of the LDS images: the column length rounded up to the lane-group size (the tail is kept zero, so the
// register path needs no predication), odd so that the columns of a round start on different banks
__host__ __device__ inline int jacobiLd(int n) { return ((n + 15) & ~15) | 1; }
// Lanes per column pair: 16 (one DPP row).  A tournament round is one dependent chain per wave (LDS read -> inner
// products -> reduce -> rotation -> LDS write -> barrier) and what it costs is that chain's instruction count, not VALU
// throughput: 8 lanes per pair (twice the per-lane column length) measured 11 % slower at n = 105.
// Pointers into the LDS images carry their address space: through a generic double* every access is a FLAT
// instruction (aperture check in the texture addresser, counted on vmcnt and lgkmcnt), several times slower than ds_*.
using lds_double = __attribute__((address_space(3))) double;
__device__ __forceinline__ lds_double* toLds(double* p) { return (lds_double*)p; }

// one copy for all instantiations of the solver
struct JacobiShared {
  double nullTol2;
  int anyRotation, anyLargeRotation;
};
__shared__ JacobiShared gJacobiShared;

// LPG lanes per pair; P = lds_double* (the LDS images: every column has jacobiLd(n) addressable entries with a zero
// tail) or double* (global memory, leading dimension n).
// Returns the number of sweeps.  kHasQ: rotate the rows of Q along with G (compile-time: each use gets a straight-line round).
// NU > 0: the columns have exactly NU slices of LPG entries (compile-time: the register loops of a round then have no exit
// test between their LDS reads -- with the test each slice was its own LDS round trip, three in a row at n = 45)
template <int LPG, class P, bool kHasQ, int NU = 0>
__device__ int jacobiEigBlock(P G, P Q, int n, int ld, int* flag) {
  constexpr bool padded = std::is_same<P, lds_double*>::value;
  constexpr int kRegCols = kJacobiRegLen / LPG;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nWaves = blockDim.x >> 6;
  if (n <= 1) return 0;
  const int np = (n & 1) ? n + 1 : n;  // phantom player when n is odd
  constexpr bool inRegs = padded;   // the LDS images exist for n <= 136 < kJacobiRegLen only
  int sweeps = 0;
  // Columns whose norm (= |eigenvalue|) is below eps*n*max-norm belong to the numerical null space: the callers zero
  // those eigenvalues anyway, and rotating two such columns against each other only chases rounding noise (it used
  // to keep the solver busy for all 40 sweeps).  Pairs with at least one significant column are still rotated.
  double& nullTol2 = gJacobiShared.nullTol2;
  int& anyRotation = gJacobiShared.anyRotation;
  // (Skipping pairs whose columns did not change since they were last found orthogonal buys nothing: the sweeps are
  // dense, nearly every column still moves a little in every one of the 14 .. 19 sweeps.)
  for (int sweep = 0; sweep < 40; ++sweep) {
    __syncthreads();
    if (threadIdx.x == 0) { anyRotation = 0; gJacobiShared.anyLargeRotation = 0; nullTol2 = 0.0; }
    __syncthreads();
    {
      double mx = 0;
      for (int j = wave; j < n; j += nWaves) {
        double a2 = 0;
        for (int i = lane; i < n; i += 64) { const double x = G[(size_t)j * ld + i]; a2 += x * x; }
        a2 = waveSumM(a2);
        mx = fmax(mx, a2);
      }
      // non-negative doubles order like their bit patterns
      if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(&nullTol2), (unsigned long long)__double_as_longlong(mx));
    }
    __syncthreads();
    const double eps_n = 2.220446049250313e-16 * n;
    const double tol2 = nullTol2 * eps_n * eps_n;
    // the n/2 disjoint pairs of a round run side by side, one lane group each
    const int grp = threadIdx.x / LPG, gl = threadIdx.x % LPG, nGroups = blockDim.x / LPG;
    bool rotated = false, large = false;
    for (int round = 0; round < np - 1; ++round) {
      for (int k = grp; k < np / 2; k += nGroups) {
        int a, b;
        jacobiPair(np, round, k, a, b);
        if (a >= n || b >= n) continue;
        const int pI = a < b ? a : b, qI = a < b ? b : a;
        P gp = G + pI * ld + gl;
        P gq = G + qI * ld + gl;
        double al = 0, be = 0, ga = 0;
        double xs[kRegCols], ys[kRegCols];
        if (inRegs) {
          if (NU > 0) {
#pragma unroll
            for (int u = 0; u < (NU > 0 ? NU : 1); ++u) { xs[u] = gp[LPG * u]; ys[u] = gq[LPG * u]; }
#pragma unroll
            for (int u = 0; u < (NU > 0 ? NU : 1); ++u) { al += xs[u] * xs[u]; be += ys[u] * ys[u]; ga += xs[u] * ys[u]; }
          } else {
#pragma unroll
            for (int u = 0; u < kRegCols; ++u) {
              if (LPG * u >= n) break;
              const double x = gp[LPG * u], y = gq[LPG * u];
              xs[u] = x; ys[u] = y;
              al += x * x; be += y * y; ga += x * y;
            }
          }
        } else {
          for (int i = gl; i < n; i += LPG) { const double x = gp[i - gl], y = gq[i - gl]; al += x * x; be += y * y; ga += x * y; }
        }
        static_assert(LPG == 16, "one DPP row per column pair");
        al = rowSum16(al); be = rowSum16(be); ga = rowSum16(ga);
        if (ga * ga <= (kJacobiOrthTol * kJacobiOrthTol) * (al * be) || al == 0.0 || be == 0.0 || (al <= tol2 && be <= tol2))
          continue;
        rotated = true;
        large = large || ga * ga >= (kJacobiFinalCos * kJacobiFinalCos) * (al * be);
        double c, s;
        jacobiRotation(al, be, ga, c, s);
        if (inRegs) {
          if (NU > 0) {
#pragma unroll
            for (int u = 0; u < (NU > 0 ? NU : 1); ++u) {
              gp[LPG * u] = c * xs[u] - s * ys[u];
              gq[LPG * u] = s * xs[u] + c * ys[u];
            }
          } else {
#pragma unroll
            for (int u = 0; u < kRegCols; ++u) {
              if (LPG * u >= n) break;
              gp[LPG * u] = c * xs[u] - s * ys[u];
              gq[LPG * u] = s * xs[u] + c * ys[u];
            }
          }
        } else {
          for (int i = gl; i < n; i += LPG) {
            const double x = gp[i - gl], y = gq[i - gl];
            gp[i - gl] = c * x - s * y; gq[i - gl] = s * x + c * y;
          }
        }
        if (kHasQ) {
          P vp = Q + pI * ld;
          P vq = Q + qI * ld;
          for (int i = gl; i < n; i += LPG) {
            const double u = vp[i], w = vq[i];
            vp[i] = c * u - s * w; vq[i] = s * u + c * w;
          }
        }
      }
      if (padded) ldsBarrier(); else __syncthreads();   // padded = G and Q are LDS images
    }
    if (rotated) anyRotation = 1;   // racing stores of the same value
    if (large) gJacobiShared.anyLargeRotation = 1;
    if (threadIdx.x == 0) flag[1] = sweep + 1;
    sweeps = sweep + 1;
    __syncthreads();
    // Converged when nothing was rotated - or when every rotation of this sweep was by less than kJacobiFinalCos: the
    // pairs it left alone were orthogonal to kJacobiOrthTol when visited and have since moved by products of two such
    // angles at most (n of them: 1e-15), so the sweep that would only confirm it is not run.
    if (anyRotation == 0 || gJacobiShared.anyLargeRotation == 0) break;
  }
  __syncthreads();
  return sweeps;
}
constexpr int kJacobiLanes = 16;

// Same, with G and Q staged through LDS when the kernel was launched with 2*n*jacobiLd(n) doubles of dynamic shared
// memory (lds != nullptr): every round of the tournament is one LDS round trip instead of a global-memory one
// (12 sweeps x 50 rounds at n = 51: 1.9 ms -> 0.2 ms).
__device__ void jacobiEig(double* G, double* Q, int n, int* flag, double* lds) {
  if (!lds) { jacobiEigBlock<kJacobiLanes, double*, true>(G, Q, n, n, flag); return; }
  const int ld = jacobiLd(n);
  lds_double* sG = toLds(lds);
  lds_double* sQ = sG + n * ld;
  for (int idx = threadIdx.x; idx < n * ld; idx += blockDim.x) {
    const int i = idx / ld, j = idx - i * ld;
    sG[idx] = j < n ? G[i * n + j] : 0.0;
    sQ[idx] = j < n ? Q[i * n + j] : 0.0;
  }
  __syncthreads();
  jacobiEigBlock<kJacobiLanes, lds_double*, true>(sG, sQ, n, ld, flag);
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
    const int i = idx / n, j = idx - i * n;
    G[idx] = sG[i * ld + j];
    Q[idx] = sQ[i * ld + j];
  }
  __syncthreads();
}
constexpr size_t kJacobiLdsLimit = 158 * 1024 - sizeof(JacobiShared);   // 160 KB per workgroup less the static __shared__ scalars
static size_t jacobiLdsBytes(int n) {
  const size_t b = (size_t)2 * n * jacobiLd(n) * sizeof(double);
  return b <= kJacobiLdsLimit ? b : 0;
}
static size_t jacobiLdsBytesGOnly(int n) {
  const size_t b = (size_t)n * jacobiLd(n) * sizeof(double);
  return b <= kJacobiLdsLimit && n <= kJacobiRegLen ? b : 0;
}

// ---------------------------------------------------------------- M2: landmark part (:557-619)
// per landmark: p_b from diag(V), V' = V/(p_b p_b^T), (V')^+ by 3x3 eigendecomposition with tolerance
// eps*3*lmax, N = D_b^-1 U_e diag(sqrt(1/l)|0);  Mu = W_l N (m x 3) overwrites W_l; vb = N N^T bb_l.
__global__ void k_marg_lm_prepare(MargDev md, double* vb) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= md.Lm) return;
  const double* V = md.V + 9 * (size_t)l;
  double pb[3];
  for (int a = 0; a < 3; ++a) pb[a] = (V[a * 4] > 1.0e-9) ? sqrt(V[a * 4]) : 1.0e-3;
  double Vs[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) Vs[a * 3 + b] = V[a * 3 + b] / (pb[a] * pb[b]);
  double ev[3], Q[9];
  symEig3(Vs, ev, Q);
  const double mx = fmax(ev[0], fmax(ev[1], ev[2]));
  const double tol = 2.220446049250313e-16 * 3 * mx;
  double Nm[9];
  for (int j = 0; j < 3; ++j) {
    const double sc = (ev[j] > tol) ? sqrt(1.0 / ev[j]) : 0.0;
    for (int a = 0; a < 3; ++a) Nm[a * 3 + j] = Q[a * 3 + j] * sc / pb[a];
  }
  // vb = N N^T bb
  const double* bb = md.bb + 3 * l;
  double t[3];
  for (int j = 0; j < 3; ++j) t[j] = Nm[j] * bb[0] + Nm[3 + j] * bb[1] + Nm[6 + j] * bb[2];
  for (int a = 0; a < 3; ++a) vb[3 * l + a] = Nm[a * 3] * t[0] + Nm[a * 3 + 1] * t[1] + Nm[a * 3 + 2] * t[2];
  // store N in V (no longer needed)
  double* Vw = md.V + 9 * (size_t)l;
  for (int k = 0; k < 9; ++k) Vw[k] = Nm[k];
}
// ba -= W vb (uses the original W) ; then W_l <- W_l N_l.  One workgroup per row of W, the landmarks dealt over its threads,
// the row's sum reduced in a fixed order (deterministic); one THREAD per row walking all landmarks took 130-220 us.
__global__ __launch_bounds__(256) void k_marg_lm_apply(MargDev md, const double* vb) {
  __shared__ double red[256];
  const int i = blockIdx.x, t = threadIdx.x;
  double* Wr = md.W + (size_t)i * 3 * md.Lm;
  double s = 0;
  for (int l = t; l < md.Lm; l += 256) {
    const double w0 = Wr[3 * l], w1 = Wr[3 * l + 1], w2 = Wr[3 * l + 2];
    s += w0 * vb[3 * l] + w1 * vb[3 * l + 1] + w2 * vb[3 * l + 2];
    const double* Nm = md.V + 9 * (size_t)l;
    Wr[3 * l] = w0 * Nm[0] + w1 * Nm[3] + w2 * Nm[6];
    Wr[3 * l + 1] = w0 * Nm[1] + w1 * Nm[4] + w2 * Nm[7];
    Wr[3 * l + 2] = w0 * Nm[2] + w1 * Nm[5] + w2 * Nm[8];
  }
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) md.ba[i] -= red[0];
}
// U -= Mu Mu^T: 16 x 16 tiles on v_mfma_f64_16x16x4 (A[i = l & 15][k = l >> 4], B[k][j = l & 15], C: column l & 15, row
// (l >> 4) + 4 reg), one wave per tile over all 3 Lm columns of Mu in order (deterministic).  One thread per entry streaming
// two rows of Mu from global memory took 100-180 us.
__global__ __launch_bounds__(64) void k_marg_lm_update(MargDev md) {
  const int tr = (md.m + 15) >> 4, ti = blockIdx.x / tr, tj = blockIdx.x - ti * tr, l = threadIdx.x;
  const int K = 3 * md.Lm;
  const int ra = ti * 16 + (l & 15), rb = tj * 16 + (l & 15);
  const double* pa = md.W + (size_t)min(ra, md.m - 1) * K;
  const double* pb = md.W + (size_t)min(rb, md.m - 1) * K;
  const bool okA = ra < md.m, okB = rb < md.m;
  typedef double d4m __attribute__((ext_vector_type(4)));
  d4m acc = {0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < K; kk += 4) {
    const int k = kk + (l >> 4);
    const double av = (okA && k < K) ? pa[k] : 0.0, bv = (okB && k < K) ? pb[k] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int r = ti * 16 + (l >> 4) + 4 * rg, c = tj * 16 + (l & 15);
    if (r < md.m && c < md.m) md.U[(size_t)r * md.m + c] -= acc[rg];
  }
}

static size_t margCholLdsBytes(int n) {
  const size_t nT = ((size_t)n + 15) / 16, NP = 16 * nT;
  return (NP * (NP + 1) + nT * 16 * kPanelLd + NP) * sizeof(double);
}
// ---------------------------------------------------------------- M2 dense part (:622-667) + M3 (:725-758)
// Single workgroup.  keep/marg index lists select rows of U (m x m).  Outputs the reduced Hk (nk x nk), bk.
struct DenseArgs {
  int m, nk, nm;
  const int* keep; const int* marg;
  const double* U; const double* ba;
  double *Hk, *bk;          // outputs
  double *Vm, *Qm, *tmp;    // scratch: nm x nm, nm x nm, nk x nm + 2 nm
  int* flag;
};
__global__ __launch_bounds__(1024) void k_marg_dense(DenseArgs a, int useLds, int prodLds, int tileChol) {
  extern __shared__ double jacobiLds[];
  const int t = threadIdx.x, nt = blockDim.x, nm = a.nm, nk = a.nk, m = a.m;
  double* pm = a.tmp;                 // nm
  double* tv = a.tmp + nm;            // nm
  double* Mu = a.tmp + 2 * nm;        // nk x nm
  for (int i = t; i < nm; i += nt) {
    const double hd = a.U[(size_t)a.marg[i] * m + a.marg[i]];
    pm[i] = (hd > 1.0e-9) ? sqrt(hd) : 1.0e-3;
  }
  __syncthreads();
  // V' = 0.5 (V + V^T) scaled ; Q = I
  for (int idx = t; idx < nm * nm; idx += nt) {
    const int i = idx / nm, j = idx % nm;
    const double v = 0.5 * (a.U[(size_t)a.marg[i] * m + a.marg[j]] + a.U[(size_t)a.marg[j] * m + a.marg[i]]);
    a.Vm[idx] = v / (pm[i] * pm[j]);
    a.Qm[idx] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  // The pseudo-inverse of V' drops the eigenvalues <= eps nm lambda_max (MarginalizationError.hpp:48-220).  The marginalised
  // pose / speed-bias block is well conditioned in every frame of the sliding windows (eigenvalues of V' 0.5 .. 1.7, unit
  // diagonal), and then NOTHING is dropped and (V')^+ = (V')^-1 = R^-1 R^-T: a Cholesky factor is a proof -- lambda_min >=
  // 1 / trace(V'^-1) = 1 / |R^-1|_F^2 and lambda_max <= trace(V') = nm, so 1 / |R^-1|_F^2 > eps nm^2 certifies the rank rule's
  // outcome without an eigenvalue -- and N = D^-1 R^-1 replaces D^-1 Q sqrt(1 / lambda) (N N^T is the same matrix).  The
  // eigen-solve below remains for blocks the certificate does not cover (a non-positive pivot, a gauge freedom in the block).
  __shared__ int sDirect;
  if (t == 0) sDirect = 0;
  __syncthreads();
  if (tileChol) {   // blocked on 16 x 16 MFMA tiles (tileCholFactor / tileCholInverse above; the unblocked first version took ~40 us at nm = 27)
    const int nTm = (nm + 15) >> 4, NPm = 16 * nTm, ldm = NPm + 1;
    lds_double* A = toLds(jacobiLds);
    double* DgGen = jacobiLds + (size_t)NPm * ldm;
    double* dinvGen = DgGen + nTm * 16 * kPanelLd;
    const lds_double* Dg = toLds(DgGen);
    const lds_double* dinv = toLds(dinvGen);
    __shared__ int sFailD;
    __shared__ double sFroD[16];
    if (t == 0) sFailD = 0;
    for (int idx = t; idx < NPm * ldm; idx += nt) {
      const int r = idx / ldm, c = idx - r * ldm;
      A[idx] = (r < nm && c < nm) ? a.Vm[(size_t)r * nm + c] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    tileCholFactor<16>(A, ldm, nTm, DgGen, dinvGen, &sFailD);
    if (!sFailD) {   // (uniform)
      const double fro = waveSumM(tileCholInverse<16>(A, ldm, nTm, nm, Dg, dinv));
      if ((t & 63) == 0) sFroD[t >> 6] = fro;
      __syncthreads();
      if (t == 0) {
        double f2 = 0;
        for (int w = 0; w < 16; ++w) f2 += sFroD[w];
        sDirect = (f2 > 0.0 && f2 < 1.0e300 && 1.0 / f2 > 2.220446049250313e-16 * (double)nm * (double)nm) ? 1 : 0;
      }
      __syncthreads();
      if (sDirect) {   // N^T row j = column j of N = D^-1 R^-1 e_j, R^-1 = L^-T: N^T[j][i] = L^-1[j][i] / pm[i], i <= j
        for (int idx = t; idx < nm * nm; idx += nt) {
          const int j = idx / nm, i = idx - j * nm;
          double y = 0.0;
          if (i <= j) y = ((i >> 4) < (j >> 4)) ? (double)A[i * ldm + j] : tileLinvAt(Dg, dinv, j >> 4, j & 15, i & 15);
          a.Qm[idx] = y / pm[i];
        }
        if (t == 0) a.flag[5] = 1;
      }
    }
    __syncthreads();
  }
  if (!sDirect) {
  jacobiEig(a.Vm, a.Qm, nm, a.flag, useLds ? jacobiLds : nullptr);
  // eigenvalues, tolerance, N = D^-1 Q diag(sqrt(1/l)|0)  (column j of N = eigenvector j scaled)
  __shared__ double smax;
  for (int j = t; j < nm; j += nt) {
    double s = 0;
    for (int i = 0; i < nm; ++i) s += a.Qm[(size_t)j * nm + i] * a.Vm[(size_t)j * nm + i];
    tv[j] = s;
  }
  __syncthreads();
  if (t == 0) { double mx = tv[0]; for (int j = 1; j < nm; ++j) mx = fmax(mx, tv[j]); smax = mx; }
  __syncthreads();
  const double tol = 2.220446049250313e-16 * nm * smax;
  // overwrite Q rows: N^T row j = sqrt(1/l_j) * q_j / pm
  for (int idx = t; idx < nm * nm; idx += nt) {
    const int j = idx / nm, i = idx % nm;
    const double sc = (tv[j] > tol) ? sqrt(1.0 / tv[j]) : 0.0;
    a.Qm[idx] = a.Qm[idx] * sc / pm[i];
  }
  if (t == 0) a.flag[5] = 0;
  }
  __syncthreads();
  // Mu = W N  (W = U[keep, marg]) : Mu[r][j] = sum_i W[r][i] N[i][j] = sum_i W[r][i] Qm[j][i]; then Hk = U[keep, keep] - Mu Mu^T,
  // bk = ba[keep] - Mu (N^T b_m).  With W, N^T and Mu staged in LDS when the launch has room for them (prodLds: every operand
  // used to come from global memory once per multiply-add -- a third of this kernel's 220 us at nk = 117, nm = 27).
  if (prodLds) {
    __syncthreads();
    double* sW = jacobiLds;                    // nk x nm
    double* sN = sW + (size_t)nk * nm;         // nm x nm (row j = column j of N)
    double* sMu = sN + (size_t)nm * nm;        // nk x nm
    for (int idx = t; idx < nk * nm; idx += nt) { const int r = idx / nm, i = idx - r * nm; sW[idx] = a.U[(size_t)a.keep[r] * m + a.marg[i]]; }
    for (int idx = t; idx < nm * nm; idx += nt) sN[idx] = a.Qm[idx];
    __syncthreads();
    for (int idx = t; idx < nk * nm; idx += nt) {
      const int r = idx / nm, j = idx - r * nm;
      double s = 0;
      for (int i = 0; i < nm; ++i) s += sW[r * nm + i] * sN[j * nm + i];
      sMu[idx] = s;
      Mu[idx] = s;
    }
    for (int j = t; j < nm; j += nt) {
      double s = 0;
      for (int i = 0; i < nm; ++i) s += sN[j * nm + i] * a.ba[a.marg[i]];
      pm[j] = s;  // pm reused
    }
    __syncthreads();
    for (int idx = t; idx < nk * nk; idx += nt) {
      const int r = idx / nk, c = idx - r * nk;
      double s = 0;
      for (int j = 0; j < nm; ++j) s += sMu[r * nm + j] * sMu[c * nm + j];
      a.Hk[idx] = a.U[(size_t)a.keep[r] * m + a.keep[c]] - s;
    }
    for (int r = t; r < nk; r += nt) {
      double s = 0;
      for (int j = 0; j < nm; ++j) s += sMu[r * nm + j] * pm[j];
      a.bk[r] = a.ba[a.keep[r]] - s;
    }
    return;
  }
  for (int idx = t; idx < nk * nm; idx += nt) {
    const int r = idx / nm, j = idx % nm;
    double s = 0;
    for (int i = 0; i < nm; ++i) s += a.U[(size_t)a.keep[r] * m + a.marg[i]] * a.Qm[(size_t)j * nm + i];
    Mu[idx] = s;
  }
  // tv2 = N^T b_m
  __syncthreads();
  for (int j = t; j < nm; j += nt) {
    double s = 0;
    for (int i = 0; i < nm; ++i) s += a.Qm[(size_t)j * nm + i] * a.ba[a.marg[i]];
    pm[j] = s;  // pm reused
  }
  __syncthreads();
  for (int idx = t; idx < nk * nk; idx += nt) {
    const int r = idx / nk, c = idx % nk;
    double s = 0;
    for (int j = 0; j < nm; ++j) s += Mu[(size_t)r * nm + j] * Mu[(size_t)c * nm + j];
    a.Hk[idx] = a.U[(size_t)a.keep[r] * m + a.keep[c]] - s;
  }
  for (int r = t; r < nk; r += nt) {
    double s = 0;
    for (int j = 0; j < nm; ++j) s += Mu[(size_t)r * nm + j] * pm[j];
    a.bk[r] = a.ba[a.keep[r]] - s;
  }
}

// M3: H = U S U^T of the Jacobi-preconditioned H; J = (p U sqrt(S))^T, e0 = -(sqrt(S)^+ U^T p^-1) b0;
// plus the H-space form used by the solver: Ht = J^T J, bp = J^T e0, c0 = e0.e0 (out[0]).
struct FinalArgs {
  int n;
  const double* H; const double* b0;
  double *G, *Q, *J, *e0, *Ht, *bp, *scal, *tmp;
  int* flag;
};
__device__ __forceinline__ double margScale(double hd) { return (hd > 1.0e-9) ? sqrt(hd) : 1.0e-3; }

// ---- Cholesky-preconditioned solve (mode 4, Veselic / Hari): A + delta I = R^T R, then one-sided Jacobi on the rows of R
// (the columns of L = R^T).  R V = Sigma U^T with U the eigenvectors of A and sigma_j^2 = lambda_j + delta, so the
// eigenvectors come out of the rotated rows themselves - no Q, no rotation log, no second workgroup - and because the
// rows of R carry the square roots of the spectrum (condition 1e6 instead of 1e15 for the columns of A) the tournament
// converges in 11-12 sweeps instead of 18-19.  delta = 64 n eps (A has a unit diagonal) keeps the factorisation away
// from the rounding-negative eigenvalues of the numerically semidefinite A and is removed again from sigma^2 exactly
// (sigma^2 >= delta is computed to relative precision); an eigen-direction is reliable to eps |R| / sigma_j, i.e. 1e-10 for
// the null space (which is dropped) and better than 1e-12 for everything above 1e-8.  Returns false (uniformly) if a
// pivot is not positive; the caller then runs the two-phase solve.
// P = lds_double* (the image in LDS, n <= 136) or double* (the image in a.G: larger priors; the same arithmetic at
// global-memory latency, still 2-3x faster than mode 0 for its fewer sweeps and the missing Q).
template <class P>
__device__ bool margFinalCholesky(const FinalArgs& a, P lds, int ld) {
  constexpr bool inLds = std::is_same<P, lds_double*>::value;
  const int t = threadIdx.x, nt = blockDim.x, n = a.n;
  const int wave = t >> 6, lane = t & 63, nWaves = nt >> 6;
  const int grp = t >> 4, gl = t & 15, nGroups = nt >> 4;
  double* p = a.tmp;            // n
  double* ev = a.tmp + n;       // n
  double* rsig = a.tmp + 2 * n; // n: 1 / sigma_j
  const double delta = 64.0 * n * 2.220446049250313e-16;
  const long long tStart = wall_clock64();
  for (int i = t; i < n; i += nt) p[i] = margScale(a.H[(size_t)i * n + i]);
  __syncthreads();
  // upper triangle of A + delta I, the rest (lower triangle and the padding) zero
  for (int idx = t; idx < n * ld; idx += nt) {
    const int r = idx / ld, c = idx - r * ld;
    double v = 0.0;
    if (c >= r && c < n)
      v = 0.5 * (a.H[(size_t)r * n + c] + a.H[(size_t)c * n + r]) / (p[r] * p[c]) + (r == c ? delta : 0.0);
    lds[idx] = v;
  }
  __syncthreads();
  // right-looking Cholesky on the upper triangle; row k is left unscaled (A[j][i] -= A[k][j] A[k][i] / A[k][k]) so that
  // a step is one pass and one barrier, and scaled afterwards
  for (int k = 0; k < n - 1; ++k) {
    const double pivot = lds[k * ld + k];
    if (!(pivot > 0.0)) return false;
    const double rp = 1.0 / pivot;
    for (int j = k + 1 + wave; j < n; j += nWaves) {
      const double f = lds[k * ld + j] * rp;
      for (int i = j + lane; i < n; i += 64) lds[j * ld + i] -= f * lds[k * ld + i];
    }
    if (inLds) ldsBarrier(); else __syncthreads();
  }
  if (!(lds[(n - 1) * ld + (n - 1)] > 0.0)) return false;
  for (int k = wave; k < n; k += nWaves) {
    const double rs = rsqrt(lds[k * ld + k]);
    for (int i = k + lane; i < n; i += 64) lds[k * ld + i] *= rs;
  }
  __syncthreads();
  const long long tPrep = wall_clock64(), cPrep = clock64();
  if (inLds) {
    switch ((n + kJacobiLanes - 1) / kJacobiLanes) {   // slices per column: <= 9 for the LDS image (n <= 136)
      case 1: jacobiEigBlock<kJacobiLanes, P, false, 1>(lds, (P) nullptr, n, ld, a.flag); break;
      case 2: jacobiEigBlock<kJacobiLanes, P, false, 2>(lds, (P) nullptr, n, ld, a.flag); break;
      case 3: jacobiEigBlock<kJacobiLanes, P, false, 3>(lds, (P) nullptr, n, ld, a.flag); break;
      case 4: jacobiEigBlock<kJacobiLanes, P, false, 4>(lds, (P) nullptr, n, ld, a.flag); break;
      case 5: jacobiEigBlock<kJacobiLanes, P, false, 5>(lds, (P) nullptr, n, ld, a.flag); break;
      case 6: jacobiEigBlock<kJacobiLanes, P, false, 6>(lds, (P) nullptr, n, ld, a.flag); break;
      case 7: jacobiEigBlock<kJacobiLanes, P, false, 7>(lds, (P) nullptr, n, ld, a.flag); break;
      case 8: jacobiEigBlock<kJacobiLanes, P, false, 8>(lds, (P) nullptr, n, ld, a.flag); break;
      default: jacobiEigBlock<kJacobiLanes, P, false, 9>(lds, (P) nullptr, n, ld, a.flag); break;
    }
  } else {
    jacobiEigBlock<kJacobiLanes, P, false>(lds, (P) nullptr, n, ld, a.flag);
  }
  const long long tEig = wall_clock64(), cEig = clock64();
  // row j = sigma_j u_j
  for (int j = grp; j < n; j += nGroups) {
    double s = 0;
    for (int i = gl; i < n; i += 16) { const double x = lds[j * ld + i]; s += x * x; }
    s = rowSum16(s);
    if (gl == 0) { ev[j] = s - delta; rsig[j] = rsqrt(s); }
  }
  __syncthreads();
  if (t < 64) {
    double mx = -1.0e300, mn = 1.0e300;
    for (int j = t; j < n; j += 64) { mx = fmax(mx, ev[j]); mn = fmin(mn, ev[j]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); mn = fmin(mn, __shfl_xor(mn, o, 64)); }
    const double tl = 2.220446049250313e-16 * n * mx;
    int c = 0;
    for (int j = t; j < n; j += 64) c += ev[j] <= tl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (t == 0) { gJacobiShared.nullTol2 = mx; a.flag[2] = c; a.scal[1] = mn; a.scal[2] = mx; }
  }
  __syncthreads();
  const double tol = 2.220446049250313e-16 * n * gJacobiShared.nullTol2;
  for (int i = grp; i < n; i += nGroups) {
    double e = 0;
    for (int j = gl; j < n; j += 16) e += lds[i * ld + j] * (a.b0[j] / p[j]);
    e = rowSum16(e);
    if (gl == 0) a.e0[i] = ev[i] > tol ? -sqrt(1.0 / ev[i]) * (e * rsig[i]) : 0.0;
  }
  __syncthreads();
  // J = (p U sqrt(S))^T in place of the rows
  for (int idx = t; idx < n * n; idx += nt) {
    const int i = idx / n, j = idx - i * n;
    const double s = ev[i] > tol ? sqrt(ev[i]) * rsig[i] : 0.0;
    const double v = p[j] * lds[i * ld + j] * s;
    a.J[idx] = v;
    lds[i * ld + j] = v;
  }
  __syncthreads();
  for (int idx = t; idx < n * n; idx += nt) {
    const int i = idx / n, j = idx - i * n;
    double s = 0;
    for (int k = 0; k < n; ++k) s += lds[k * ld + i] * lds[k * ld + j];
    a.Ht[idx] = s;
  }
  for (int i = t; i < n; i += nt) {
    double s = 0;
    for (int k = 0; k < n; ++k) s += lds[k * ld + i] * a.e0[k];
    a.bp[i] = s;
  }
  if (t < 64) {
    double c = 0;
    for (int k = t; k < n; k += 64) c += a.e0[k] * a.e0[k];
    c = waveSumM(c);
    if (t == 0) {
      a.scal[0] = c;
      a.scal[3] = (double)(tPrep - tStart); a.scal[4] = (double)(tEig - tPrep); a.scal[5] = (double)(wall_clock64() - tEig);
      a.scal[6] = (double)(cEig - cPrep);
      a.scal[7] = (double)n;
      a.flag[3] = inLds ? -4 : -6;   // marks the mode in the SVIN_MARG_TIMING line
    }
  }
  return true;
}

// ---- M3 without an eigenvalue, when the prior has full numerical rank (round 5).  The reference drops the eigen-directions of the
// Jacobi-scaled H = p A p with lambda <= eps n lambda_max (MarginalizationError.cpp:739-742) and takes J = (p U sqrt(S))^T,
// e0 = -(sqrt(S)^+ U^T p^-1) b0.  The optimiser sees the prior through J^T J, J^T e0 and e0.e0 only, which ANY square root of
// the kept part delivers -- and when nothing is dropped, the Cholesky factor is one: A = R^T R, J = R p, e0 = -R^-T p^-1 b0,
// J^T J = H, J^T e0 = -b0.  "Nothing is dropped" is certified, not assumed: lambda_min >= 1 / trace(A^-1) = 1 / |R^-1|_F^2 and
// lambda_max <= trace(A) = n (unit diagonal), so 1 / |R^-1|_F^2 > eps n^2 implies every eigenvalue is above the reference's
// threshold.  Steady-state priors of both sliding windows pass (smallest eigenvalue 1e-8, threshold 9e-14); a prior with a
// gauge freedom, a non-positive pivot or a failed certificate writes flag[4] = 0 and k_marg_final_dc, enqueued behind, does
// the eigen-solve.  One workgroup, everything in 16 x 16 tiles on v_mfma_f64_16x16x4 (the unblocked first version -- one barrier
// per column, R^-1 column by column on 8 lanes -- took 107 + 154 us at n = 117):
//   * A = L L^T blocked right-looking over the lower tiles of the LDS image (identity padding to a multiple of 16): per block
//     column the diagonal tile by wave 0 out of registers (cholDiag16Acc of the dense solvers, tile16.hpp: L and L^-1 of the tile
//     into a 16 x 17 scratch tile), the panel tiles X = A_IK L_KK^-T one per wave, the trailing tiles A_IJ -= X_I X_J^T dealt over
//     the 16 waves; three barriers per block column, eight block columns at n = 117;
//   * Y = L^-1 one block column per wave without a barrier: Y_JJ is the scratch tile's inverse, Y_IJ = -L_II^-1 sum_K L_IK Y_KJ
//     with the running sum in the accumulator layout, which IS the B operand of the product with L_II^-1; Y_IJ^T goes to the upper
//     tile (J, I), which the factorisation never reads, so element Y[i][j] sits at image[j][i];
//   * |Y|_F^2 (the certificate), e0 = -Y (b0 / p) (8 lanes per row), J = L^T p, Ht = the symmetrised H, bp = -b0, c0 = e0.e0.
__global__ __launch_bounds__(1024) void k_marg_final_chol(FinalArgs a) {
  extern __shared__ double jacobiLds[];
  const int t = threadIdx.x, nt = 1024, n = a.n, nT = (n + 15) >> 4, NP = 16 * nT, ld = NP + 1;
  const int wave = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
  lds_double* A = toLds(jacobiLds);                       // NP x ld image
  double* DgGen = jacobiLds + (size_t)NP * ld;            // nT scratch tiles (16 x kPanelLd): L (lower) | L^-1 transposed (strict upper)
  double* dinvGen = DgGen + nT * 16 * kPanelLd;           // 1 / L_ii
  lds_double* Dg = toLds(DgGen);
  lds_double* dinv = toLds(dinvGen);
  double* p = a.tmp;            // n
  double* bt = a.tmp + n;       // n: b0 / p
  __shared__ double sFro[16];
  __shared__ int sOk, sFail;
  const long long tStart = wall_clock64();
  if (t == 0) { a.flag[4] = 0; sFail = 0; }
  for (int i = t; i < n; i += nt) { const double pi = margScale(a.H[(size_t)i * n + i]); p[i] = pi; bt[i] = a.b0[i] / pi; }
  __syncthreads();
  for (int idx = t; idx < NP * ld; idx += nt) {
    const int r = idx / ld, cc = idx - r * ld;
    A[idx] = (r < n && cc < n) ? 0.5 * (a.H[(size_t)r * n + cc] + a.H[(size_t)cc * n + r]) / (p[r] * p[cc]) : ((r == cc) ? 1.0 : 0.0);
  }
  __syncthreads();
  tileCholFactor<16>(A, ld, nT, DgGen, dinvGen, &sFail);
  if (sFail) return;   // (uniform) a pivot was not positive: the eigen-solve behind this launch decides
  const long long tChol = wall_clock64();
  {
    const double fro = waveSumM(tileCholInverse<16>(A, ld, nT, n, Dg, dinv));
    if (lane == 0) sFro[wave] = fro;
  }
  __syncthreads();
  // e0 = -Y (b0 / p): 8 lanes per row
  {
    const int i = t >> 3, sub = t & 7;
    double e = 0.0;
    if (i < n) {
      const int bi = i >> 4;
      for (int j = sub; j <= i; j += 8) {
        const double yv = ((j >> 4) < bi) ? (double)A[j * ld + i] : tileLinvAt(Dg, dinv, bi, i & 15, j & 15);
        e = __builtin_fma(yv, bt[j], e);
      }
    }
    e = symeig::sum8(e);
    if (i < n && sub == 0) a.e0[i] = -e;
  }
  if (t == 0) {
    double f2 = 0;
    for (int w = 0; w < 16; ++w) f2 += sFro[w];
    sOk = (f2 > 0.0 && f2 < 1.0e300 && 1.0 / f2 > 2.220446049250313e-16 * (double)n * (double)n) ? 1 : 0;
    a.scal[1] = f2 > 0.0 ? 1.0 / f2 : 0.0;   // (a lower bound of the smallest eigenvalue, an upper bound of the largest)
    a.scal[2] = (double)n;
  }
  __syncthreads();
  if (!sOk) return;
  // J = L^T p (row i = column i of L, columns scaled), Ht = the symmetrised H itself (= J^T J), bp = J^T e0 = -b0, c0 = e0.e0
  for (int idx = t; idx < n * n; idx += nt) {
    const int i = idx / n, j = idx - i * n;
    double lji = 0.0;   // L[j][i], j >= i
    if (j >= i) lji = ((j >> 4) > (i >> 4)) ? (double)A[j * ld + i] : (double)Dg[(j >> 4) * 16 * kPanelLd + (j & 15) * kPanelLd + (i & 15)];
    a.J[idx] = lji * p[j];
    a.Ht[idx] = 0.5 * (a.H[idx] + a.H[(size_t)j * n + i]);
  }
  for (int i = t; i < n; i += nt) a.bp[i] = -a.b0[i];
  __syncthreads();
  if (t < 64) {
    double cs = 0;
    for (int k = t; k < n; k += 64) cs += a.e0[k] * a.e0[k];
    cs = waveSumM(cs);
    if (t == 0) {
      a.scal[0] = cs;
      a.scal[3] = 0.0; a.scal[4] = (double)(tChol - tStart); a.scal[5] = (double)(wall_clock64() - tChol);
      a.scal[6] = 0.0;
      a.scal[7] = (double)n;
      a.flag[1] = 0; a.flag[2] = 0;
      a.flag[3] = -8;   // marks the mode in the SVIN_MARG_TIMING line
      __threadfence();
      a.flag[4] = 1;
    }
  }
}

// ---- M3 by a direct eigen-solve (round 5): tridiagonalisation + divide and conquer (symeig.hpp) instead of ~1 400 dependent
// Jacobi rounds.  H = p A p (A with a unit diagonal), A = U S U^T; J = (p U sqrt(S))^T, e0 = -(sqrt(S)^+ U^T p^-1) b0 with the
// eigenvalues <= eps n lambda_max dropped (MarginalizationError.cpp:739-742), and the H-space form the solver reads: Ht = J^T J,
// bp = J^T e0, c0 = e0.e0.  Writes flag[4] = 1 when it has produced the prior; k_marg_final, enqueued right behind it with
// `skipIfDone`, then returns at once -- otherwise (a non-finite result) it runs the Jacobi solve as before.
__global__ __launch_bounds__(1024) void k_marg_final_dc(FinalArgs a, int skipIfDone) {
  extern __shared__ double jacobiLds[];
  if (skipIfDone && a.flag[4] == 1) return;   // k_marg_final_chol, enqueued ahead of this launch, has produced the prior
  lds_double* X = toLds(jacobiLds);
  const int t = threadIdx.x, n = a.n, ld = n | 1;
  double* p = a.tmp;            // n
  double* ev = a.tmp + n;       // n
  const long long tStart = wall_clock64();
  for (int i = t; i < n; i += 1024) p[i] = margScale(a.H[(size_t)i * n + i]);
  __syncthreads();
  for (int idx = t; idx < n * ld; idx += 1024) {
    const int r = idx / ld, c = idx - r * ld;
    X[idx] = (c < n) ? 0.5 * (a.H[(size_t)r * n + c] + a.H[(size_t)c * n + r]) / (p[r] * p[c]) : 0.0;
  }
  __syncthreads();
  const long long tPrep = wall_clock64();
  const bool ok = symeig::solve(X, n, ld, a.G);
  const long long tEig = wall_clock64();
  if (!ok) {
    if (t == 0) a.flag[4] = 0;
    return;
  }
  // X[i * ld + j] = component i of eigenvector j, symeig::gS.d[j] = eigenvalue j (ascending).  Per-vector scalars once, in LDS
  // (the solver's per-phase arrays are free again): sq = sqrt(l) or 0 (dropped), lamp = l or 0, bt = b0 / p
  double* sq = symeig::gS.u.dc.z;
  double* lamp = symeig::gS.u.dc.dcur;
  double* bt = symeig::gS.u.dc.ztil;
  double* e0s = symeig::gS.u.dc.cd;
  const double mx = symeig::gS.d[n - 1], mn = symeig::gS.d[0];
  const double tol = 2.220446049250313e-16 * n * mx;
  if (t < n) {
    const double l = symeig::gS.d[t];
    ev[t] = l;
    lamp[t] = l > tol ? l : 0.0;
    sq[t] = l > tol ? sqrt(l) : 0.0;
    bt[t] = a.b0[t] / p[t];
  }
  if (t < 64) {
    int c = 0;
    for (int j = t; j < n; j += 64) c += symeig::gS.d[j] <= tol;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (t == 0) { a.flag[2] = c; a.flag[1] = 0; a.scal[1] = mn; a.scal[2] = mx; }
  }
  symeig::ldsBarrier();
  // e0_j = -(1 / sqrt(l_j)) u_j . (b0 / p): 8 lanes per eigenvector
  {
    const int j = t >> 3, sub = t & 7;
    double s = 0;
    if (j < n)
      for (int i = sub; i < n; i += 8) s += X[i * ld + j] * bt[i];
    s = symeig::sum8(s);
    if (j < n && sub == 0) { const double e = sq[j] > 0.0 ? -s / sq[j] : 0.0; a.e0[j] = e; e0s[j] = e; }
  }
  // J = (p U sqrt(S))^T: row j = eigen-direction j
  for (int idx = t; idx < n * n; idx += 1024) {
    const int j = idx / n, i = idx - j * n;
    a.J[idx] = p[i] * X[i * ld + j] * sq[j];
  }
  symeig::ldsBarrier();
  // Ht = J^T J = p (U S U^T) p: 16 x 16 tiles on v_mfma_f64_16x16x4, both operands rows of X
  {
    const int wave = t >> 6, l = t & 63, tr = (n + 15) >> 4;
    for (int id = wave; id < tr * tr; id += 16) {
      const int ra = (id / tr) * 16 + (l & 15), cb = (id % tr) * 16 + (l & 15);
      symeig::d4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int kk = 0; kk < n; kk += 4) {
        const int j = kk + (l >> 4);
        const double av = (ra < n && j < n) ? (double)X[ra * ld + j] : 0.0;
        const double bv = (cb < n && j < n) ? (double)X[cb * ld + j] * lamp[j] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = (id / tr) * 16 + (l >> 4) + 4 * rg;
        if (r < n && cb < n) a.Ht[(size_t)r * n + cb] = acc[rg] * p[r] * p[cb];
      }
    }
  }
  // bp = J^T e0
  {
    const int i = t >> 3, sub = t & 7;
    double s = 0;
    if (i < n)
      for (int j = sub; j < n; j += 8) s += X[i * ld + j] * (sq[j] * e0s[j]);
    s = symeig::sum8(s);
    if (i < n && sub == 0) a.bp[i] = p[i] * s;
  }
  if (t < 64) {
    double c = 0;
    for (int k = t; k < n; k += 64) c += e0s[k] * e0s[k];
    c = waveSumM(c);
    if (t == 0) {
      a.scal[0] = c;
      a.scal[3] = (double)(tPrep - tStart); a.scal[4] = (double)(tEig - tPrep); a.scal[5] = (double)(wall_clock64() - tEig);
      a.scal[6] = 0.0;
      a.scal[7] = (double)n;
      a.flag[3] = -7;   // marks the mode in the SVIN_MARG_TIMING line
      __threadfence();
      a.flag[4] = 1;
    }
  }
}
constexpr int kSymEigMaxN = symeig::kMaxN;
static size_t symEigLdsBytes(int n) { return (size_t)n * (n | 1) * sizeof(double); }

// inspection hook (svin_ba_debug_sym_eig): the solver on an arbitrary symmetric matrix
__global__ __launch_bounds__(1024) void k_sym_eig_debug(int n, const double* A, double* lam, double* Xout, double* scratch, int* okOut) {
  extern __shared__ double jacobiLds[];
  lds_double* X = toLds(jacobiLds);
  const int t = threadIdx.x, ld = n | 1;
  for (int idx = t; idx < n * ld; idx += 1024) {
    const int r = idx / ld, c = idx - r * ld;
    X[idx] = (c < n) ? 0.5 * (A[(size_t)r * n + c] + A[(size_t)c * n + r]) : 0.0;
  }
  __syncthreads();
  const bool ok = symeig::solve(X, n, ld, scratch);
  if (t == 0) *okOut = ok ? 1 : 0;
  if (t < n) lam[t] = symeig::gS.d[t];
#ifdef SVIN_SYMEIG_TIMING
  if (t < 80) scratch[t] = t < symeig::gS.nstamp ? (double)(symeig::gS.stamp[t] - symeig::gS.stamp[0]) : -1.0;
#endif
  for (int idx = t; idx < n * n; idx += 1024) { const int i = idx / n, j = idx - i * n; Xout[idx] = X[i * ld + j]; }
}
int debugSymEig(int n, const double* A, double* lam, double* X, double* deviceMs) {
  if (n < 1 || n > kSymEigMaxN) return 0;
  // (a test hook that bench and tests call repeatedly: buffers and events are released on every path, also when a HIP call throws)
  struct Scratch {
    double* dA = nullptr; int* dOk = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (dA) (void)hipFree(dA);
      if (dOk) (void)hipFree(dOk);
    }
  } sc;
  const size_t n2 = (size_t)n * n;
  HIP_OK(hipMalloc(&sc.dA, sizeof(double) * (3 * n2 + n)));
  double *dA = sc.dA, *dX = dA + n2, *dS = dX + n2, *dLam = dS + n2;
  HIP_OK(hipMalloc(&sc.dOk, sizeof(int)));
  int* dOk = sc.dOk;
  HIP_OK(hipMemcpy(dA, A, sizeof(double) * n2, hipMemcpyHostToDevice));
  const size_t lds = symEigLdsBytes(n);
  ensureDynamicLds((const void*)k_sym_eig_debug, lds);
  HIP_OK(hipEventCreate(&sc.e0)); HIP_OK(hipEventCreate(&sc.e1));
  const hipEvent_t e0 = sc.e0, e1 = sc.e1;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {   // (the first launch pays the code upload)
    HIP_OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_sym_eig_debug, dim3(1), dim3(1024), lds, 0, n, dA, dLam, dX, dS, dOk);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  int ok = 0;
  HIP_OK(hipMemcpy(&ok, dOk, sizeof(int), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(lam, dLam, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(X, dX, sizeof(double) * n2, hipMemcpyDeviceToHost));
#ifdef SVIN_SYMEIG_TIMING
  if (n2 >= 80) {   // stage stamps of the last launch (100 MHz ticks since the first), one line
    double st[80];
    HIP_OK(hipMemcpy(st, dS, sizeof(st), hipMemcpyDeviceToHost));
    std::printf("[symeig] n %d stamps(us):", n);
    for (int i = 0; i < 80 && st[i] >= 0; ++i) std::printf(" %.1f", st[i] / 100.0);
    std::printf("\n");
  }
#endif
  if (deviceMs) *deviceMs = best;
  return ok ? 1 : -1;
}

__global__ __launch_bounds__(1024) void k_marg_final(FinalArgs a, int useLds, int fallbackLds, int skipIfDone) {
  extern __shared__ double jacobiLds[];
  if (skipIfDone && a.flag[4] == 1) return;   // k_marg_final_dc, enqueued ahead of this launch, has produced the prior
  // modes: 4 = Cholesky-preconditioned Jacobi, image in LDS; 6 = the same with the image in global memory (priors too
  // large for LDS); 1 / 0 = the fall-back, one-sided Jacobi on A itself with G and Q in LDS / in global memory
  // (also what runs when a pivot of the factorisation is not positive: `fallbackLds` says whether G and Q both fit)
  if (useLds == 6) {
    if (margFinalCholesky<double*>(a, a.G, a.n)) return;
    __syncthreads();
    useLds = 0;
  } else if (useLds == 4) {
    if (margFinalCholesky<lds_double*>(a, toLds(jacobiLds), jacobiLd(a.n))) return;
    __syncthreads();
    useLds = fallbackLds ? 1 : 0;
  }
  const int t = threadIdx.x, nt = blockDim.x, n = a.n;
  double* p = a.tmp;        // n
  double* ev = a.tmp + n;   // n
  const long long tStart = wall_clock64();
  for (int i = t; i < n; i += nt) {
    const double hd = a.H[(size_t)i * n + i];
    p[i] = (hd > 1.0e-9) ? sqrt(hd) : 1.0e-3;
  }
  __syncthreads();
  for (int idx = t; idx < n * n; idx += nt) {
    const int i = idx / n, j = idx % n;
    a.G[idx] = 0.5 * (a.H[idx] + a.H[(size_t)j * n + i]) / (p[i] * p[j]);
    a.Q[idx] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const long long tPrep = wall_clock64(), cPrep = clock64();
  jacobiEig(a.G, a.Q, n, a.flag, useLds ? jacobiLds : nullptr);
  const long long tEig = wall_clock64(), cEig = clock64();
  __shared__ double smax;
  const int grp = t >> 4, gl = t & 15, nGroups = nt >> 4;   // 16-lane groups (one DPP row each)
  for (int j = grp; j < n; j += nGroups) {
    double s = 0;
    for (int i = gl; i < n; i += 16) s += a.Q[(size_t)j * n + i] * a.G[(size_t)j * n + i];
    s = rowSum16(s);
    if (gl == 0) ev[j] = s;
  }
  __syncthreads();
  if (t < 64) {
    double mx = -1.0e300, mn = 1.0e300;
    for (int j = t; j < n; j += 64) { mx = fmax(mx, ev[j]); mn = fmin(mn, ev[j]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); mn = fmin(mn, __shfl_xor(mn, o, 64)); }
    const double tl = 2.220446049250313e-16 * n * mx;
    int c = 0;
    for (int j = t; j < n; j += 64) c += ev[j] <= tl;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (t == 0) { smax = mx; a.flag[2] = c; a.scal[1] = mn; a.scal[2] = mx; }
  }
  __syncthreads();
  const double tol = 2.220446049250313e-16 * n * smax;
  // J = (p U sqrt(S))^T: row i = eigen-direction i; kept in LDS (when the launch has it) for the J^T J below
  const int ld = jacobiLd(n);
  lds_double* sJ = useLds ? toLds(jacobiLds) : nullptr;
  for (int idx = t; idx < n * n; idx += nt) {
    const int i = idx / n, j = idx - i * n;
    const double s = ev[i] > tol ? sqrt(ev[i]) : 0.0;
    const double v = p[j] * a.Q[idx] * s;
    a.J[idx] = v;
    if (sJ) sJ[i * ld + j] = v;
  }
  for (int i = grp; i < n; i += nGroups) {
    double e = 0;
    for (int j = gl; j < n; j += 16) e += a.Q[(size_t)i * n + j] * (a.b0[j] / p[j]);
    e = rowSum16(e);
    if (gl == 0) a.e0[i] = ev[i] > tol ? -sqrt(1.0 / ev[i]) * e : 0.0;
  }
  __syncthreads();
  if (sJ) {
    for (int idx = t; idx < n * n; idx += nt) {
      const int i = idx / n, j = idx - i * n;
      double s = 0;
      for (int k = 0; k < n; ++k) s += sJ[k * ld + i] * sJ[k * ld + j];
      a.Ht[idx] = s;
    }
    for (int i = t; i < n; i += nt) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += sJ[k * ld + i] * a.e0[k];
      a.bp[i] = s;
    }
  } else {
    for (int idx = t; idx < n * n; idx += nt) {
      const int i = idx / n, j = idx % n;
      double s = 0;
      for (int k = 0; k < n; ++k) s += a.J[(size_t)k * n + i] * a.J[(size_t)k * n + j];
      a.Ht[idx] = s;
    }
    for (int i = t; i < n; i += nt) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += a.J[(size_t)k * n + i] * a.e0[k];
      a.bp[i] = s;
    }
  }
  if (t < 64) {
    double c = 0;
    for (int k = t; k < n; k += 64) c += a.e0[k] * a.e0[k];
    c = waveSumM(c);
    if (t == 0) {
      a.scal[0] = c;
      // 100 MHz ticks: prepare, eigen-solve, everything after it (printed under SVIN_MARG_TIMING)
      a.scal[3] = (double)(tPrep - tStart); a.scal[4] = (double)(tEig - tPrep); a.scal[5] = (double)(wall_clock64() - tEig);
      a.scal[6] = (double)(cEig - cPrep);   // shader clocks of the eigen-solve
      a.scal[7] = (double)n;
    }
  }
}

// ================================================================ host: policy + job assembly
namespace {
template <class T>
bool contains(const std::vector<T>& v, const T& q) {
  for (const T& e : v) if (e == q) return true;
  return false;
}
}  // namespace

static double nowSec() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int Window::applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, std::vector<uint64_t>& removed) {
  quiesce();
  const bool timing = optOn(kOptMargTiming);
  const double tm0 = nowSec();
  double tm1 = tm0, tm2 = tm0, tm3 = tm0, tm4 = tm0;
  // ---- policy (Estimator.cpp:495-770), operating on the host graph only
  auto rit = states_.rbegin();
  for (size_t k = 0; k < numImuFrames; k++) {
    rit++;
    if (rit == states_.rend()) return 1;
  }
  // A pose on a reduced manifold (Map::resetParameterization, a Map-level feature okvis::Estimator never uses: Estimator.cpp:801 is
  // commented out): the prior's columns would have to be the 3 / 4 / 2 minimal ones (MarginalizationError.cpp:147-160 takes
  // minimalDimension()).  Not built; refuse before anything is modified.
  for (const auto& kv : factors_)
    if (kv.second.kind == F_HOST)
      throw std::runtime_error("applyMarginalizationStrategy: a residual evaluated by a host cost function is in the window (its linearisation is not available to the marginalisation kernels)");
  for (const auto& kv : blocks_)
    if (kv.second.lock != 0)
      throw std::runtime_error("applyMarginalizationStrategy: a pose block on a reduced manifold (Pose3d / Pose4d / Pose2d) is in the window");
  if (numLandmarkPriors_ > 0 || numFixedLandmarks_ > 0) {
    // Landmarks that carry a HomogeneousPointError stay out of the marginalisation: the reference's policy loop assumes
    // every residual of a landmark is a ReprojectionError (Estimator.cpp:689-698) and never meets one, because Estimator
    // adds none.  Observed from a frame that is about to leave, such a landmark cannot be handled: refuse before anything
    // is modified.
    std::vector<uint64_t> leaving;
    size_t kept = 0;
    for (auto r2 = rit; r2 != states_.rend(); ++r2) {
      if (!r2->second.isKeyframe || kept >= numKeyframes) leaving.push_back(r2->second.id);
      else kept++;
    }
    for (const auto& kv : landmarks_) {
      if (kv.second.priors.empty() && !kv.second.fixed) continue;
      for (const Observation& o : kv.second.obs)
        if (std::find(leaving.begin(), leaving.end(), o.poseId) != leaving.end()) {
          lastError() = "applyMarginalizationStrategy: a landmark with a HomogeneousPointError, or a constant one, is observed from a frame that leaves the window";
          return -1;
        }
    }
  }
  // :509-514 the old prior leaves the graph; its content is re-used below
  const bool hadPrior = hasPrior_;
  if (hadPrior) {
    for (const PriorBlockHost& pb : priorBlocks_)
      if (Block* b = findBlock(pb.id)) {
        for (size_t i = 0; i < b->residuals.size(); ++i)
          if (b->residuals[i] == priorResId_) { b->residuals.erase(b->residuals.begin() + i); break; }
      }
    priorResId_ = 0;
  }
  std::vector<uint64_t> removeFrames, removeAllButPose, allLinearizedFrames;
  size_t countedKeyframes = 0;
  while (rit != states_.rend()) {
    if (!rit->second.isKeyframe || countedKeyframes >= numKeyframes) removeFrames.push_back(rit->second.id);
    else countedKeyframes++;
    removeAllButPose.push_back(rit->second.id);
    allLinearizedFrames.push_back(rit->second.id);
    ++rit;
  }
  // the job: residuals to linearise (copied out of the graph as they are removed) and blocks to marginalise
  struct JobObs { Observation o; uint64_t lmId, extId; };
  std::vector<Factor> jobFactors;
  std::vector<JobObs> jobObs;
  std::vector<uint64_t> toMarginalize;
  // connected dense blocks in insertion order (after the blocks of the old prior) and landmarks
  std::vector<PriorBlockHost> dense = priorBlocks_;
  std::vector<uint64_t> lmOrder;
  std::unordered_map<uint64_t, std::vector<double>> lmLin;  // linearisation points of marginalised landmarks
  // (one flag per block handle: the landmark pass asks "is it connected already" twice per linearised residual)
  std::vector<char> connected[3];
  for (int k = 0; k < 3; ++k) connected[k].assign(std::max(nextBlockH_[k], 1), 0);
  for (const PriorBlockHost& pb : dense)
    if (const Block* b = findBlock(pb.id)) connected[b->kind][b->handle] = 1;
  auto connectBlock = [&](const Block& b) {
    if (connected[b.kind][b.handle]) return;
    connected[b.kind][b.handle] = 1;
    PriorBlockHost pb;
    pb.id = b.id; pb.kind = b.kind; pb.dim = (b.kind == B_SB) ? 9 : 7;
    pb.mdim = b.fixed ? 0 : ((b.kind == B_SB) ? 9 : 6);
    std::memcpy(pb.lin, b.x, sizeof(double) * pb.dim);
    dense.push_back(pb);
  };
  auto connectDense = [&](uint64_t id) {
    for (const PriorBlockHost& pb : dense) if (pb.id == id) return;
    const Block& b = blocks_.at(id);
    connected[b.kind][b.handle] = 1;
    PriorBlockHost pb;
    pb.id = id; pb.kind = b.kind; pb.dim = (b.kind == B_SB) ? 9 : 7;
    pb.mdim = b.fixed ? 0 : ((b.kind == B_SB) ? 9 : 6);
    std::memcpy(pb.lin, b.x, sizeof(double) * pb.dim);
    dense.push_back(pb);
  };
  auto addFactorToJob = [&](uint64_t rid) {
    auto it = factors_.find(rid);
    if (it == factors_.end()) return;
    for (int b = 0; b < it->second.nblk; ++b) connectDense(it->second.blocks[b]);
    jobFactors.push_back(it->second);
    removeFactor(rid);
  };
  auto isReproj = [&](uint64_t rid) { return obsRes2Lm_.count(rid); };
  auto isPrior = [&](uint64_t rid) { return rid != 0 && !isReproj(rid) && !factors_.count(rid); };

  for (uint64_t fid : removeAllButPose) {  // :541-605
    auto it = states_.find(fid);
    for (size_t j = 0; j < it->second.sb.size(); ++j) {
      StateInfo& si = it->second.sb[j];
      if (!si.exists) continue;
      if (blocks_.at(si.id).fixed) continue;
      auto checkit = it;
      checkit++;
      if (checkit != states_.end() && checkit->second.sb.size() > j && checkit->second.sb[j].exists &&
          checkit->second.sb[j].id == si.id) continue;
      si.exists = false;
      toMarginalize.push_back(si.id);
      const std::vector<uint64_t> res = blocks_.at(si.id).residuals;
      for (uint64_t rid : res)
        if (!isReproj(rid) && !isPrior(rid)) addFactorToJob(rid);
    }
  }
  bool reDoFixation = false;
  bool landmarksDone = false;
  // The job's observation tables come out of the device-resident CSR when it mirrors the graph as it is now (nothing was
  // added or removed since the last solve); otherwise they are assembled here from the records the policy removes.
  const bool deviceJob = residentValid_ && residentUsed_ && addLog_.empty() && remLog_.empty() && setLog_.empty() && res_.N == (int)numObs_;
  ++pathCounters_[deviceJob ? 2 : 3];
  if (!deviceJob) syncLandmarks();   // the linearisation points of the marginalised landmarks are read from the host graph
  std::vector<unsigned char> poseClass;
  std::vector<uint64_t> margLandmarks;
  int nJobObs = 0;
  for (uint64_t fid : removeFrames) {  // :607-770
    auto it = states_.find(fid);
    it->second.pose.exists = false;
    toMarginalize.push_back(it->second.pose.id);
    {
      const std::vector<uint64_t> res = blocks_.at(it->second.pose.id).residuals;
      for (uint64_t rid : res) {
        auto fit = factors_.find(rid);
        if (fit != factors_.end() && fit->second.kind == F_POSE_PRIOR) {  // :624-629
          removeFactor(rid);
          reDoFixation = true;
          continue;
        }
        if (!isReproj(rid) && fit != factors_.end()) addFactorToJob(rid);
      }
    }
    for (size_t j = 0; j < it->second.ext.size(); ++j) {  // :638-664
      StateInfo& si = it->second.ext[j];
      if (!si.exists) continue;
      if (blocks_.at(si.id).fixed) continue;
      auto checkit = it;
      checkit++;
      if (checkit != states_.end() && checkit->second.ext[j].exists && checkit->second.ext[j].id == si.id) continue;
      si.exists = false;
      toMarginalize.push_back(si.id);
      const std::vector<uint64_t> res = blocks_.at(si.id).residuals;
      for (uint64_t rid : res)
        if (!isReproj(rid) && factors_.count(rid)) addFactorToJob(rid);
    }
    if (!landmarksDone) {  // :671-766
      // The reference walks every landmark once per leaving frame.  What it does with a landmark depends only on the three
      // frame sets, which are fixed for the whole call, and applying it a second time changes nothing -- so ONE pass, over the
      // landmarks a leaving frame has seen (Block::seenLm) and the ones without observations, in handle order (the order of
      // the device-resident CSR, whose gather kernel picks the same residuals: resident.hpp margObsAction).
      landmarksDone = true;
      const double tl0 = nowSec();
      const size_t obs0 = numObs_;
      const uint64_t currentKfId = allLinearizedFrames.at(0);
      std::vector<unsigned char>& cls = poseClass;
      cls.assign(std::max(nextBlockH_[B_POSE], 1), 0);
      for (uint64_t f : removeFrames) cls[blocks_.at(f).handle] |= kMargRemove;
      for (uint64_t f : allLinearizedFrames) cls[blocks_.at(f).handle] |= kMargLin;
      for (auto st = states_.lower_bound(currentKfId); st != states_.end(); ++st)
        if (const Block* pb = findBlock(st->second.pose.id)) cls[pb->handle] |= kMargNew;
      std::vector<int> cand;
      ++visitStamp_;
      auto consider = [&](int h) {
        Landmark* lp = (h >= 0 && h < (int)lmByHandle_.size()) ? lmByHandle_[h] : nullptr;
        if (!lp || lp->visit == visitStamp_) return;
        lp->visit = visitStamp_;
        cand.push_back(h);
      };
      for (uint64_t f : removeFrames) for (int h : blocks_.at(f).seenLm) consider(h);
      for (int h : emptyLm_) {
        Landmark* lp = (h >= 0 && h < (int)lmByHandle_.size()) ? lmByHandle_[h] : nullptr;
        if (lp && lp->obs.empty()) consider(h);
      }
      emptyLm_.clear();
      std::sort(cand.begin(), cand.end());
      const double tl1 = nowSec();
      for (int h : cand) {
        Landmark& lm = *lmByHandle_[h];
        if (!lm.priors.empty() || lm.fixed) continue;   // see the guard at the top
        if (lm.obs.empty()) {
          removed.push_back(lm.id);
          eraseLandmark(lm);
          continue;
        }
        bool skipLandmark = true, hasNewObservations = false, marginalize = true, errorTermAdded = false;
        int obsCount = 0;
        for (const Observation& o : lm.obs) {
          const int c = cls[o.poseH];
          if (c & kMargRemove) skipLandmark = false;
          if (c & kMargNew) { marginalize = false; hasNewObservations = true; }
          if (c & kMargLin) obsCount++;
        }
        if (skipLandmark) continue;
        size_t keep = 0;   // the list is compacted once (an erase per removed record moves its tail every time)
        for (size_t i = 0; i < lm.obs.size(); ++i) {
          const Observation& ob = lm.obs[i];
          const int action = margObsAction(cls[ob.poseH], hasNewObservations, marginalize, obsCount);
          if (action == 0) {
            if (keep != i) lm.obs[keep] = ob;
            ++keep;
            continue;
          }
          if (action == 2) {
            errorTermAdded = true;
            connectBlock(*blockByHandle_[B_POSE][ob.poseH]);
            connectBlock(*blockByHandle_[B_EXT][ob.extH]);
            if (lmOrder.empty() || lmOrder.back() != lm.id) {
              lmOrder.push_back(lm.id);
              if (!deviceJob) lmLin[lm.id].assign(lm.hp, lm.hp + 4);
            }
            if (!deviceJob) jobObs.push_back({ob, lm.id, extIdOf(ob)});
            ++nJobObs;
          }
          detachObsRecord(lm, ob);
        }
        if (keep != lm.obs.size()) {
          lm.obs.resize(keep);
          afterObsRemoval(lm);
        }
        if (lm.obs.empty() && !errorTermAdded) {   // every residual was dropped: "justDelete"
          removed.push_back(lm.id);
          eraseLandmark(lm);
        } else if (marginalize && errorTermAdded) {
          toMarginalize.push_back(lm.id);
          margLandmarks.push_back(lm.id);
          removed.push_back(lm.id);
          eraseLandmark(lm);
        }
      }
      if (timing)
        std::printf("[marg] landmark pass: %zu candidates collected in %.0f us, handled in %.0f us; %zu observations removed, %zu landmarks erased\n",
                    cand.size(), 1e6 * (tl1 - tl0), 1e6 * (nowSec() - tl1), obs0 - numObs_, removed.size());
    }
    states_.erase(it->second.id);
  }

  // ---- device job (M1-M3)
  tm1 = nowSec();
  std::sort(removed.begin(), removed.end());   // PointMap order (the reference erases while walking its std::map)
  std::sort(margLandmarks.begin(), margLandmarks.end());
  obsCachePose_ = 0;
  std::sort(toMarginalize.begin(), toMarginalize.end());
  toMarginalize.erase(std::unique(toMarginalize.begin(), toMarginalize.end()), toMarginalize.end());
  bool anyWork = !toMarginalize.empty();
  // ordering of the dense part
  int m = 0;
  for (PriorBlockHost& pb : dense) { pb.ord = m; m += pb.mdim; }
  const int Lm = (int)lmOrder.size();
  std::vector<double> Hk, bk;
  std::vector<PriorBlockHost> kept;
  if (m > 0 || Lm > 0) {
    hipStream_t s = stream_;
    // sub-problem tables at the linearisation points
    std::vector<uint64_t> jPose, jExt, jSb;
    std::unordered_map<uint64_t, int> sPose, sExt, sSb, sLm;
    std::vector<double> hPose, hExt, hSb, hLm;
    std::vector<int> oPose, oExt, oSb;
    for (const PriorBlockHost& pb : dense) {
      const int off = pb.mdim > 0 ? pb.ord : -1;
      if (pb.kind == B_POSE) { sPose[pb.id] = (int)jPose.size(); jPose.push_back(pb.id); hPose.insert(hPose.end(), pb.lin, pb.lin + 7); oPose.push_back(off); }
      else if (pb.kind == B_EXT) { sExt[pb.id] = (int)jExt.size(); jExt.push_back(pb.id); hExt.insert(hExt.end(), pb.lin, pb.lin + 7); oExt.push_back(off); }
      else { sSb[pb.id] = (int)jSb.size(); jSb.push_back(pb.id); hSb.insert(hSb.end(), pb.lin, pb.lin + 9); oSb.push_back(off); }
    }
    if (hExt.empty()) { hExt.assign(7, 0.0); hExt[6] = 1.0; oExt.push_back(-1); }
    if (hPose.empty()) { hPose.assign(7, 0.0); hPose[6] = 1.0; oPose.push_back(-1); }
    if (!deviceJob)
      for (int l = 0; l < Lm; ++l) {
        sLm[lmOrder[l]] = l;
        const std::vector<double>& hp = lmLin.at(lmOrder[l]);
        hLm.insert(hLm.end(), hp.begin(), hp.end());
      }
    // observations sorted landmark-major (the policy hands them over landmark by landmark already)
    if (!deviceJob)
      std::stable_sort(jobObs.begin(), jobObs.end(), [&](const JobObs& a, const JobObs& b) { return sLm.at(a.lmId) < sLm.at(b.lmId); });
    const int N = nJobObs;
    std::vector<double> hUv, hW;
    std::vector<uint32_t> hIdx;
    std::vector<int> hObsLm, hLmPtr(deviceJob ? 0 : Lm + 1, 0);
    bool anyExtVar = false;
    // device job: the tables below are written by k_window_marg_gather from the resident CSR; it needs to know the class of
    // every pose handle and where a block sits in the job's tables
    std::vector<int> jobPoseSlot, jobExtSlot;
    if (deviceJob) {
      jobPoseSlot.assign(std::max(nextBlockH_[B_POSE], 1), -1); jobExtSlot.assign(std::max(nextBlockH_[B_EXT], 1), -1);
      for (const auto& kv : sPose) jobPoseSlot[blocks_.at(kv.first).handle] = kv.second;
      for (const auto& kv : sExt) jobExtSlot[blocks_.at(kv.first).handle] = kv.second;
    }
    for (const JobObs& jo : jobObs) {
      hUv.push_back(jo.o.uv[0]); hUv.push_back(jo.o.uv[1]);
      hW.push_back(obsWeight(jo.o.size));
      const int es = sExt.count(jo.extId) ? sExt.at(jo.extId) : 0;
      hIdx.push_back(packObs(sPose.at(jo.o.poseId), es, jo.o.cam));
      hObsLm.push_back(sLm.at(jo.lmId));
      hLmPtr[sLm.at(jo.lmId) + 1]++;
    }
    if (!deviceJob) for (int l = 0; l < Lm; ++l) hLmPtr[l + 1] += hLmPtr[l];
    for (int o : oExt) if (o >= 0) anyExtVar = true;
    // factors
    std::vector<DevFactor> hFac;
    std::vector<DevImu> hImu;
    std::vector<uint32_t> hImuT;
    std::vector<double> hImuM;
    for (Factor& f : jobFactors) {
      DevFactor df;
      std::memset(&df, 0, sizeof(df));
      df.kind = f.kind; df.nblk = f.nblk; df.m = f.m; df.imuIndex = -1;
      for (int b = 0; b < f.nblk; ++b) {
        const uint64_t id = f.blocks[b];
        if (sPose.count(id)) { df.blkKind[b] = B_POSE; df.blkSlot[b] = sPose.at(id); }
        else if (sExt.count(id)) { df.blkKind[b] = B_EXT; df.blkSlot[b] = sExt.at(id); }
        else { df.blkKind[b] = B_SB; df.blkSlot[b] = sSb.at(id); }
      }
      std::memcpy(df.meas, f.meas, sizeof(df.meas));
      std::memcpy(df.aux, f.aux, sizeof(df.aux));
      std::memcpy(df.sqrtInfo, f.sqrtInfo, sizeof(df.sqrtInfo));
      if (f.kind == F_IMU) {
        df.imuIndex = (int)hImu.size();
        f.imu.sampleStart = (int)(hImuT.size() / 2);
        f.imu.sampleCount = (int)(f.imuT.size() / 2);
        hImu.push_back(f.imu);
        hImuT.insert(hImuT.end(), f.imuT.begin(), f.imuT.end());
        hImuM.insert(hImuM.end(), f.imuMeas.begin(), f.imuMeas.end());
      }
      hFac.push_back(df);
    }
    const int F = (int)hFac.size();
    // The job is asynchronous (nothing below waits for the device): every table travels in the one pinned block of
    // flushStaged, which waits for the previous block's DMA itself.
    tm2 = nowSec();
    double* dbgScal = nullptr;
    // persistent job buffers (grow-only): hipMalloc / hipFree per call used to cost more than the algebra
    MargBuffers& mb = margBuf_;
    auto &bPose = mb.bPose, &bExt = mb.bExt, &bSb = mb.bSb, &bLm = mb.bLm, &bUv = mb.bUv, &bW = mb.bW, &bLin = mb.bLin,
         &bU = mb.bU, &bW2 = mb.bW2, &bV = mb.bV, &bVec = mb.bVec, &bScratch = mb.bScratch;
    auto &bOP = mb.bOP, &bOE = mb.bOE, &bOS = mb.bOS, &bLmPtr = mb.bLmPtr, &bObsLm = mb.bObsLm, &bIdxList = mb.bIdxList,
         &bFlag = mb.bFlag;
    auto &bIdx = mb.bIdx, &bImuT = mb.bImuT;
    auto& bFac = mb.bFac;
    auto& bFacLin = mb.bFacLin;
    auto& bImu = mb.bImu;
    auto &bImuM = mb.bImuM, &bPartial = mb.bPartial;
    auto& bScal = mb.bScal;
    // M2 dense part: which rows stay (the new prior) and which are marginalised
    std::vector<int> keepIdx, margIdx;
    for (const PriorBlockHost& pb : dense) {
      const bool marg = std::binary_search(toMarginalize.begin(), toMarginalize.end(), pb.id);
      for (int k = 0; k < pb.mdim; ++k) (marg ? margIdx : keepIdx).push_back(pb.ord + k);
      if (!marg) kept.push_back(pb);
    }
    const int nk = (int)keepIdx.size(), nm = (int)margIdx.size();
    std::vector<int> idxLists(keepIdx);
    idxLists.insert(idxLists.end(), margIdx.begin(), margIdx.end());
    const size_t mm = std::max(m, 1), L3 = std::max(3 * Lm, 1);
    bU.reserve(mm * mm + 2); bW2.reserve(mm * L3 + 2); bV.reserve((size_t)9 * std::max(Lm, 1) + 2); bVec.reserve(mm + 2 * L3 + 16);
    auto pendingPtr = std::make_shared<std::vector<StagedCopy>>();
    std::vector<StagedCopy>& pending = *pendingPtr;
    {   // the job's tables: one pinned block, one DMA, one scatter kernel (17 pageable copies cost ~100 us of enqueueing)
      // U, W, V and the vectors start from zero: clears riding in the same launch
      pending.push_back({nullptr, sizeof(double) * mm * mm, bU.p});
      pending.push_back({nullptr, sizeof(double) * mm * L3, bW2.p});
      pending.push_back({nullptr, sizeof(double) * 9 * (size_t)std::max(Lm, 1), bV.p});
      pending.push_back({nullptr, sizeof(double) * (mm + 2 * L3 + 14), bVec.p});
      auto stage = [&](auto& buf, const auto& host) {
        using T = typename std::decay_t<decltype(host)>::value_type;
        buf.reserve(std::max<size_t>(host.size() + 16 / sizeof(T) + 1, 1));   // room for the 16-byte rounding of the copy
        if (!host.empty()) pending.push_back({host.data(), sizeof(T) * host.size(), buf.p});
      };
      stage(bPose, hPose); stage(bExt, hExt); stage(bSb, hSb); stage(bLm, hLm); stage(bUv, hUv); stage(bW, hW);
      stage(bOP, oPose); stage(bOE, oExt); stage(bOS, oSb); stage(bLmPtr, hLmPtr); stage(bObsLm, hObsLm); stage(bIdx, hIdx);
      stage(bFac, hFac); stage(bImu, hImu); stage(bImuT, hImuT); stage(bImuM, hImuM);
      stage(dCams_, cameras_);
      stage(bIdxList, idxLists);
      if (deviceJob) {
        stage(res_.poseClass, poseClass); stage(res_.poseSlotOfH, jobPoseSlot); stage(res_.extSlotOfH, jobExtSlot);
        bLm.reserve(std::max<size_t>((size_t)4 * Lm, 1)); bUv.reserve(std::max<size_t>((size_t)2 * N, 1)); bW.reserve(std::max<size_t>(N, 1));
        bLmPtr.reserve((size_t)Lm + 1); bObsLm.reserve(std::max<size_t>(N, 1)); bIdx.reserve(std::max<size_t>(N, 1));
        res_.margScratch.reserve(std::max<size_t>((size_t)2 * res_.L, 1));
      }
    }
    // ---- everything the launches below need is fixed from here on; no allocation and no host table is touched after this point,
    // so the launches themselves (a DMA, the scatter and 6-13 kernels: 25-50 us of API calls) can be issued by the handle's
    // enqueue thread while this call goes on with the graph update and returns (Window::quiesce joins it before the stream, the
    // job buffers or the prior are touched again)
    bLin.reserve(std::max<size_t>((size_t)32 * N, 1));
    bFacLin.reserve(std::max(F, 1));
    bPartial.reserve((size_t)16 * 4096);
    bScal.reserve(1);
    bFlag.reserve(8);
    int ordk = 0;
    for (PriorBlockHost& pb : kept) { pb.ord = ordk; ordk += pb.mdim; }
    Hk.assign((size_t)nk * nk, 0.0);
    bk.assign(nk, 0.0);
    auto &bHk = mb.bHk, &bOut = mb.bOut;
    const int oldPriorM = (hadPrior && priorM_ > 0) ? priorM_ : 0;
    const double* oldPrior = mb.bHk.p;   // (H | b0) of the previous prior: bHk is only ever re-allocated below when it has to grow ...
    DevBuf<double> oldPriorKeep;         // ... and then the old allocation is kept alive until the job has copied out of it
    if (std::max<size_t>((size_t)nk * nk + nk, 1) > bHk.cap && oldPriorM > 0) { std::swap(oldPriorKeep.p, bHk.p); std::swap(oldPriorKeep.cap, bHk.cap); }
    bHk.reserve(std::max<size_t>((size_t)nk * nk + nk, 1));
    if (nk > 0 && nm > 0) bScratch.reserve((size_t)2 * nm * nm + (size_t)nk * nm + 2 * nm + 16);
    const size_t n2k = (size_t)nk * nk;
    if (nk > 0) bOut.reserve(5 * n2k + 4 * nk + 16);
    if (nk > 0) priorHostValid_ = false;  // results stay on the device (solver reads Ht / bp / c0 in place); getPrior() fetches
    const int nPoseJ = (int)(hPose.size() / 7), nExtJ = (int)(hExt.size() / 7), nSbJ = (int)jSb.size(), nImuJ = (int)hImu.size();
    const int nCamJ = (int)cameras_.size();
    const bool keepPre = optOn(kOptMargKeepPre);
    const int margEig = debugOption(kOptMargEig);   // read HERE, on the caller's thread: the job may be issued by the enqueue thread
    // the host tables the staged block is filled from must outlive this call when the job is issued by the enqueue thread
    struct JobTables {
      std::vector<double> hPose, hExt, hSb, hLm, hUv, hW, hImuM;
      std::vector<int> oPose, oExt, oSb, hObsLm, hLmPtr, idxLists, jobPoseSlot, jobExtSlot;
      std::vector<uint32_t> hIdx, hImuT;
      std::vector<DevFactor> hFac;
      std::vector<DevImu> hImu;
      std::vector<unsigned char> poseClass;
      std::vector<CameraModel> cams;
      DevBuf<double> oldPriorKeep;
    };
    auto tables = std::make_shared<JobTables>();
    const bool syncJob = optOn(kOptMargSyncEnqueue);   // A/B switch: issue the launches from this thread
    const bool runInline = timing || keepPre || syncJob;
    double** dbgScalPtr = runInline ? &dbgScal : nullptr;
    // (explicit captures: the job buffers are reached through `this` -- a by-value capture of the DevBuf aliases above would copy,
    // and later free, the buffers themselves; lmOrder / dense / toMarginalize are only read by the inline inspection path)
    auto launchJob = [this, tables, pendingPtr, N, Lm, m, F, nk, nm, mm, L3, n2k, oldPriorM, oldPrior, anyExtVar, deviceJob, nPoseJ, nExtJ,
                      nSbJ, nImuJ, nCamJ, keepPre, margEig, dbgScalPtr, s, &lmOrder, &dense, &toMarginalize]() {
      MargBuffers& mb = margBuf_;
      auto &bPose = mb.bPose, &bExt = mb.bExt, &bSb = mb.bSb, &bLm = mb.bLm, &bUv = mb.bUv, &bW = mb.bW, &bLin = mb.bLin,
           &bU = mb.bU, &bW2 = mb.bW2, &bV = mb.bV, &bVec = mb.bVec, &bScratch = mb.bScratch;
      auto &bOP = mb.bOP, &bOE = mb.bOE, &bOS = mb.bOS, &bLmPtr = mb.bLmPtr, &bObsLm = mb.bObsLm, &bIdxList = mb.bIdxList, &bFlag = mb.bFlag;
      auto &bIdx = mb.bIdx, &bImuT = mb.bImuT;
      auto& bFac = mb.bFac;
      auto& bFacLin = mb.bFacLin;
      auto& bImu = mb.bImu;
      auto &bImuM = mb.bImuM, &bPartial = mb.bPartial;
      auto& bScal = mb.bScal;
      auto &bHk = mb.bHk, &bOut = mb.bOut;
      flushStaged(*pendingPtr, s);
      if (deviceJob && (N > 0 || Lm > 0)) {
        MargGatherArgs ga;
        std::memset(&ga, 0, sizeof(ga));
        ga.L = res_.L; ga.H = res_.H; ga.expectN = N; ga.expectLm = Lm;
        ga.lmPtr = res_.lmPtr[res_.cur].p; ga.handleOfSlot = res_.handleOfSlot[res_.cur].p;
        ga.uv = res_.uv[res_.cur].p; ga.w = res_.w[res_.cur].p; ga.hnd = res_.hnd[res_.cur].p;
        ga.lmHp = res_.lmHp.p;
        ga.poseClass = res_.poseClass.p; ga.jobPoseSlot = res_.poseSlotOfH.p; ga.jobExtSlot = res_.extSlotOfH.p;
        ga.jLmPtr = bLmPtr.p; ga.jObsLm = bObsLm.p; ga.jIdx = bIdx.p; ga.jUv = bUv.p; ga.jW = bW.p; ga.jLm = bLm.p;
        ga.scratch = res_.margScratch.p;
        ga.status = resStatusDev_;
        launchWindowMargGather(ga, s);
      }
    {   // (the four clears rode in the scatter launch of the staged block) the two copies of the old prior land inside the cleared U / ba
      // old prior content (H_, b0_) occupies the leading block: it is still on the device, exactly where k_marg_dense left it
      if (oldPriorM > 0) {
        FillJobs copies;
        copies.n = 0;
        addFill(copies, bU.p, oldPrior, (size_t)oldPriorM * oldPriorM, oldPriorM, m, oldPriorM);
        addFill(copies, bVec.p, oldPrior + (size_t)oldPriorM * oldPriorM, oldPriorM);
        launchFillJobs(copies, s);
      }
    }
    DeviceProblem q;
    std::memset(&q, 0, sizeof(q));
    q.nPose = nPoseJ; q.nExt = nExtJ; q.nSb = nSbJ;
    q.L = Lm; q.N = N; q.F = F; q.nImu = nImuJ; q.d = m; q.dC = 0; q.nCam = nCamJ;
    q.anyExtVariable = anyExtVar ? 1 : 0;
    q.ownsCamera = 1;
    q.pose = bPose.p; q.ext = bExt.p; q.sb = bSb.p; q.lm = bLm.p;
    q.poseC = bPose.p; q.extC = bExt.p; q.sbC = bSb.p; q.lmC = bLm.p;
    q.poseOff = bOP.p; q.extOff = bOE.p; q.sbOff = bOS.p;
    q.cams = dCams_.p;
    q.lmPtr = bLmPtr.p; q.obsUv = bUv.p; q.obsW = bW.p; q.obsIdx = bIdx.p; q.obsLm = bObsLm.p;
    q.rCur = bLin.p; q.JpCur = bLin.p + (size_t)2 * N; q.JlCur = bLin.p + (size_t)14 * N; q.JeCur = bLin.p + (size_t)20 * N;
    q.factors = bFac.p; q.linCur = bFacLin.p; q.linCand = bFacLin.p;
    q.imus = bImu.p; q.imuT = bImuT.p; q.imuMeas = bImuM.p;
    q.scal = bScal.p; q.partial = bPartial.p;
    MargDev md;
    md.m = m; md.Lm = Lm; md.N = N; md.F = F;
    md.U = bU.p; md.ba = bVec.p; md.W = bW2.p; md.V = bV.p; md.bb = bVec.p + mm;
    double* vb = bVec.p + mm + L3;
    // M1: evaluate at the linearisation points (Cauchy corrector as in :283-330) and accumulate
    if (N > 0) {
      launchEvalReproj(q, false, true, s);
      const dim3 camGrid(q.nPose + q.nExt, anyExtVar ? 1 + q.nCam : 1);
      if (anyExtVar) {
        hipLaunchKernelGGL(k_marg_accum_lm<true>, dim3((Lm + 63) / 64), dim3(64), 0, s, q, md);
        hipLaunchKernelGGL(k_marg_accum_cam<true>, camGrid, dim3(64), 0, s, q, md);
      } else {
        hipLaunchKernelGGL(k_marg_accum_lm<false>, dim3((Lm + 63) / 64), dim3(64), 0, s, q, md);
        hipLaunchKernelGGL(k_marg_accum_cam<false>, camGrid, dim3(64), 0, s, q, md);
      }
    }
    if (F > 0) {
      launchEvalFactors(q, false, s);
      hipLaunchKernelGGL(k_marg_accum_factors, dim3(1), dim3(256), 0, s, q, md);
    }
    if (keepPre) {   // inspection: the system after M1 (svin_ba_get_marg_pre), before anything is eliminated
      HIP_OK(hipStreamSynchronize(s));
      margPre_.m = m; margPre_.Lm = Lm;
      margPre_.U.assign((size_t)m * m, 0.0); margPre_.ba.assign(std::max(m, 1), 0.0);
      margPre_.W.assign((size_t)m * 3 * Lm, 0.0); margPre_.V.assign((size_t)9 * Lm, 0.0); margPre_.bb.assign((size_t)3 * Lm, 0.0);
      if (m > 0) {
        HIP_OK(hipMemcpy(margPre_.U.data(), bU.p, sizeof(double) * m * m, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(margPre_.ba.data(), bVec.p, sizeof(double) * m, hipMemcpyDeviceToHost));
      }
      if (Lm > 0) {
        if (m > 0) HIP_OK(hipMemcpy(margPre_.W.data(), bW2.p, sizeof(double) * m * 3 * Lm, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(margPre_.V.data(), bV.p, sizeof(double) * 9 * Lm, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(margPre_.bb.data(), md.bb, sizeof(double) * 3 * Lm, hipMemcpyDeviceToHost));
      }
      margPre_.margRows.clear();
      margPre_.denseIds.clear(); margPre_.denseOrd.clear(); margPre_.denseMdim.clear();
      margPre_.lmIds.assign(lmOrder.begin(), lmOrder.end());
      for (const PriorBlockHost& pb : dense) { margPre_.denseIds.push_back(pb.id); margPre_.denseOrd.push_back(pb.ord); margPre_.denseMdim.push_back(pb.mdim); }
      for (const PriorBlockHost& pb : dense) {
        const bool mg = std::binary_search(toMarginalize.begin(), toMarginalize.end(), pb.id);
        for (int k = 0; k < pb.mdim; ++k) margPre_.margRows.push_back(mg ? 1 : 0);
      }
    }
    // M2 landmark part
    if (Lm > 0 && m > 0) {
      hipLaunchKernelGGL(k_marg_lm_prepare, dim3((Lm + 127) / 128), dim3(128), 0, s, md, vb);
      hipLaunchKernelGGL(k_marg_lm_apply, dim3(m), dim3(256), 0, s, md, (const double*)vb);
      hipLaunchKernelGGL(k_marg_lm_update, dim3(((m + 15) / 16) * ((m + 15) / 16)), dim3(64), 0, s, md);
    }
    // M2 dense part
    if (nk > 0) {
      if (nm > 0) {
        DenseArgs da;
        da.m = m; da.nk = nk; da.nm = nm;
        da.keep = bIdxList.p; da.marg = bIdxList.p + nk;
        da.U = bU.p; da.ba = bVec.p;
        da.Hk = bHk.p; da.bk = bHk.p + (size_t)nk * nk;
        da.Vm = bScratch.p; da.Qm = bScratch.p + (size_t)nm * nm; da.tmp = bScratch.p + (size_t)2 * nm * nm;
        da.flag = bFlag.p;
        {
          const size_t ldsEig = jacobiLdsBytes(nm);
          const size_t ldsProd = sizeof(double) * ((size_t)2 * nk * nm + (size_t)nm * nm);   // W, Mu, N^T of the products
          const bool prodLds = ldsProd <= kJacobiLdsLimit;
          const size_t ldsTile = margCholLdsBytes(nm);   // the certified Cholesky route of the pseudo-inverse, on tiles
          const bool tileChol = ldsTile <= kJacobiLdsLimit;
          const size_t lds = std::max(std::max(ldsEig, prodLds ? ldsProd : (size_t)0), tileChol ? ldsTile : (size_t)0);
          if (lds) ensureDynamicLds((const void*)k_marg_dense, lds);
          hipLaunchKernelGGL(k_marg_dense, dim3(1), dim3(1024), lds, s, da, ldsEig ? 1 : 0, prodLds ? 1 : 0, tileChol ? 1 : 0);
          HIP_OK(hipGetLastError());   // (a refused launch would leave a garbage prior behind)
        }
      } else {
        HIP_OK(hipMemcpyAsync(bHk.p, bU.p, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpyAsync(bHk.p + (size_t)m * m, bVec.p, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
      }
      // M3
      const size_t n2 = n2k;
      FinalArgs fa;
      fa.n = nk; fa.H = bHk.p; fa.b0 = bHk.p + n2;
      fa.G = bOut.p; fa.Q = bOut.p + n2; fa.J = bOut.p + 2 * n2; fa.Ht = bOut.p + 3 * n2;
      fa.e0 = bOut.p + 4 * n2; fa.bp = bOut.p + 4 * n2 + nk; fa.scal = bOut.p + 4 * n2 + 2 * nk;
      fa.tmp = bOut.p + 4 * n2 + 2 * nk + 8;
      fa.flag = bFlag.p;
      if (dbgScalPtr) *dbgScalPtr = fa.scal;
      {
        // Eigen-solver of the prior (k_marg_final's `useLds`):
        //   4  A + delta I = R^T R, one-sided Jacobi on the rows of R, image in LDS (default: n <= 136)
        //   6  the same with the image in global memory (larger priors)
        //   1 / 0  the ONE fall-back: one-sided Jacobi on A itself with G and Q in LDS (n <= 96) / in global memory; taken
        //          inside the kernel when a pivot of the factorisation is not positive, or with SVIN_MARG_EIG=jacobi
        //          (the test that keeps the fall-back honest)
        const bool want = margEig != 0;   // SVIN_MARG_EIG: 1 "direct", 2 "cholesky", 3 "jacobi" (options.hpp)
        const bool forceFallback = margEig == 3;
        const size_t ldsBoth = jacobiLdsBytes(nk), ldsOne = jacobiLdsBytesGOnly(nk);
        const int mode = forceFallback ? (ldsBoth ? 1 : 0) : (ldsOne ? 4 : 6);
        const size_t lds = (mode == 4) ? std::max(ldsOne, ldsBoth) : (mode == 1 ? ldsBoth : 0);
        // default since round 5: the direct solve (tridiagonalisation + divide and conquer) for priors up to 128 unknowns, the
        // Jacobi kernel behind it as the fall-back for a non-finite result (it returns at once when flag[4] says "done");
        // SVIN_MARG_EIG=cholesky / jacobi select the round-2 solvers alone
        // ... and ahead of both, for a prior of full numerical rank (every steady-state prior of the sliding windows), the Cholesky
        // factor with its certificate that the rank rule drops nothing (k_marg_final_chol); SVIN_MARG_EIG=direct skips it
        const bool wantDirect = margEig == 1;
        const bool direct = (!want || wantDirect) && nk <= kSymEigMaxN;
        const bool chol = !want && nk <= kSymEigMaxN;
        if (chol) {
          const size_t ldsC = margCholLdsBytes(nk);
          ensureDynamicLds((const void*)k_marg_final_chol, ldsC);
          hipLaunchKernelGGL(k_marg_final_chol, dim3(1), dim3(1024), ldsC, s, fa);
          HIP_OK(hipGetLastError());
        }
        if (direct) {
          const size_t ldsDc = symEigLdsBytes(nk);
          ensureDynamicLds((const void*)k_marg_final_dc, ldsDc);
          hipLaunchKernelGGL(k_marg_final_dc, dim3(1), dim3(1024), ldsDc, s, fa, chol ? 1 : 0);
          HIP_OK(hipGetLastError());
        }
        if (lds) ensureDynamicLds((const void*)k_marg_final, lds);
        hipLaunchKernelGGL(k_marg_final, dim3(1), dim3(1024), lds, s, fa, mode, (mode == 4 && ldsBoth) ? 1 : 0, direct ? 1 : 0);
        HIP_OK(hipGetLastError());
      }
    }
    };
    // move the host tables behind the staged pointers into the job (a vector's storage does not move with it)
    tables->hPose = std::move(hPose); tables->hExt = std::move(hExt); tables->hSb = std::move(hSb); tables->hLm = std::move(hLm);
    tables->hUv = std::move(hUv); tables->hW = std::move(hW); tables->hImuM = std::move(hImuM);
    tables->oPose = std::move(oPose); tables->oExt = std::move(oExt); tables->oSb = std::move(oSb); tables->hObsLm = std::move(hObsLm);
    tables->hLmPtr = std::move(hLmPtr); tables->idxLists = std::move(idxLists); tables->jobPoseSlot = std::move(jobPoseSlot);
    tables->jobExtSlot = std::move(jobExtSlot); tables->hIdx = std::move(hIdx); tables->hImuT = std::move(hImuT);
    tables->hFac = std::move(hFac); tables->hImu = std::move(hImu); tables->poseClass = std::move(poseClass);
    std::swap(tables->oldPriorKeep.p, oldPriorKeep.p); std::swap(tables->oldPriorKeep.cap, oldPriorKeep.cap);
    if (runInline) launchJob();
    else enqueueAsync(std::move(launchJob));
    tm3 = nowSec();
    if (timing) {
      HIP_OK(hipStreamSynchronize(s));
      int fl[4] = {0, 0, 0, 0};
      HIP_OK(hipMemcpy(fl, bFlag.p, sizeof(fl), hipMemcpyDeviceToHost));
      double sc3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (mb.bOut.p) HIP_OK(hipMemcpy(sc3, dbgScal, sizeof(sc3), hipMemcpyDeviceToHost));
      std::printf("[marg] m %d Lm %d; Jacobi sweeps of the last eigen-solve: %d; eigenvalues <= tol: %d (min %.3e max %.3e)\n", m, Lm,
                  fl[1], fl[2], sc3[1], sc3[2]);
      if (mb.bOut.p && sc3[7] > 0 && fl[3] != -8) {   // the eigenvalues sit behind p[] in the kernel's scratch (flag[3] == -8: the certified
                                                        // Cholesky route produced the prior -- tmp + n holds b0 / p there, no eigenvalues exist)
        const int nn = (int)sc3[7];
        std::vector<double> evh(nn);
        HIP_OK(hipMemcpy(evh.data(), dbgScal + 8 + nn, sizeof(double) * nn, hipMemcpyDeviceToHost));
        std::sort(evh.begin(), evh.end());
        std::printf("[marg] smallest eigenvalues:");
        for (int i = 0; i < std::min(nn, 8); ++i) std::printf(" %.3e", evh[i]);
        std::printf("  (tol %.3e)\n", 2.220446049250313e-16 * nn * sc3[2]);
      }
      std::printf("[marg] k_marg_final: n %d, prepare %.0f us, eigen-solve %.0f us (two-phase: phase 1 %d us; %.0f shader clocks per us), "
                  "J / e0 / J^T J %.0f us\n", (int)sc3[7], sc3[3] / 100.0, sc3[4] / 100.0, fl[3], sc3[6] / std::max(1.0, sc3[4] / 100.0),
                  sc3[5] / 100.0);
    }
    (void)anyWork;
  }
  tm4 = nowSec();
  // ---- graph update (:710-716, :788-811)
  for (uint64_t id : toMarginalize)
    if (!std::binary_search(margLandmarks.begin(), margLandmarks.end(), id)) removeBlock(id);
  int nk = 0;
  for (const PriorBlockHost& pb : kept) nk += pb.mdim;
  if (nk > 0) {
    hasPrior_ = true;
    priorBlocks_ = kept;
    priorM_ = nk;
    priorResId_ = nextResId_++;
    for (const PriorBlockHost& pb : priorBlocks_) blocks_.at(pb.id).residuals.push_back(priorResId_);
  } else {
    hasPrior_ = false;
    priorBlocks_.clear();
    priorM_ = 0;
  }
  if (reDoFixation && !states_.empty()) {
    const uint64_t firstId = states_.begin()->first;
    double information[36] = {0};
    information[35] = 1.0e14; information[0] = 1.0e14; information[7] = 1.0e14; information[14] = 1.0e14;
    Factor f;
    f.kind = F_POSE_PRIOR; f.nblk = 1; f.blocks[0] = firstId; f.m = 6;
    std::memcpy(f.meas, blocks_.at(firstId).x, 7 * sizeof(double));
    // Eigen LLT early-exit semantics
    for (int i = 0; i < 36; ++i) f.sqrtInfo[i] = 0;
    f.sqrtInfo[0] = f.sqrtInfo[7] = f.sqrtInfo[14] = std::sqrt(1.0e14);
    f.sqrtInfo[35] = 1.0e14;
    addFactor(std::move(f));
  }
  if (timing)
    std::printf("[marg] policy %.0f us, job assembly %.0f us, upload+enqueue %.0f us, device wait %.0f us, graph update %.0f us\n",
                1e6 * (tm1 - tm0), 1e6 * (tm2 - tm1), 1e6 * (tm3 - tm2), 1e6 * (tm4 - tm3), 1e6 * (nowSec() - tm4));
  return 1;
}

}  // namespace svin
