// Symmetric eigen-decomposition of one n x n matrix (n <= 128) by ONE workgroup of 1024 threads, the matrix resident in LDS:
// Householder tridiagonalisation -> Cuppen's divide and conquer with Gu / Eisenstat vectors, bottom-up from 1 x 1 leaves ->
// back-transformation.  Used by the marginalisation prior (marg.hip, M3: MarginalizationError::updateErrorComputation,
// okvis_ceres/src/MarginalizationError.cpp:725-758, which calls Eigen::SelfAdjointEigenSolver = tridiagonalisation +
// implicit QL; the decomposition itself is third-party arithmetic, what the reference fixes is the use of its result).
//
// Why not Jacobi (rounds 1-4): a one-sided Jacobi sweep is n - 1 dependent tournament rounds and these priors take 10-13
// sweeps however they are preconditioned (110 of 117 eigenvalues sit in [1e-3, 3.4], 60 of them in five 12-fold clusters of
// the extrinsics chain): ~1 400 rounds of 1.0-1.6 us.  The tridiagonalisation is n dependent steps, and a tridiagonal
// eigenproblem splits: log2 n merge levels whose work is the secular equation (one root per lane group), the Loewner formula
// and one matrix product each.  Deflation takes the clusters out (numerically multiple eigenvalues are what divide and conquer
// deflates), orthogonality comes from Gu / Eisenstat's recomputed z (no re-orthogonalisation, no cluster heuristics).
// The algorithm is replayed in numpy by tools/sym_eig_dc_replay.py (tests/test_sym_eig_dc_host.py holds the replay against
// LAPACK; tests/test_gpu_sym_eig.py holds this code against LAPACK through svin_ba_debug_sym_eig).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace svin {
namespace symeig {

using lds_double = __attribute__((address_space(3))) double;
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kMaxN = 128;
constexpr int kThreads = 1024;
constexpr double kEps = 2.220446049250313e-16;

// Small per-phase arrays next to the n x ld image (static __shared__; the phases reuse the union)
struct Small {
  double d[kMaxN], e[kMaxN], tau[kMaxN];   // tridiagonal (d becomes the eigenvalues of the solved blocks), Householder scalars
  union {
    struct { double v[kMaxN], p[kMaxN], w[kMaxN]; } hh;                          // tridiagonalisation
    struct { double v[2][4][kMaxN]; } bt;                                        // back-transformation: two blocks of four reflectors
    struct {
      double z[kMaxN], dcur[kMaxN], ztil[kMaxN];     // per column of the merge: z, d after the deflation rotations, z-hat (0: deflated)
      double cd[kMaxN], cz[kMaxN];                   // compact (non-deflated, ascending) poles and weights of each block, at [lo, lo + K)
      double rtau[kMaxN], cinv[kMaxN], newd[kMaxN];  // per root: tau, 1 / |v_j|; eigenvalues of the merged block before the final sort
      double oorgd[kMaxN], otau[kMaxN], oinv[kMaxN], dnext[kMaxN];   // per OUTPUT column
      double rotC[kMaxN], rotS[kMaxN];
      double ds[kMaxN], zs[kMaxN];                   // d and z in the merged (ascending) order of each pair
      int sorted[kMaxN], kind[kMaxN], ndl[kMaxN], dfl[kMaxN], rorg[kMaxN];
      int okind[kMaxN], osrc[kMaxN], rotP[kMaxN], rotQ[kMaxN], K[kMaxN], nrot[kMaxN];
    } dc;
  } u;
  double tblk[kMaxN / 4][10];   // the 4 x 4 upper-triangular T of every block of four reflectors (compact WY)
  int bad;
  long long stamp[80];   // 100 MHz wall-clock stamps of the stages (SVIN_SYMEIG_TIMING builds: tools/symeig_time.py)
  int nstamp;
};

// one instance per kernel that uses the solver; named directly (not passed by reference) so that every access is a DS instruction:
// through a generic reference the compiler emits FLAT loads
__shared__ Small gS;

#ifdef SVIN_SYMEIG_TIMING
#define SYMEIG_STAMP() do { if (threadIdx.x == 0 && gS.nstamp < 80) gS.stamp[gS.nstamp++] = wall_clock64(); } while (0)
#else
#define SYMEIG_STAMP() do { } while (0)
#endif

__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int kCtrl>
__device__ __forceinline__ double dppMov(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_mov_dpp(lo, kCtrl, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, kCtrl, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
template <int kCtrl>
__device__ __forceinline__ int dppMovI(int v) { return __builtin_amdgcn_mov_dpp(v, kCtrl, 0xf, 0xf, false); }
// Reductions over aligned groups of 8 lanes, the result in every lane of the group and BIT-IDENTICAL in all of them (each step
// combines the same two operands in either order, and + * max are commutative): the lanes of a group take the same branches.
// row_half_mirror = 0x141 (lane i of a half row reads lane 7 - i), quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E.
__device__ __forceinline__ double sum8(double v) {
  v += dppMov<0x141>(v);
  v += dppMov<0xB1>(v);
  v += dppMov<0x4E>(v);
  return v;
}
__device__ __forceinline__ double prod8(double v) {
  v *= dppMov<0x141>(v);
  v *= dppMov<0xB1>(v);
  v *= dppMov<0x4E>(v);
  return v;
}
__device__ __forceinline__ double max8(double v) {
  v = fmax(v, dppMov<0x141>(v));
  v = fmax(v, dppMov<0xB1>(v));
  v = fmax(v, dppMov<0x4E>(v));
  return v;
}
__device__ __forceinline__ int min8i(int v) {
  v = min(v, dppMovI<0x141>(v));
  v = min(v, dppMovI<0xB1>(v));
  v = min(v, dppMovI<0x4E>(v));
  return v;
}
// the same over aligned groups of G = 1, 2, 4 or 8 lanes
template <int G> __device__ __forceinline__ double sumG(double v) {
  if (G == 8) return sum8(v);
  if (G >= 2) v += dppMov<0xB1>(v);
  if (G == 4) v += dppMov<0x4E>(v);
  return v;
}
template <int G> __device__ __forceinline__ double maxG(double v) {
  if (G == 8) return max8(v);
  if (G >= 2) v = fmax(v, dppMov<0xB1>(v));
  if (G == 4) v = fmax(v, dppMov<0x4E>(v));
  return v;
}
template <int G> __device__ __forceinline__ int minGi(int v) {
  if (G == 8) return min8i(v);
  if (G >= 2) v = min(v, dppMovI<0xB1>(v));
  if (G == 4) v = min(v, dppMovI<0x4E>(v));
  return v;
}
// 1 / d by v_rcp_f64 and two Newton steps (the IEEE division sequence is ~4 x as many instructions; these loops do K^2 of them)
__device__ __forceinline__ double fastRcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// sum over the wave, in every lane (row rotations + the four row totals through v_readlane)
__device__ __forceinline__ double readlaneD(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double waveSumAll(double v) {
  v += dppMov<0x128>(v);   // row_ror:8, 4, 2, 1
  v += dppMov<0x124>(v);
  v += dppMov<0x122>(v);
  v += dppMov<0x121>(v);
  return (readlaneD(v, 0) + readlaneD(v, 16)) + (readlaneD(v, 32) + readlaneD(v, 48));
}

__device__ __forceinline__ double waveMaxAll(double v) {
  v = fmax(v, dppMov<0x128>(v));
  v = fmax(v, dppMov<0x124>(v));
  v = fmax(v, dppMov<0x122>(v));
  v = fmax(v, dppMov<0x121>(v));
  return fmax(fmax(readlaneD(v, 0), readlaneD(v, 16)), fmax(readlaneD(v, 32), readlaneD(v, 48)));
}
// LDS traffic of ONE wave seen by its own lanes in program order (the lanes run in lockstep; the compiler must not move
// accesses across)
__device__ __forceinline__ void waveLdsFence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------- stage 1: tridiagonalisation
// Householder reduction of the symmetric matrix in Q (full storage, row-major, leading dimension ld) to tridiagonal form,
// unblocked (LAPACK dsytd2, lower): per column k one reflector H_k = I - tau v v^T, v = (1, A[k+2.., k] / (alpha - beta)),
// A22 <- A22 - v w^T - w v^T with w = p - (tau / 2)(p.v) v, p = tau A22 v.  d, e, tau go to S; v stays below the sub-diagonal
// of column k.  Four LDS-only barriers per column.
__device__ __forceinline__ void tridiagonalize(lds_double* Q, int n, int ld) {
  Small& S = gS;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (int k = 0; k + 2 < n; ++k) {
    const int m = n - k - 1;
    if (k == 0 || k == 58) SYMEIG_STAMP();
    if (wave == 0) {
      double s = 0;
      for (int i = k + 2 + lane; i < n; i += 64) { const double x = Q[i * ld + k]; s += x * x; }
      s = waveSumAll(s);
      const double alpha = Q[(k + 1) * ld + k];
      double beta = alpha, tk = 0.0, scale = 0.0;
      if (s != 0.0) {
        beta = -copysign(sqrt(alpha * alpha + s), alpha);
        tk = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      for (int i = k + 1 + lane; i < n; i += 64) {
        const double v = (i == k + 1) ? 1.0 : Q[i * ld + k] * scale;
        S.u.hh.v[i] = v;
        if (i > k + 1) Q[i * ld + k] = v;
      }
      if (lane == 0) { S.tau[k] = tk; S.e[k] = beta; S.d[k] = Q[k * ld + k]; }
    }
    ldsBarrier();
    if (k == 0 || k == 58) SYMEIG_STAMP();
    const double tk = S.tau[k];
    if (tk != 0.0) {   // (uniform)
      const int r = t >> 3, sub = t & 7, i = k + 1 + r;
      {
        double s = 0;
        if (r < m)
          for (int j = k + 1 + sub; j < n; j += 8) s += Q[i * ld + j] * S.u.hh.v[j];
        s = sum8(s);
        if (r < m && sub == 0) S.u.hh.p[i] = tk * s;
      }
      ldsBarrier();
      if (k == 0 || k == 58) SYMEIG_STAMP();
      if (wave == 0) {
        double s = 0;
        for (int q = k + 1 + lane; q < n; q += 64) s += S.u.hh.p[q] * S.u.hh.v[q];
        s = waveSumAll(s);
        const double al = -0.5 * tk * s;
        for (int q = k + 1 + lane; q < n; q += 64) S.u.hh.w[q] = S.u.hh.p[q] + al * S.u.hh.v[q];
      }
      ldsBarrier();
      if (k == 0 || k == 58) SYMEIG_STAMP();
      if (r < m) {
        const double vi = S.u.hh.v[i], wi = S.u.hh.w[i];
        for (int j = k + 1 + sub; j < n; j += 8) Q[i * ld + j] -= vi * S.u.hh.w[j] + wi * S.u.hh.v[j];
      }
      ldsBarrier();
      if (k == 0 || k == 58) SYMEIG_STAMP();
    }
  }
  if (t == 0) {
    if (n >= 2) {
      S.d[n - 2] = Q[(n - 2) * ld + (n - 2)];
      S.e[n - 2] = Q[(n - 1) * ld + (n - 2)];
      S.tau[n - 2] = 0.0;
    }
    S.d[n - 1] = Q[(n - 1) * ld + (n - 1)];
    S.e[n - 1] = 0.0;
    S.tau[n - 1] = 0.0;
  }
  ldsBarrier();
}

// ---------------------------------------------------------------------------------------------- stage 2: divide and conquer
// step x with c + S / (dI - x) + R / (dJ - x) = 0 and lo < tau + x < hi; NaN if there is none
__device__ __forceinline__ double quadRootIn(double c, double S, double dI, double R, double dJ, double lo, double hi, double tau) {
  const double a = c, b = -(c * (dI + dJ) + S + R), cc = c * dI * dJ + S * dJ + R * dI;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if (a == 0.0) {
    const double x = (b != 0.0) ? cc / (-b) : nan;
    return (x == x && lo < tau + x && tau + x < hi) ? x : nan;
  }
  const double disc = b * b - 4.0 * a * cc;
  if (!(disc >= 0.0)) return nan;
  const double q = -0.5 * (b + copysign(sqrt(disc), b));
  const double x1 = q / a, x2 = (q != 0.0) ? cc / q : nan;
  if (x1 == x1 && lo < tau + x1 && tau + x1 < hi) return x1;
  if (x2 == x2 && lo < tau + x2 && tau + x2 < hi) return x2;
  return nan;
}

// Root i (0-based, ascending) of 1 + rho sum_k z_k^2 / (d_k - lambda) over the K compact poles cd[0..K) (strictly ascending)
// with weights cz[k]^2 > 0: lambda = cd[org] + tau, org the nearer of the two poles that bracket the root (the last root: the
// last pole), so that d_k - lambda = (cd[k] - cd[org]) - tau carries no cancellation.  Executed by an aligned group of 8 lanes
// (sub = lane in the group: the pole sums are split over them, every scalar decision is taken redundantly and identically).
// Each step keeps the origin pole with its exact weight and models everything else as r + R / (d_q - x), fitted to value and
// slope, q = the pole that dominates that slope (dlaed4's fixed-weight scheme with the fitted pole chosen by dominance); the
// bracket is kept, a step that leaves it is replaced by regula falsi (Illinois).  `active` = this group has a root to find;
// inactive groups run along (the reductions need every lane) on harmless numbers.
template <int G>
__device__ __forceinline__ void secularRoot(bool active, int i, int K, const double* cd, const double* cz, double rho, int sub,
                            int& orgOut, double& tauOut) {
  if (!active) { i = 0; K = 1; }
  int org = 0, I = i;
  double tau = 0.0, lo = 0.0, hi = 0.0;
  const bool last = (i == K - 1);
  bool done = !active;
  if (K == 1) {
    const double z = active ? (double)cz[0] : 1.0;
    tau = rho * z * z;
    done = true;
  } else if (!last) {
    const int J = i + 1;
    const double dI0 = cd[I], gap = cd[J] - dI0, mid = 0.5 * gap;
    double s = 0;
    for (int k = sub; k < K; k += G)
      if (k != I && k != J) { const double z = cz[k]; s += z * z * fastRcp((cd[k] - dI0) - mid); }
    s = sumG<G>(s);
    const double zI = cz[I], zJ = cz[J];
    const double rest = 1.0 + rho * s, SI = rho * zI * zI, SJ = rho * zJ * zJ;
    const double f = rest + SI / (-mid) + SJ / (gap - mid);
    double dI, dJ;
    if (f > 0) { org = I; lo = 0.0; hi = mid; dI = 0.0; dJ = gap; }
    else { org = J; lo = -mid; hi = 0.0; dI = -gap; dJ = 0.0; }
    const double x = quadRootIn(rest, SI, dI, SJ, dJ, lo, hi, 0.0);
    tau = (x == x) ? x : 0.5 * (lo + hi);
  } else {
    org = K - 1;
    const double dO = cd[org];
    double s2 = 0;
    for (int k = sub; k < K; k += G) { const double z = cz[k]; s2 += z * z; }
    s2 = sumG<G>(s2);
    lo = 0.0; hi = rho * s2;
    const double mid = 0.5 * hi;
    double s = 0;
    for (int k = sub; k < K - 2; k += G) { const double z = cz[k]; s += z * z * fastRcp((cd[k] - dO) - mid); }
    s = sumG<G>(s);
    const double zA = cz[K - 2], zB = cz[K - 1];
    const double x = quadRootIn(1.0 + rho * s, rho * zA * zA, cd[K - 2] - dO, rho * zB * zB, 0.0, lo, hi, 0.0);
    tau = (x == x) ? x : mid;
  }
  const double dOrg = cd[org], zO = cz[org], So = rho * zO * zO;
  double flo = 0.0, fhi = 0.0;
  bool haveLo = false, haveHi = false;
  int side = 0;
  for (int it = 0; it < 48; ++it) {
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;   // (wave-uniform)
    double psi = 0, phi = 0, dpsi = 0, dphi = 0, best = -1.0;
    int bestk = 0x7fffffff;
    for (int k = sub; k < K; k += G) {
      const double den = (cd[k] - dOrg) - tau;
      const double r = fastRcp(den), z = cz[k];
      const double tt = z * z * r, t2 = tt * r;
      if (k <= I || last) { psi += tt; dpsi += t2; }
      else { phi += tt; dphi += t2; }
      if (k != org && t2 > best) { best = t2; bestk = k; }
    }
    psi = rho * sumG<G>(psi); phi = rho * sumG<G>(phi); dpsi = rho * sumG<G>(dpsi); dphi = rho * sumG<G>(dphi);
    const double bmax = maxG<G>(best);
    bestk = minGi<G>(best == bmax ? bestk : 0x7fffffff);
    if (done) continue;
    const double f = 1.0 + psi + phi;
    const double erretm = 8.0 * (fabs(psi) + fabs(phi)) + 2.0 + fabs(tau) * (dpsi + dphi);
    if (fabs(f) <= kEps * erretm || !(f == f)) { done = true; continue; }
    if (f > 0) {
      hi = tau; fhi = f; haveHi = true;
      if (side == 1 && haveLo) flo *= 0.5;
      side = 1;
    } else {
      lo = tau; flo = f; haveLo = true;
      if (side == -1 && haveHi) fhi *= 0.5;
      side = -1;
    }
    if (hi - lo <= 4.0 * kEps * fmax(fabs(lo), fabs(hi))) { done = true; continue; }
    const double dO = -tau;
    const double dq = (cd[bestk < K ? bestk : org] - dOrg) - tau;
    const double rO = fastRcp(dO);
    const double w = psi + phi - So * rO, dw = dpsi + dphi - So * rO * rO;
    const double R = dw * dq * dq, r0 = w - dw * dq;
    const double x = (bestk < K) ? quadRootIn(1.0 + r0, So, dO, R, dq, lo, hi, tau) : __longlong_as_double(0x7ff8000000000000LL);
    double nw = tau + x;
    if (!(x == x) || !(lo < nw && nw < hi)) {
      nw = 0.5 * (lo + hi);
      if (haveLo && haveHi && fhi != flo) {
        const double rf = lo - flo * (hi - lo) / (fhi - flo);
        if (lo < rf && rf < hi) nw = rf;
      }
    }
    tau = nw;
  }
  orgOut = org;
  tauOut = tau;
}

// Deflation of the pairs of 16 and more poles (the four top levels at n = 117), one WAVE per pair.  What the one-lane walk
// spends its time on is walking: 117 dependent steps at the top although a handful of poles deflate.  Here the lanes (two
// merged positions each) find the poles without weight, compact the others, and test every neighbouring pair of those against
// its ORIGINAL values in parallel; only where such a test says "deflate" does lane 0 walk on, as dlaed2 would, until a pair
// does not deflate (from there on the original values -- and so the parallel tests -- hold again).
__device__ __forceinline__ void deflateWide(int n, int b) {
  Small& S = gS;
  auto& D = S.u.dc;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, two = 2 * b;
  const int P = (n - b + two - 1) / two;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int p = wave; p < P; p += kThreads / 64) {
    const int lo = p * two, mid = lo + b, hi = min(lo + two, n), m = hi - lo;
    const double rho = 2.0 * fabs(S.e[mid - 1]);
    const int s0 = lane, s1 = lane + 64;
    const bool v0 = s0 < m, v1 = s1 < m;
    const int idx0 = v0 ? D.sorted[lo + s0] : 0, idx1 = v1 ? D.sorted[lo + s1] : 0;
    const double d0 = v0 ? (double)D.ds[lo + s0] : 0.0, z0 = v0 ? (double)D.zs[lo + s0] : 0.0;
    const double d1 = v1 ? (double)D.ds[lo + s1] : 0.0, z1 = v1 ? (double)D.zs[lo + s1] : 0.0;
    const double dmax = waveMaxAll(fmax(fabs(d0), fabs(d1))), zmax = waveMaxAll(fmax(fabs(z0), fabs(z1)));
    const double tol = 8.0 * kEps * fmax(dmax, zmax);
    const bool all = rho * zmax <= tol;
    const bool sm0 = v0 && (all || rho * fabs(z0) <= tol), sm1 = v1 && (all || rho * fabs(z1) <= tol);
    const unsigned long long S0 = __builtin_amdgcn_ballot_w64(sm0), S1 = __builtin_amdgcn_ballot_w64(sm1);
    const unsigned long long B0 = __builtin_amdgcn_ballot_w64(v0 && !sm0), B1 = __builtin_amdgcn_ballot_w64(v1 && !sm1);
    const int nSmall = __popcll(S0) + __popcll(S1), M = __popcll(B0) + __popcll(B1);
    // poles without weight: deflated as they are; the others compacted (ascending) into scratch that later steps overwrite
    int* cIdx = D.rorg + lo; double* cD = D.rtau + lo; double* cZ = D.cinv + lo;
    if (sm0) { D.kind[idx0] = 1; D.dfl[lo + __popcll(S0 & lt)] = idx0; }
    if (sm1) { D.kind[idx1] = 1; D.dfl[lo + __popcll(S0) + __popcll(S1 & lt)] = idx1; }
    if (v0 && !sm0) { const int c = __popcll(B0 & lt); cIdx[c] = idx0; cD[c] = d0; cZ[c] = z0; }
    if (v1 && !sm1) { const int c = __popcll(B0) + __popcll(B1 & lt); cIdx[c] = idx1; cD[c] = d1; cZ[c] = z1; }
    waveLdsFence();
    auto pairDeflates = [&](int i) -> bool {   // compact neighbours (i - 1, i), original values
      const double zP = cZ[i - 1], zI = cZ[i];
      const double rt = fastRcp(sqrt(zI * zI + zP * zP));
      return fabs((cD[i] - cD[i - 1]) * (zI * rt) * (zP * rt)) <= tol;
    };
    const int i0 = lane + 1, i1 = lane + 65;
    const unsigned long long F0 = __builtin_amdgcn_ballot_w64(i0 < M && pairDeflates(i0));   // bit q: pair (q, q + 1)
    const unsigned long long F1 = __builtin_amdgcn_ballot_w64(i1 < M && pairDeflates(i1));   // bit q: pair (q + 64, q + 65)
    unsigned long long G0 = 0, G1 = 0;   // compact poles deflated into their right-hand neighbour
    int nr = 0, nd2 = 0;
    if ((F0 | F1) != 0ull) {
      if (lane == 0) {
        auto nextFlag = [&](int from) -> int {   // first flagged pair index i >= from (pair (i - 1, i)), M if none
          if (from <= 64) {
            const unsigned long long r = F0 >> (from - 1);
            if (r) return from + __builtin_ctzll(r);
            from = 65;
          }
          if (from - 65 < 64) {
            const unsigned long long r = F1 >> (from - 65);
            if (r) return from + __builtin_ctzll(r);
          }
          return M;
        };
        int i = nextFlag(1);
        while (i < M) {
          int pj = i - 1, cur = i;
          double dP = cD[pj], zP = cZ[pj];
          while (cur < M) {
            const double dI = cD[cur], zI = cZ[cur];
            const double tau = sqrt(zI * zI + zP * zP), rt = fastRcp(tau);
            const double cs = zI * rt, sn = -zP * rt, tt = dI - dP;
            if (!(fabs(tt * cs * sn) <= tol)) break;
            const int colP = cIdx[pj];
            D.rotP[lo + nr] = colP; D.rotQ[lo + nr] = cIdx[cur]; D.rotC[lo + nr] = cs; D.rotS[lo + nr] = sn; ++nr;
            D.dcur[colP] = dP * cs * cs + dI * sn * sn;
            D.z[colP] = 0.0;
            D.kind[colP] = 1; D.dfl[lo + nSmall + nd2++] = colP;
            if (pj < 64) G0 |= 1ull << pj; else G1 |= 1ull << (pj - 64);
            dP = dP * sn * sn + dI * cs * cs; zP = tau; pj = cur; ++cur;
          }
          cD[pj] = dP; cZ[pj] = zP;
          i = nextFlag(cur + 1);
        }
      }
      G0 = __shfl(G0, 0); G1 = __shfl(G1, 0);   // (64-bit shuffles: two ds_bpermute each)
      nr = __shfl(nr, 0); nd2 = __shfl(nd2, 0);
      waveLdsFence();
    }
    // the survivors, in order
    const bool sv0 = lane < M && !((G0 >> lane) & 1ull), sv1 = lane + 64 < M && !((G1 >> lane) & 1ull);
    const unsigned long long V0 = __builtin_amdgcn_ballot_w64(sv0), V1 = __builtin_amdgcn_ballot_w64(sv1);
    const int K = __popcll(V0) + __popcll(V1);
    // (read everything the survivors need before cd / cz are written: cD / cZ / cIdx alias later-step arrays, not these)
    if (sv0) {
      const int q = __popcll(V0 & lt), col = cIdx[lane];
      const double d = cD[lane], z = cZ[lane];
      D.ndl[lo + q] = col; D.cd[lo + q] = d; D.cz[lo + q] = z; D.kind[col] = 0; D.dcur[col] = d; D.z[col] = z;
    }
    if (sv1) {
      const int q = __popcll(V0) + __popcll(V1 & lt), col = cIdx[lane + 64];
      const double d = cD[lane + 64], z = cZ[lane + 64];
      D.ndl[lo + q] = col; D.cd[lo + q] = d; D.cz[lo + q] = z; D.kind[col] = 0; D.dcur[col] = d; D.z[col] = z;
    }
    waveLdsFence();
    const int nd = nSmall + nd2;
    for (int q = lane; q < nd; q += 64) { const int c = D.dfl[lo + q]; D.newd[lo + K + q] = D.dcur[c]; D.ztil[c] = 0.0; }
    if (lane == 0) { D.K[lo] = K; D.nrot[lo] = nr; }
  }
}

// One level of the bottom-up recursion: every pair of solved neighbouring blocks [lo, lo + b) | [lo + b, min(lo + 2b, n)) is
// merged.  T restricted to the pair = diag(T1', T2') + |e_k| w w^T, w = e_k-hat + sign(e_k) e_(k+1)-hat, k = lo + b - 1 (the
// tear was subtracted from d_k, d_(k+1) before the leaves were "solved"), so in the basis of the two solved halves the pair is
// D + rho z z^T with z = (last row of Q1 | sign * first row of Q2) / sqrt 2, rho = 2 |e_k|  (LAPACK dlaed1 .. dlaed3).
__device__ __forceinline__ void mergeLevel(lds_double* Q, int n, int ld, int b) {
  Small& S = gS;
  const int t = threadIdx.x;
  auto& D = S.u.dc;
  const int two = 2 * b;
  // ---- 1: z, the merged order of the poles
  if (t < n) {
    const int c = t, lo = (c / two) * two, mid = lo + b, hi = min(lo + two, n);
    if (mid < n) {
      const int k = mid - 1;
      const double sg = (S.e[k] >= 0.0) ? 1.0 : -1.0;
      const double z = (c < mid) ? Q[k * ld + c] : sg * Q[(k + 1) * ld + c];
      D.z[c] = z * 0.70710678118654752440;
      const double dc = S.d[c];
      D.dcur[c] = dc;
      // both halves are sorted: own position + the number of elements of the other half in front (ties: the lower index first)
      int rank, a0, a1;
      if (c < mid) { rank = c - lo; a0 = mid; a1 = hi; } else { rank = c - mid; a0 = lo; a1 = mid; }
      {
        int l0 = a0, l1 = a1;   // first q in [a0, a1) that is NOT in front of c
        while (l0 < l1) {
          const int q = (l0 + l1) >> 1;
          const double dq = S.d[q];
          if (dq < dc || (dq == dc && q < c)) l0 = q + 1; else l1 = q;
        }
        rank += l0 - a0;
      }
      D.sorted[lo + rank] = c;
      D.ds[lo + rank] = dc;
      D.zs[lo + rank] = z * 0.70710678118654752440;
    }
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 2: deflation (dlaed2), one lane per pair of blocks, walking the merged order.  The state of the walk (the last pole that
  // still carries weight: index, d, z) lives in registers and the next element is requested before the current one is
  // processed, so a step is ~40 dependent instructions instead of four dependent LDS round trips.
  if (two >= 16) deflateWide(n, b);
  else if (t < n && (t % two) == 0 && t + b < n) {
    const int lo = t, mid = lo + b, hi = min(lo + two, n), m = hi - lo, k = mid - 1;
    const double rho = 2.0 * fabs(S.e[k]);
    double dmax = 0, zmax = 0;
    for (int q = lo; q < hi; ++q) { dmax = fmax(dmax, fabs(D.ds[q])); zmax = fmax(zmax, fabs(D.zs[q])); }
    const double tol = 8.0 * kEps * fmax(dmax, zmax);
    int K = 0, nd = 0, nr = 0;
    if (rho * zmax <= tol) {
      for (int s = 0; s < m; ++s) { const int idx = D.sorted[lo + s]; D.kind[idx] = 1; D.dfl[lo + nd++] = idx; }
    } else {
      int pj = -1;
      double dP = 0, zP = 0;
      int idxN = D.sorted[lo];
      double dN = D.ds[lo], zN = D.zs[lo];
      for (int s = 0; s < m; ++s) {
        const int idx = idxN;
        const double dI = dN, zI = zN;
        if (s + 1 < m) { idxN = D.sorted[lo + s + 1]; dN = D.ds[lo + s + 1]; zN = D.zs[lo + s + 1]; }
        if (rho * fabs(zI) <= tol) { D.kind[idx] = 1; D.dfl[lo + nd++] = idx; continue; }
        if (pj < 0) { pj = idx; dP = dI; zP = zI; continue; }
        const double tau = sqrt(zI * zI + zP * zP), rt = fastRcp(tau);   // (|z| <= 1: no overflow to guard against)
        const double cs = zI * rt, sn = -zP * rt, tt = dI - dP;
        if (fabs(tt * cs * sn) <= tol) {   // two close poles: rotate the weight of pj into this one
          D.rotP[lo + nr] = pj; D.rotQ[lo + nr] = idx; D.rotC[lo + nr] = cs; D.rotS[lo + nr] = sn; ++nr;
          D.dcur[pj] = dP * cs * cs + dI * sn * sn;
          D.z[pj] = 0.0;
          D.kind[pj] = 1; D.dfl[lo + nd++] = pj;
          dP = dP * sn * sn + dI * cs * cs; zP = tau; pj = idx;
        } else {
          D.kind[pj] = 0; D.ndl[lo + K] = pj; D.cd[lo + K] = dP; D.cz[lo + K] = zP; D.dcur[pj] = dP; D.z[pj] = zP; ++K;
          pj = idx; dP = dI; zP = zI;
        }
      }
      if (pj >= 0) { D.kind[pj] = 0; D.ndl[lo + K] = pj; D.cd[lo + K] = dP; D.cz[lo + K] = zP; D.dcur[pj] = dP; D.z[pj] = zP; ++K; }
    }
    for (int q = 0; q < nd; ++q) { const int c = D.dfl[lo + q]; D.newd[lo + K + q] = D.dcur[c]; D.ztil[c] = 0.0; }
    D.K[lo] = K; D.nrot[lo] = nr;
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 3: the deflation rotations on the columns of Q (one thread per row, the list in order)
  if (t < n) {
    const int r = t, lo = (r / two) * two;
    if (lo + b < n) {
      const int nr = D.nrot[lo];
      for (int q = 0; q < nr; ++q) {
        const int p = D.rotP[lo + q], qq = D.rotQ[lo + q];
        const double cs = D.rotC[lo + q], sn = D.rotS[lo + q];
        const double x = Q[r * ld + p], y = Q[r * ld + qq];
        Q[r * ld + p] = cs * x + sn * y;
        Q[r * ld + qq] = cs * y - sn * x;
      }
    }
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 4: secular equation.  The scalar part of an iteration (~250 instructions: the quadratic, the safeguards) is executed
  // by every wave that holds a root: one lane per root up to 8 poles, two for 16, eight beyond (measured at n = 117, us per level
  // with 8 lanes everywhere: 3.5 10.3 10.0 12.8 15.7 23.0 19.6; with one lane up to 16 poles and two beyond: 1.9 4.7 7.6 14.9 22.5
  // 57.7 41.2; four lanes beyond 16 poles: 25 31 at the two top levels against 18 23 -- a lane's pole sum is a dependent chain,
  // long sums want many lanes, short ones few waves)
  {
    auto run = [&](auto lprTag) {
      constexpr int G = decltype(lprTag)::value;
      const int g = t / G, sub = t % G;
      const int lo = (g / two) * two;
      const bool merged = g < n && lo + b < n;
      const int K = merged ? gS.u.dc.K[lo] : 0, i = g - lo;
      const bool active = merged && i < K;
      const double rho = merged ? 2.0 * fabs(gS.e[lo + b - 1]) : 1.0;
      int org; double tau;
      secularRoot<G>(active, i, K, gS.u.dc.cd + (merged ? lo : 0), gS.u.dc.cz + (merged ? lo : 0), rho, sub, org, tau);
      if (active && sub == 0) { gS.u.dc.rorg[g] = org; gS.u.dc.rtau[g] = tau; }
    };
    if (two <= 8) run(std::integral_constant<int, 1>());
    else if (two == 16) run(std::integral_constant<int, 2>());
    else run(std::integral_constant<int, 8>());
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 5: Gu / Eisenstat: the weights for which the computed roots are the exact eigenvalues
  {
    const int g = t >> 3, sub = t & 7;
    const int lo = (g / two) * two;
    const bool merged = g < n && lo + b < n;
    const int K = merged ? D.K[lo] : 0, k = g - lo;
    const bool active = merged && k < K;
    double prod = 1.0;
    if (active) {
      const double dk = D.cd[lo + k];
      for (int j = sub; j < K; j += 8) {
        const double num = (D.cd[lo + D.rorg[lo + j]] - dk) + D.rtau[lo + j];   // lambda_j - d_k
        prod *= (j == k) ? num : num * fastRcp(D.cd[lo + j] - dk);
      }
    }
    prod = prod8(prod);
    if (active && sub == 0) {
      const double rho = 2.0 * fabs(S.e[lo + b - 1]);
      D.ztil[D.ndl[lo + k]] = copysign(sqrt(fabs(prod) / rho), D.cz[lo + k]);
    }
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 6: the norms of the new vectors, the new eigenvalues
  {
    const int g = t >> 3, sub = t & 7;
    const int lo = (g / two) * two;
    const bool merged = g < n && lo + b < n;
    const int K = merged ? D.K[lo] : 0, j = g - lo;
    const bool active = merged && j < K;
    double s = 0;
    if (active) {
      const double dO = D.cd[lo + D.rorg[g]], tau = D.rtau[g];
      for (int k = sub; k < K; k += 8) {
        const double v = D.ztil[D.ndl[lo + k]] * fastRcp((D.cd[lo + k] - dO) - tau);
        s += v * v;
      }
    }
    s = sum8(s);
    if (active && sub == 0) {
      D.cinv[g] = 1.0 / sqrt(s);
      D.newd[g] = D.cd[lo + D.rorg[g]] + D.rtau[g];
    }
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 7: where each new column goes (ascending eigenvalues), what it is made of
  if (t < n) {
    const int s = t, lo = (s / two) * two, hi = min(lo + two, n);
    if (lo + b < n) {
      const int K = D.K[lo], q = s - lo;
      const double v = D.newd[s];
      // newd = [the K roots, ascending | the deflated values in the order they were deflated]: binary search in the roots,
      // a walk over the (few) deflated ones; ties: the lower slot first
      int rank = 0;
      {
        int l0 = lo, l1 = lo + K;
        while (l0 < l1) {
          const int r = (l0 + l1) >> 1;
          const double w = D.newd[r];
          if (w < v || (w == v && r < s)) l0 = r + 1; else l1 = r;
        }
        rank = l0 - lo;
        for (int r = lo + K; r < hi; ++r) { const double w = D.newd[r]; rank += (w < v || (w == v && r < s)) ? 1 : 0; }
      }
      const int o = lo + rank;
      D.dnext[o] = v;
      if (q < K) {
        D.okind[o] = 0;
        D.oorgd[o] = D.cd[lo + D.rorg[s]];
        D.otau[o] = D.rtau[s];
        D.oinv[o] = D.cinv[s];
      } else {
        D.okind[o] = 1;
        D.osrc[o] = D.dfl[lo + q - K];
      }
    }
  }
  ldsBarrier();
  SYMEIG_STAMP();
  // ---- 8: Q <- Q [V | deflated columns] per pair of blocks, 16 x 16 tiles on v_mfma_f64_16x16x4 (A[i = l & 15][k = l >> 4],
  // B[k = l >> 4][j = l & 15], C: column l & 15, row (l >> 4) + 4 reg).  The B operand is generated on the fly: for a root column
  // z-hat_k / ((d_k - d_org) - tau) / |v|, for a deflated column a unit vector.  At most four tiles per wave, all products in
  // registers before anything is written (a row of the new Q depends on the same row of the old one only).  Pairs of more than
  // 32 columns: a wave takes one column tile and up to four row tiles, so that one generated B serves four products.
  {
    const int wave = t >> 6, l = t & 63;
    d4 acc[4];
    int tLo[4], tRt[4], tCt[4], tM[4], nTiles = 0;
    const int P = (n - b + two - 1) / two;                 // merged pairs of this level
    const int loLast = (P - 1) * two, mLast = min(loLast + two, n) - loLast;
    const int trF = (two + 15) >> 4, trL = (mLast + 15) >> 4;
    const bool wide = two > 32;
    if (wide) {   // unit = (pair, column tile, half of the row tiles): at most 16 of them for n <= 128
      int u = wave;
      for (int p = 0; p < P; ++p) {
        const int tr = (p == P - 1) ? trL : trF, nh = (tr + 3) >> 2, cnt = tr * nh;
        if (u >= 0 && u < cnt) {
          const int ct = u / nh, rh = u - ct * nh;
          for (int q = 0; q < 4; ++q)
            if (4 * rh + q < tr) { tLo[nTiles] = p * two; tRt[nTiles] = 4 * rh + q; tCt[nTiles] = ct; tM[nTiles] = (p == P - 1) ? mLast : two; ++nTiles; }
        }
        u -= cnt;
      }
    } else {
      const int perF = trF * trF, total = (P - 1) * perF + trL * trL;
      for (int id = wave; id < total && nTiles < 4; id += 16) {
        const int p = min(id / perF, P - 1), r = id - p * perF, tr = (p == P - 1) ? trL : trF;
        tLo[nTiles] = p * two; tRt[nTiles] = r / tr; tCt[nTiles] = r - (r / tr) * tr; tM[nTiles] = (p == P - 1) ? mLast : two; ++nTiles;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    auto genB = [&](int o, bool oOk, int ok, int osrc, double od, double ot, double oi, int kc, bool kOk) -> double {
      double bv = 0.0;
      if (kOk && oOk) {
        if (ok == 1) bv = (kc == osrc) ? 1.0 : 0.0;
        else if (D.kind[kc] == 0) bv = D.ztil[kc] * fastRcp((D.dcur[kc] - od) - ot) * oi;
      }
      return bv;
    };
    if (wide) {
      if (nTiles > 0) {
        const int lo = tLo[0], m = tM[0], hi = lo + m;
        const int o = lo + tCt[0] * 16 + (l & 15);
        const bool oOk = o < hi;
        const int ok = oOk ? D.okind[o] : 1, osrc = oOk ? D.osrc[o] : -1;
        const double od = oOk ? D.oorgd[o] : 0.0, ot = oOk ? D.otau[o] : 1.0, oi = oOk ? D.oinv[o] : 0.0;
        const int r0 = lo + tRt[0] * 16 + (l & 15);
        for (int kk = 0; kk < m; kk += 4) {
          const int kc = lo + kk + (l >> 4);
          const bool kOk = kc < hi;
          const double bv = genB(o, oOk, ok, osrc, od, ot, oi, kc, kOk);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nTiles) {
              const int r = r0 + 16 * q;
              const double a = (r < hi && kOk) ? (double)Q[r * ld + kc] : 0.0;
              acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[q], 0, 0, 0);
            }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < nTiles) {
          const int lo = tLo[q], m = tM[q], hi = lo + m;
          const int r = lo + tRt[q] * 16 + (l & 15), o = lo + tCt[q] * 16 + (l & 15);
          const bool rOk = r < hi, oOk = o < hi;
          const int ok = oOk ? D.okind[o] : 1, osrc = oOk ? D.osrc[o] : -1;
          const double od = oOk ? D.oorgd[o] : 0.0, ot = oOk ? D.otau[o] : 1.0, oi = oOk ? D.oinv[o] : 0.0;
          for (int kk = 0; kk < m; kk += 4) {
            const int kc = lo + kk + (l >> 4);
            const bool kOk = kc < hi;
            const double a = (rOk && kOk) ? (double)Q[r * ld + kc] : 0.0;
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, genB(o, oOk, ok, osrc, od, ot, oi, kc, kOk), acc[q], 0, 0, 0);
          }
        }
    }
    ldsBarrier();
    SYMEIG_STAMP();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nTiles) {
        const int lo = tLo[q], hi = lo + tM[q];
        const int o = lo + tCt[q] * 16 + (l & 15);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = lo + tRt[q] * 16 + (l >> 4) + 4 * rg;
          if (r < hi && o < hi) Q[r * ld + o] = acc[q][rg];
        }
      }
    if (t < n) {
      const int lo = (t / two) * two;
      if (lo + b < n) S.d[t] = D.dnext[t];
    }
  }
  ldsBarrier();
}

// The triangular factors of the compact WY form of every block of four consecutive reflectors (LAPACK dlarft, forward /
// columnwise): H_a H_(a+1) H_(a+2) H_(a+3) = I - V T V^T, T_cc = tau_c, T(0:c, c) = -tau_c T(0:c, 0:c) (V^T v_c)(0:c).  Computed
// while the reflectors still sit below the sub-diagonal of the image: 8 lanes per inner product v_c . v_c' (six per block).
__device__ __forceinline__ void tBlocks(lds_double* Q, int n, int ld) {
  Small& S = gS;
  const int t = threadIdx.x, nref = n - 2, nblk = (nref + 3) >> 2;
  if (nref <= 0) return;
  auto vAt = [&](int k, int i) -> double {   // component i of v_k
    return (k >= nref || S.tau[k] == 0.0) ? 0.0 : (i == k + 1 ? 1.0 : (i > k + 1 ? (double)Q[i * ld + k] : 0.0));
  };
  for (int base = 0; base < nblk * 6; base += kThreads / 8) {
    const int id = base + (t >> 3), sub = t & 7;
    const int blk = id / 6, pr = id - 6 * blk;
    const int c0 = pr < 3 ? 0 : (pr < 5 ? 1 : 2), c1 = pr < 3 ? pr + 1 : (pr < 5 ? pr - 1 : 3);   // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    double g = 0;
    if (blk < nblk) {
      const int k0 = 4 * blk + c0, k1 = 4 * blk + c1;
      for (int i = k1 + 1 + sub; i < n; i += 8) g += vAt(k0, i) * vAt(k1, i);
    }
    g = sum8(g);
    if (blk < nblk && sub == 0) S.tblk[blk][pr < 3 ? pr + 1 : (pr < 5 ? pr + 2 : 8)] = g;   // G01 G02 G03 -> [1..3], G12 G13 -> [5..6], G23 -> [8]
  }
  ldsBarrier();
  if (t < nblk) {
    double* T = S.tblk[t];
    auto tauAt = [&](int c) { const int k = 4 * t + c; return k < nref ? S.tau[k] : 0.0; };
    const double g01 = T[1], g02 = T[2], g03 = T[3], g12 = T[5], g13 = T[6], g23 = T[8];
    const double t0 = tauAt(0), t1 = tauAt(1), t2 = tauAt(2), t3 = tauAt(3);
    const double T00 = t0, T11 = t1, T22 = t2, T33 = t3;
    const double T01 = -t1 * T00 * g01;
    const double T02 = -t2 * (T00 * g02 + T01 * g12), T12 = -t2 * T11 * g12;
    const double T03 = -t3 * (T00 * g03 + T01 * g13 + T02 * g23), T13 = -t3 * (T11 * g13 + T12 * g23), T23 = -t3 * T22 * g23;
    T[0] = T00; T[1] = T01; T[2] = T02; T[3] = T03; T[4] = T11; T[5] = T12; T[6] = T13; T[7] = T22; T[8] = T23; T[9] = T33;
  }
  ldsBarrier();
}

// Eigen-decomposition of the symmetric matrix in Q (LDS, n x ld, full storage).  On return Q[i * ld + j] = component i of
// eigenvector j, S.d[j] = eigenvalue j (ascending).  gV: n * n doubles of global scratch (the Householder vectors wait there
// while the image is the eigenvector matrix of T); the small arrays live in gS.  All kThreads threads of the workgroup must call.  Returns false (uniformly)
// if a result is not finite.
__device__ __forceinline__ bool solve(lds_double* Q, int n, int ld, double* gV) {
  Small& S = gS;
  const int t = threadIdx.x;
  if (t == 0) { S.bad = 0; S.nstamp = 0; }
  ldsBarrier();
  SYMEIG_STAMP();
  tridiagonalize(Q, n, ld);
  tBlocks(Q, n, ld);
  SYMEIG_STAMP();
  for (int idx = t; idx < n * n; idx += kThreads) { const int i = idx / n, j = idx - i * n; gV[idx] = Q[i * ld + j]; }
  __syncthreads();
  // leaves: every adjacent pair is torn at some level, so leaf j starts as d_j - |e_(j-1)| - |e_j|
  if (t < n) S.d[t] = S.d[t] - (t > 0 ? fabs(S.e[t - 1]) : 0.0) - (t + 1 < n ? fabs(S.e[t]) : 0.0);
  for (int idx = t; idx < n * ld; idx += kThreads) { const int i = idx / ld, j = idx - i * ld; Q[idx] = (i == j) ? 1.0 : 0.0; }
  ldsBarrier();
  SYMEIG_STAMP();
  for (int b = 1; b < n; b <<= 1) mergeLevel(Q, n, ld, b);
  SYMEIG_STAMP();
  // back-transformation: X = H_0 H_1 ... H_(n-3) Z, four reflectors at a time in compact WY form: H_a .. H_(a+3) = I - V T V^T
  // (T from tBlocks()).  Eight consecutive lanes own one column of Z (the rows dealt round robin), so V^T z, T (V^T z) and the
  // update need no exchange beyond three DPP steps; the only barrier per block is the one behind the staging of the next V.
  {
    auto& B = S.u.bt;
    const int j = t >> 3, sub = t & 7;
    const int nref = n - 2, nblk = (nref + 3) >> 2;
    auto stage = [&](int blk, int buf) {   // V of block blk: v_c[i] for the rows i of the block's support, c = 0 .. 3
      const int a = 4 * blk;
      for (int idx = t; idx < 4 * n; idx += kThreads) {
        const int c = idx / n, i = idx - c * n, k = a + c;
        double v = 0.0;
        if (k < nref && S.tau[k] != 0.0) v = (i == k + 1) ? 1.0 : (i > k + 1 ? gV[i * n + k] : 0.0);
        B.v[buf][c][i] = v;
      }
    };
    if (nblk > 0) stage(nblk - 1, 0);
    ldsBarrier();
    for (int blk = nblk - 1, buf = 0; blk >= 0; --blk, buf ^= 1) {
      if (blk > 0) stage(blk - 1, buf ^ 1);
      const int a = 4 * blk;
      if (j < n) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int i = a + 1 + sub; i < n; i += 8) {
          const double q = Q[i * ld + j];
          s0 += B.v[buf][0][i] * q; s1 += B.v[buf][1][i] * q; s2 += B.v[buf][2][i] * q; s3 += B.v[buf][3][i] * q;
        }
        s0 = sum8(s0); s1 = sum8(s1); s2 = sum8(s2); s3 = sum8(s3);
        const double* T = S.tblk[blk];   // T00 T01 T02 T03 T11 T12 T13 T22 T23 T33
        const double y0 = T[0] * s0 + T[1] * s1 + T[2] * s2 + T[3] * s3, y1 = T[4] * s1 + T[5] * s2 + T[6] * s3,
                     y2 = T[7] * s2 + T[8] * s3, y3 = T[9] * s3;
        for (int i = a + 1 + sub; i < n; i += 8)
          Q[i * ld + j] -= B.v[buf][0][i] * y0 + B.v[buf][1][i] * y1 + B.v[buf][2][i] * y2 + B.v[buf][3][i] * y3;
      } else {
        (void)sum8(0.0); (void)sum8(0.0); (void)sum8(0.0); (void)sum8(0.0);
      }
      ldsBarrier();
    }
  }
  SYMEIG_STAMP();
  if (t < n) {
    const double lam = S.d[t];
    double cs = 0;
    for (int i = 0; i < n; ++i) cs += Q[i * ld + t];
    if (!(fabs(lam) < 1.0e300) || !(fabs(cs) < 1.0e300)) S.bad = 1;
  }
  __syncthreads();
  return S.bad == 0;
}

}  // namespace symeig
}  // namespace svin
