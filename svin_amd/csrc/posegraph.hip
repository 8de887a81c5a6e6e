// svin_amd global pose-graph optimisation: the numerical core of pose_graph's optimisation thread
// (/root/reference/pose_graph/src/pose_graph/PoseGraph.cpp:226-543) on gfx950 behind include/svin_pg.h.
//
//   host    keyframe list -> local problem exactly like the reference (sequential edges to the 2 / 4 predecessors of
//           the same sequence, loop edges, constant first keyframe), Levenberg-Marquardt control flow of Ceres 2.2
//           (TrustRegionMinimizer + LevenbergMarquardtStrategy, default options), one scalar read-back per iteration;
//           symbolic step once per call: the keyframe chain is cut into pieces (<= 64 keyframes) by separators =
//           w consecutive keyframes every piece (w = 2 / 4, the reach of the sequential edges) + a vertex cover of
//           the long (loop) edges, so that every piece's interior only talks to itself and to separators
//   device  k_pg_eval            one thread per edge: FourDOFError / FourDOFWeightError / PoseGraph3dErrorTerm
//                                residual, analytic minimal Jacobians (the reference uses AutoDiff: same derivatives),
//                                HuberLoss(0.1) corrector on loop edges, cost partials
//           k_pg_node            one thread per free keyframe: gradient, column norms, Jacobi scaling, J^T J block
//                                (incident edges in fixed order: deterministic)
//           k_pg_assemble_*      damped normal equations scattered into: the dense separator system, the pieces'
//                                banded interior blocks, the pieces' interior x separator coupling columns
//           k_pg_piece_factor    one workgroup per piece: banded Cholesky in LDS, forward substitution of the coupling
//                                and right-hand-side columns (one column per thread): Y = L^-1 [C | r]
//           k_pg_piece_schur     Y^T Y per piece (16x16 tiles), k_pg_sep_gather subtracts them from the separator
//                                system in a fixed order (deterministic)
//           solve                the dense reduced-system solver of the BA backend on the separator system (LDS-
//                                resident or multi-workgroup blocked Cholesky on v_mfma_f64_16x16x4_f64, kernels.hip)
//           k_pg_piece_back      interior unknowns: x_I = L^-T (y_r - Y_C x_S)
//           k_pg_model / k_pg_plus / k_pg_reduce   model cost change, candidate = Plus(x, delta), norms
// Graphs with <= 128 free keyframes skip the pieces (every keyframe is a separator: one dense solve).
#include "kernels.hpp"
#include "dmath.hpp"
#include "../../include/svin_pg.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace svin {
namespace pg {

#define PG_HIP_OK(expr)                                                                             \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

constexpr double kPgPi = 3.14159265358979323846;

struct PgPiece {
  int rows, cols, ld, colPtr;          // interior unknowns, coupling columns + 1 (rhs last), row stride of Y / S_p
  int rowPtr, pad;
  long long bandOff, yOff, sOff;       // offsets (doubles) into band / Y / Sp
};
struct PgDev {
  int nn, ne, n, m, six, D, R;
  int BW, nS, nPieces, maxRows;         // band half-width (elements), separator unknowns, pieces, max interior rows
  double *yaw, *pitch, *roll, *t, *q;   // current point (yaw in degrees; q = [x y z w])
  double *yawC, *tC, *qC;               // candidate
  int* off;                             // tangent offset per node, -1 = constant
  int *ea, *eb, *eloop;
  double *et, *eyaw, *epitch, *eroll, *eq, *esq;
  double *res, *Ja, *Jb;                // robustified residuals and Jacobian blocks (R x D per edge and side)
  int *nodePtr, *nodeEdge;              // incident edges per node: edge * 2 + side (0 = a, 1 = b)
  double *scale, *g, *colsq, *nodeBlk, *y;
  double *partial, *scal;
  // partition
  int* sepOff;                          // per node: offset in the separator system, -1 = interior / constant
  int *nodePiece, *nodeRow;             // per node: piece and first interior row, -1 = not interior
  const PgPiece* pieces;
  int4* edgeDst;                        // {kind | rowIsB << 4, piece, rowOff, colOff}; kind 0 none, 1 sep-sep, 2 band, 3 coupling
  int *colSep, *rowTan;                 // per piece column -> separator offset; per piece row -> tangent index
  int4* tileWork;                       // {piece, ti, tj, 0}
  int *gPtr; int2* gDst; int4* gSrc;    // separator block <- sum of piece Schur blocks {piece, rowOff, colOff, 0}
  int *rPtr; int2* rSrc;                // separator node rhs <- {piece, colOff}
  double *band, *Y, *Sp, *HS, *rhsS, *yS;
  int *fail, *ticket;
};
enum PgScal : int { PG_COST = 0, PG_GRADMAX = 1, PG_MODEL = 2, PG_STEP2 = 3, PG_X2 = 4, PG_NSCAL = 8 };
constexpr int kPgMaxPartials = 4096;

__device__ __forceinline__ double pgNormalizeAngle(double deg) {  // PoseGraph.h:85-93
  if (deg > 180.0) return deg - 360.0;
  if (deg < -180.0) return deg + 360.0;
  return deg;
}
// PoseGraph.h:110-127 (Rz Ry Rx, degrees) and its derivative with respect to the yaw angle (degrees)
__device__ __forceinline__ void pgYpr2R(double yaw, double pitch, double roll, double* R, double* dR) {
  const double y = yaw / 180.0 * kPgPi, p = pitch / 180.0 * kPgPi, r = roll / 180.0 * kPgPi;
  const double cy = cos(y), sy = sin(y), cp = cos(p), sp = sin(p), cr = cos(r), sr = sin(r);
  R[0] = cy * cp; R[1] = -sy * cr + cy * sp * sr; R[2] = sy * sr + cy * sp * cr;
  R[3] = sy * cp; R[4] = cy * cr + sy * sp * sr;  R[5] = -cy * sr + sy * sp * cr;
  R[6] = -sp;     R[7] = cp * sr;                 R[8] = cp * cr;
  const double k = kPgPi / 180.0;
  dR[0] = -sy * cp * k; dR[1] = (-cy * cr - sy * sp * sr) * k; dR[2] = (cy * sr - sy * sp * cr) * k;
  dR[3] = cy * cp * k;  dR[4] = (-sy * cr + cy * sp * sr) * k; dR[5] = (sy * sr + cy * sp * cr) * k;
  dR[6] = 0; dR[7] = 0; dR[8] = 0;
}
// q_a * x = plus(q_a) x ; x * q_b = oplus(q_b) x  (rows / columns in [x y z w] order)
__device__ __forceinline__ void pgPlusMat(const Quat& q, double* Q) {
  Q[0] = q.w; Q[1] = -q.z; Q[2] = q.y; Q[3] = q.x;
  Q[4] = q.z; Q[5] = q.w; Q[6] = -q.x; Q[7] = q.y;
  Q[8] = -q.y; Q[9] = q.x; Q[10] = q.w; Q[11] = q.z;
  Q[12] = -q.x; Q[13] = -q.y; Q[14] = -q.z; Q[15] = q.w;
}
__device__ __forceinline__ void pgOplusMat(const Quat& q, double* Q) {
  Q[0] = q.w; Q[1] = q.z; Q[2] = -q.y; Q[3] = q.x;
  Q[4] = -q.z; Q[5] = q.w; Q[6] = q.x; Q[7] = q.y;
  Q[8] = q.y; Q[9] = -q.x; Q[10] = q.w; Q[11] = q.z;
  Q[12] = -q.x; Q[13] = -q.y; Q[14] = -q.z; Q[15] = q.w;
}
__device__ __forceinline__ double pgBlockSum(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x + 63) / 64; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ double pgBlockMax(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x + 63) / 64; ++i) s = fmax(s, red[i]);
  return s;
}

// residual and minimal Jacobians of one edge (tangent order 4-DoF: [yaw, t]; 6-DoF: [t, dq], q <- [sin|d| d/|d|, cos|d|] q)
template <bool SIX>
__device__ __forceinline__ void pgEdge(const PgDev& p, int e, bool cand, double* r, double* Ja, double* Jb) {
  const int a = p.ea[e], b = p.eb[e];
  const double* T = cand ? p.tC : p.t;
  const double d[3] = {T[3 * b] - T[3 * a], T[3 * b + 1] - T[3 * a + 1], T[3 * b + 2] - T[3 * a + 2]};
  if constexpr (!SIX) {
    const double* Y = cand ? p.yawC : p.yaw;
    double R[9], dR[9];
    pgYpr2R(Y[a], p.epitch[e], p.eroll[e], R, dR);
    const bool loop = p.eloop[e] != 0;
    const double wt = 1.0, wy = loop ? 0.1 : 1.0;  // FourDOFWeightError: weight 1, yaw / 10 (PoseGraph.h:182, :206)
    for (int k = 0; k < 3; ++k) {
      const double ti = R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2];    // (R^T d)[k]
      const double dti = dR[k] * d[0] + dR[3 + k] * d[1] + dR[6 + k] * d[2];
      r[k] = (ti - p.et[3 * e + k]) * wt;
      Ja[k * 4 + 0] = dti * wt;
      Jb[k * 4 + 0] = 0.0;
      for (int c = 0; c < 3; ++c) { Ja[k * 4 + 1 + c] = -R[c * 3 + k] * wt; Jb[k * 4 + 1 + c] = R[c * 3 + k] * wt; }
    }
    r[3] = pgNormalizeAngle(Y[b] - Y[a] - p.eyaw[e]) * wy;
    Ja[12] = -wy; Ja[13] = Ja[14] = Ja[15] = 0.0;
    Jb[12] = wy;  Jb[13] = Jb[14] = Jb[15] = 0.0;
  } else {
    const double* Qn = cand ? p.qC : p.q;
    const Quat qa{Qn[4 * a], Qn[4 * a + 1], Qn[4 * a + 2], Qn[4 * a + 3]}, qb{Qn[4 * b], Qn[4 * b + 1], Qn[4 * b + 2], Qn[4 * b + 3]};
    const Quat qm{p.eq[4 * e], p.eq[4 * e + 1], p.eq[4 * e + 2], p.eq[4 * e + 3]};
    const double* si = p.esq + 6 * e;
    const Mat3 Ra = quatToR(qa);
    double pab[3];
    for (int k = 0; k < 3; ++k) pab[k] = Ra.m[k] * d[0] + Ra.m[3 + k] * d[1] + Ra.m[6 + k] * d[2];
    const Quat qac{-qa.x, -qa.y, -qa.z, qa.w}, qbc{-qb.x, -qb.y, -qb.z, qb.w};
    const Quat qab = qmul(qac, qb);
    const Quat dq = qmul(qm, Quat{-qab.x, -qab.y, -qab.z, qab.w});
    const double u[6] = {pab[0] - p.et[3 * e], pab[1] - p.et[3 * e + 1], pab[2] - p.et[3 * e + 2], 2 * dq.x, 2 * dq.y, 2 * dq.z};
    for (int k = 0; k < 6; ++k) r[k] = si[k] * u[k];
    for (int k = 0; k < 36; ++k) { Ja[k] = 0.0; Jb[k] = 0.0; }
    // d(R_a^T d)/d(delta_a) = 2 R_a^T [d]x
    const double dx[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) {
        const double rt = Ra.m[c * 3 + k];   // (R^T)[k][c]
        Ja[k * 6 + c] = -rt * si[k];
        Jb[k * 6 + c] = rt * si[k];
        double s = 0;
        for (int j = 0; j < 3; ++j) s += Ra.m[j * 3 + k] * dx[j * 3 + c];
        Ja[k * 6 + 3 + c] = 2.0 * s * si[k];
      }
    double PA[16], OB[16];
    pgPlusMat(qmul(qm, qbc), PA);
    pgOplusMat(qa, OB);
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int j = 0; j < 4; ++j) s += PA[k * 4 + j] * OB[j * 4 + c];
        Ja[(3 + k) * 6 + 3 + c] = 2.0 * s * si[3 + k];
        Jb[(3 + k) * 6 + 3 + c] = -2.0 * s * si[3 + k];
      }
  }
}

// the iteration's sums (and the linearisation's gradient max) from the block partials, fixed order; one block
__device__ void pgFinalReduce(const PgDev& p, int nEdgeBlocks, int nNodeBlocks, double* red) {
  const int slots[5] = {PG_MODEL, PG_STEP2, PG_X2, PG_COST, PG_GRADMAX};
  for (int k = 0; k < 5; ++k) {
    const int slot = slots[k], n = (slot == PG_MODEL || slot == PG_COST) ? nEdgeBlocks : nNodeBlocks;
    const bool isMax = slot == PG_GRADMAX;
    double v = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double x = p.partial[slot * kPgMaxPartials + i];
      v = isMax ? fmax(v, x) : v + x;
    }
    const double s = isMax ? pgBlockMax(v, red) : pgBlockSum(v, red);
    if (threadIdx.x == 0) p.scal[slot] = s;
  }
}
template <bool SIX>
__global__ __launch_bounds__(128) void k_pg_eval(PgDev p, int cand, int withJac, int finalReduceNodeBlocks) {
  constexpr int D = SIX ? 6 : 4, R = D, RD = R * D;
  __shared__ double red[2];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0;
  if (e < p.ne) {
    double r[R], Ja[RD], Jb[RD];
    pgEdge<SIX>(p, e, cand != 0, r, Ja, Jb);
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += r[k] * r[k];
    double sc = 1.0;
    if (p.eloop[e]) {  // HuberLoss(0.1) + Corrector (rho'' <= 0: plain sqrt(rho') scaling)
      const double a = 0.1, b = a * a;
      if (s > b) {
        const double rt = sqrt(s);
        cost = 0.5 * (2.0 * a * rt - b);
        sc = sqrt(fmax(2.2250738585072014e-308, a / rt));
      } else {
        cost = 0.5 * s;
      }
    } else {
      cost = 0.5 * s;
    }
    if (withJac) {
#pragma unroll
      for (int k = 0; k < R; ++k) p.res[(size_t)e * R + k] = sc * r[k];
#pragma unroll
      for (int k = 0; k < RD; ++k) { p.Ja[(size_t)e * RD + k] = sc * Ja[k]; p.Jb[(size_t)e * RD + k] = sc * Jb[k]; }
    }
  }
  const double bs = pgBlockSum(cost, red);
  if (threadIdx.x == 0) p.partial[PG_COST * kPgMaxPartials + blockIdx.x] = bs;
  if (finalReduceNodeBlocks <= 0) return;
  // candidate evaluation = last kernel of an iteration: whichever block finishes last reduces every scalar of the
  // iteration (one launch less per iteration)
  __shared__ int isLast;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) isLast = atomicAdd(p.ticket, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!isLast) return;
  __threadfence();
  pgFinalReduce(p, gridDim.x, finalReduceNodeBlocks, red);
  if (threadIdx.x == 0) *p.ticket = 0;
}

// one thread per node: gradient, column norms, (first iteration) Jacobi scaling, raw J^T J block
template <int D>
__global__ __launch_bounds__(128) void k_pg_node(PgDev p, int initScale) {
  constexpr int R = D;
  __shared__ double red[2];
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  double gmax = 0;
  if (k < p.nn && p.off[k] >= 0) {
    const int o = p.off[k];
    double g[D], blk[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = 0;
#pragma unroll
    for (int i = 0; i < D * D; ++i) blk[i] = 0;
    for (int it = p.nodePtr[k]; it < p.nodePtr[k + 1]; ++it) {
      const int e = p.nodeEdge[it] >> 1, side = p.nodeEdge[it] & 1;
      const double* Jg = (side ? p.Jb : p.Ja) + (size_t)e * R * D;
      const double* rg = p.res + (size_t)e * R;
      double J[R * D], r[R];
#pragma unroll
      for (int i = 0; i < R * D; ++i) J[i] = Jg[i];
#pragma unroll
      for (int i = 0; i < R; ++i) r[i] = rg[i];
#pragma unroll
      for (int c1 = 0; c1 < D; ++c1) {
        double gs = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) gs += J[q * D + c1] * r[q];
        g[c1] += gs;
#pragma unroll
        for (int c2 = 0; c2 <= c1; ++c2) {
          double s = 0;
#pragma unroll
          for (int q = 0; q < R; ++q) s += J[q * D + c1] * J[q * D + c2];
          blk[c1 * D + c2] += s;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
      if (initScale) p.scale[o + c] = 1.0 / (1.0 + sqrt(blk[c * D + c]));
      p.g[o + c] = g[c];
      p.colsq[o + c] = blk[c * D + c];
      gmax = fmax(gmax, fabs(g[c]));
    }
#pragma unroll
    for (int c1 = 0; c1 < D; ++c1)
#pragma unroll
      for (int c2 = 0; c2 <= c1; ++c2) p.nodeBlk[(size_t)k * 36 + c1 * D + c2] = blk[c1 * D + c2];   // lower triangle
  }
  const double bm = pgBlockMax(gmax, red);
  if (threadIdx.x == 0) p.partial[PG_GRADMAX * kPgMaxPartials + blockIdx.x] = bm;
}

// damped, scaled diagonal block and right-hand side of one free node -> separator system or its piece
__device__ __forceinline__ void pgAssembleNode(const PgDev& p, double radius, int vb) {
  const int k = vb * blockDim.x + threadIdx.x;
  if (k >= p.nn || p.off[k] < 0) return;
  const int D = p.D, o = p.off[k], so = p.sepOff[k];
  const double* blk = p.nodeBlk + (size_t)k * 36;
  double sc[6];
  for (int c = 0; c < D; ++c) sc[c] = p.scale[o + c];
  const PgPiece pc = so < 0 ? p.pieces[p.nodePiece[k]] : PgPiece{};
  const int row = so < 0 ? p.nodeRow[k] : 0, LDB = p.BW + 1;
  for (int c1 = 0; c1 < D; ++c1) {
    for (int c2 = 0; c2 <= c1; ++c2) {
      double v = blk[c1 * D + c2] * sc[c1] * sc[c2];
      if (c1 == c2) v += fmin(fmax(blk[c1 * D + c1] * sc[c1] * sc[c1], 1e-6), 1e32) / radius;  // LM diagonal
      if (so >= 0) p.HS[(size_t)(so + c1) * p.nS + so + c2] = v;
      else p.band[pc.bandOff + (size_t)(row + c1) * LDB + (c2 - c1 + p.BW)] = v;
    }
    const double r = p.g[o + c1] * sc[c1];
    if (so >= 0) p.rhsS[so + c1] = r;
    else p.Y[pc.yOff + (size_t)(row + c1) * pc.ld + pc.cols - 1] = r;
  }
}

// one thread per edge between two free nodes: the off-diagonal block J_b^T J_a (rows of b, columns of a), scaled,
// added to its destination.  At most two edges share a destination block (a sequential and a loop edge between the
// same pair), so the atomic adds commute exactly.
template <int D>
__device__ __forceinline__ void pgAssembleEdge(const PgDev& p, int vb) {
  constexpr int R = D;
  const int e = vb * blockDim.x + threadIdx.x;
  if (e >= p.ne) return;
  const int4 dst = p.edgeDst[e];
  const int kind = dst.x & 15, rowIsB = (dst.x >> 4) & 1;
  if (kind == 0) return;
  const int oa = p.off[p.ea[e]], ob = p.off[p.eb[e]];
  double Ja[R * D], Jb[R * D], sa[D], sb[D];
#pragma unroll
  for (int i = 0; i < R * D; ++i) { Ja[i] = p.Ja[(size_t)e * R * D + i]; Jb[i] = p.Jb[(size_t)e * R * D + i]; }
#pragma unroll
  for (int i = 0; i < D; ++i) { sa[i] = p.scale[oa + i]; sb[i] = p.scale[ob + i]; }
  double* base;
  size_t ld;
  if (kind == 1) { base = p.HS + (size_t)dst.z * p.nS + dst.w; ld = p.nS; }
  else {
    const PgPiece pc = p.pieces[dst.y];
    if (kind == 2) { base = p.band + pc.bandOff + (size_t)dst.z * (p.BW + 1) + (dst.w - dst.z + p.BW); ld = p.BW; }  // (r, c) -> r * LDB + c - r + BW
    else { base = p.Y + pc.yOff + (size_t)dst.z * pc.ld + dst.w; ld = pc.ld; }
  }
#pragma unroll
  for (int c1 = 0; c1 < D; ++c1)
#pragma unroll
    for (int c2 = 0; c2 < D; ++c2) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < R; ++q) s += Jb[q * D + c1] * Ja[q * D + c2];
      s *= sb[c1] * sa[c2];
      const int r = rowIsB ? c1 : c2, c = rowIsB ? c2 : c1;
      atomicAdd(base + (size_t)r * ld + c, s);
    }
}
// the damped normal equations of one linearisation in one launch: node blocks first, then edge blocks
template <int D>
__global__ __launch_bounds__(128) void k_pg_assemble(PgDev p, double radius, int nNodeBlocks) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.fail = 0;   // (read back after the solve; set by the piece / dense factorisations)
  if ((int)blockIdx.x < nNodeBlocks) pgAssembleNode(p, radius, blockIdx.x);
  else pgAssembleEdge<D>(p, blockIdx.x - nNodeBlocks);
}

// ---------------------------------------------------------------- pieces
constexpr int kPgPieceThreads = 256; // = max coupling columns + 1 of a piece
__device__ __forceinline__ double pgRsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = __builtin_fma(-h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ void pgWaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// band geometry of a chain whose keyframes (D unknowns) talk to their W predecessors: row r keeps columns r-BW..r
template <int D, int W>
struct PgBand {
  static constexpr int BW = W * D + D - 1, LDB = BW + 1, P = W * D, NPAIR = P * (P + 1) / 2;
};

// Block-banded Cholesky of the piece's interior (one keyframe = D columns at a time: the D x D diagonal block in
// registers, the W*D rows below it one per lane, the trailing update one (row, column) pair per lane), then
// Y = L^-1 [C | r]: one column per thread, the last BW values of its column in registers.
template <int D, int W>
__global__ __launch_bounds__(kPgPieceThreads) void k_pg_piece_factor(PgDev p) {
  using G = PgBand<D, W>;
  constexpr int BW = G::BW, LDB = G::LDB, P = G::P, NPAIR = G::NPAIR;
  constexpr bool kAllWaves = NPAIR > 128;  // the trailing update of a 6-DoF keyframe has 300 pairs: use the workgroup
  // With one factorising wave (4-DoF) the other waves run the forward substitution BEHIND it: keyframe j's rows are
  // final once wave 0 has finished steps A and B of keyframe j (`progress`), long before the whole band is done.
  __shared__ int progress;
  extern __shared__ double sm[];
  const PgPiece pc = p.pieces[blockIdx.x];
  const int n = pc.rows, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* Lb = sm;
  double* dinv = Lb + (size_t)p.maxRows * LDB;
  unsigned short* tab = reinterpret_cast<unsigned short*>(dinv + p.maxRows);
  double* band = p.band + pc.bandOff;
  for (int i = tid; i < n * LDB; i += kPgPieceThreads) Lb[i] = band[i];
  for (int i = tid; i < NPAIR; i += kPgPieceThreads) {   // (s, t), 0 <= t <= s < P
    int s_ = 0, rem = i;
    while (rem > s_) { rem -= s_ + 1; ++s_; }
    tab[i] = (unsigned short)(s_ | (rem << 8));
  }
  if (tid == 0) progress = 0;
  __syncthreads();
  int bad = 0;
  constexpr int kStride = kAllWaves ? kPgPieceThreads : 64, kRounds = (NPAIR + kStride - 1) / kStride;
  int pairS[kRounds], pairT[kRounds];
#pragma unroll
  for (int rd = 0; rd < kRounds; ++rd) {
    const int pi = (kAllWaves ? tid : lane) + rd * kStride;
    pairS[rd] = pi < NPAIR ? (tab[pi] & 255) : -1;
    pairT[rd] = pi < NPAIR ? (tab[pi] >> 8) : 0;
  }
  if (kAllWaves || wave == 0) {
    for (int j0 = 0; j0 < n; j0 += D) {
      if (wave == 0) {
        // the D x D diagonal block, factorised redundantly by every lane (broadcast LDS reads)
        double a[D][D], di[D];
#pragma unroll
        for (int x = 0; x < D; ++x)
#pragma unroll
          for (int y = 0; y <= x; ++y) a[x][y] = Lb[(j0 + x) * LDB + BW - (x - y)];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          double sdiag = a[k][k];
#pragma unroll
          for (int m = 0; m < k; ++m) sdiag -= a[k][m] * a[k][m];
          const bool ok = sdiag > 0.0;
          bad |= ok ? 0 : 1;
          di[k] = pgRsqrt(ok ? sdiag : 1.0);
          a[k][k] = sdiag * di[k];
#pragma unroll
          for (int x = k + 1; x < D; ++x) {
            double v = a[x][k];
#pragma unroll
            for (int m = 0; m < k; ++m) v -= a[x][m] * a[k][m];
            a[x][k] = v * di[k];
          }
        }
        if (lane < D) {
#pragma unroll
          for (int x = 0; x < D; ++x)
            if (lane == x) {
#pragma unroll
              for (int y = 0; y <= x; ++y) Lb[(j0 + x) * LDB + BW - (x - y)] = a[x][y];
              dinv[j0 + x] = di[x];
            }
        }
        // the rows below: X = A L_jj^-T, one row per lane
        const int r = j0 + D + lane;
        if (lane < P && r < n) {
          double* row = Lb + r * LDB + BW - D - lane;   // columns j0 .. j0 + D - 1 of row r
          double x[D];
#pragma unroll
          for (int b = 0; b < D; ++b) {
            double v = row[b];
#pragma unroll
            for (int k = 0; k < b; ++k) v -= x[k] * a[b][k];
            x[b] = v * di[b];
          }
#pragma unroll
          for (int b = 0; b < D; ++b) row[b] = x[b];
        }
      }
      if (kAllWaves) __syncthreads();
      else {
        pgWaveSync();
        if (lane == 0) __hip_atomic_store(&progress, j0 / D + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      // trailing update: A[j0+D+s][j0+D+t] -= X_s . X_t
#pragma unroll
      for (int rd = 0; rd < kRounds; ++rd) {
        const int s_ = pairS[rd], t_ = pairT[rd];   // (this thread's pairs never change: decoded once, before the loop)
        const int rs = j0 + D + s_;
        if (s_ >= 0 && rs < n) {
          const double* xs = Lb + rs * LDB + BW - D - s_;
          const double* xt = Lb + (j0 + D + t_) * LDB + BW - D - t_;
          double acc = 0;
#pragma unroll
          for (int b = 0; b < D; ++b) acc += xs[b] * xt[b];
          Lb[rs * LDB + BW - (s_ - t_)] -= acc;
        }
      }
      if (kAllWaves) __syncthreads(); else pgWaveSync();
    }
    if (bad && tid == 0) atomicOr(p.fail, 1);
  }
  if (kAllWaves) __syncthreads();
  // column -> thread: waves 1..3 take the first 192 columns (they are free while wave 0 factorises), wave 0 the rest
  const int col = kAllWaves ? tid : (tid + kPgPieceThreads - 64) % kPgPieceThreads;
  if (col < pc.cols) {
    double* Yc = p.Y + pc.yOff + col;
    // keyframe by keyframe: hist = y of the W keyframes before (oldest first, zero before the first row), shifted by
    // one keyframe per step so that every register index is static
    double hist[P], cur[D], v[D];
#pragma unroll
    for (int k = 0; k < P; ++k) hist[k] = 0.0;
#pragma unroll
    for (int b = 0; b < D; ++b) v[b] = Yc[(size_t)b * pc.ld];
    for (int j0 = 0; j0 < n; j0 += D) {
      double vn[D];
#pragma unroll
      for (int b = 0; b < D; ++b) vn[b] = j0 + D + b < n ? Yc[(size_t)(j0 + D + b) * pc.ld] : 0.0;   // next keyframe's, in flight
      if (!kAllWaves && wave != 0) {   // (wave 0 only gets here after its own loop)
        while (__hip_atomic_load(&progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < j0 / D + 1) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int b = 0; b < D; ++b) {
        const double* row = Lb + (j0 + b) * LDB;
        double x0 = v[b], x1 = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) {   // L[r][j0 - P + k] sits at row[D - 1 - b + k]
          const double term = row[D - 1 - b + k] * hist[k];
          if (k & 1) x1 -= term; else x0 -= term;
        }
#pragma unroll
        for (int k = 0; k < b; ++k) x0 -= row[BW - b + k] * cur[k];
        cur[b] = (x0 + x1) * dinv[j0 + b];
        Yc[(size_t)(j0 + b) * pc.ld] = cur[b];
      }
#pragma unroll
      for (int k = 0; k < P - D; ++k) hist[k] = hist[k + D];
#pragma unroll
      for (int b = 0; b < D; ++b) { hist[P - D + b] = cur[b]; v[b] = vn[b]; }
    }
  }
  __syncthreads();
  for (int i = tid; i < n * LDB; i += kPgPieceThreads) band[i] = Lb[i];
}

// S_p = Y^T Y, one 16x16 tile per workgroup (lower tiles only; the rhs column is the last row of S_p)
__global__ __launch_bounds__(256) void k_pg_piece_schur(PgDev p) {
  const int4 wk = p.tileWork[blockIdx.x];
  const PgPiece pc = p.pieces[wk.x];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const double* Ya = p.Y + pc.yOff + 16 * wk.y + ty;
  const double* Yb = p.Y + pc.yOff + 16 * wk.z + tx;
  double acc = 0;
#pragma unroll 8
  for (int r = 0; r < pc.rows; ++r) acc += Ya[(size_t)r * pc.ld] * Yb[(size_t)r * pc.ld];
  p.Sp[pc.sOff + (size_t)(16 * wk.y + ty) * pc.ld + 16 * wk.z + tx] = acc;
}

// separator system -= sum over pieces of their Schur blocks, contributions in the host's fixed order
__device__ __forceinline__ void pgSepGather(const PgDev& p, int nDest, int vb) {
  const int idx = vb * blockDim.x + threadIdx.x, DD = p.D * p.D;
  if (idx >= nDest * DD) return;
  const int b = idx / DD, e = idx - b * DD, r = e / p.D, c = e - r * p.D;
  const int2 dst = p.gDst[b];
  if (dst.x == dst.y && c > r) return;
  double acc = 0;
  for (int i = p.gPtr[b]; i < p.gPtr[b + 1]; ++i) {
    const int4 src = p.gSrc[i];
    const PgPiece pc = p.pieces[src.x];
    acc += p.Sp[pc.sOff + (size_t)(src.y + r) * pc.ld + src.z + c];
  }
  p.HS[(size_t)(dst.x + r) * p.nS + dst.y + c] -= acc;
}
__device__ __forceinline__ void pgSepGatherRhs(const PgDev& p, int vb) {
  const int idx = vb * blockDim.x + threadIdx.x;
  if (idx >= p.nS) return;
  const int sn = idx / p.D, c = idx - sn * p.D;
  double acc = 0;
  for (int i = p.rPtr[sn]; i < p.rPtr[sn + 1]; ++i) {
    const int2 src = p.rSrc[i];
    const PgPiece pc = p.pieces[src.x];
    acc += p.Sp[pc.sOff + (size_t)(pc.cols - 1) * pc.ld + src.y + c];
  }
  p.rhsS[idx] -= acc;
}
__global__ __launch_bounds__(256) void k_pg_sep_gather(PgDev p, int nDest, int nDestBlocks) {
  if ((int)blockIdx.x < nDestBlocks) pgSepGather(p, nDest, blockIdx.x);
  else pgSepGatherRhs(p, blockIdx.x - nDestBlocks);
}

// level 2: the blocks of a level-2 piece (band, coupling columns, right-hand side) copied out of the separator system
__global__ void k_pg_l2_extract(PgDev p, const int4* xDst, const int2* xSrc, int nX) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, DD = p.D * p.D;
  if (idx >= nX * DD) return;
  const int b = idx / DD, e = idx - b * DD, r = e / p.D, c = e - r * p.D;
  const int4 dst = xDst[b];
  const int2 src = xSrc[b];
  const int kind = dst.x & 15;
  const PgPiece pc = p.pieces[dst.y];
  if (kind == 2) {        // band block (rows of the later keyframe); the diagonal block only has its lower triangle
    if (dst.z == dst.w && c > r) return;
    p.band[pc.bandOff + (size_t)(dst.z + r) * (p.BW + 1) + (dst.w + c - dst.z - r + p.BW)] = p.HS[(size_t)(src.x + r) * p.nS + src.y + c];
  } else if (kind == 3) { // coupling to a root separator: stored transposed in the lower triangle of the system
    p.Y[pc.yOff + (size_t)(dst.z + r) * pc.ld + dst.w + c] = p.HS[(size_t)(src.x + c) * p.nS + src.y + r];
  } else if (c == 0) {    // right-hand side column
    p.Y[pc.yOff + (size_t)(dst.z + r) * pc.ld + dst.w] = p.rhsS[src.x + r];
  }
}

// interior unknowns of one piece: x_I = L^-T (y_r - Y_C x_S)
template <int D, int W>
__global__ __launch_bounds__(kPgPieceThreads) void k_pg_piece_back(PgDev p) {
  using G = PgBand<D, W>;
  constexpr int BW = G::BW, LDB = G::LDB, P = G::P;
  extern __shared__ double sm[];
  if ((int)blockIdx.x >= p.nPieces) {   // level 1 only: extra blocks copy the separators' solution to tangent order
    const int k = ((int)blockIdx.x - p.nPieces) * blockDim.x + threadIdx.x;
    if (k < p.nn && p.off[k] >= 0 && p.sepOff[k] >= 0)
      for (int c = 0; c < p.D; ++c) p.y[p.off[k] + c] = p.yS[p.sepOff[k] + c];
    return;
  }
  const PgPiece pc = p.pieces[blockIdx.x];
  const int n = pc.rows, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* Lb = sm;
  double* z = Lb + (size_t)p.maxRows * LDB;   // P zeros of padding behind the last row
  double* xA = z + p.maxRows + P;
  const double* band = p.band + pc.bandOff;
  for (int i = tid; i < n * LDB; i += kPgPieceThreads) Lb[i] = band[i];
  for (int c = tid; c < pc.cols - 1; c += kPgPieceThreads) xA[c] = p.yS[p.colSep[pc.colPtr + c]];
  if (tid < P) z[n + tid] = 0.0;
  __syncthreads();
  const double* Y = p.Y + pc.yOff;
  constexpr int kRowsInFlight = 8;   // one wave per row (coalesced dot product), 8 rows' loads in flight together
  for (int r0 = wave * kRowsInFlight; r0 < n; r0 += kRowsInFlight * (kPgPieceThreads / 64)) {
    double acc[kRowsInFlight], rhs[kRowsInFlight];
#pragma unroll
    for (int q = 0; q < kRowsInFlight; ++q) { acc[q] = 0; rhs[q] = (lane == 0 && r0 + q < n) ? Y[(size_t)(r0 + q) * pc.ld + pc.cols - 1] : 0.0; }
    for (int c = lane; c < pc.cols - 1; c += 64) {
      const double x = xA[c];
      double yv[kRowsInFlight];
#pragma unroll
      for (int q = 0; q < kRowsInFlight; ++q) yv[q] = r0 + q < n ? Y[(size_t)(r0 + q) * pc.ld + c] : 0.0;
#pragma unroll
      for (int q = 0; q < kRowsInFlight; ++q) acc[q] += yv[q] * x;
    }
#pragma unroll
    for (int q = 0; q < kRowsInFlight; ++q) {
      double a = acc[q];
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0 && r0 + q < n) z[r0 + q] = rhs[q] - a;
    }
  }
  __syncthreads();
  if (wave == 0) {
    // keyframe by keyframe from the last: lane b < D gathers row j0 + b of L^T x over the W keyframes behind it,
    // then every lane solves the D x D triangle redundantly
    for (int j0 = n - D; j0 >= 0; j0 -= D) {
      if (lane < D) {
        double a0 = z[j0 + lane], a1 = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const int r = j0 + D + k;
          const double l = r < n ? Lb[r * LDB + BW - D - k + lane] : 0.0;   // L[r][j0 + lane]
          if (k & 1) a1 -= l * z[r]; else a0 -= l * z[r];
        }
        z[j0 + lane] = a0 + a1;
      }
      pgWaveSync();
      double x[D];
#pragma unroll
      for (int b = D - 1; b >= 0; --b) {
        double v = z[j0 + b];
#pragma unroll
        for (int k = b + 1; k < D; ++k) v -= Lb[(j0 + k) * LDB + BW - (k - b)] * x[k];
        x[b] = v / Lb[(j0 + b) * LDB + BW];
      }
      pgWaveSync();
      if (lane < D) {
#pragma unroll
        for (int b = 0; b < D; ++b)
          if (lane == b) z[j0 + b] = x[b];
      }
      pgWaveSync();
    }
  }
  __syncthreads();
  for (int r = tid; r < n; r += kPgPieceThreads) p.y[p.rowTan[pc.rowPtr + r]] = z[r];
}
__global__ void k_pg_scatter_sep(PgDev p) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.nn || p.off[k] < 0 || p.sepOff[k] < 0) return;
  for (int c = 0; c < p.D; ++c) p.y[p.off[k] + c] = p.yS[p.sepOff[k] + c];
}

// the step in the tangent space: delta = -y * scale (y solves the scaled, damped normal equations)
__device__ __forceinline__ double pgDelta(const PgDev& p, int i) { return -p.y[i] * p.scale[i]; }

// model cost change = -(J delta).(r + J delta / 2) over the edges
template <int D>
__device__ __forceinline__ void pgModel(const PgDev& p, int vb, double* red) {
  constexpr int R = D;
  const int e = vb * blockDim.x + threadIdx.x;
  double acc = 0;
  if (e < p.ne) {
    const int oa = p.off[p.ea[e]], ob = p.off[p.eb[e]];
    double da[D], db[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { da[c] = oa >= 0 ? pgDelta(p, oa + c) : 0.0; db[c] = ob >= 0 ? pgDelta(p, ob + c) : 0.0; }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      double mr = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) mr += p.Ja[((size_t)e * R + q) * D + c] * da[c] + p.Jb[((size_t)e * R + q) * D + c] * db[c];
      acc += mr * (p.res[(size_t)e * R + q] + 0.5 * mr);
    }
  }
  const double bs = pgBlockSum(acc, red);
  if (threadIdx.x == 0) p.partial[PG_MODEL * kPgMaxPartials + vb] = bs;
}

// candidate = Plus(x, delta) (YawAngleFunctor PoseGraph.h:95-108 / EigenQuaternionManifold::Plus); |x_c - x|^2, |x|^2
__device__ __forceinline__ void pgPlus(const PgDev& p, int vb, double* red) {
  const int k = vb * blockDim.x + threadIdx.x;
  double st = 0, xn = 0;
  if (k < p.nn) {
    const int o = p.off[k];
    if (!p.six) {
      const double y0 = p.yaw[k];
      const double y1 = o >= 0 ? pgNormalizeAngle(y0 + pgDelta(p, o)) : y0;
      p.yawC[k] = y1;
      if (o >= 0) { st += (y1 - y0) * (y1 - y0); xn += y0 * y0; }
      for (int c = 0; c < 3; ++c) {
        const double x0 = p.t[3 * k + c], x1 = o >= 0 ? x0 + pgDelta(p, o + 1 + c) : x0;
        p.tC[3 * k + c] = x1;
        if (o >= 0) { st += (x1 - x0) * (x1 - x0); xn += x0 * x0; }
      }
    } else {
      for (int c = 0; c < 3; ++c) {
        const double x0 = p.t[3 * k + c], x1 = o >= 0 ? x0 + pgDelta(p, o + c) : x0;
        p.tC[3 * k + c] = x1;
        if (o >= 0) { st += (x1 - x0) * (x1 - x0); xn += x0 * x0; }
      }
      const Quat q0{p.q[4 * k], p.q[4 * k + 1], p.q[4 * k + 2], p.q[4 * k + 3]};
      Quat q1 = q0;
      if (o >= 0) {
        const double dx = pgDelta(p, o + 3), dy = pgDelta(p, o + 4), dz = pgDelta(p, o + 5);
        const double nd = sqrt(dx * dx + dy * dy + dz * dz);
        if (nd > 0.0) {
          const double s = sin(nd) / nd;
          q1 = qmul(Quat{s * dx, s * dy, s * dz, cos(nd)}, q0);
        }
        st += (q1.x - q0.x) * (q1.x - q0.x) + (q1.y - q0.y) * (q1.y - q0.y) + (q1.z - q0.z) * (q1.z - q0.z) + (q1.w - q0.w) * (q1.w - q0.w);
        xn += q0.x * q0.x + q0.y * q0.y + q0.z * q0.z + q0.w * q0.w;
      }
      p.qC[4 * k] = q1.x; p.qC[4 * k + 1] = q1.y; p.qC[4 * k + 2] = q1.z; p.qC[4 * k + 3] = q1.w;
    }
  }
  const double a = pgBlockSum(st, red);
  const double b = pgBlockSum(xn, red);
  if (threadIdx.x == 0) { p.partial[PG_STEP2 * kPgMaxPartials + vb] = a; p.partial[PG_X2 * kPgMaxPartials + vb] = b; }
}
// model cost change (edge blocks) and candidate (node blocks) in one launch
template <int D>
__global__ __launch_bounds__(128) void k_pg_model_plus(PgDev p, int nEdgeBlocks) {
  __shared__ double red[2];
  if ((int)blockIdx.x < nEdgeBlocks) pgModel<D>(p, blockIdx.x, red);
  else pgPlus(p, blockIdx.x - nEdgeBlocks, red);
}

// single-block final reductions (fixed order): slot -> scal[slot]; isMax selects max instead of sum
__global__ __launch_bounds__(256) void k_pg_reduce(PgDev p, int slot, int n, int isMax) {
  __shared__ double red[4];
  double v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double x = p.partial[slot * kPgMaxPartials + i];
    v = isMax ? fmax(v, x) : v + x;
  }
  const double s = isMax ? pgBlockMax(v, red) : pgBlockSum(v, red);
  if (threadIdx.x == 0) p.scal[slot] = s;
}

// ---------------------------------------------------------------- host
struct Keyframe {
  int index = 0, sequence = 0;
  double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1};   // Keyframe::getSVInPose: the optimisation's input, never modified
  double P[3] = {0, 0, 0}, Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // Keyframe::getPose (updatePose)
  double ypr[3] = {0, 0, 0}, Rs[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; // R2ypr / rotation matrix of the SVIn pose (cached at add time)
  bool hasLoop = false;
  int loopIndex = -1;
  double loopT[3] = {0, 0, 0}, loopQ[4] = {0, 0, 0, 1}, loopYaw = 0;
};

template <class T>
struct Buf {
  T* p = nullptr;
  size_t cap = 0;
  ~Buf() { if (p) (void)hipFree(p); }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipFree(p);
    size_t c = cap ? cap : 64;
    while (c < n) c *= 2;
    if (hipMalloc(&p, c * sizeof(T)) != hipSuccess) { p = nullptr; cap = 0; throw std::runtime_error("hipMalloc failed"); }
    cap = c;
  }
  void upload(const std::vector<T>& h, hipStream_t s) {
    reserve(std::max<size_t>(h.size(), 1));
    if (!h.empty()) PG_HIP_OK(hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s));
  }
};

static void hostR2ypr(const double* q, double* ypr) {  // Utils.h:71-86 on Quaterniond::toRotationMatrix()
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                       2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  const double yw = std::atan2(R[3], R[0]);
  const double p = std::atan2(-R[6], R[0] * std::cos(yw) + R[3] * std::sin(yw));
  const double r = std::atan2(R[2] * std::sin(yw) - R[5] * std::cos(yw), -R[1] * std::sin(yw) + R[4] * std::cos(yw));
  ypr[0] = yw / kPgPi * 180.0; ypr[1] = p / kPgPi * 180.0; ypr[2] = r / kPgPi * 180.0;
}
static double hostYawOfR(const double* R) { return std::atan2(R[3], R[0]) / kPgPi * 180.0; }
static void hostQmul(const double* a, const double* b, double* o) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void hostQ2R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void hostYpr2R(double yaw, double pitch, double roll, double* R) {
  const double y = yaw / 180.0 * kPgPi, p = pitch / 180.0 * kPgPi, r = roll / 180.0 * kPgPi;
  const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
  R[0] = cy * cp; R[1] = -sy * cr + cy * sp * sr; R[2] = sy * sr + cy * sp * cr;
  R[3] = sy * cp; R[4] = cy * cr + sy * sp * sr;  R[5] = -cy * sr + sy * sp * cr;
  R[6] = -sp;     R[7] = cp * sr;                 R[8] = cp * cr;
}
static void hostR2q(const double* R, double* qo) {  // Eigen::Quaterniond(Matrix3d)
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    qo[3] = 0.5 * s; s = 0.5 / s;
    qo[0] = (R[7] - R[5]) * s; qo[1] = (R[2] - R[6]) * s; qo[2] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    qo[i] = 0.5 * s; s = 0.5 / s;
    qo[3] = (R[k * 3 + j] - R[j * 3 + k]) * s;
    qo[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    qo[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
  }
}

class PoseGraph {
 public:
  PoseGraph(int device, bool six, int maxIter) : six_(six), maxIter_(maxIter > 0 ? maxIter : (six ? 5 : 10)) {
    // measured on the config-#5 graph (tools/pgtime.py sweeps): short level-1 pieces pay off once the cuts are
    // eliminated at level 2; the 6-DoF band is wider, so its pieces stay at the 256-row limit
    pieceLen_ = six ? 64 : 32;
    l2Len_ = six ? 32 : 16;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      throw std::runtime_error("svin_pg: no HIP device available (this backend has no CPU fallback)");
    if (device < 0 || device >= count) throw std::runtime_error("svin_pg: invalid device index");
    PG_HIP_OK(hipSetDevice(device));
    PG_HIP_OK(hipStreamCreate(&s_));
    for (int i = 0; i < kPgMaxTimed; ++i) { PG_HIP_OK(hipEventCreate(&evA_[i])); PG_HIP_OK(hipEventCreate(&evB_[i])); }
  }
  ~PoseGraph() {
    for (int i = 0; i < kPgMaxTimed; ++i) { if (evA_[i]) (void)hipEventDestroy(evA_[i]); if (evB_[i]) (void)hipEventDestroy(evB_[i]); }
    if (s_) (void)hipStreamDestroy(s_);
  }

  std::vector<Keyframe> kfs;
  double summary[8] = {0, 0, 0, 1, 0, 0, 0, 0};   // [6] = seconds of the symbolic step (host)
  int partition[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // [7] level-2 pieces, [8] root unknowns; [5] = unknowns of the dense solve       // free keyframes, separator keyframes, pieces, max piece rows, Schur tiles,
                                                  // separator unknowns, timed dense solves ([7] of summary = their seconds)
  int pieceLen_ = 64, denseNodes_ = 128, l2Len_ = 32;
  bool level2_ = true;
  // drift of the odometry frame against the optimised map (PoseGraph.cpp:356-363 / :521-526)
  double yawDrift = 0, rDrift[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tDrift[3] = {0, 0, 0};

  // addKeyframe's pose update (PoseGraph.cpp:127-132): pose = drift * SVIn pose
  void applyDrift(Keyframe& kf) const {
    const double* R = kf.Rs;
    for (int r = 0; r < 3; ++r) {
      kf.P[r] = rDrift[3 * r] * kf.t[0] + rDrift[3 * r + 1] * kf.t[1] + rDrift[3 * r + 2] * kf.t[2] + tDrift[r];
      for (int c = 0; c < 3; ++c) kf.Rp[3 * r + c] = rDrift[3 * r] * R[c] + rDrift[3 * r + 1] * R[3 + c] + rDrift[3 * r + 2] * R[6 + c];
    }
  }

  int optimize(int earliest, int cur) {
    const auto tCall0 = std::chrono::steady_clock::now();
    // ---- local problem (PoseGraph.cpp:262-332 / :436-489)
    std::vector<double> yaw, pitch, roll, t, q;
    std::vector<int> off, ea, eb, eloop, seq, localOf(kfs.size(), -1), kfOfLocal;
    std::vector<char> fixed;
    std::vector<double> et, eyaw, epitch, eroll, eq, esq;
    std::unordered_map<int, size_t> posOfIndex;
    posOfIndex.reserve(kfs.size() * 2);
    for (size_t k = 0; k < kfs.size(); ++k) posOfIndex[kfs[k].index] = k;
    int i = 0;
    for (size_t k = 0; k < kfs.size(); ++k) {
      const Keyframe& kf = kfs[k];
      if (kf.index < earliest) continue;
      localOf[k] = i;
      kfOfLocal.push_back((int)k);
      const double* ypr = kf.ypr;
      yaw.push_back(ypr[0]); pitch.push_back(ypr[1]); roll.push_back(ypr[2]);
      t.insert(t.end(), kf.t, kf.t + 3);
      q.insert(q.end(), kf.q, kf.q + 4);
      seq.push_back(kf.sequence);
      fixed.push_back(six_ ? (kf.index == earliest || kf.sequence == 0) : (kf.index <= earliest));
      const int nSeq = six_ ? 4 : 2;
      for (int j = 1; j <= nSeq; ++j) {
        if (i - j >= 0 && seq[i] == seq[i - j]) {
          const double* qa = &q[4 * (i - j)];
          const double* Ra = kfs[kfOfLocal[i - j]].Rs;
          const double d[3] = {t[3 * i] - t[3 * (i - j)], t[3 * i + 1] - t[3 * (i - j) + 1], t[3 * i + 2] - t[3 * (i - j) + 2]};
          for (int c = 0; c < 3; ++c) et.push_back(Ra[c] * d[0] + Ra[3 + c] * d[1] + Ra[6 + c] * d[2]);
          ea.push_back(i - j); eb.push_back(i); eloop.push_back(0);
          eyaw.push_back(yaw[i] - yaw[i - j]); epitch.push_back(pitch[i - j]); eroll.push_back(roll[i - j]);
          const double n2 = qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2] + qa[3] * qa[3];
          const double qai[4] = {-qa[0] / n2, -qa[1] / n2, -qa[2] / n2, qa[3] / n2};
          double rq[4];
          hostQmul(qai, &q[4 * i], rq);
          eq.insert(eq.end(), rq, rq + 4);
          const double si[6] = {20, 20, 20, 100, 100, 57.3};
          esq.insert(esq.end(), si, si + 6);
        }
      }
      if (kf.hasLoop) {
        int ci = -1;   // getKeyframe(loop_index)->local_index: the last keyframe of the list with that index
        const auto lit = posOfIndex.find(kf.loopIndex);
        if (lit != posOfIndex.end()) ci = localOf[lit->second];
        if (ci >= 0) {
          ea.push_back(ci); eb.push_back(i); eloop.push_back(1);
          et.insert(et.end(), kf.loopT, kf.loopT + 3);
          eyaw.push_back(kf.loopYaw); epitch.push_back(pitch[ci]); eroll.push_back(roll[ci]);
          eq.insert(eq.end(), kf.loopQ, kf.loopQ + 4);
          const double si[6] = {20, 20, 20, 100, 100, 100};
          esq.insert(esq.end(), si, si + 6);
        }
      }
      if (kf.index == cur) { ++i; break; }
      ++i;
    }
    const int nn = (int)fixed.size(), ne = (int)ea.size(), D = six_ ? 6 : 4, R = D;
    off.assign(nn, -1);
    int n = 0;
    for (int k = 0; k < nn; ++k)
      if (!fixed[k]) { off[k] = n; n += D; }
    summary[0] = summary[1] = 0; summary[2] = 0; summary[3] = 0; summary[4] = 0; summary[5] = 0;
    if (n == 0 || ne == 0) {  // nothing to optimise: the poses are still written back and the drift updated
      writeBack(yaw, pitch, roll, t, q, kfOfLocal, cur);
      return 1;
    }
    // incident edges per node (edge order = insertion order: deterministic accumulation)
    std::vector<int> nodePtr(nn + 1, 0), nodeEdge(2 * (size_t)ne);
    for (int e = 0; e < ne; ++e) { nodePtr[ea[e] + 1]++; nodePtr[eb[e] + 1]++; }
    for (int k = 0; k < nn; ++k) nodePtr[k + 1] += nodePtr[k];
    {
      std::vector<int> curp(nodePtr.begin(), nodePtr.end() - 1);
      for (int e = 0; e < ne; ++e) { nodeEdge[curp[ea[e]]++] = 2 * e; nodeEdge[curp[eb[e]]++] = 2 * e + 1; }
    }
    const auto tSym0 = std::chrono::steady_clock::now();
    // ---- symbolic step: separators (cuts of w keyframes + a vertex cover of the long edges) and pieces
    const int w = six_ ? 4 : 2, BW = w * D + D - 1;
    const int pieceLen = std::max(8, std::min(pieceLen_, 256 / D));           // interior keyframes per piece
    const int maxAdj = (kPgPieceThreads - 1) / D;                              // separator keyframes a piece may touch
    std::vector<int> freeNodes, pos(nn, -1);
    for (int k = 0; k < nn; ++k)
      if (!fixed[k]) { pos[k] = (int)freeNodes.size(); freeNodes.push_back(k); }
    const int F = (int)freeNodes.size();
    std::vector<char> isSep(nn, 0), isCover(nn, 0);
    std::vector<int> pieceOf(nn, -1);
    int nPieces = 0;
    if (F <= denseNodes_) {
      for (int k : freeNodes) isSep[k] = 1;
    } else {
      std::vector<int> longDeg(nn, 0);
      auto isLong = [&](int e) { return pos[ea[e]] >= 0 && pos[eb[e]] >= 0 && std::abs(pos[ea[e]] - pos[eb[e]]) > w; };
      for (int e = 0; e < ne; ++e)
        if (isLong(e)) { longDeg[ea[e]]++; longDeg[eb[e]]++; }
      for (int e = 0; e < ne; ++e)
        if (isLong(e) && !isSep[ea[e]] && !isSep[eb[e]]) {
          const int c = longDeg[ea[e]] >= longDeg[eb[e]] ? ea[e] : eb[e];
          isSep[c] = 1; isCover[c] = 1;
        }
      // walk the chain: close a piece after pieceLen interior keyframes (or when it touches too many separators)
      std::vector<int> stamp(nn, -1);
      int cnt = 0, adj = 2 * w;
      for (int i = 0; i < F; ++i) {
        const int k = freeNodes[i];
        if (isSep[k]) {
          if (stamp[k] != nPieces) { stamp[k] = nPieces; ++adj; }   // (conservative: counted even if not adjacent)
          continue;
        }
        pieceOf[k] = nPieces;
        ++cnt;
        for (int it = nodePtr[k]; it < nodePtr[k + 1]; ++it) {
          const int e = nodeEdge[it] >> 1, o = (nodeEdge[it] & 1) ? ea[e] : eb[e];
          if (pos[o] >= 0 && isSep[o] && stamp[o] != nPieces) { stamp[o] = nPieces; ++adj; }
        }
        if ((cnt >= pieceLen || adj + 1 >= maxAdj) && i + w < F - 1) {
          for (int j = 1; j <= w; ++j) isSep[freeNodes[i + j]] = 1;
          i += w;
          ++nPieces; cnt = 0; adj = 2 * w;
        }
      }
      if (cnt > 0) ++nPieces;
      else {  // the walk ended right after a cut (or on separators): the last piece id may be unused
        bool used = false;
        for (int k : freeNodes) used |= pieceOf[k] == nPieces;
        if (used) ++nPieces;
      }
    }
    // ---- level 2 (optional): the cut keyframes form a chain of their own (a cut group only talks to the next one
    // through the piece between them: reach W2 = 2w - 1 cut keyframes); they are eliminated the same way, so the
    // dense root only holds the loop cover and a few level-2 cuts
    const int W2 = 2 * w - 1, BW2 = W2 * D + D - 1;
    std::vector<std::vector<int>> pieceAdjNodes(nPieces);
    for (int e = 0; e < ne; ++e) {
      const int a = ea[e], b = eb[e];
      if (pos[a] < 0 || pos[b] < 0) continue;
      if (!isSep[a] && !isSep[b] && pieceOf[a] != pieceOf[b]) throw std::runtime_error("svin_pg: partition left an edge between two pieces");
      if (!isSep[a] && isSep[b]) pieceAdjNodes[pieceOf[a]].push_back(b);
      if (!isSep[b] && isSep[a]) pieceAdjNodes[pieceOf[b]].push_back(a);
    }
    for (auto& A : pieceAdjNodes) { std::sort(A.begin(), A.end()); A.erase(std::unique(A.begin(), A.end()), A.end()); }
    std::vector<std::vector<int>> sepAdj(nn);   // structure of the separator system after level 1 (node ids)
    std::vector<int> l2PieceOf(nn, -1);
    std::vector<char> isRoot(nn, 0);            // separators that stay in the dense root
    int nPieces2 = 0;
    {
      std::vector<int> cutList;
      for (int k : freeNodes)
        if (isSep[k] && !isCover[k]) cutList.push_back(k);
      bool useL2 = level2_ && nPieces >= 8 && (int)cutList.size() >= 3 * l2Len_;
      if (useL2) {
        for (const auto& A : pieceAdjNodes)
          for (int x : A)
            for (int y : A)
              if (x != y) sepAdj[x].push_back(y);
        for (int e = 0; e < ne; ++e) {
          const int a = ea[e], b = eb[e];
          if (pos[a] >= 0 && pos[b] >= 0 && isSep[a] && isSep[b]) { sepAdj[a].push_back(b); sepAdj[b].push_back(a); }
        }
        for (int k : freeNodes)
          if (isSep[k]) { auto& A = sepAdj[k]; std::sort(A.begin(), A.end()); A.erase(std::unique(A.begin(), A.end()), A.end()); }
        const int maxAdj2 = (kPgPieceThreads - 1) / D;
        std::vector<int> stamp(nn, -1);
        std::vector<char> isL2Cut(nn, 0);
        int cnt = 0, adj = 2 * W2;
        const int C = (int)cutList.size();
        for (int i = 0; i < C; ++i) {
          const int k = cutList[i];
          l2PieceOf[k] = nPieces2;
          ++cnt;
          for (int o : sepAdj[k])
            if (isCover[o] && stamp[o] != nPieces2) { stamp[o] = nPieces2; ++adj; }
          if ((cnt >= l2Len_ || adj + 12 >= maxAdj2) && i + W2 < C - 1) {
            for (int j = 1; j <= W2; ++j) isL2Cut[cutList[i + j]] = 1;
            i += W2;
            ++nPieces2; cnt = 0; adj = 2 * W2;
          }
        }
        if (cnt > 0) ++nPieces2;
        for (int k : freeNodes)
          if (isSep[k] && (isCover[k] || isL2Cut[k])) { isRoot[k] = 1; l2PieceOf[k] = -1; }
        // validate: interior cut keyframes only talk to their own level-2 piece (within W2) or to root separators,
        // and no piece exceeds the column budget; otherwise fall back to one level
        std::vector<int> rank(nn, -1);
        for (int i = 0, r = 0; i < C; ++i)
          if (!isRoot[cutList[i]]) rank[cutList[i]] = r++;
        std::vector<std::vector<int>> a2(nPieces2);
        for (int k : cutList) {
          if (isRoot[k]) continue;
          for (int o : sepAdj[k]) {
            if (isRoot[o]) a2[l2PieceOf[k]].push_back(o);
            else if (l2PieceOf[o] != l2PieceOf[k] || std::abs(rank[o] - rank[k]) > W2) useL2 = false;
          }
        }
        for (auto& A : a2) {
          std::sort(A.begin(), A.end());
          A.erase(std::unique(A.begin(), A.end()), A.end());
          if ((int)A.size() * D + 1 > kPgPieceThreads) useL2 = false;
        }
        if (nPieces2 < 2) useL2 = false;
      }
      if (!useL2) {
        nPieces2 = 0;
        for (int k : freeNodes) { l2PieceOf[k] = -1; isRoot[k] = isSep[k]; }
      }
    }
    std::vector<int> sepOff(nn, -1), nodePiece(nn, -1), nodeRow(nn, -1);
    int nS = 0;
    for (int k : freeNodes)
      if (isSep[k] && !isRoot[k]) { sepOff[k] = nS; nS += D; }
    const int offR = nS;   // the root = trailing block of the separator system
    for (int k : freeNodes)
      if (isSep[k] && isRoot[k]) { sepOff[k] = nS; nS += D; }
    const int nR = nS - offR;
    // piece descriptors of one level from the per-piece row counts and adjacent separator offsets
    struct LevelHost {
      std::vector<PgPiece> pieces;
      std::vector<std::vector<int>> adj;   // per piece: sorted offsets of the separators it touches
      std::vector<int> rowMap, colSep, gPtr, rPtr;
      std::vector<int4> tileWork, gSrc;
      std::vector<int2> gDst, rSrc;
      int maxRows = 0, nDest = 0;
      size_t bandTot = 1, yTot = 1, spTot = 1;
    };
    auto finishLevel = [&](LevelHost& Lh, const std::vector<int>& rows, int bw) {
      const int np = (int)rows.size();
      Lh.pieces.resize(np);
      long long bandOff = 0, yOff = 0, sOff = 0;
      for (int pi = 0; pi < np; ++pi) {
        auto& A = Lh.adj[pi];
        std::sort(A.begin(), A.end());
        A.erase(std::unique(A.begin(), A.end()), A.end());
        PgPiece& pc = Lh.pieces[pi];
        pc.rows = rows[pi];
        pc.cols = (int)A.size() * D + 1;
        if (pc.cols > kPgPieceThreads) throw std::runtime_error("svin_pg: a piece touches too many separators");
        pc.ld = (pc.cols + 15) / 16 * 16;
        pc.colPtr = (int)Lh.colSep.size();
        pc.rowPtr = (int)Lh.rowMap.size();
        pc.pad = 0;
        pc.bandOff = bandOff; pc.yOff = yOff; pc.sOff = sOff;
        bandOff += (long long)pc.rows * (bw + 1);
        yOff += (long long)pc.rows * pc.ld;
        sOff += (long long)pc.ld * pc.ld;
        for (int so : A)
          for (int c = 0; c < D; ++c) Lh.colSep.push_back(so + c);
        Lh.rowMap.resize(Lh.rowMap.size() + pc.rows);
        Lh.maxRows = std::max(Lh.maxRows, pc.rows);
      }
      Lh.bandTot = (size_t)std::max<long long>(bandOff, 1); Lh.yTot = (size_t)std::max<long long>(yOff, 1);
      Lh.spTot = (size_t)std::max<long long>(sOff, 1);
      // work lists: Schur tiles, and per separator block / separator keyframe the pieces that contribute to it
      struct Contrib { long long key; int piece, ro, co; };
      std::vector<Contrib> cs;
      std::vector<std::vector<int2>> rl(nS / D);
      for (int pi = 0; pi < np; ++pi) {
        const PgPiece& pc = Lh.pieces[pi];
        const int nt = pc.ld / 16;
        for (int ti = 0; ti < nt; ++ti)
          for (int tj = 0; tj <= ti; ++tj) Lh.tileWork.push_back(make_int4(pi, ti, tj, 0));
        const auto& A = Lh.adj[pi];
        for (size_t i = 0; i < A.size(); ++i) {
          rl[A[i] / D].push_back(make_int2(pi, (int)i * D));
          for (size_t j = 0; j <= i; ++j)
            cs.push_back({(long long)A[i] * nS + A[j], pi, (int)i * D, (int)j * D});
        }
      }
      std::stable_sort(cs.begin(), cs.end(), [](const Contrib& x, const Contrib& y) { return x.key < y.key; });
      Lh.gPtr.assign(1, 0);
      for (size_t i = 0; i < cs.size(); ++i) {
        if (i == 0 || cs[i].key != cs[i - 1].key) {
          if (i) Lh.gPtr.push_back((int)Lh.gSrc.size());
          Lh.gDst.push_back(make_int2((int)(cs[i].key / nS), (int)(cs[i].key % nS)));
        }
        Lh.gSrc.push_back(make_int4(cs[i].piece, cs[i].ro, cs[i].co, 0));
      }
      if (!cs.empty()) Lh.gPtr.push_back((int)Lh.gSrc.size());
      Lh.rPtr.assign(nS / D + 1, 0);
      for (int sn = 0; sn < nS / D; ++sn) {
        Lh.rPtr[sn + 1] = Lh.rPtr[sn] + (int)rl[sn].size();
        Lh.rSrc.insert(Lh.rSrc.end(), rl[sn].begin(), rl[sn].end());
      }
      Lh.nDest = (int)Lh.gDst.size();
    };
    auto colIn = [&](const std::vector<int>& A, int so) { return (int)(std::lower_bound(A.begin(), A.end(), so) - A.begin()) * D; };
    // level 1
    LevelHost L1;
    {
      std::vector<int> rows(nPieces, 0);
      for (int k : freeNodes)
        if (!isSep[k]) { nodePiece[k] = pieceOf[k]; nodeRow[k] = rows[pieceOf[k]]; rows[pieceOf[k]] += D; }
      L1.adj.resize(nPieces);
      for (int pi = 0; pi < nPieces; ++pi)
        for (int o : pieceAdjNodes[pi]) L1.adj[pi].push_back(sepOff[o]);
      finishLevel(L1, rows, BW);
      for (int k : freeNodes)
        if (!isSep[k])
          for (int c = 0; c < D; ++c) L1.rowMap[L1.pieces[pieceOf[k]].rowPtr + nodeRow[k] + c] = off[k] + c;
    }
    std::vector<int4> edgeDst(ne);
    for (int e = 0; e < ne; ++e) {
      const int a = ea[e], b = eb[e];
      int4 d4 = make_int4(0, 0, 0, 0);
      if (pos[a] >= 0 && pos[b] >= 0) {
        if (isSep[a] && isSep[b]) {
          const bool rowB = sepOff[b] > sepOff[a];
          d4 = make_int4(1 | (rowB ? 16 : 0), 0, rowB ? sepOff[b] : sepOff[a], rowB ? sepOff[a] : sepOff[b]);
        } else if (!isSep[a] && !isSep[b]) {
          const bool rowB = nodeRow[b] > nodeRow[a];
          d4 = make_int4(2 | (rowB ? 16 : 0), pieceOf[a], rowB ? nodeRow[b] : nodeRow[a], rowB ? nodeRow[a] : nodeRow[b]);
        } else {
          const bool rowB = !isSep[b];   // the interior node owns the rows
          const int in = rowB ? b : a, sp = rowB ? a : b;
          d4 = make_int4(3 | (rowB ? 16 : 0), pieceOf[in], nodeRow[in], colIn(L1.adj[pieceOf[in]], sepOff[sp]));
        }
      }
      edgeDst[e] = d4;
    }
    // level 2: pieces of cut keyframes; their blocks are copied out of the level-1 separator system
    LevelHost L2;
    std::vector<int4> xDst;   // {kind | transposed << 4, piece, dstRow, dstCol}: 2 band, 3 coupling, 4 right-hand side
    std::vector<int2> xSrc;   // {row, col} offsets in the separator system
    if (nPieces2 > 0) {
      std::vector<int> rows(nPieces2, 0), l2Row(nn, -1);
      for (int k : freeNodes)
        if (l2PieceOf[k] >= 0) { l2Row[k] = rows[l2PieceOf[k]]; rows[l2PieceOf[k]] += D; }
      L2.adj.resize(nPieces2);
      for (int k : freeNodes)
        if (l2PieceOf[k] >= 0)
          for (int o : sepAdj[k])
            if (isRoot[o]) L2.adj[l2PieceOf[k]].push_back(sepOff[o]);
      finishLevel(L2, rows, BW2);
      for (int k : freeNodes) {
        if (l2PieceOf[k] < 0) continue;
        const int pi = l2PieceOf[k];
        for (int c = 0; c < D; ++c) L2.rowMap[L2.pieces[pi].rowPtr + l2Row[k] + c] = sepOff[k] + c;
        xDst.push_back(make_int4(2, pi, l2Row[k], l2Row[k]));   // diagonal block
        xSrc.push_back(make_int2(sepOff[k], sepOff[k]));
        xDst.push_back(make_int4(4, pi, l2Row[k], L2.pieces[pi].cols - 1));
        xSrc.push_back(make_int2(sepOff[k], 0));
        for (int o : sepAdj[k]) {
          if (isRoot[o]) {
            xDst.push_back(make_int4(3 | 16, pi, l2Row[k], colIn(L2.adj[pi], sepOff[o])));
            xSrc.push_back(make_int2(sepOff[o], sepOff[k]));
          } else if (l2Row[o] < l2Row[k]) {
            xDst.push_back(make_int4(2, pi, l2Row[k], l2Row[o]));
            xSrc.push_back(make_int2(sepOff[k], sepOff[o]));
          }
        }
      }
    }
    const std::vector<int4>& tileWork = L1.tileWork;
    const int nDest = L1.nDest, maxRows = L1.maxRows;
    const size_t bandTot = L1.bandTot, yTot = L1.yTot, spTot = L1.spTot;
    partition[7] = nPieces2; partition[8] = nR;
    partition[0] = F; partition[1] = nS / D; partition[2] = nPieces; partition[3] = maxRows; partition[4] = (int)tileWork.size();
    summary[6] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tSym0).count();
    // ---- upload
    dYaw_.upload(yaw, s_); dPitch_.upload(pitch, s_); dRoll_.upload(roll, s_); dT_.upload(t, s_); dQ_.upload(q, s_);
    dYawC_.reserve(nn); dTC_.reserve(3 * (size_t)nn); dQC_.reserve(4 * (size_t)nn);
    dOff_.upload(off, s_); dEa_.upload(ea, s_); dEb_.upload(eb, s_); dEloop_.upload(eloop, s_);
    dEt_.upload(et, s_); dEyaw_.upload(eyaw, s_); dEpitch_.upload(epitch, s_); dEroll_.upload(eroll, s_);
    dEq_.upload(eq, s_); dEsq_.upload(esq, s_);
    dNodePtr_.upload(nodePtr, s_); dNodeEdge_.upload(nodeEdge, s_);
    dSepOff_.upload(sepOff, s_); dNodePiece_.upload(nodePiece, s_); dNodeRow_.upload(nodeRow, s_);
    dPieces_.upload(L1.pieces, s_); dEdgeDst_.upload(edgeDst, s_); dColSep_.upload(L1.colSep, s_); dRowTan_.upload(L1.rowMap, s_);
    dTileWork_.upload(L1.tileWork, s_); dGPtr_.upload(L1.gPtr, s_); dGDst_.upload(L1.gDst, s_); dGSrc_.upload(L1.gSrc, s_);
    dRPtr_.upload(L1.rPtr, s_); dRSrc_.upload(L1.rSrc, s_);
    if (nPieces2 > 0) {
      dPieces2_.upload(L2.pieces, s_); dColSep2_.upload(L2.colSep, s_); dRowMap2_.upload(L2.rowMap, s_);
      dTileWork2_.upload(L2.tileWork, s_); dGPtr2_.upload(L2.gPtr, s_); dGDst2_.upload(L2.gDst, s_); dGSrc2_.upload(L2.gSrc, s_);
      dRPtr2_.upload(L2.rPtr, s_); dRSrc2_.upload(L2.rSrc, s_); dXDst_.upload(xDst, s_); dXSrc_.upload(xSrc, s_);
      dSp2_.reserve(L2.spTot);
    }
    const size_t RD = (size_t)R * D;
    dRes_.reserve((size_t)ne * R); dJa_.reserve(ne * RD); dJb_.reserve(ne * RD);
    // the separator system and band | Y of both levels share one allocation: one memset per iteration
    const size_t workTot = bandTot + yTot + (nPieces2 > 0 ? L2.bandTot + L2.yTot : 0);
    dHS_.reserve((size_t)nS * nS + workTot); dVec_.reserve((size_t)5 * n + 4 * (size_t)nS + 64); dNodeBlk_.reserve((size_t)nn * 36);
    dSp_.reserve(spTot);
    dChol_.reserve(solveReducedScratchDoubles(nR));
    dPartial_.reserve((size_t)8 * kPgMaxPartials);
    dScal_.reserve(PG_NSCAL + sizeof(SolverScalars) / sizeof(double) + 3);   // LM scalars, then the dense solver's: one read-back
    SolverScalars* const dSol = reinterpret_cast<SolverScalars*>(dScal_.p + PG_NSCAL);
    PgDev p;
    std::memset(&p, 0, sizeof(p));
    p.nn = nn; p.ne = ne; p.n = n; p.m = ne * R; p.six = six_ ? 1 : 0; p.D = D; p.R = R;
    p.BW = BW; p.nS = nS; p.nPieces = nPieces; p.maxRows = maxRows;
    p.yaw = dYaw_.p; p.pitch = dPitch_.p; p.roll = dRoll_.p; p.t = dT_.p; p.q = dQ_.p;
    p.yawC = dYawC_.p; p.tC = dTC_.p; p.qC = dQC_.p;
    p.off = dOff_.p; p.ea = dEa_.p; p.eb = dEb_.p; p.eloop = dEloop_.p;
    p.et = dEt_.p; p.eyaw = dEyaw_.p; p.epitch = dEpitch_.p; p.eroll = dEroll_.p; p.eq = dEq_.p; p.esq = dEsq_.p;
    p.res = dRes_.p; p.Ja = dJa_.p; p.Jb = dJb_.p; p.nodePtr = dNodePtr_.p; p.nodeEdge = dNodeEdge_.p;
    p.scale = dVec_.p; p.g = dVec_.p + n; p.colsq = dVec_.p + 2 * (size_t)n; p.y = dVec_.p + 3 * (size_t)n;
    p.rhsS = dVec_.p + 5 * (size_t)n; p.yS = p.rhsS + nS;
    double* ones = p.yS + nS;    // htilC stand-in for the solver's v_C output
    double* vdump = ones + nS;
    p.nodeBlk = dNodeBlk_.p;
    p.partial = dPartial_.p; p.scal = dScal_.p;
    p.sepOff = dSepOff_.p; p.nodePiece = dNodePiece_.p; p.nodeRow = dNodeRow_.p; p.pieces = dPieces_.p;
    p.edgeDst = dEdgeDst_.p; p.colSep = dColSep_.p; p.rowTan = dRowTan_.p; p.tileWork = dTileWork_.p;
    p.gPtr = dGPtr_.p; p.gDst = dGDst_.p; p.gSrc = dGSrc_.p; p.rPtr = dRPtr_.p; p.rSrc = dRSrc_.p;
    p.HS = dHS_.p; p.band = dHS_.p + (size_t)nS * nS; p.Y = p.band + bandTot; p.Sp = dSp_.p;
    p.fail = &dSol->cholFail;
    p.ticket = reinterpret_cast<int*>(dSol + 1);
    PG_HIP_OK(hipMemsetAsync(dSol, 0, sizeof(SolverScalars) + 2 * sizeof(double), s_));
    {
      std::vector<double> one(nS, 1.0);
      PG_HIP_OK(hipMemcpyAsync(ones, one.data(), sizeof(double) * nS, hipMemcpyHostToDevice, s_));
      PG_HIP_OK(hipStreamSynchronize(s_));
    }
    // the dense solver of the BA backend sees the separator system through a DeviceProblem view
    DeviceProblem dp;
    std::memset(&dp, 0, sizeof(dp));
    dp.d = nR; dp.S = p.HS + (size_t)offR * nS + offR; dp.ldS = nS;   // the root: trailing block, solved in place
    dp.gRed = p.rhsS + offR; dp.gFull = p.rhsS + offR; dp.htilC = ones; dp.yC = p.yS + offR; dp.vC = vdump;
    dp.cholL = dChol_.p; dp.scal = dSol;
    // level 2 runs the same piece kernels on a second view: its pieces' rows are separator unknowns (y = yS)
    PgDev p2 = p;
    const int nX = (int)xDst.size();
    if (nPieces2 > 0) {
      p2.BW = BW2; p2.nPieces = nPieces2; p2.maxRows = L2.maxRows;
      p2.pieces = dPieces2_.p; p2.colSep = dColSep2_.p; p2.rowTan = dRowMap2_.p; p2.tileWork = dTileWork2_.p;
      p2.gPtr = dGPtr2_.p; p2.gDst = dGDst2_.p; p2.gSrc = dGSrc2_.p; p2.rPtr = dRPtr2_.p; p2.rSrc = dRSrc2_.p;
      p2.band = p.band + bandTot + yTot; p2.Y = p2.band + L2.bandTot; p2.Sp = dSp2_.p; p2.y = p.yS;
    }
    const size_t ldsFactor2 = (size_t)L2.maxRows * (BW2 + 2) * 8 + 2 * 1024;
    const size_t ldsBack2 = ((size_t)L2.maxRows * (BW2 + 2) + W2 * D + kPgPieceThreads) * 8;
    if (nPieces2 > 0) {
      (void)hipFuncSetAttribute(six_ ? (const void*)k_pg_piece_factor<6, 7> : (const void*)k_pg_piece_factor<4, 3>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFactor2);
      (void)hipFuncSetAttribute(six_ ? (const void*)k_pg_piece_back<6, 7> : (const void*)k_pg_piece_back<4, 3>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBack2);
    }
    const int gE = (ne + 127) / 128, gN = (nn + 127) / 128;
    if (gE > kPgMaxPartials || gN > kPgMaxPartials) throw std::runtime_error("svin_pg: graph too large for the reduction scratch");
    const size_t ldsFactor = (size_t)maxRows * (BW + 2) * 8 + 2 * 512;
    const size_t ldsBack = ((size_t)maxRows * (BW + 2) + w * D + kPgPieceThreads) * 8;
    if (nPieces > 0) {
      (void)hipFuncSetAttribute(six_ ? (const void*)k_pg_piece_factor<6, 4> : (const void*)k_pg_piece_factor<4, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFactor);
      (void)hipFuncSetAttribute(six_ ? (const void*)k_pg_piece_back<6, 4> : (const void*)k_pg_piece_back<4, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBack);
    }
    struct HostScal { double sc[PG_NSCAL]; SolverScalars sol; } hs;
    static_assert(sizeof(SolverScalars) % sizeof(double) == 0, "SolverScalars must pack behind the LM scalars");
    double* const sc = hs.sc;
    auto readScal = [&]() {   // the one blocking read-back of an iteration
      PG_HIP_OK(hipMemcpyAsync(&hs, p.scal, sizeof(double) * PG_NSCAL + sizeof(SolverScalars), hipMemcpyDeviceToHost, s_));
      PG_HIP_OK(hipStreamSynchronize(s_));
    };
    // residuals (+ Jacobians) at the current point or the candidate; finalReduce: the launch also sums the iteration's scalars
    auto evaluate = [&](bool cand, bool withJac, bool finalReduce) {
      if (six_) hipLaunchKernelGGL(k_pg_eval<true>, dim3(gE), dim3(128), 0, s_, p, cand ? 1 : 0, withJac ? 1 : 0, finalReduce ? gN : 0);
      else hipLaunchKernelGGL(k_pg_eval<false>, dim3(gE), dim3(128), 0, s_, p, cand ? 1 : 0, withJac ? 1 : 0, finalReduce ? gN : 0);
    };
    // damped normal equations at the current linearisation -> y (tangent order, scaled space)
    auto solveNormalEquations = [&](double radius, int& nSolves) {
      PG_HIP_OK(hipMemsetAsync(p.HS, 0, sizeof(double) * ((size_t)nS * nS + (nPieces > 0 ? workTot : 0)), s_));   // H_S | band | Y (| level 2)
      if (six_) hipLaunchKernelGGL(k_pg_assemble<6>, dim3(gN + gE), dim3(128), 0, s_, p, radius, gN);
      else hipLaunchKernelGGL(k_pg_assemble<4>, dim3(gN + gE), dim3(128), 0, s_, p, radius, gN);
      if (nPieces > 0) {
        if (six_) hipLaunchKernelGGL((k_pg_piece_factor<6, 4>), dim3(nPieces), dim3(kPgPieceThreads), ldsFactor, s_, p);
        else hipLaunchKernelGGL((k_pg_piece_factor<4, 2>), dim3(nPieces), dim3(kPgPieceThreads), ldsFactor, s_, p);
        hipLaunchKernelGGL(k_pg_piece_schur, dim3((unsigned)tileWork.size()), dim3(256), 0, s_, p);
        hipLaunchKernelGGL(k_pg_sep_gather, dim3((nDest * D * D + 255) / 256 + (nS + 255) / 256), dim3(256), 0, s_, p, nDest,
                           (nDest * D * D + 255) / 256);
      }
      if (nPieces2 > 0) {
        hipLaunchKernelGGL(k_pg_l2_extract, dim3((nX * D * D + 255) / 256), dim3(256), 0, s_, p2, (const int4*)dXDst_.p,
                           (const int2*)dXSrc_.p, nX);
        if (six_) hipLaunchKernelGGL((k_pg_piece_factor<6, 7>), dim3(nPieces2), dim3(kPgPieceThreads), ldsFactor2, s_, p2);
        else hipLaunchKernelGGL((k_pg_piece_factor<4, 3>), dim3(nPieces2), dim3(kPgPieceThreads), ldsFactor2, s_, p2);
        hipLaunchKernelGGL(k_pg_piece_schur, dim3((unsigned)L2.tileWork.size()), dim3(256), 0, s_, p2);
        hipLaunchKernelGGL(k_pg_sep_gather, dim3((L2.nDest * D * D + 255) / 256 + (nS + 255) / 256), dim3(256), 0, s_, p2, L2.nDest,
                           (L2.nDest * D * D + 255) / 256);
      }
      PG_HIP_OK(hipEventRecord(evA_[nSolves % kPgMaxTimed], s_));
      launchSolveReduced(dp, s_, 0.0, false, false);
      PG_HIP_OK(hipEventRecord(evB_[nSolves % kPgMaxTimed], s_));
      ++nSolves;
      if (nPieces2 > 0) {
        if (six_) hipLaunchKernelGGL((k_pg_piece_back<6, 7>), dim3(nPieces2), dim3(kPgPieceThreads), ldsBack2, s_, p2);
        else hipLaunchKernelGGL((k_pg_piece_back<4, 3>), dim3(nPieces2), dim3(kPgPieceThreads), ldsBack2, s_, p2);
      }
      if (nPieces > 0) {
        const int nScatter = (nn + kPgPieceThreads - 1) / kPgPieceThreads;
        if (six_) hipLaunchKernelGGL((k_pg_piece_back<6, 4>), dim3(nPieces + nScatter), dim3(kPgPieceThreads), ldsBack, s_, p);
        else hipLaunchKernelGGL((k_pg_piece_back<4, 2>), dim3(nPieces + nScatter), dim3(kPgPieceThreads), ldsBack, s_, p);
      } else {
        hipLaunchKernelGGL(k_pg_scatter_sep, dim3(gN), dim3(128), 0, s_, p);
      }
    };
    int nSolves = 0;
    const auto t0 = std::chrono::steady_clock::now();
    // ---- Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy, default options
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    evaluate(false, true, false);
    hipLaunchKernelGGL(k_pg_reduce, dim3(1), dim3(256), 0, s_, p, (int)PG_COST, gE, 0);
    readScal();
    double x_cost = sc[PG_COST];
    summary[0] = x_cost;
    bool needLinearize = true, initScale = true;
    int iteration = 0, invalid = 0, successful = 0, termination = 1;
    while (true) {
      const bool freshLinearization = needLinearize;
      if (needLinearize) {  // gradient / column norms / J^T J blocks of the current linearisation
        if (six_) hipLaunchKernelGGL(k_pg_node<6>, dim3(gN), dim3(128), 0, s_, p, initScale ? 1 : 0);
        else hipLaunchKernelGGL(k_pg_node<4>, dim3(gN), dim3(128), 0, s_, p, initScale ? 1 : 0);
        initScale = false;
      }
      if (iteration >= maxIter_) { termination = 1; break; }
      if (radius <= min_radius) { termination = 0; break; }
      // The step is enqueued before the gradient check's scalar is back (one read-back per iteration instead of
      // three); when the gradient test fires the step is simply not used -- Ceres would not have computed it.
      solveNormalEquations(radius, nSolves);
      if (six_) hipLaunchKernelGGL(k_pg_model_plus<6>, dim3(gE + gN), dim3(128), 0, s_, p, gE);
      else hipLaunchKernelGGL(k_pg_model_plus<4>, dim3(gE + gN), dim3(128), 0, s_, p, gE);
      evaluate(true, false, true);   // + the iteration's scalars (and the gradient max of this linearisation)
      readScal();
      if (freshLinearization && sc[PG_GRADMAX] <= gradient_tolerance) { termination = 0; break; }
      ++iteration;
      const double model_cost_change = -sc[PG_MODEL];
      needLinearize = false;
      if (hs.sol.cholFail != 0 || !(model_cost_change > 0.0) || !std::isfinite(sc[PG_STEP2])) {  // HandleInvalidStep
        if (++invalid >= 5) { termination = 3; break; }
        radius /= decrease_factor; decrease_factor *= 2.0;
        continue;
      }
      invalid = 0;
      const double step_norm = std::sqrt(sc[PG_STEP2]), x_norm = std::sqrt(sc[PG_X2]);
      if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { termination = 0; break; }
      const double cand_cost = sc[PG_COST];
      const double cost_change = x_cost - cand_cost;
      if (std::fabs(cost_change) <= function_tolerance * x_cost) { termination = 0; break; }
      const double rel = cost_change / model_cost_change;
      if (rel > min_relative_decrease) {
        std::swap(p.yaw, p.yawC); std::swap(p.t, p.tC); std::swap(p.q, p.qC);
        evaluate(false, true, false);   // residuals + Jacobians at the accepted point (HandleSuccessfulStep)
        x_cost = cand_cost;
        radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
        decrease_factor = 2.0;
        needLinearize = true;
        ++successful;
      } else {
        radius /= decrease_factor; decrease_factor *= 2.0;
      }
    }
    PG_HIP_OK(hipStreamSynchronize(s_));
    summary[5] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    summary[1] = x_cost; summary[2] = iteration; summary[3] = termination; summary[4] = successful;
    summary[7] = 0;
    for (int i = 0; i < std::min(nSolves, kPgMaxTimed); ++i) {   // HIP events on the solve's own stream
      float ms = 0;
      PG_HIP_OK(hipEventElapsedTime(&ms, evA_[i], evB_[i]));
      summary[7] += 1e-3 * ms;
    }
    partition[5] = nR; partition[6] = std::min(nSolves, kPgMaxTimed);
    // ---- write back, drift update, keyframes after cur (PoseGraph.cpp:340-375 / :504-534)
    std::vector<double> hy(nn), ht(3 * (size_t)nn), hq(4 * (size_t)nn);
    PG_HIP_OK(hipMemcpy(hy.data(), p.yaw, sizeof(double) * nn, hipMemcpyDeviceToHost));
    PG_HIP_OK(hipMemcpy(ht.data(), p.t, sizeof(double) * 3 * nn, hipMemcpyDeviceToHost));
    PG_HIP_OK(hipMemcpy(hq.data(), p.q, sizeof(double) * 4 * nn, hipMemcpyDeviceToHost));
    const auto tWb0 = std::chrono::steady_clock::now();
    writeBack(hy, pitch, roll, ht, hq, kfOfLocal, cur);
    if (optOn(kOptPgTiming)) {
      auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return 1e3 * std::chrono::duration<double>(b - a).count(); };
      const auto tEnd = std::chrono::steady_clock::now();
      std::printf("[pg] problem construction %.3f ms, symbolic %.3f ms, upload + setup %.3f ms, LM loop %.3f ms, download %.3f ms, write back %.3f ms\n",
                  sec(tCall0, tSym0), 1e3 * summary[6], sec(tSym0, t0) - 1e3 * summary[6], 1e3 * summary[5], sec(t0, tWb0) - 1e3 * summary[5], sec(tWb0, tEnd));
    }
    return 1;
  }

  void writeBack(const std::vector<double>& hy, const std::vector<double>& pitch, const std::vector<double>& roll,
                 const std::vector<double>& ht, const std::vector<double>& hq, const std::vector<int>& kfOfLocal, int cur) {
    const int nn = (int)kfOfLocal.size();
    for (int k = 0; k < nn; ++k) {
      Keyframe& kf = kfs[kfOfLocal[k]];
      for (int c = 0; c < 3; ++c) kf.P[c] = ht[3 * k + c];
      if (!six_) {
        double Rm[9], qq[4];
        hostYpr2R(hy[k], pitch[k], roll[k], Rm);
        hostR2q(Rm, qq);   // tmp_q = ypr2R(...); tmp_r = tmp_q.toRotationMatrix()
        hostQ2R(qq, kf.Rp);
      } else {
        hostQ2R(&hq[4 * k], kf.Rp);
      }
    }
    if (nn == 0 || kfs[kfOfLocal[nn - 1]].index != cur) return;
    const Keyframe& ck = kfs[kfOfLocal[nn - 1]];
    const double* Rs = ck.Rs;
    if (!six_) {
      yawDrift = hostYawOfR(ck.Rp) - hostYawOfR(Rs);
      hostYpr2R(yawDrift, 0, 0, rDrift);
    } else {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) rDrift[3 * r + c] = ck.Rp[r] * Rs[c] + ck.Rp[3 + r] * Rs[3 + c] + ck.Rp[6 + r] * Rs[6 + c];
      yawDrift = hostYawOfR(rDrift);
    }
    for (int r = 0; r < 3; ++r)
      tDrift[r] = ck.P[r] - (rDrift[3 * r] * ck.t[0] + rDrift[3 * r + 1] * ck.t[1] + rDrift[3 * r + 2] * ck.t[2]);
    for (size_t k = (size_t)kfOfLocal[nn - 1] + 1; k < kfs.size(); ++k) applyDrift(kfs[k]);
  }

 private:
  bool six_;
  int maxIter_;
  hipStream_t s_ = nullptr;
  static constexpr int kPgMaxTimed = 64;
  hipEvent_t evA_[kPgMaxTimed] = {}, evB_[kPgMaxTimed] = {};
  Buf<double> dYaw_, dPitch_, dRoll_, dT_, dQ_, dYawC_, dTC_, dQC_, dEt_, dEyaw_, dEpitch_, dEroll_, dEq_, dEsq_;
  Buf<double> dRes_, dJa_, dJb_, dHS_, dVec_, dChol_, dPartial_, dScal_, dNodeBlk_, dSp_;
  Buf<int> dOff_, dEa_, dEb_, dEloop_, dNodePtr_, dNodeEdge_, dSepOff_, dNodePiece_, dNodeRow_, dColSep_, dRowTan_, dGPtr_, dRPtr_;
  Buf<PgPiece> dPieces_, dPieces2_;
  Buf<int4> dEdgeDst_, dTileWork_, dGSrc_, dTileWork2_, dGSrc2_, dXDst_;
  Buf<int2> dGDst_, dRSrc_, dGDst2_, dRSrc2_, dXSrc_;
  Buf<int> dColSep2_, dRowMap2_, dGPtr2_, dRPtr2_;
  Buf<double> dSp2_;
};

}  // namespace pg
}  // namespace svin

// ---------------------------------------------------------------- C ABI
struct svin_pg {
  svin::pg::PoseGraph g;
  svin_pg(int device, bool six, int it) : g(device, six, it) {}
};
static thread_local std::string g_pgError;
extern "C" {
svin_pg* svin_pg_create(int device, int six_dof, int max_iterations) {
  try {
    return new svin_pg(device, six_dof != 0, max_iterations);
  } catch (const std::exception& e) {
    g_pgError = e.what();
    return nullptr;
  }
}
void svin_pg_destroy(svin_pg* h) { delete h; }
const char* svin_pg_last_error(void) { return g_pgError.c_str(); }
int svin_pg_add_keyframe(svin_pg* h, int index, int sequence, const double* t, const double* q, int loop_index,
                         const double* loop_rel_t, const double* loop_rel_q, double loop_rel_yaw_deg) try {
  if (!h || !t || !q) return -1;
  svin::pg::Keyframe kf;
  kf.index = index; kf.sequence = sequence;
  std::memcpy(kf.t, t, sizeof(kf.t));
  std::memcpy(kf.q, q, sizeof(kf.q));
  if (loop_index >= 0) {
    if (!loop_rel_t || !loop_rel_q) return -1;
    kf.hasLoop = true; kf.loopIndex = loop_index;
    std::memcpy(kf.loopT, loop_rel_t, sizeof(kf.loopT));
    std::memcpy(kf.loopQ, loop_rel_q, sizeof(kf.loopQ));
    kf.loopYaw = loop_rel_yaw_deg;
  }
  svin::pg::hostR2ypr(kf.q, kf.ypr);
  svin::pg::hostQ2R(kf.q, kf.Rs);
  h->g.applyDrift(kf);
  h->g.kfs.push_back(kf);
  return 1;
} catch (const std::exception& e) {   // (allocation failure of the keyframe list: nothing crosses the C ABI)
  g_pgError = e.what();
  return -3;
}
int svin_pg_num_keyframes(const svin_pg* h) { return h ? (int)h->g.kfs.size() : -1; }
int svin_pg_optimize(svin_pg* h, int earliest_loop_index, int cur_index) {
  if (!h) return -1;
  try {
    return h->g.optimize(earliest_loop_index, cur_index);
  } catch (const std::exception& e) {
    g_pgError = e.what();
    return -3;
  }
}
int svin_pg_get_pose(const svin_pg* h, int k, double* t, double* q) {
  if (!h || k < 0 || k >= (int)h->g.kfs.size()) return -2;
  if (t) std::memcpy(t, h->g.kfs[k].P, sizeof(double) * 3);
  if (q) svin::pg::hostR2q(h->g.kfs[k].Rp, q);
  return 1;
}
int svin_pg_get_poses(const svin_pg* h, int n, double* t, double* q) {
  if (!h || n < 0 || n > (int)h->g.kfs.size()) return -2;
  for (int k = 0; k < n; ++k) {
    if (t) std::memcpy(t + 3 * (size_t)k, h->g.kfs[k].P, sizeof(double) * 3);
    if (q) svin::pg::hostR2q(h->g.kfs[k].Rp, q + 4 * (size_t)k);
  }
  return 1;
}
int svin_pg_get_drift(const svin_pg* h, double* yaw_drift_deg, double* r_drift, double* t_drift) {
  if (!h) return -1;
  if (yaw_drift_deg) *yaw_drift_deg = h->g.yawDrift;
  if (r_drift) std::memcpy(r_drift, h->g.rDrift, sizeof(double) * 9);
  if (t_drift) std::memcpy(t_drift, h->g.tDrift, sizeof(double) * 3);
  return 1;
}
int svin_pg_set_partition(svin_pg* h, int piece_keyframes, int dense_keyframes) {
  if (!h || piece_keyframes < 0 || dense_keyframes < 0) return -1;
  if (piece_keyframes > 0) h->g.pieceLen_ = piece_keyframes;
  h->g.denseNodes_ = dense_keyframes;
  return 1;
}
int svin_pg_set_levels(svin_pg* h, int levels, int level2_piece_keyframes) {
  if (!h || levels < 1 || levels > 2 || level2_piece_keyframes < 0) return -1;
  h->g.level2_ = levels == 2;
  if (level2_piece_keyframes > 0) h->g.l2Len_ = level2_piece_keyframes;
  return 1;
}
int svin_pg_get_partition(const svin_pg* h, double* out10) {
  if (!h || !out10) return -1;
  for (int i = 0; i < 5; ++i) out10[i] = h->g.partition[i];
  out10[5] = h->g.summary[6];
  out10[6] = h->g.partition[5];
  out10[7] = h->g.partition[6];
  out10[8] = h->g.summary[7];
  out10[9] = h->g.partition[7];
  return 1;
}
int svin_pg_summary(const svin_pg* h, double* out6) {
  if (!h || !out6) return -1;
  std::memcpy(out6, h->g.summary, sizeof(double) * 6);
  return 1;
}
}
