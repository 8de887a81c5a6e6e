// svin_amd HIP kernels for gfx950 (CDNA4, wave64).  No CUDA shims, no CPU fallbacks.
//
// K1  k_eval_reproj      reprojection residual + minimal Jacobians + Cauchy corrector (fused K4)
// K2  k_eval_factors     IMU (incl. conditional re-preintegration) and the small unary/binary factors
// K3  k_prior_*          marginalisation prior in H-space
// K5  k_schur_dense      landmark elimination as a Gram matrix S = A - G G^T on v_mfma_f64_16x16x4_f64 (narrow windows);
//     k_schur_panels     the same in 96-row panel pairs (wide windows); k_schur: pairwise blocks with LDS atomics (fallback)
//     k_reduce_slabs     deterministic sum of the per-workgroup slabs
// K6  k_chol_solve_lds   blocked Cholesky of the reduced system in LDS (d <= 176; <true>: d = 177 .. 180, the rows beyond eliminated while loading), trailing update on MFMA
//     k_chol_solve_ll    (176 < d <= 272) left-looking variant: at most 72 live tiles in LDS behind a slot map
//     k_big_chol_chain   (d > 272) one-launch tile Cholesky over many workgroups + super-panel backward substitution
//     k_sb_factor / k_sb_forward / k_sb_load / k_sb_back   wide windows: the chain of 9x9 speed / bias blocks eliminated by cyclic
//                        reduction ahead of whichever of the three dense solvers the kept rows select
// K7  k_post_solve       back-substitution, J*v / J*y sums, dogleg step and retraction (k_step_retract for rejected steps)
// K8  cost reductions    per-block partials + single-block final reduce (deterministic)
// K9  k_landmark_quality
// Reference arithmetic: see dmath.hpp and the per-kernel comments.
#include "kernels.hpp"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include "tile16.hpp"

#ifndef SVIN_BATCH_OCC_SCHUR
#define SVIN_BATCH_OCC_SCHUR 2   // waves per SIMD the batched Schur / post-solve forms are held to (register budget 512 / n)
#endif
#ifndef SVIN_BATCH_OCC_POST
#define SVIN_BATCH_OCC_POST 2
#endif
namespace svin {

// hipFuncSetAttribute once per kernel and size (it is a driver call: ~2 us on the host path of every launch otherwise)
void ensureDynamicLds(const void* fn, size_t bytes) {
  static std::mutex mtx;
  static std::unordered_map<unsigned long long, size_t> granted;   // (device, kernel) -> bytes
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long key = (unsigned long long)reinterpret_cast<uintptr_t>(fn) * 64ull + (unsigned long long)dev;
  std::lock_guard<std::mutex> lock(mtx);
  auto it = granted.find(key);
  if (it != granted.end() && it->second >= bytes) return;
  // a refused attribute is not cached: the launch that follows fails, its hipGetLastError() check reports it, and the
  // next caller tries again instead of trusting a size that was never granted
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize, ") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  granted[key] = bytes;
}

// (the A/B switches of the reduced solve -- SVIN_NO_LL, SVIN_NO_SB_ELIM, SVIN_NO_LDS_BORDER -- live in the library's one option
// table since round 6: options.hpp)


// ---------------------------------------------------------------- small helpers
// Cross-lane sums without LDS: __shfl_xor compiles to ds_bpermute (one LDS round trip per step and 32-bit half), DPP
// row rotations are plain VALU operand modifiers.  kCtrl: row_ror:N = 0x120 + N (lane i of a 16-lane row reads lane
// (i - N) & 15).  Every lane of the row (rowSum16 / rowMax16) or of the wave (waveSum / waveMax) must be active.
template <int kCtrl>
__device__ __forceinline__ double dppRowMov(double v) {
  // (only used with row rotations: every lane receives a value, so no `old` operand has to be zeroed first)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), kCtrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), kCtrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rowSum16(double v) {  // sum over the 16 lanes of a DPP row, in every lane
  v += dppRowMov<0x128>(v);
  v += dppRowMov<0x124>(v);
  v += dppRowMov<0x122>(v);
  v += dppRowMov<0x121>(v);
  return v;
}
__device__ __forceinline__ double rowMax16(double v) {
  v = fmax(v, dppRowMov<0x128>(v));
  v = fmax(v, dppRowMov<0x124>(v));
  v = fmax(v, dppRowMov<0x122>(v));
  v = fmax(v, dppRowMov<0x121>(v));
  return v;
}
// Workgroup barrier that only orders LDS traffic: __syncthreads() also waits for every outstanding global access of the
// thread, so a global store issued just before it holds the whole workgroup for a memory round trip.
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// DPP move for wave-wide scans: kCtrl row_shr:N = 0x110 + N (within 16-lane rows), row_bcast:15 = 0x142 (lane 15 of a row
// to the whole next row; row mask 0xa = rows 1 and 3), row_bcast:31 = 0x143 (lane 31 to rows 2 and 3: row mask 0xc),
// wave_shr:1 = 0x138.  Lanes that receive nothing get 0.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dppScanMov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double waveSum(double v) {
  v = rowSum16(v);
  return (readlaneD(v, 0) + readlaneD(v, 16)) + (readlaneD(v, 32) + readlaneD(v, 48));
}
__device__ __forceinline__ double waveMax(double v) {
  v = rowMax16(v);
  return fmax(fmax(readlaneD(v, 0), readlaneD(v, 16)), fmax(readlaneD(v, 32), readlaneD(v, 48)));
}
// block-wide sum; result valid in thread 0.  `red` must hold blockDim/64 doubles.
__device__ __forceinline__ double blockSum(double v, double* red) {
  v = waveSum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x + 63) / 64; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Kernel-argument fields a kernel needs first, fetched in ONE batch of scalar loads: left to the compiler every field of the
// by-value DeviceProblem is loaded at its first use with its own s_waitcnt -- about ten serial scalar-memory latencies before
// a block issues its first real load (the empty "camera block" of k_post_solve finished 2.5 us after it started).
#define SVIN_ARGS(...) asm volatile("" ::__VA_ARGS__)
#define SA(x) "s"(x)

#ifdef SVIN_TRACE
// in-kernel timeline (instrumented variant builds only, tools/build_variant.sh trace -DSVIN_TRACE): 100 MHz wall clock
// stamps of the LAST launch, slot 2k = earliest / slot 2k+1 = latest stamp any block recorded at trace point k
__device__ unsigned long long g_trace[128];
__device__ __forceinline__ void tracePoint(int k) {
  if (threadIdx.x == 0) {
    const unsigned long long now = wall_clock64();
    atomicMin(&g_trace[2 * k], now);
    atomicMax(&g_trace[2 * k + 1], now);
  }
}
extern "C" void svin_debug_trace(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long init[128];
    for (int i = 0; i < 128; ++i) init[i] = (i & 1) ? 0ull : ~0ull;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), init, sizeof(init));
    return;
  }
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), 128 * 8);
}
#define TRACE(k) tracePoint(k)
#else
#define TRACE(k) ((void)0)
#endif

// side stream of a solver stream (one per device and stream, created on first use, kept for the life of the process) with the
// events of the fork / join around what runs on it
struct SideLane { hipStream_t side = nullptr; hipEvent_t fork = nullptr, mid = nullptr, join = nullptr; };
#define HIP_LAUNCH_OK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
static std::mutex g_sideLaneMutex;
static std::map<std::pair<int, hipStream_t>, SideLane> g_sideLanes;
// (called by the owner of `s` before it destroys the stream: the side stream has nothing in flight that `s` has not waited for)
void releaseSideLane(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_sideLaneMutex);
  for (auto it = g_sideLanes.begin(); it != g_sideLanes.end();) {
    if (it->first.second != s) { ++it; continue; }
    SideLane& l = it->second;
    if (l.side) { (void)hipStreamSynchronize(l.side); (void)hipStreamDestroy(l.side); }
    if (l.fork) (void)hipEventDestroy(l.fork);
    if (l.mid) (void)hipEventDestroy(l.mid);
    if (l.join) (void)hipEventDestroy(l.join);
    it = g_sideLanes.erase(it);
  }
}
static SideLane& sideLaneOf(hipStream_t s) {
  int dev = 0;
  HIP_LAUNCH_OK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_sideLaneMutex);
  SideLane& l = g_sideLanes[std::make_pair(dev, s)];
  if (!l.side) {
    HIP_LAUNCH_OK(hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking));
    HIP_LAUNCH_OK(hipEventCreateWithFlags(&l.fork, hipEventDisableTiming));
    HIP_LAUNCH_OK(hipEventCreateWithFlags(&l.mid, hipEventDisableTiming));
    HIP_LAUNCH_OK(hipEventCreateWithFlags(&l.join, hipEventDisableTiming));
  }
  return l;
}
// partial-sum slots in p.partial (each slot holds up to kMaxPartials doubles)
constexpr int kMaxPartials = 4096;
enum PartialSlot : int {
  PS_COST_REPROJ = 0, PS_COST_FACTORS = 1, PS_JV_SQ = 2, PS_JV_DOT = 3, PS_STEP = 4, PS_XNORM = 5,
  PS_GHAT = 6, PS_GNHAT = 7, PS_GDOTGN = 8, PS_GRADMAX = 9, PS_JY_SQ = 10, PS_JVJY = 11, PS_JY_DOT = 12, PS_COUNT = 13
};

enum Ticket : int { TK_POST = 0, TK_STEP = 1, TK_EVAL = 2 };
// last-block-done: returns true in every thread of the block that finishes last (its loads see all other
// blocks' partials); the summation order over the partial slots is fixed, so the result is deterministic
__device__ __forceinline__ bool lastBlockDone(unsigned int* ticket, int* flagLds) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int tk = atomicAdd(ticket, 1u);
    *flagLds = (tk == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  const bool last = *flagLds != 0;
  if (last) __threadfence();
  return last;
}

// The same ticket without the L2 write-back: for kernels whose blocks hand ONLY values written with cstore() (agent-
// scope relaxed atomics: sc1, coherent by themselves) to the last block, which reads them with cload().  A plain
// __threadfence() makes every block write back its XCD's dirty L2 lines -- megabytes of Jacobians right after K1.
__device__ __forceinline__ void cstore(double* q, double v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double cload(const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool lastBlockDoneLight(unsigned int* ticket, int* flagLds) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this thread's cstore()s have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int tk = atomicAdd(ticket, 1u);
    *flagLds = (tk == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  return *flagLds != 0;
}

// total cost = reprojection partials (nA blocks) + factor partials (nB) + prior, into SolverScalars (whole block)
__global__ __launch_bounds__(64) void k_publish_scalars(const SolverScalars* scal, ScalarMailbox* mailbox, unsigned long long seq) {
  constexpr int nD = (int)(sizeof(SolverScalars) / sizeof(double));
  const int t = threadIdx.x;
  if (t < nD) {
    const double v = __hip_atomic_load(reinterpret_cast<const double*>(scal) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    reinterpret_cast<volatile double*>(&mailbox->scal)[t] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) *reinterpret_cast<volatile unsigned long long*>(&mailbox->seq) = seq;
}
// Sharded mode: the message of the per-iteration all-reduce is the LOWER triangle of S (the solvers read nothing else)
// followed by gRed | gFull | hC, packed row by row into p.cholL (free until the solver starts): d (d + 1) / 2 + 3 d doubles
// instead of d^2 + 3 d.  One workgroup per row; block d copies the vectors.
__global__ __launch_bounds__(256) void k_pack_lower(DeviceProblem p, int unpack) {
  const int d = p.d, i = blockIdx.x, ld = p.ldS ? p.ldS : d;
  double* msg = p.cholL;
  if (i < d) {
    double* row = p.S + (size_t)i * ld;
    double* m = msg + (size_t)i * (i + 1) / 2;
    for (int j = threadIdx.x; j <= i; j += blockDim.x) {
      if (unpack) row[j] = m[j]; else m[j] = row[j];
    }
  } else {
    double* m = msg + (size_t)d * (d + 1) / 2;
    for (int j = threadIdx.x; j < 3 * d; j += blockDim.x) {
      double* v = (j < d) ? p.gRed + j : ((j < 2 * d) ? p.gFull + (j - d) : p.hC + (j - 2 * d));
      if (unpack) *v = m[j]; else m[j] = *v;
    }
  }
}
size_t packedSystemDoubles(const DeviceProblem& p) { return (size_t)p.d * (p.d + 1) / 2 + (size_t)3 * p.d; }
void launchPackSystem(const DeviceProblem& p, bool unpack, hipStream_t s) {
  if (p.d <= 0) return;
  hipLaunchKernelGGL(k_pack_lower, dim3(p.d + 1), dim3(256), 0, s, p, unpack ? 1 : 0);
}
// Sharded mode: the wall-clock time limit (Estimator::setOptimizationTimeLimit) is the one host decision that is not a
// function of all-reduced numbers.  Each rank writes its own vote into the free slots of scalar group A right before that
// group's all-reduce; every rank then reads the same sum and stops (or not) together.
__global__ __launch_bounds__(64) void k_set_stop_vote(SolverScalars* scal, double vote) {
  if (threadIdx.x == 0) { scal->spareA0 = vote; scal->spareA1 = 0.0; }
}
void launchSetStopVote(SolverScalars* scal, double vote, hipStream_t s) {
  hipLaunchKernelGGL(k_set_stop_vote, dim3(1), dim3(64), 0, s, scal, vote);
}
void launchPublishScalars(const SolverScalars* scal, ScalarMailbox* mailbox, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(64), 0, s, scal, mailbox, seq);
}

// Block-wide reduction of K values at once: out[k] valid in threads 0..K-1 (as `mine`), `red` holds 4*K doubles.
template <int K>
__device__ __forceinline__ double blockSumK(const double (&v)[K], double* red, int kMaxIndex) {
  // 16-lane row sums by DPP, the rows (four per wave) meet in LDS: `red` holds 4 * (blockDim / 64) * K doubles.
  // (Finishing each wave's sum in registers first costs 11 more instructions per value on what is usually a serial tail.)
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 4, nRows = (blockDim.x + 15) >> 4;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = (k == kMaxIndex) ? rowMax16(v[k]) : rowSum16(v[k]);
    if ((lane & 15) == 0) red[row * K + k] = s;
  }
  __syncthreads();
  double mine = 0;
  if ((int)threadIdx.x < K) {
    if (nRows == 16) {   // 256 threads: all sixteen partials requested at once (with a run-time bound: one LDS latency per term)
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = red[i * K + threadIdx.x];
#pragma unroll
      for (int i = 0; i < 16; ++i) mine = ((int)threadIdx.x == kMaxIndex) ? fmax(mine, x[i]) : mine + x[i];
    } else {
      for (int i = 0; i < nRows; ++i) {
        const double x = red[i * K + threadIdx.x];
        mine = ((int)threadIdx.x == kMaxIndex) ? fmax(mine, x) : mine + x;
      }
    }
  }
  return mine;
}

// Tail of the cost evaluation (last block): everything it needs from memory is requested in ONE round trip -- the
// two partial lists AND the other scalars of the record (written by earlier kernels) -- then one block sum; the record
// is published from registers (no store -> fence -> re-load of the freshly written cost fields).  `red`: >= 72 doubles.
// nDefer > 0: the reprojection blocks also took the landmark half of the fused step; their step / state norm partials are
// added to the (block-only) sums k_post_solve left in the record.
__device__ __forceinline__ void reduceCost(const DeviceProblem& p, int nA, int nB, double* red, int nDefer = 0) {
  const int t = threadIdx.x;
  constexpr int nD = (int)(sizeof(SolverScalars) / sizeof(double));
  static_assert(nD <= 64, "SolverScalars must fit one wave-wide store");
  double s[4] = {0, 0, 0, 0};
  for (int i = t; i < nA; i += blockDim.x) s[0] += cload(p.partial + (size_t)PS_COST_REPROJ * kMaxPartials + i);
  for (int i = t; i < nB; i += blockDim.x) s[1] += cload(p.partial + (size_t)PS_COST_FACTORS * kMaxPartials + i);
  for (int i = t; i < nDefer; i += blockDim.x) {
    s[2] += cload(p.partial + (size_t)PS_STEP * kMaxPartials + i);
    s[3] += cload(p.partial + (size_t)PS_XNORM * kMaxPartials + i);
  }
  double rec = (t < nD) ? cload(reinterpret_cast<const double*>(p.scal) + t) : 0.0;  // slot 3 = costPrior of this evaluation (cstore()d)
  const double mine = blockSumK<4>(s, red, -1);   // red[0..63]
  double* fields = red + 64;
  if (t < 4 && t != 3) fields[t == 2 ? 4 : t] = mine;
  if (t == 3) { fields[5] = mine; fields[3] = (p.ownsCamera && p.priorM > 0) ? rec : 0.0; }
  __syncthreads();
  const double a = fields[0], bf = fields[1], pr = fields[3];   // (sharded: this rank's factors -- the sum over ranks follows)
  if (t == 0) rec = a + bf + pr;
  if (t == 1) rec = a;
  if (t == 2) rec = bf;
  if (t == 3) rec = pr;
  if (nDefer > 0 && t == 4) rec += fields[4];   // stepNormSq
  if (nDefer > 0 && t == 5) rec += fields[5];   // xNormSq
  if (t < 4 || (nDefer > 0 && t < 6)) reinterpret_cast<double*>(p.scal)[t] = rec;
  if (p.mailbox) {
    // publish everything the host needs for its accept/reject decision: the scalars as ONE wave-wide store to the
    // pinned host page (25 serial stores + two system fences cost ~18 us of every iteration), then the sequence number
    TRACE(21);
    if (t < nD) reinterpret_cast<volatile double*>(&p.mailbox->scal)[t] = rec;
    __threadfence_system();
    TRACE(22);
    __syncthreads();
    if (t == 0) *reinterpret_cast<volatile unsigned long long*>(&p.mailbox->seq) = p.mailboxSeq;
  }
}


// 1/x to about one ulp without the IEEE division sequence: v_rcp_f64 and two Newton steps
__device__ __forceinline__ double rcpNewton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
// ---- 16x16 diagonal block in registers (wave 0, lane i = row i), cross-lane traffic through v_readlane.
#ifdef SVIN_CHOL_TIMING
__device__ double g_cholDbg[4];
void debugCholTiming(double* out, bool reset) {
  if (reset) { double z[4] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cholDbg), z, sizeof(z)); return; }
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cholDbg), 32);
}
#endif
// One DPP move of a double (two 32-bit halves); rows outside kRowMask keep `old`
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dppMovD(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), kCtrl, kRowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), kCtrl, kRowMask, 0xf, false);
  return __hiloint2double(hi, lo);
}
// a DPP move that writes every lane (no `old` operand to initialise)
template <int kCtrl>
__device__ __forceinline__ double dppBcastD(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), kCtrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), kCtrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// v replicated over the four lane rows (lane (g, c) holds v[c]) -> out[q] = v[4 q + g]: row g rotated left by g (row_ror
// by 16 - g, on rows 1-3), then lane 4 q of every row broadcast to its row (row_newbcast)
__device__ __forceinline__ void colToRowForm(double v, double (&out)[4]) {
  double u = v;
  u = dppMovD<0x120 + 15, 0x2>(u, v);
  u = dppMovD<0x120 + 14, 0x4>(u, v);
  u = dppMovD<0x120 + 13, 0x8>(u, v);
  out[0] = dppBcastD<0x150 + 0>(u);
  out[1] = dppBcastD<0x150 + 4>(u);
  out[2] = dppBcastD<0x150 + 8>(u);
  out[3] = dppBcastD<0x150 + 12>(u);
}
// sum of a value over the four lane rows (lanes c, 16 + c, 32 + c, 48 + c), on every lane: three swaps, two additions
__device__ __forceinline__ double sumLaneRows(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // rows [v0 v0 v2 v2], [v1 v1 v3 v3]
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);  // [s01 s01 s23 s23]
  const unsigned slo = (unsigned)__double2loint(s), shi = (unsigned)__double2hiint(s);
  const auto l32 = __builtin_amdgcn_permlane32_swap(slo, slo, false, false);  // [s01 x4], [s23 x4]
  const auto h32 = __builtin_amdgcn_permlane32_swap(shi, shi, false, false);
  return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}
__device__ __forceinline__ void cholDiag16Reg(double* D, double* dinv, int lane, int* failFlag) {
  d4_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = D[((lane >> 4) + 4 * r) * kPanelLd + (lane & 15)];
  cholDiag16Acc(acc, D, dinv, lane, failFlag);
}

// ================================================================ K1: reprojection evaluation
// The linearisation buffers are write-once streams (160-256 B per observation, nothing of them is re-read by this
// kernel): non-temporal stores keep them from thrashing L2 on their way to HBM -- measured 4.3 -> 6.4 TB/s on the
// HBM-resident batch, neutral for the single cache-resident window of the solver (profiles/r01_k1_variants.txt).
// Round 4: a tile-major output (the 20 rows of 64 observations kept together: one contiguous 10 KB tile per wave instead of twenty
// streams N x 8 bytes apart) was measured on the roofline batch and changes nothing (6.4 vs 6.25 TB/s at 1 GB, 4.3 vs 4.3 at 4 GB):
// what bounds the kernel beyond ~1.5 GB of footprint is the memory system, not the layout -- a plain device-to-device copy falls
// from 5.4 to 4.4 TB/s over the same range (tools/k1_sweep.py).
#define K1_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
// One block of the reprojection evaluation: `block` is the block index within the evaluation (not necessarily
// blockIdx.x: the fused evaluation kernel runs these next to the small-factor blocks), `smem` holds
// (nPose + nExt) * 7 doubles + nCam camera models, `red` 4 doubles.
// deferred landmark retraction (fused step): inputs of x_cand = x + (cg v_l - cn y_l) and where the results go
struct LmDefer {   // passed BY VALUE (through a pointer it lands in scratch and every access turns into scratch + flat loads)
  int on = 0;
  double cg = 0, cn = 0;
  const double* vL = nullptr;
  const double* yL = nullptr;
  const int* lmPtr = nullptr;
  double* lmC = nullptr;
  double* stepPartial = nullptr;
  double* xPartial = nullptr;
};
template <bool ROBUST, bool WITH_EXT>
__device__ __forceinline__ void evalReprojBlock(int block, double* smem, double* red, int N, int nPose, int nExt, int nCam,
                                                const double* __restrict__ pose, const double* __restrict__ ext,
                                                const double* __restrict__ lm, const CameraModel* __restrict__ cams,
                                                const double* __restrict__ obsUv, const double* __restrict__ obsW,
                                                const uint32_t* __restrict__ obsIdx, const int* __restrict__ obsLm,
                                                double* __restrict__ r, double* __restrict__ Jp, double* __restrict__ Jl,
                                                double* __restrict__ Je, double* __restrict__ costPartial, size_t stride,
                                                const LmDefer df = LmDefer(), const double* __restrict__ lmPrior = nullptr) {
  double* sPose = smem;                      // nPose*7
  double* sExt = sPose + nPose * 7;          // nExt*7
  CameraModel* sCam = reinterpret_cast<CameraModel*>(sExt + nExt * 7);  // nCam
  for (int i = threadIdx.x; i < nPose * 7; i += blockDim.x) sPose[i] = pose[i];
  for (int i = threadIdx.x; i < nExt * 7; i += blockDim.x) sExt[i] = ext[i];
  {
    const double* src = reinterpret_cast<const double*>(cams);
    double* dst = reinterpret_cast<double*>(sCam);
    for (int i = threadIdx.x; i < nCam * (int)(sizeof(CameraModel) / 8); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int i = block * blockDim.x + threadIdx.x;
  double cost = 0, stepSq = 0, xSq = 0;
  if (i < N) {
    const uint32_t idx = obsIdx[i];
    const int ps = idx & 0xfff, es = (idx >> 12) & 0xfff, cs = (idx >> 24) & 0xf;
    const double2 uv = reinterpret_cast<const double2*>(obsUv)[i];
    // a NEGATIVE weight marks an observation of a landmark that is held constant (Map::setParameterBlockConstant on a
    // HomogeneousPointParameterBlock, TestMap.cpp:93): the residual and the pose / extrinsics Jacobians are the usual ones, the
    // landmark Jacobian is written as zero -- V_l, b_l and the landmark's columns of the Schur complement then vanish exactly and
    // its step is zero, without a second code path in any kernel downstream
    const double wRaw = obsW[i];
    const bool lmConstant = wRaw < 0.0;
    const double w = fabs(wRaw);
    const int lmi = obsLm[i];
    const double4 hp = reinterpret_cast<const double4*>(lm)[lmi];
    double hpw[4] = {hp.x, hp.y, hp.z, hp.w};
    if (df.on) {
      // the landmark half of the fused step (k_post_solve left it to this kernel): x_cand = x + (cg v_l - cn y_l), the
      // same arithmetic as retractItem; the lane holding the landmark's first observation stores it and counts the norms
      double xo[4];
#pragma unroll
      for (int k = 0; k < 3; ++k) xo[k] = hpw[k] + (df.cg * df.vL[3 * lmi + k] - df.cn * df.yL[3 * lmi + k]);
      xo[3] = hpw[3] + 0.0;
      if (df.lmPtr[lmi] == i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { stepSq += (hpw[k] - xo[k]) * (hpw[k] - xo[k]); xSq += hpw[k] * hpw[k]; }
        xSq += hpw[3] * hpw[3];
        reinterpret_cast<double4*>(df.lmC)[lmi] = double4{xo[0], xo[1], xo[2], xo[3]};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) hpw[k] = xo[k];
    }
    double rr[2], jp[12], jl[6], je[12];
    const bool isPrior = cs == kPriorCam;
    if (isPrior) {
      // HomogeneousPointError (HomogeneousPointError.cpp:77-117) as two pseudo-observations of its landmark: e = lm - meas
      // (first three components), r = S e with S the upper-triangular square-root information; uv = (index into the
      // prior table, part): part 0 carries rows 0 and 1 of S, part 1 row 2 and a zero row.  No loss function.
      const double* pr = lmPrior + 12 * (int)uv.x;
      const bool second = uv.y != 0.0;
      const double e0 = hpw[0] - pr[0], e1 = hpw[1] - pr[1], e2 = hpw[2] - pr[2];
      const double* Sa = pr + 3 + (second ? 6 : 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) { jl[k] = Sa[k]; jl[3 + k] = second ? 0.0 : pr[6 + k]; }
      rr[0] = jl[0] * e0 + jl[1] * e1 + jl[2] * e2;
      rr[1] = jl[3] * e0 + jl[4] * e1 + jl[5] * e2;
#pragma unroll
      for (int k = 0; k < 12; ++k) { jp[k] = 0.0; je[k] = 0.0; }
    } else {
      reprojEval(sCam[cs], sPose + ps * 7, hpw, sExt + es * 7, uv.x, uv.y, w, rr, jp, jl, je);
    }
    const double s = rr[0] * rr[0] + rr[1] * rr[1];
    if (ROBUST && !isPrior) {
      // Ceres Corrector for CauchyLoss(1): rho'' < 0 always -> residual and Jacobian scale by sqrt(rho')
      double rho0, rho1, rho2;
      cauchyLoss(s, rho0, rho1, rho2);
      cost = 0.5 * rho0;
      const double sc = sqrt(rho1);
      rr[0] *= sc; rr[1] *= sc;
#pragma unroll
      for (int k = 0; k < 12; ++k) jp[k] *= sc;
#pragma unroll
      for (int k = 0; k < 6; ++k) jl[k] *= sc;
      if (WITH_EXT) {
#pragma unroll
        for (int k = 0; k < 12; ++k) je[k] *= sc;
      }
    } else {
      cost = 0.5 * s;
    }
    if (lmConstant) {
#pragma unroll
      for (int k = 0; k < 6; ++k) jl[k] = 0.0;
    }
    K1_STORE(&r[i], rr[0]);
    K1_STORE(&r[stride + i], rr[1]);
#pragma unroll
    for (int k = 0; k < 12; ++k) K1_STORE(&Jp[k * stride + i], jp[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) K1_STORE(&Jl[k * stride + i], jl[k]);
    if (WITH_EXT) {
#pragma unroll
      for (int k = 0; k < 12; ++k) K1_STORE(&Je[k * stride + i], je[k]);
    }
  }
  if (costPartial) {
    if (df.on) {  // red: >= 48 doubles
      const double v3[3] = {cost, stepSq, xSq};
      const double mine = blockSumK<3>(v3, red, -1);
      if (threadIdx.x == 0) cstore(costPartial + block, mine);
      if (threadIdx.x == 1) cstore(df.stepPartial + block, mine);
      if (threadIdx.x == 2) cstore(df.xPartial + block, mine);
    } else {
      const double bs = blockSum(cost, red);
      if (threadIdx.x == 0) cstore(costPartial + block, bs);
    }
  }
}

template <bool ROBUST, bool WITH_EXT>
__global__ __launch_bounds__(128) void k_eval_reproj(int N, int nPose, int nExt, int nCam, const double* __restrict__ pose,
                                                     const double* __restrict__ ext, const double* __restrict__ lm,
                                                     const CameraModel* __restrict__ cams,
                                                     const double* __restrict__ obsUv, const double* __restrict__ obsW,
                                                     const uint32_t* __restrict__ obsIdx, const int* __restrict__ obsLm,
                                                     double* __restrict__ r, double* __restrict__ Jp,
                                                     double* __restrict__ Jl, double* __restrict__ Je,
                                                     double* __restrict__ costPartial, size_t stride,
                                                     const double* __restrict__ lmPrior) {
  extern __shared__ double smem[];
  __shared__ double red[4];
  evalReprojBlock<ROBUST, WITH_EXT>(blockIdx.x, smem, red, N, nPose, nExt, nCam, pose, ext, lm, cams, obsUv, obsW, obsIdx,
                                    obsLm, r, Jp, Jl, Je, costPartial, stride, LmDefer(), lmPrior);
}

static int evalGrid(int N) { return (N + 127) / 128; }

void launchEvalReproj(const DeviceProblem& p, bool cand, bool robust, hipStream_t s) {
  if (p.N == 0) return;
  const size_t smem = (size_t)(p.nPose * 7 + p.nExt * 7) * 8 + (size_t)p.nCam * sizeof(CameraModel);
  const int grid = evalGrid(p.N);
  const double* pose = cand ? p.poseC : p.pose;
  const double* ext = cand ? p.extC : p.ext;
  const double* lm = cand ? p.lmC : p.lm;
  double* r = cand ? p.rCand : p.rCur;
  double* Jp = cand ? p.JpCand : p.JpCur;
  double* Jl = cand ? p.JlCand : p.JlCur;
  double* Je = cand ? p.JeCand : p.JeCur;
  double* cp = p.partial + (size_t)PS_COST_REPROJ * kMaxPartials;
#define LAUNCH(R, E)                                                                                              \
  hipLaunchKernelGGL((k_eval_reproj<R, E>), dim3(grid), dim3(128), smem, s, p.N, p.nPose, p.nExt, p.nCam, pose, ext, \
                     lm, p.cams, p.obsUv, p.obsW, p.obsIdx, p.obsLm, r, Jp, Jl, Je, cp, (size_t)p.N, p.lmPrior)
  if (robust) { if (p.anyExtVariable) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (p.anyExtVariable) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
}

// Jacobian-evaluation roofline kernel on `copies` independent replicas of the window (HBM-resident
// working set): identical arithmetic, outputs offset per replica.
void launchEvalReprojBatched(const DeviceProblem& p, int copies, double* rOut, double* JpOut, double* JlOut,
                             double* JeOut, hipStream_t s) {
  // a replica = the same N observations; the SoA stride is the full batch so every replica owns
  // distinct output lines.  Inputs are replicated by the caller (obs arrays of size copies*N).
  const size_t smem = (size_t)(p.nPose * 7 + p.nExt * 7) * 8 + (size_t)p.nCam * sizeof(CameraModel);
  const int NB = p.N * copies;
  const int grid = evalGrid(NB);
  if (p.anyExtVariable)
    hipLaunchKernelGGL((k_eval_reproj<true, true>), dim3(grid), dim3(128), smem, s, NB, p.nPose, p.nExt, p.nCam, p.pose,
                       p.ext, p.lm, p.cams, p.obsUv, p.obsW, p.obsIdx, p.obsLm, rOut, JpOut, JlOut, JeOut,
                       (double*)nullptr, (size_t)NB, p.lmPrior);
  else
    hipLaunchKernelGGL((k_eval_reproj<true, false>), dim3(grid), dim3(128), smem, s, NB, p.nPose, p.nExt, p.nCam,
                       p.pose, p.ext, p.lm, p.cams, p.obsUv, p.obsW, p.obsIdx, p.obsLm, rOut, JpOut, JlOut, JeOut,
                       (double*)nullptr, (size_t)NB, p.lmPrior);
}

// ================================================================ K2: small factors (one workgroup each)
__device__ __forceinline__ double dtSecDev(const uint32_t* a, const uint32_t* b) {
  long long s = (long long)a[0] - (long long)b[0];
  long long ns = (long long)a[1] - (long long)b[1];
  while (ns < 0) { ns += 1000000000LL; s -= 1; }
  while (ns >= 1000000000LL) { ns -= 1000000000LL; s += 1; }
  return (double)s + 1e-9 * (double)ns;
}
__device__ __forceinline__ bool timeLess(const uint32_t* a, const uint32_t* b) {
  return a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]);
}
__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void crossMxDev(double x, double y, double z, double* C) {
  C[0] = 0; C[1] = -z; C[2] = y; C[3] = z; C[4] = 0; C[5] = -x; C[6] = -y; C[7] = x; C[8] = 0;
}
// rightJacobian (okvis::kinematics::rightJacobian as ImuError uses it): I + a [phi]x + b [phi]x^2 with
// a = -(1 - cos Phi) / Phi^2, b = (Phi - sin Phi) / Phi^3 and the constants -1/2, 1/6 below Phi = 1e-4 like the reference.
// Between 1e-4 and 1/2 rad -- every IMU step: Phi = |omega| dt -- the two coefficients come from their Taylor series:
// the closed forms cancel catastrophically there (relative error ~1e-16 / Phi^2) and cost two library calls on the
// one-thread-per-step stage of the re-integration.
__device__ __forceinline__ void rightJacobianDev(double x, double y, double z, double* J) {
  const double Phi2 = x * x + y * y + z * z;
  double X[9], X2[9];
  crossMxDev(x, y, z, X);
  mm3(X, X, X2);
  double a, b;
  if (Phi2 < 1.0e-8) { a = -0.5; b = 1.0 / 6.0; }
  else if (Phi2 < 0.25) {
    // a = -(1/2 - P/24 + P^2/720 - ...), b = 1/6 - P/120 + P^2/5040 - ...   (P = Phi^2)
    double ta = 1.0 / 20922789888000.0;   // 1/16!
    ta = __builtin_fma(ta, -Phi2, 1.0 / 87178291200.0);
    ta = __builtin_fma(ta, -Phi2, 1.0 / 479001600.0);
    ta = __builtin_fma(ta, -Phi2, 1.0 / 3628800.0);
    ta = __builtin_fma(ta, -Phi2, 1.0 / 40320.0);
    ta = __builtin_fma(ta, -Phi2, 1.0 / 720.0);
    ta = __builtin_fma(ta, -Phi2, 1.0 / 24.0);
    ta = __builtin_fma(ta, -Phi2, 0.5);
    a = -ta;
    double tb = 1.0 / 355687428096000.0;  // 1/17!
    tb = __builtin_fma(tb, -Phi2, 1.0 / 1307674368000.0);
    tb = __builtin_fma(tb, -Phi2, 1.0 / 6227020800.0);
    tb = __builtin_fma(tb, -Phi2, 1.0 / 39916800.0);
    tb = __builtin_fma(tb, -Phi2, 1.0 / 362880.0);
    tb = __builtin_fma(tb, -Phi2, 1.0 / 5040.0);
    tb = __builtin_fma(tb, -Phi2, 1.0 / 120.0);
    tb = __builtin_fma(tb, -Phi2, 1.0 / 6.0);
    b = tb;
  } else {
    const double Phi = sqrt(Phi2), Phi3 = Phi2 * Phi;
    a = -(1.0 - cos(Phi)) / Phi2;
    b = (Phi - sin(Phi)) / Phi3;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = a * X[i] + b * X2[i];
  J[0] += 1; J[4] += 1; J[8] += 1;
}
// Quaternion exponential on the device (retraction, IMU bias correction): sin(h)/h and cos(h) from their Taylor series for the
// half-angles a trust-region step produces (h^2 < 1/4: truncation below 1e-21), the library functions beyond.  The
// library versions cost several hundred instructions each, and this sits on the serial tail of every iteration.
__device__ __forceinline__ Quat deltaQDev(double ax, double ay, double az) {
  const double h2 = 0.25 * (ax * ax + ay * ay + az * az);
  double sc, c;
  if (h2 < 0.25) {
    sc = -1.0 / 121645100408832000.0;  // 1/19!
    sc = __builtin_fma(sc, h2, 1.0 / 355687428096000.0);
    sc = __builtin_fma(sc, h2, -1.0 / 1307674368000.0);
    sc = __builtin_fma(sc, h2, 1.0 / 6227020800.0);
    sc = __builtin_fma(sc, h2, -1.0 / 39916800.0);
    sc = __builtin_fma(sc, h2, 1.0 / 362880.0);
    sc = __builtin_fma(sc, h2, -1.0 / 5040.0);
    sc = __builtin_fma(sc, h2, 1.0 / 120.0);
    sc = __builtin_fma(sc, h2, -1.0 / 6.0);
    sc = __builtin_fma(sc, h2, 1.0);
    c = -1.0 / 6402373705728000.0;  // 1/18!
    c = __builtin_fma(c, h2, 1.0 / 20922789888000.0);
    c = __builtin_fma(c, h2, -1.0 / 87178291200.0);
    c = __builtin_fma(c, h2, 1.0 / 479001600.0);
    c = __builtin_fma(c, h2, -1.0 / 3628800.0);
    c = __builtin_fma(c, h2, 1.0 / 40320.0);
    c = __builtin_fma(c, h2, -1.0 / 720.0);
    c = __builtin_fma(c, h2, 1.0 / 24.0);
    c = __builtin_fma(c, h2, -0.5);
    c = __builtin_fma(c, h2, 1.0);
  } else {
    const double h = sqrt(h2);
    sc = sin(h) / h;
    c = cos(h);
  }
  const double s = 0.5 * sc;
  return Quat{s * ax, s * ay, s * az, c};
}
// quatPlusMat3 / quatOplusMat3: dmath.hpp
__device__ __forceinline__ void quatPlusMat4(const Quat& q, double* Q) {
  Q[0] = q.w; Q[1] = -q.z; Q[2] = q.y; Q[3] = q.x;
  Q[4] = q.z; Q[5] = q.w; Q[6] = -q.x; Q[7] = q.y;
  Q[8] = -q.y; Q[9] = q.x; Q[10] = q.w; Q[11] = q.z;
  Q[12] = -q.x; Q[13] = -q.y; Q[14] = -q.z; Q[15] = q.w;
}
__device__ __forceinline__ void quatOplusMat4(const Quat& q, double* Q) {
  Q[0] = q.w; Q[1] = q.z; Q[2] = -q.y; Q[3] = q.x;
  Q[4] = -q.z; Q[5] = q.w; Q[6] = q.x; Q[7] = q.y;
  Q[8] = q.y; Q[9] = -q.x; Q[10] = q.w; Q[11] = q.z;
  Q[12] = -q.x; Q[13] = -q.y; Q[14] = -q.z; Q[15] = q.w;
}
__device__ __forceinline__ void mm4(const double* A, const double* B, double* C) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
      C[i * 4 + j] = s;
    }
}

// LDS layout of the factor kernel
struct FactorShared {
  double W[225];      // sqrtInfo (m x m)
  double F[15 * 30];  // un-weighted Jacobian blocks (m x ncols)
  double e[15];       // un-weighted error
  double P[225], T[225], Fd[225];  // IMU covariance propagation
  double st[72];      // staged IMU pre-integration state (Delta_t .. dp_db_g, 56 doubles)
  double xs[32];      // staged parameter blocks x0(7) s0(9) x1(7) s1(9)
  double rw[15];      // weighted residual
  // re-preintegration (imuIntegrate): per-step inputs (P0), the two serial chains (P1), the blocks of F_delta
  // per step (P2/P3, double buffered for the covariance wave) and the published integrals
  double pre[64 * 33];
  double fb[128 * 93];   // 128 steps x kFbLd
  double seq[65 * 13];
  double tot[64];
  double tile[16 * kPanelLd];  // 15x15 covariance / information padded to the 16x16 wave-level factorisation
  double dinv[16];
  int flag, used;
};

// normalised pose -> Transformation(r, q) semantics (q normalised, C from normalised q)
struct TF { double r[3]; Quat q; Mat3 C; };
__device__ __forceinline__ TF makeTF(const double* x) {
  TF t;
  t.r[0] = x[0]; t.r[1] = x[1]; t.r[2] = x[2];
  t.q = qnormalized(Quat{x[3], x[4], x[5], x[6]});
  t.C = quatToR(t.q);
  return t;
}

// IMU integration loop shared by redoPreintegration (ImuError.cpp:76-263, REDO=true) and
// propagation (:266-476, REDO=false), executed by the whole workgroup: every thread carries the small
// 3x3 state redundantly in registers; the 15x15 covariance lives in LDS (sh.P), one entry per thread.
// The two flavours differ exactly where the reference does: dalpha_db_g accumulates
// C_1*rightJacobian*dt (:189) vs dt*C_1 (:384); sigma2_v = dt*sigma_a_c^2 (:215) vs
// dt*sigma_a_c*par.sigma_a_c (:412).
struct ImuState {
  Quat Dq;
  double Ci[9], Cdi[9], ai[3], adi[3], dal[9], dv[9], dp[9], Delta_t;
  int used;
};
constexpr int kImuSuper = 64;  // integration steps per round of the per-step phases P0..P3
constexpr int kImuBlock = 128;  // integration steps per block of the covariance stage (sh.fb)
constexpr int kPreLd = 33;     // odd row strides keep the one-thread-per-step phases off the same LDS banks
constexpr int kFbLd = 93;   // 87 entries of the step + the five process-noise values of P4 (87..91); sh.fb is sized for it
constexpr int kSeqLd = 13;

// -crossMx(v)[i][j] = sign * v[comp]  (sign 0 on the diagonal)
__device__ __forceinline__ void negCrossDesc(int i, int j, int& comp, double& sign) {
  comp = 0; sign = 0.0;
  if (i == j) return;
  comp = 3 - i - j;
  sign = ((j - i + 3) % 3 == 1) ? 1.0 : -1.0;
}

#ifdef SVIN_IMU_TIMING
__device__ double g_imuDbg[16];
#define IMU_TICK(var) long long var = __builtin_readcyclecounter()
#define IMU_ACC(slot, a, b, cond) if (cond) atomicAdd(&g_imuDbg[slot], (double)((b) - (a)))
#else
#define IMU_TICK(var)
#define IMU_ACC(slot, a, b, cond)
#endif

// Re-preintegration (ImuError.cpp:118-243 redo / :309-452 propagation) restructured so that only the two
// genuinely serial chains stay serial, every round handling kImuSuper integration steps:
//   P0  one thread per step : time bookkeeping, interpolation at the interval ends, saturation test,
//                             omega/acc minus bias, dq, R(dq^-1), rightJacobian*dt                -> sh.pre
//   P1  wave 0 / wave 1     : Delta_q_{k+1} = Delta_q_k (x) dq_k  /  cross_{k+1} = R(dq^-1) cross_k + rJ dt
//                             (the only recurrences that are not plain sums)                       -> sh.seq
//   P2  one thread per step : C, C_1, the per-step increments of every integral and the blocks of F_delta
//                             that do not involve running sums                                      -> sh.fb
//   P3  wave 0, one lane per scalar component: the running sums C_integral, acc_integral, dv_db_g (in step
//                             order, like the reference) and the F_delta blocks built from them    -> sh.fb
//   P4  all four waves      : P <- F P F^T + Q as v_mfma_f64_16x16x4_f64 products with P held in the MFMA accumulator
//                             layout; the steps of a block are split into four segments (one per SIMD), each wave
//                             carries its segment's covariance and transition product, wave 0 chains the segments
template <bool REDO>
__device__ void imuIntegrate(const DevImu& im, const uint32_t* __restrict__ T, const double* __restrict__ M,
                             const double* sb, FactorShared& sh, ImuState& st) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int n = im.sampleCount;
  const uint32_t t0[2] = {im.t0[0], im.t0[1]}, end[2] = {im.t1[0], im.t1[1]};
  const double sgw2 = im.par.sigma_gw_c * im.par.sigma_gw_c, saw2 = im.par.sigma_aw_c * im.par.sigma_aw_c;
  IMU_TICK(qStart);

  Quat Dq = {0, 0, 0, 1};  // wave 0
  double cross[9] = {0};   // wave 1
  // wave 0, P3: lane c owns one scalar component: run1 (first sum) and run2 (second sum)
  //   c 0..8 C_integral/C_doubleintegral, 9..11 acc_integral/acc_doubleintegral, 12..20 dv_db_g/dp_db_g,
  //   21..29 dalpha_db_g, 30 Delta_t, 31 number of executed steps
  double run1 = 0, run2 = 0;
  int iInc = 86, iQ = 86, iOut = 85, tot1 = 60, tot2 = 61;
  double incSign = 1.0, outSign = 1.0;
  if (lane < 9) { iInc = 43 + lane; incSign = -1.0; iQ = 55 + lane; iOut = 13 + lane; outSign = -1.0; tot1 = 4 + lane; tot2 = 13 + lane; }
  else if (lane < 12) { const int k = lane - 9; iInc = 31 + k; iQ = 64 + k; iOut = k; tot1 = 22 + k; tot2 = 25 + k; }
  else if (lane < 21) { const int k = lane - 12; iInc = 34 + k; iQ = 67 + k; iOut = 4 + k; tot1 = 37 + k; tot2 = 46 + k; }
  else if (lane < 30) { const int k = lane - 21; iInc = 76 + k; tot1 = 28 + k; }
  else if (lane == 30) { iInc = 3; tot1 = 55; }
  else if (lane == 31) { iInc = 54; tot1 = 56; }
  // wave 0, P4: X = P (parity 0) or P^T (parity 1) in the accumulator layout
  // X[(lane>>4)+4r][lane&15]; F_delta entries F[lane&15][(lane>>4)+4q] = fC0 + fS * fb[fIdx]
  d4_t X = {0, 0, 0, 0};
  int parity = 0;
  int fIdx[4], qDiag = -1, dKind = 0;
  double fS[4], fC0[4];
  for (int q = 0; q < 4; ++q) {
    const int i = lane & 15, c = (lane >> 4) + 4 * q;
    fIdx[q] = 86; fS[q] = 0.0; fC0[q] = (i == c && i < 15) ? 1.0 : 0.0;
    if (i == c && i < 15) { qDiag = q; dKind = i / 3 + 1; }
    if (i < 9 && c < 15) {
      const int br = i / 3, ii = i % 3, bc = c / 3, jj = c % 3;
      int comp; double sg;
      if (br == 0) {
        if (bc == 1) { negCrossDesc(ii, jj, comp, sg); fIdx[q] = comp; fS[q] = sg; }
        else if (bc == 2) { if (ii == jj) { fIdx[q] = 3; fS[q] = 1.0; } }
        else if (bc == 3) { fIdx[q] = 4 + ii * 3 + jj; fS[q] = 1.0; }
        else if (bc == 4) { fIdx[q] = 13 + ii * 3 + jj; fS[q] = 1.0; }
      } else if (br == 1) {
        if (bc == 3) { fIdx[q] = 22 + ii * 3 + jj; fS[q] = 1.0; }
      } else {
        if (bc == 1) { negCrossDesc(ii, jj, comp, sg); fIdx[q] = 31 + comp; fS[q] = sg; }
        else if (bc == 3) { fIdx[q] = 34 + ii * 3 + jj; fS[q] = 1.0; }
        else if (bc == 4) { fIdx[q] = 43 + ii * 3 + jj; fS[q] = 1.0; }
      }
    }
  }
  // (qDiag, dKind): accumulator register q holds row (lane>>4)+4q, column lane&15 -- the F descriptor's pairing
  // with rows and columns swapped; the diagonal test is symmetric, so one loop serves both.
  // P4 segments: N = F_delta - I transposed in the A-operand layout, N[(lane>>4)+4q][lane&15] = tS * fb[tIdx] (rows 0..8
  // only: q < 3), and where the process noise of row (lane>>4)+4q sits in the step's fb row
  int tIdx[4], rowKind[4];
  double tS[4];
  for (int q = 0; q < 4; ++q) {
    const int i = (lane >> 4) + 4 * q, c = lane & 15;   // row, column of F
    tIdx[q] = 86; tS[q] = 0.0;
    rowKind[q] = (i < 15) ? 87 + i / 3 : 86;   // index of this row's process noise in the step's fb row (86: zero)
    if (i < 9 && c < 15) {
      const int br = i / 3, ii = i % 3, bc = c / 3, jj = c % 3;
      int comp; double sg;
      if (br == 0) {
        if (bc == 1) { negCrossDesc(ii, jj, comp, sg); tIdx[q] = comp; tS[q] = sg; }
        else if (bc == 2) { if (ii == jj) { tIdx[q] = 3; tS[q] = 1.0; } }
        else if (bc == 3) { tIdx[q] = 4 + ii * 3 + jj; tS[q] = 1.0; }
        else if (bc == 4) { tIdx[q] = 13 + ii * 3 + jj; tS[q] = 1.0; }
      } else if (br == 1) {
        if (bc == 3) { tIdx[q] = 22 + ii * 3 + jj; tS[q] = 1.0; }
      } else {
        if (bc == 1) { negCrossDesc(ii, jj, comp, sg); tIdx[q] = 31 + comp; tS[q] = sg; }
        else if (bc == 3) { tIdx[q] = 34 + ii * 3 + jj; tS[q] = 1.0; }
        else if (bc == 4) { tIdx[q] = 43 + ii * 3 + jj; tS[q] = 1.0; }
      }
    }
  }

  for (int blk0 = 0; blk0 < n; blk0 += kImuBlock) {
   const int nb = min(kImuBlock, n - blk0);
   for (int s0 = blk0; s0 < blk0 + nb; s0 += kImuSuper) {
    const int ns = min(kImuSuper, blk0 + nb - s0);
    double* fbw = sh.fb + (size_t)(s0 - blk0) * kFbLd;
    // ---------------- P0
    IMU_TICK(qp0);
    if (t < ns) {
      const int it = s0 + t;
      double w0[3] = {M[6 * it], M[6 * it + 1], M[6 * it + 2]};
      double a0[3] = {M[6 * it + 3], M[6 * it + 4], M[6 * it + 5]};
      const bool last = (it + 1 == n);
      const int nx = last ? it : it + 1;
      double w1[3] = {M[6 * nx], M[6 * nx + 1], M[6 * nx + 2]};
      double a1[3] = {M[6 * nx + 3], M[6 * nx + 4], M[6 * nx + 5]};
      const uint32_t tIt[2] = {T[2 * it], T[2 * it + 1]};
      // `time` of the reference loop: t0 until the first executed step, the sample time afterwards
      const bool prevStarted = (it > 0) && timeLess(t0, tIt);
      const uint32_t time[2] = {prevStarted ? tIt[0] : t0[0], prevStarted ? tIt[1] : t0[1]};
      uint32_t nexttime[2] = {last ? end[0] : T[2 * nx], last ? end[1] : T[2 * nx + 1]};
      double dt = dtSecDev(nexttime, time);
      if (timeLess(end, nexttime)) {
        const double interval = dtSecDev(nexttime, tIt);
        nexttime[0] = end[0]; nexttime[1] = end[1];
        dt = dtSecDev(nexttime, time);
        const double rr = dt / interval;
        for (int k = 0; k < 3; ++k) { w1[k] = (1.0 - rr) * w0[k] + rr * w1[k]; a1[k] = (1.0 - rr) * a0[k] + rr * a1[k]; }
      }
      const bool exec = (dt > 0.0) && timeLess(time, end);
      if (exec && !prevStarted) {
        const double rr = dt / dtSecDev(nexttime, tIt);
        for (int k = 0; k < 3; ++k) { w0[k] = rr * w0[k] + (1.0 - rr) * w1[k]; a0[k] = rr * a0[k] + (1.0 - rr) * a1[k]; }
      }
      double sigma_g_c = im.par.sigma_g_c, sigma_a_c = im.par.sigma_a_c;
      bool gs = false, as = false;
      for (int k = 0; k < 3; ++k) {
        gs = gs || fabs(w0[k]) > im.par.g_max || fabs(w1[k]) > im.par.g_max;
        as = as || fabs(a0[k]) > im.par.a_max || fabs(a1[k]) > im.par.a_max;
      }
      if (gs) sigma_g_c *= 100;
      if (as) sigma_a_c *= 100;
      const double wt[3] = {0.5 * (w0[0] + w1[0]) - sb[3], 0.5 * (w0[1] + w1[1]) - sb[4], 0.5 * (w0[2] + w1[2]) - sb[5]};
      const double at[3] = {0.5 * (a0[0] + a1[0]) - sb[6], 0.5 * (a0[1] + a1[1]) - sb[7], 0.5 * (a0[2] + a1[2]) - sb[8]};
      // dq = [sinc(theta/2) omega dt / 2, cos(theta/2)]  (ImuError.cpp:151-157) = the quaternion exponential of omega dt
      const Quat dq = deltaQDev(wt[0] * dt, wt[1] * dt, wt[2] * dt);
      const double dqn = 1.0 / (dq.x * dq.x + dq.y * dq.y + dq.z * dq.z + dq.w * dq.w);
      const Mat3 Rdqi = quatToR(Quat{-dq.x * dqn, -dq.y * dqn, -dq.z * dqn, dq.w * dqn});
      double RJ[9];
      rightJacobianDev(wt[0] * dt, wt[1] * dt, wt[2] * dt, RJ);
      double* pr = sh.pre + t * kPreLd;
      pr[0] = dt; pr[1] = at[0]; pr[2] = at[1]; pr[3] = at[2];
      pr[4] = dq.x; pr[5] = dq.y; pr[6] = dq.z; pr[7] = dq.w;
      for (int k = 0; k < 9; ++k) { pr[8 + k] = Rdqi.m[k]; pr[17 + k] = RJ[k] * dt; }
      pr[26] = dt * sigma_g_c * sigma_g_c;
      pr[27] = REDO ? dt * sigma_a_c * sigma_a_c : dt * sigma_a_c * im.par.sigma_a_c;
      pr[28] = exec ? 1.0 : 0.0;
    }
    IMU_TICK(qp1);
    IMU_ACC(0, qp0, qp1, t == 0);
    __syncthreads();
    IMU_TICK(qp2);
    // ---------------- P1: the two serial chains, side by side
    if (wave == 0 && ns > 0) {
      // Delta_q chain as a wave-wide inclusive scan (quaternion products are associative): lane i carries dq_i
      // (identity for a step that is not executed), six combine levels instead of ns serial products
      const double* pr = sh.pre + min(lane, ns - 1) * kPreLd;
      const bool act = lane < ns && pr[28] != 0.0;
      Quat q = act ? Quat{pr[4], pr[5], pr[6], pr[7]} : Quat{0, 0, 0, 1};
      // inclusive scan without LDS (__shfl_up is ds_bpermute: 8 LDS round trips per level here): four levels inside the
      // 16-lane rows (row_shr), then the row totals travel down the wave (row_bcast:15, row_bcast:31)
#define SVIN_QSCAN(CTRL, MASK, COND)                                                                                 \
      {                                                                                                             \
        const Quat lo = {dppScanMov<CTRL, MASK>(q.x), dppScanMov<CTRL, MASK>(q.y), dppScanMov<CTRL, MASK>(q.z),     \
                         dppScanMov<CTRL, MASK>(q.w)};                                                              \
        const Quat c = qmul(lo, q); /* earlier steps on the left */                                                 \
        if (COND) q = c;                                                                                            \
      }
      SVIN_QSCAN(0x111, 0xf, (lane & 15) >= 1)
      SVIN_QSCAN(0x112, 0xf, (lane & 15) >= 2)
      SVIN_QSCAN(0x114, 0xf, (lane & 15) >= 4)
      SVIN_QSCAN(0x118, 0xf, (lane & 15) >= 8)
      SVIN_QSCAN(0x142, 0xa, (lane >> 4) & 1)
      SVIN_QSCAN(0x143, 0xc, lane >= 32)
#undef SVIN_QSCAN
      Quat e = {dppScanMov<0x138, 0xf>(q.x), dppScanMov<0x138, 0xf>(q.y), dppScanMov<0x138, 0xf>(q.z), dppScanMov<0x138, 0xf>(q.w)};
      if (lane == 0) e = Quat{0, 0, 0, 1};
      const Quat before = qmul(Dq, e);     // Delta_q before step `lane`
      if (lane < ns) { double* sq = sh.seq + lane * kSeqLd; sq[0] = before.x; sq[1] = before.y; sq[2] = before.z; sq[3] = before.w; }
      const Quat tot = {readlaneD(q.x, ns - 1), readlaneD(q.y, ns - 1), readlaneD(q.z, ns - 1), readlaneD(q.w, ns - 1)};
      Dq = qmul(Dq, tot);
      if (lane == 0) { double* sq = sh.seq + ns * kSeqLd; sq[0] = Dq.x; sq[1] = Dq.y; sq[2] = Dq.z; sq[3] = Dq.w; }
    }
    if (wave == 1 && ns > 0) {
      // cross chain M <- R_i M + B_i (3x3, B_i = rightJacobian dt): affine maps compose associatively,
      // (R_l, B_l) after (R_e, B_e) = (R_l R_e, R_l B_e + B_l); same scan, lane i carries step i's map
      const double* pr = sh.pre + min(lane, ns - 1) * kPreLd;
      const bool act = lane < ns && pr[28] != 0.0;
      double R[9], B[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { R[k] = act ? pr[8 + k] : ((k % 4 == 0) ? 1.0 : 0.0); B[k] = act ? pr[17 + k] : 0.0; }
#define SVIN_ASCAN(CTRL, MASK, COND)                                                                                 \
      {                                                                                                             \
        double Re[9], Be[9], Rn[9], Bn[9];                                                                          \
        _Pragma("unroll") for (int k = 0; k < 9; ++k) { Re[k] = dppScanMov<CTRL, MASK>(R[k]); Be[k] = dppScanMov<CTRL, MASK>(B[k]); } \
        _Pragma("unroll") for (int r = 0; r < 3; ++r)                                                               \
          _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                           \
            Rn[3 * r + c] = R[3 * r] * Re[c] + R[3 * r + 1] * Re[3 + c] + R[3 * r + 2] * Re[6 + c];                 \
            Bn[3 * r + c] = (R[3 * r] * Be[c] + R[3 * r + 1] * Be[3 + c] + R[3 * r + 2] * Be[6 + c]) + B[3 * r + c]; \
          }                                                                                                         \
        if (COND) {                                                                                                 \
          _Pragma("unroll") for (int k = 0; k < 9; ++k) { R[k] = Rn[k]; B[k] = Bn[k]; }                             \
        }                                                                                                           \
      }
      SVIN_ASCAN(0x111, 0xf, (lane & 15) >= 1)
      SVIN_ASCAN(0x112, 0xf, (lane & 15) >= 2)
      SVIN_ASCAN(0x114, 0xf, (lane & 15) >= 4)
      SVIN_ASCAN(0x118, 0xf, (lane & 15) >= 8)
      SVIN_ASCAN(0x142, 0xa, (lane >> 4) & 1)
      SVIN_ASCAN(0x143, 0xc, lane >= 32)
#undef SVIN_ASCAN
      // exclusive prefix applied to the state at the start of the round; the inclusive one of the last step ends it
      double Re[9], Be[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { Re[k] = dppScanMov<0x138, 0xf>(R[k]); Be[k] = dppScanMov<0x138, 0xf>(B[k]); }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { Re[k] = (k % 4 == 0) ? 1.0 : 0.0; Be[k] = 0.0; }
      }
      double after[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double bef = (Re[3 * r] * cross[c] + Re[3 * r + 1] * cross[3 + c] + Re[3 * r + 2] * cross[6 + c]) + Be[3 * r + c];
          after[3 * r + c] = (R[3 * r] * cross[c] + R[3 * r + 1] * cross[3 + c] + R[3 * r + 2] * cross[6 + c]) + B[3 * r + c];
          if (lane < ns) sh.seq[lane * kSeqLd + 4 + 3 * r + c] = bef;
        }
#pragma unroll
      for (int k = 0; k < 9; ++k) cross[k] = readlaneD(after[k], ns - 1);
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (lane == k) sh.seq[ns * kSeqLd + 4 + k] = cross[k];
    }
    IMU_TICK(qp3);
    IMU_ACC(1, qp2, qp3, t == 0);
    IMU_ACC(6, qp2, qp3, t == 64);
    __syncthreads();
    IMU_TICK(qp4);
    // ---------------- P2
    if (t < ns) {
      const double* pr = sh.pre + t * kPreLd;
      const double* sq = sh.seq + t * kSeqLd;
      double* fb = fbw + t * kFbLd;
      if (pr[28] == 0.0) {
        for (int k = 0; k < kFbLd; ++k) fb[k] = 0.0;
      } else {
        const double dt = pr[0];
        const double at[3] = {pr[1], pr[2], pr[3]};
        const Mat3 C = quatToR(Quat{sq[0], sq[1], sq[2], sq[3]});
        const Mat3 C1 = quatToR(Quat{sq[kSeqLd], sq[kSeqLd + 1], sq[kSeqLd + 2], sq[kSeqLd + 3]});
        double Cs[9];
        for (int k = 0; k < 9; ++k) Cs[k] = C.m[k] + C1.m[k];
        const double Csa[3] = {Cs[0] * at[0] + Cs[1] * at[1] + Cs[2] * at[2], Cs[3] * at[0] + Cs[4] * at[1] + Cs[5] * at[2],
                               Cs[6] * at[0] + Cs[7] * at[1] + Cs[8] * at[2]};
        double dalInc[9];
        if (REDO) mm3(C1.m, pr + 17, dalInc);  // C_1 * rightJacobian * dt
        else for (int k = 0; k < 9; ++k) dalInc[k] = dt * C1.m[k];
        double ax[9], t1[9], t2[9], Mm[9];
        crossMxDev(at[0], at[1], at[2], ax);
        mm3(C.m, ax, t1);
        mm3(t1, sq + 4, Mm);
        mm3(C1.m, ax, t1);
        mm3(t1, sq + kSeqLd + 4, t2);
        for (int k = 0; k < 9; ++k) Mm[k] += t2[k];
        fb[0] = fb[1] = fb[2] = 0.0;
        fb[3] = dt;
        for (int k = 0; k < 9; ++k) {
          fb[4 + k] = 0.0; fb[13 + k] = 0.0;
          fb[22 + k] = -dt * C1.m[k];
          fb[34 + k] = 0.5 * dt * Mm[k];
          fb[43 + k] = -0.5 * Cs[k] * dt;
          fb[55 + k] = 0.25 * Cs[k] * dt * dt;
          fb[67 + k] = 0.25 * dt * dt * Mm[k];
          fb[76 + k] = dalInc[k];
        }
        for (int k = 0; k < 3; ++k) { fb[31 + k] = 0.5 * Csa[k] * dt; fb[64 + k] = 0.25 * Csa[k] * dt * dt; }
        fb[52] = pr[26]; fb[53] = pr[27]; fb[54] = 1.0;
        fb[85] = 0.0; fb[86] = 0.0;
        // Q_delta by row kind (P4): position, angle, velocity, gyro bias, accelerometer bias
        fb[87] = 0.5 * dt * dt * pr[27]; fb[88] = pr[26]; fb[89] = pr[27]; fb[90] = dt * sgw2; fb[91] = dt * saw2;
      }
    }
    IMU_TICK(qp5);
    IMU_ACC(7, qp4, qp5, t == 0);
    __syncthreads();
    IMU_TICK(qp6);
    // ---------------- P3: running sums in step order; B012 / pterm / F09 need the sums *before* the step
    if (wave == 0 && ns > 0) {
      // eight steps per group; the three inputs of the NEXT group are requested before the current group's results are
      // stored (the stores and the loads hit the same LDS array: in source order the compiler would have to finish one
      // group, latency and all, before starting the next)
      constexpr int kG = 8;
      // Eight steps per group, row pointers advance by a group, inside a group every access is base + compile-time offset.
      // The inputs of the NEXT group are requested before the current group's results are stored (same LDS array: in source
      // order the compiler would finish one group, latency and all, before starting the next); two register sets alternate
      // (no copies), full groups run without per-step conditions -- a branch per step made every step its own basic block
      // with no overlap between the steps' dependent chains (160 cycles per step), selects cost six instructions per step --
      // and only the last, partial group of a round is masked.
      const double* pDt = fbw + 3;
      const double* pInc = fbw + iInc;
      const double* pQ = fbw + iQ;
      double* pOut = fbw + iOut;
      auto loadGroup = [&](double (&dt8)[kG], double (&in8)[kG], double (&q8)[kG]) {
#pragma unroll
        for (int u = 0; u < kG; ++u) { dt8[u] = pDt[u * kFbLd]; in8[u] = pInc[u * kFbLd]; q8[u] = pQ[u * kFbLd]; }
        pDt += kG * kFbLd; pInc += kG * kFbLd; pQ += kG * kFbLd;
      };
      auto runGroup = [&](const double (&dt8)[kG], const double (&in8)[kG], const double (&q8)[kG]) {
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          const double a = run1 * dt8[u];
          pOut[u * kFbLd] = outSign * a + q8[u];
          run2 += a + q8[u];
          run1 += incSign * in8[u];
        }
        pOut += kG * kFbLd;
      };
      double dtA[kG], inA[kG], qA[kG], dtB[kG], inB[kG], qB[kG];
      const int nFull = ns / kG;   // full groups
      int gI = 0;
      if (nFull > 0) loadGroup(dtA, inA, qA);
      while (gI + 2 <= nFull) {
        loadGroup(dtB, inB, qB);
        runGroup(dtA, inA, qA);
        if (gI + 2 < nFull) loadGroup(dtA, inA, qA);
        runGroup(dtB, inB, qB);
        gI += 2;
      }
      if (gI < nFull) { runGroup(dtA, inA, qA); ++gI; }
      const int rem = ns - nFull * kG;   // steps of the partial group: masked (rows past the end are read but contribute zeros
      if (rem > 0) {                     // and are stored into rows nobody reads; a round's rows end inside sh.fb)
        loadGroup(dtA, inA, qA);
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          const bool ok = u < rem;
          dtA[u] = ok ? dtA[u] : 0.0; inA[u] = ok ? inA[u] : 0.0; qA[u] = ok ? qA[u] : 0.0;
        }
        runGroup(dtA, inA, qA);
      }
    }
    IMU_TICK(qp7);
    IMU_ACC(5, qp6, qp7, t == 0);
    __syncthreads();
   }
   // ---------------- P4: covariance of the block.  The recurrence P <- F P F^T + Q is split into four segments,
   // one per wave (= per SIMD, each with its own matrix pipe).  A segment starts from zero, so its covariance is the sum
   //   P_w = sum_k L_k Q_k L_k^T,   L_k = F_hi-1 ... F_k+1  (the steps after k),
   // and its transition product is Phi_w = L_lo-1.  Walking the steps BACKWARDS with M = L_k^T in the accumulator layout,
   //   P_w += M^T Q_k M   (both MFMA operands are M's own registers: lane (c, g) register q = M[g + 4q][c]),
   //   M   <- F_k^T M     (A operand = F_k^T gathered from LDS, B operand = M's registers),
   // costs 8 MFMAs per step in two chains that overlap (the forward form F (F P)^T + the product for Phi took 12 in a
   // single dependent chain of 8), and nothing but the entries of F_k crosses LDS.  Wave 0 then chains the segments:
   // P <- Phi_w P Phi_w^T + P_w.  (Exact in exact arithmetic; the additions associate differently from the step-by-step
   // recurrence, i.e. rounding-level differences only.)
   {
    IMU_TICK(qc0);
    const int seg = (nb + 3) / 4;
    const int lo = min(nb, wave * seg), hi = min(nb, lo + seg);
    d4_t Xs = {0, 0, 0, 0}, Ph;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int row = (lane >> 4) + 4 * q, col = lane & 15; Ph[q] = (row == col && row < 15) ? 1.0 : 0.0; }
    const int par = 0;
    if (lo < hi) {
      // inputs of the next step (the one before) are gathered from LDS while the MFMAs of the current one run
      const double* row = sh.fb + (size_t)(hi - 1) * kFbLd;
      double g[3], qd[4];
#pragma unroll
      for (int q = 0; q < 3; ++q) g[q] = row[tIdx[q]];
#pragma unroll
      for (int q = 0; q < 4; ++q) qd[q] = row[rowKind[q]];
      for (int i = hi - 1; i >= lo; --i) {
        const double* rn = sh.fb + (size_t)max(i - 1, lo) * kFbLd;
        double gn[3], qn[4];
#pragma unroll
        for (int q = 0; q < 3; ++q) gn[q] = rn[tIdx[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) qn[q] = rn[rowKind[q]];
        // (a step that was not executed has an all-zero row: N = 0, Q = 0 -- its products add nothing, so no branch: a
        // branch per step makes every step its own basic block and nothing of one step overlaps with the next)
        {
          double aq[4], nq[3];
#pragma unroll
          for (int q = 0; q < 4; ++q) aq[q] = Ph[q] * qd[q];
#pragma unroll
          for (int q = 0; q < 3; ++q) nq[q] = tS[q] * g[q];
          // F^T M = M + N^T M; N has rows 0..8 only, i.e. three of the four k-chunks.  The chain through M goes first,
          // the noise term fills the matrix pipe behind it
          d4_t Mn = Ph;
#pragma unroll
          for (int q = 0; q < 3; ++q) Mn = __builtin_amdgcn_mfma_f64_16x16x4f64(nq[q], Ph[q], Mn, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) Xs = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], Ph[q], Xs, 0, 0, 0);   // M^T Q M
          Ph = Mn;
          // an MFMA holds the matrix pipe for 64 cycles = 16 issue slots of this wave: the gathers and moves of the next step
          // are slotted between the seven MFMAs instead of queueing behind the last one
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) g[q] = gn[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) qd[q] = qn[q];
      }
    }
    IMU_TICK(qcs);
    IMU_ACC(14, qc0, qcs, t == 0);
    // publish Phi_w and P_w (true orientation) as 16x16 tiles; sh.pre is free between the rounds
    double* phiT = sh.pre + wave * 512;
    double* pT = phiT + 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (lane >> 4) + 4 * q, col = lane & 15;
      phiT[col * 16 + row] = Ph[q];   // Ph = Phi^T
      pT[par ? col * 16 + row : row * 16 + col] = Xs[q];
    }
    __syncthreads();
    if (wave == 0) {
      for (int w = 0; w < 4; ++w) {
        const double* ph = sh.pre + w * 512;
        const double* pw = ph + 256;
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = ph[(lane & 15) * 16 + (lane >> 4) + 4 * q];
        d4_t V = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) V = __builtin_amdgcn_mfma_f64_16x16x4f64(X[q], a[q], V, 0, 0, 0);   // X^T Phi^T
        d4_t Y = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], V[q], Y, 0, 0, 0);   // Phi X^T Phi^T
        parity ^= 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (lane >> 4) + 4 * q, col = lane & 15;
          X[q] = Y[q] + pw[parity ? col * 16 + row : row * 16 + col];
        }
      }
    }
    __syncthreads();  // sh.pre is written again by the next block's P0
    IMU_TICK(qc1);
    IMU_ACC(2, qc0, qc1, t == 0);
   }
  }
  // publish: covariance and integrals (wave 0)
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (lane >> 4) + 4 * q, col = lane & 15;
      if (row < 15 && col < 15) sh.P[parity ? col * 15 + row : row * 15 + col] = X[q];
    }
  }
  if (wave == 0) {
    if (lane < 32) { sh.tot[tot1] = run1; if (lane < 21) sh.tot[tot2] = run2; }
    if (lane == 32) { sh.tot[0] = Dq.x; sh.tot[1] = Dq.y; sh.tot[2] = Dq.z; sh.tot[3] = Dq.w; }
  }
  __syncthreads();
  st.Dq = Quat{sh.tot[0], sh.tot[1], sh.tot[2], sh.tot[3]};
  for (int k = 0; k < 9; ++k) {
    st.Ci[k] = sh.tot[4 + k]; st.Cdi[k] = sh.tot[13 + k]; st.dal[k] = sh.tot[28 + k]; st.dv[k] = sh.tot[37 + k]; st.dp[k] = sh.tot[46 + k];
  }
  for (int k = 0; k < 3; ++k) { st.ai[k] = sh.tot[22 + k]; st.adi[k] = sh.tot[25 + k]; }
  st.Delta_t = sh.tot[55];
  st.used = (int)sh.tot[56];
  IMU_TICK(qEnd);
  IMU_ACC(3, qStart, qEnd, t == 0);
}

__device__ void imuRedoPreintegration(DevImu& im, const uint32_t* __restrict__ imuT, const double* __restrict__ imuM,
                                      const double* sb, FactorShared& sh) {
  const int t = threadIdx.x;
  ImuState st;
  imuIntegrate<true>(im, imuT + 2 * (size_t)im.sampleStart, imuM + 6 * (size_t)im.sampleStart, sb, sh, st);
  // symmetrise P, information = P^-1 (via Cholesky), symmetrise, sqrtInfo = chol(information)^T  (:246-258).
  // Both factorisations run on the register-resident 16x16 routine of the reduced solver (one wave, identity
  // padding): it returns L and L^-1 together, so information = L^-T L^-1 needs no triangular solves.
  __syncthreads();
  IMU_TICK(qPost0);
  const int wave = t >> 6, lane = t & 63;
  if (t < 225) sh.T[t] = 0.5 * sh.P[t] + 0.5 * sh.P[(t % 15) * 15 + t / 15];
  __syncthreads();
  const double pDeltaMine = (t < 225) ? sh.T[t] : 0.0;   // (all global stores of this routine wait until its end: one ahead of
                                                           // a barrier holds the whole workgroup until the store has completed)
  {
    const int r = t >> 4, c = t & 15;
    sh.tile[r * kPanelLd + c] = (r < 15 && c < 15) ? sh.T[r * 15 + c] : ((r == c) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (wave == 0) cholDiag16Reg(sh.tile, sh.dinv, lane, &sh.flag);
  __syncthreads();
  if (t < 225) {  // information = Linv^T Linv; Linv[k][a] sits at tile[a][k] for k > a, dinv[a] on the diagonal
    const int a = t / 15, b = t % 15, k0 = a > b ? a : b;
    double s = 0;
    for (int k = k0; k < 15; ++k) {
      const double xa = (k > a) ? sh.tile[a * kPanelLd + k] : sh.dinv[a];
      const double xb = (k > b) ? sh.tile[b * kPanelLd + k] : sh.dinv[b];
      s += xa * xb;
    }
    sh.Fd[t] = s;
  }
  __syncthreads();
  if (t < 225) { sh.P[t] = 0.5 * sh.Fd[t] + 0.5 * sh.Fd[(t % 15) * 15 + t / 15]; }
  __syncthreads();
  const double infoMine = (t < 225) ? sh.P[t] : 0.0;
  {
    const int r = t >> 4, c = t & 15;
    sh.tile[r * kPanelLd + c] = (r < 15 && c < 15) ? sh.P[r * 15 + c] : ((r == c) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (wave == 0) cholDiag16Reg(sh.tile, sh.dinv, lane, &sh.flag);
  __syncthreads();
  if (t < 225) {
    const int a = t / 15, b = t % 15;
    const double w = (b >= a) ? sh.tile[b * kPanelLd + a] : 0.0;  // L^T
    im.sqrtInfo[t] = w;
    sh.W[t] = w;   // the factor evaluation that follows reads the weights and the state from LDS, not back from memory
    im.P_delta[t] = pDeltaMine;
    im.information[t] = infoMine;
  }
  if (t == 64) {   // staged copy of the pre-integrated state, in the order of the DevImu fields from Delta_t on
    double* q = sh.st;
    q[0] = st.Delta_t; q[1] = st.Dq.x; q[2] = st.Dq.y; q[3] = st.Dq.z; q[4] = st.Dq.w;
    for (int k = 0; k < 9; ++k) { q[5 + k] = st.Ci[k]; q[14 + k] = st.Cdi[k]; q[29 + k] = st.dal[k]; q[38 + k] = st.dv[k]; q[47 + k] = st.dp[k]; }
    for (int k = 0; k < 3; ++k) { q[23 + k] = st.ai[k]; q[26 + k] = st.adi[k]; }
  }
  // the pre-integrated state (thread 0) -- every thread holds identical values
  if (t == 0) {
    im.Delta_q[0] = st.Dq.x; im.Delta_q[1] = st.Dq.y; im.Delta_q[2] = st.Dq.z; im.Delta_q[3] = st.Dq.w;
    for (int k = 0; k < 9; ++k) { im.C_integral[k] = st.Ci[k]; im.C_doubleintegral[k] = st.Cdi[k]; im.dalpha_db_g[k] = st.dal[k]; im.dv_db_g[k] = st.dv[k]; im.dp_db_g[k] = st.dp[k]; }
    for (int k = 0; k < 3; ++k) { im.acc_integral[k] = st.ai[k]; im.acc_doubleintegral[k] = st.adi[k]; }
    for (int k = 0; k < 9; ++k) im.sb_ref[k] = sb[k];
    im.Delta_t = st.Delta_t;
  }
  __syncthreads();
  IMU_TICK(qPost1);
  IMU_ACC(4, qPost0, qPost1, t == 0);
}
#ifdef SVIN_IMU_TIMING
void debugImuTiming(double* out, bool reset) {
  if (reset) { double z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_imuDbg), z, sizeof(z)); return; }
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_imuDbg), 128);
}
#endif

// ImuError::propagation (ImuError.cpp:266-476, :479-697): io[0..6] T_WS, io[7..15] speed/bias (in/out), io[16..22] integrals (out);
// out[0] = number of integration steps (or -1), optional 15x15 jacobian / covariance.
__global__ __launch_bounds__(256) void k_imu_propagation(const DevImu* imPtr, const uint32_t* __restrict__ T,
                                                         const double* __restrict__ M, double* io, double* jac,
                                                         double* cov, int* used) {
  __shared__ FactorShared sh;
  const int t = threadIdx.x;
  const DevImu& im = *imPtr;
  // sanity (:279): the last measurement must not be older than t_end
  if (im.sampleCount <= 0 || timeLess(T + 2 * (im.sampleCount - 1), im.t1)) {
    if (t == 0) *used = -1;
    return;
  }
  double sb[9];
  for (int k = 0; k < 9; ++k) sb[k] = io[7 + k];
  ImuState st;
  imuIntegrate<false>(im, T, M, sb, sh, st);
  const TF T0 = makeTF(io);
  const double gz = im.par.g * (6371009.0 / sqrt(6371009.0 * 6371009.0));
  const double gW[3] = {im.par.g * 0.0, im.par.g * 0.0, gz};
  const double Dt = st.Delta_t;
  const Vec3 c2 = rotate(T0.C, Vec3{st.adi[0], st.adi[1], st.adi[2]});
  const Vec3 c1 = rotate(T0.C, Vec3{st.ai[0], st.ai[1], st.ai[2]});
  const double C2[3] = {c2.x, c2.y, c2.z}, C1v[3] = {c1.x, c1.y, c1.z};
  __syncthreads();
  if (t == 0) {
    *used = st.used;
    for (int k = 0; k < 3; ++k) io[k] = T0.r[k] + sb[k] * Dt + C2[k] - 0.5 * gW[k] * Dt * Dt;
    const Quat qn = qnormalized(qmul(T0.q, st.Dq));
    io[3] = qn.x; io[4] = qn.y; io[5] = qn.z; io[6] = qn.w;
    for (int k = 0; k < 3; ++k) io[7 + k] = sb[k] + C1v[k] - gW[k] * Dt;
    // second overload (ImuError.cpp:664-667): acc_doubleintegral, acc_integral, Delta_t for Estimator::imuIntegralsMap_
    for (int k = 0; k < 3; ++k) { io[16 + k] = st.adi[k]; io[19 + k] = st.ai[k]; }
    io[22] = Dt;
    if (jac) {
      for (int k = 0; k < 225; ++k) jac[k] = (k / 15 == k % 15) ? 1.0 : 0.0;
      auto setB = [&](int r0, int c0, const double* B, double s) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) jac[(r0 + a) * 15 + c0 + b] = s * B[a * 3 + b];
      };
      double X[9], T9[9];
      const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      crossMxDev(C2[0], C2[1], C2[2], X); setB(0, 3, X, -1.0);
      setB(0, 6, I3, Dt);
      mm3(T0.C.m, st.dp, T9); setB(0, 9, T9, 1.0);
      mm3(T0.C.m, st.Cdi, T9); setB(0, 12, T9, -1.0);
      mm3(T0.C.m, st.dal, T9); setB(3, 9, T9, -1.0);
      crossMxDev(C1v[0], C1v[1], C1v[2], X); setB(6, 3, X, -1.0);
      mm3(T0.C.m, st.dv, T9); setB(6, 9, T9, 1.0);
      mm3(T0.C.m, st.Ci, T9); setB(6, 12, T9, -1.0);
    }
  }
  if (cov) {
    // P = T P_delta T^T with T = blockdiag(C, C, C, I, I)
    __syncthreads();
    if (t < 225) {
      const int a = t / 15, b = t % 15;
      sh.Fd[t] = (a < 9 && b < 9 && a / 3 == b / 3) ? T0.C.m[(a % 3) * 3 + (b % 3)] : ((a == b) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t < 225) {
      const int a = t / 15, b = t % 15;
      double s = 0;
      for (int k = 0; k < 15; ++k) s += sh.Fd[a * 15 + k] * sh.P[k * 15 + b];
      sh.T[t] = s;
    }
    __syncthreads();
    if (t < 225) {
      const int a = t / 15, b = t % 15;
      double s = 0;
      for (int k = 0; k < 15; ++k) s += sh.T[a * 15 + k] * sh.Fd[b * 15 + k];
      cov[t] = s;
    }
  }
}

__device__ __forceinline__ const double* blockPtr(const DeviceProblem& p, bool cand, int kind, int slot) {
  if (kind == B_POSE) return (cand ? p.poseC : p.pose) + (size_t)slot * 7;
  if (kind == B_EXT) return (cand ? p.extC : p.ext) + (size_t)slot * 7;
  return (cand ? p.sbC : p.sb) + (size_t)slot * 9;
}
__device__ __forceinline__ int blockOff(const DeviceProblem& p, int kind, int slot) {
  if (kind == B_POSE) return p.poseOff[slot];
  if (kind == B_EXT) return p.extOff[slot];
  return p.sbOff[slot];
}

__device__ __forceinline__ void evalFactorBlock(const DeviceProblem& p, int cand, int f, FactorShared& sh) {
  const int t = threadIdx.x;
  const DevFactor& fac = p.factors[f];
  FactorLin& lin = (cand ? p.linCand : p.linCur)[f];
  const int m = fac.m;
  if (t == 0) sh.flag = 0;
  for (int i = t; i < 15 * 30; i += blockDim.x) sh.F[i] = 0;
  int ncols = 0;
  for (int b = 0; b < fac.nblk; ++b) ncols += (fac.blkKind[b] == B_SB) ? 9 : 6;
  // the block table of the linearisation record: four threads of a wave that has nothing else to do (one dependent
  // load each; thread 0 doing it cost four serial memory round trips per evaluation).  The loads go out here, the
  // stores at the very end: a global store ahead of a barrier holds the whole workgroup until it has completed.
  int tblOff = -1, tblDim = 0;
  if (t >= 192 && t < 196) {
    const int b = t - 192;
    if (b < fac.nblk) { tblOff = blockOff(p, fac.blkKind[b], fac.blkSlot[b]); tblDim = (fac.blkKind[b] == B_SB) ? 9 : 6; }
  }

  if (fac.kind == F_HOST) {
    // a caller-supplied cost function (Map::addResidualBlock with a ::ceres::CostFunction the library does not know, Map.cpp:341-376):
    // the host has evaluated it at this point's parameter blocks and written r, J and the block table of `lin` before this launch
    // (Window::evaluateHostFactors); what is left is the factor's share of the cost
    if (t < 64) {
      const double rv = (t < m && t < 16) ? lin.r[t] : 0.0;
      const double c = rowSum16(rv * rv);
      if (t == 0) cstore(p.partial + (size_t)PS_COST_FACTORS * kMaxPartials + f, 0.5 * c);
    }
    return;
  }
  IMU_TICK(qe0);
  if (fac.kind == F_IMU) {
    DevImu& im = p.imus[fac.imuIndex];
    const double* x0 = blockPtr(p, cand, fac.blkKind[0], fac.blkSlot[0]);
    const double* s0 = blockPtr(p, cand, fac.blkKind[1], fac.blkSlot[1]);
    const double* x1 = blockPtr(p, cand, fac.blkKind[2], fac.blkSlot[2]);
    const double* s1 = blockPtr(p, cand, fac.blkKind[3], fac.blkSlot[3]);
    // ONE memory round trip for everything the block needs -- parameter blocks, pre-integration state, weights and
    // the inputs of the redo test (the state / weights are re-staged if the test fires)
    auto stage = [&]() {
      if (t < 225) sh.W[t] = im.sqrtInfo[t];
      if (t < 56) sh.st[t] = (&im.Delta_t)[t];
    };
    stage();
    if (t >= 64 && t < 71) sh.xs[t - 64] = x0[t - 64];
    if (t >= 71 && t < 80) sh.xs[7 + t - 71] = s0[t - 71];
    if (t >= 80 && t < 87) sh.xs[16 + t - 80] = x1[t - 80];
    if (t >= 87 && t < 96) sh.xs[23 + t - 87] = s1[t - 87];
    if (t >= 96 && t < 102) sh.st[56 + t - 96] = im.sb_ref[3 + t - 96];
    if (t == 102) sh.st[62] = (double)im.redo;
    if (t == 103) sh.st[63] = dtSecDev(im.t1, im.t0);
    if (t == 104) sh.st[64] = im.par.g;
    __syncthreads();
    const double Delta_t = sh.st[63];
    double Db[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Db[k] = sh.xs[7 + 3 + k] - sh.st[56 + k];
    // ImuError.cpp:739: redo_ || |Delta_b_g| * Delta_t > 1e-4   (uniform across the workgroup)
    const bool redo = sh.st[62] != 0.0 || (sqrt(Db[0] * Db[0] + Db[1] * Db[1] + Db[2] * Db[2]) * Delta_t > 0.0001);
    if (redo) {
      double sbl[9];
      for (int k = 0; k < 9; ++k) sbl[k] = sh.xs[7 + k];
      __syncthreads();
      imuRedoPreintegration(im, p.imuT, p.imuMeas, sbl, sh);
      if (t == 0) { im.redo = 0; im.redoCounter++; }
#pragma unroll
      for (int k = 0; k < 6; ++k) Db[k] = 0;
      // (imuRedoPreintegration left the new weights in sh.W and the new state in sh.st and ended on a barrier)
    }
    IMU_TICK(qe1);
    IMU_ACC(8, qe0, qe1, t == 0 && !redo);
    // ImuError.cpp:751-791.  The un-weighted Jacobian F = [F0 | F1] (15 x 30, columns pose0(6) sb0(9) pose1(6) sb1(9);
    // sh.F was zeroed above) and the error are built by the first lane of each of the four waves side by side: every
    // one derives the few shared quantities itself (a sync would cost more) and then fills its share of the blocks.
    if ((t & 63) == 0) {
      const int part = t >> 6;
      const double* xs0 = sh.xs;
      const double* ss0 = sh.xs + 7;
      const double* xs1 = sh.xs + 16;
      const double* ss1 = sh.xs + 23;
      const double* imDelta_q = sh.st + 1;
      const double* imC_integral = sh.st + 5;
      const double* imC_doubleintegral = sh.st + 14;
      const double* imacc_integral = sh.st + 23;
      const double* imacc_doubleintegral = sh.st + 26;
      const double* imdalpha_db_g = sh.st + 29;
      const double* imdv_db_g = sh.st + 38;
      const double* imdp_db_g = sh.st + 47;
      double* F0 = sh.F;
      double* F1 = sh.F + 15;
      auto setB = [](double* F, int r0, int c0, const double* B, double s) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) F[(r0 + a) * 30 + c0 + b] = s * B[a * 3 + b];
      };
      const double n0 = 1.0 / sqrt(xs0[3] * xs0[3] + xs0[4] * xs0[4] + xs0[5] * xs0[5] + xs0[6] * xs0[6]);
      const Quat q0 = {xs0[3] * n0, xs0[4] * n0, xs0[5] * n0, xs0[6] * n0};
      const double n1 = 1.0 / sqrt(xs1[3] * xs1[3] + xs1[4] * xs1[4] + xs1[5] * xs1[5] + xs1[6] * xs1[6]);
      const Quat q1inv = {-xs1[3] * n1, -xs1[4] * n1, -xs1[5] * n1, xs1[6] * n1};  // inverse of the normalised q1
      const double a3[3] = {-(imdalpha_db_g[0] * Db[0] + imdalpha_db_g[1] * Db[1] + imdalpha_db_g[2] * Db[2]),
                            -(imdalpha_db_g[3] * Db[0] + imdalpha_db_g[4] * Db[1] + imdalpha_db_g[5] * Db[2]),
                            -(imdalpha_db_g[6] * Db[0] + imdalpha_db_g[7] * Db[1] + imdalpha_db_g[8] * Db[2])};
      const Quat Dq = qmul(deltaQDev(a3[0], a3[1], a3[2]), Quat{imDelta_q[0], imDelta_q[1], imDelta_q[2], imDelta_q[3]});
      IMU_TICK(qeb);
      IMU_ACC(12, qe1, qeb, !redo && part == 0);
      if (part == 0) {
        // position / velocity rows: everything that carries C_S0_W = C_WS_0^T, and the error vector
        const Mat3 C0 = quatToR(q0);
        double Ct[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) Ct[a * 3 + b] = C0.m[b * 3 + a];
        const double gpar = sh.st[64];
        const double gz = gpar * (6371009.0 / sqrt(6371009.0 * 6371009.0));
        const double gW[3] = {gpar * 0.0, gpar * 0.0, gz};
        double dpv[3], dvv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dpv[k] = xs0[k] - xs1[k] + ss0[k] * Delta_t - 0.5 * gW[k] * Delta_t * Delta_t;
          dvv[k] = ss0[k] - ss1[k] - gW[k] * Delta_t;
        }
        double X[9], T9[9];
        setB(F0, 0, 0, Ct, 1.0);
        crossMxDev(dpv[0], dpv[1], dpv[2], X); mm3(Ct, X, T9); setB(F0, 0, 3, T9, 1.0);
        setB(F0, 0, 6, Ct, Delta_t);
        crossMxDev(dvv[0], dvv[1], dvv[2], X); mm3(Ct, X, T9); setB(F0, 6, 3, T9, 1.0);
        setB(F0, 6, 6, Ct, 1.0);
        IMU_TICK(qec);
        IMU_ACC(13, qeb, qec, !redo);
        const Mat3 CtM = {{Ct[0], Ct[1], Ct[2], Ct[3], Ct[4], Ct[5], Ct[6], Ct[7], Ct[8]}};
        const Vec3 v1 = rotate(CtM, Vec3{dpv[0], dpv[1], dpv[2]});
        const Vec3 v2 = rotate(CtM, Vec3{dvv[0], dvv[1], dvv[2]});
        const double v1a[3] = {v1.x, v1.y, v1.z}, v2a[3] = {v2.x, v2.y, v2.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          // bias columns of the position / velocity rows: [dp_db_g | -C_doubleintegral], [dv_db_g | -C_integral]
          double s1 = 0, s2 = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            s1 += imdp_db_g[a * 3 + k] * Db[k] - imC_doubleintegral[a * 3 + k] * Db[3 + k];
            s2 += imdv_db_g[a * 3 + k] * Db[k] - imC_integral[a * 3 + k] * Db[3 + k];
          }
          sh.e[a] = v1a[a] + imacc_doubleintegral[a] + s1;
          sh.e[6 + a] = v2a[a] + imacc_integral[a] + s2;
        }
      } else if (part == 1) {
        // d e_q / d alpha_0 = [plus(Dq q1^-1) oplus(q0)]_3x3, the identity diagonals, the pre-integral bias blocks
#pragma unroll
        for (int k = 9; k < 15; ++k) { F0[k * 31] = 1.0; F1[k * 31] = -1.0; }  // bias rows (the other diagonal blocks are written whole)
        double Qp[16], Qo[16], Q44[16];
        quatPlusMat4(qmul(Dq, q1inv), Qp);
        quatOplusMat4(q0, Qo);
        mm4(Qp, Qo, Q44);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) F0[(3 + a) * 30 + 3 + b] = Q44[a * 4 + b];
        setB(F0, 0, 9, imdp_db_g, 1.0);
        setB(F0, 0, 12, imC_doubleintegral, -1.0);
        setB(F0, 6, 9, imdv_db_g, 1.0);
        setB(F0, 6, 12, imC_integral, -1.0);
      } else if (part == 2) {
        // d e_q / d b_g = [oplus(q1^-1 q0) oplus(Dq)]_3x3 (-dalpha_db_g)
        double Qo1[16], Qo2[16], Q44[16];
        quatOplusMat4(qmul(q1inv, q0), Qo1);
        quatOplusMat4(Dq, Qo2);
        mm4(Qo1, Qo2, Q44);
        double TL[9], nd[9], T9[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) { TL[a * 3 + b] = Q44[a * 4 + b]; nd[a * 3 + b] = -imdalpha_db_g[a * 3 + b]; }
        mm3(TL, nd, T9);
        setB(F0, 3, 9, T9, 1.0);
      } else {
        // d e_q / d alpha_1 = -[plus(Dq) oplus(q0) plus(q1^-1)]_3x3
        double Qo[16], Qp2[16], Qp3[16], Qt[16], Q44[16];
        quatOplusMat4(q0, Qo);
        quatPlusMat4(Dq, Qp2);
        quatPlusMat4(q1inv, Qp3);
        mm4(Qp2, Qo, Qt);
        mm4(Qt, Qp3, Q44);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) F1[(3 + a) * 30 + 3 + b] = -Q44[a * 4 + b];
        // (this lane is the least loaded of the four: it also takes the -C_S0_W blocks of the second pose and the
        // orientation / bias rows of the error vector)
        const Mat3 C0 = quatToR(q0);
        double nCt[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) nCt[a * 3 + b] = -C0.m[b * 3 + a];
        setB(F1, 0, 0, nCt, 1.0);
        setB(F1, 6, 6, nCt, 1.0);
        const Quat qd = qmul(Dq, qmul(q1inv, q0));
        sh.e[3] = 2 * qd.x; sh.e[4] = 2 * qd.y; sh.e[5] = 2 * qd.z;
#pragma unroll
        for (int k = 0; k < 6; ++k) sh.e[9 + k] = ss0[3 + k] - ss1[3 + k];
      }
      IMU_TICK(qe2);
      IMU_ACC(9, qe1, qe2, !redo && part == 0);
      IMU_ACC(11, qe1, qe1 + 1, !redo && part == 0);
      IMU_ACC(15, qe1, qe2, !redo && part == 3);
      IMU_ACC(14, qe1, qe2, !redo && part == 1);
    }
  } else {
    for (int k = t; k < m * m; k += blockDim.x) sh.W[k] = fac.sqrtInfo[k];
    if (t == 0) {
      const double* x0 = blockPtr(p, cand, fac.blkKind[0], fac.blkSlot[0]);
      if (fac.kind == F_POSE_PRIOR) {  // PoseError.cpp:87-132
        poseErrorEval(fac.meas, x0, sh.e, sh.F);   // dmath.hpp: the function svin_host_pose_error runs on the CPU
      } else if (fac.kind == F_SB_PRIOR) {  // SpeedAndBiasError.cpp:83-113
        for (int k = 0; k < 9; ++k) { sh.e[k] = fac.meas[k] - x0[k]; sh.F[k * 9 + k] = -1.0; }
      } else if (fac.kind == F_RELPOSE) {  // RelativePoseError.cpp:79-147
        const double* x1 = blockPtr(p, cand, fac.blkKind[1], fac.blkSlot[1]);
        const TF T0 = makeTF(x0), T1 = makeTF(x1);
        const Quat dq = qnormalized(qmul(T1.q, qnormalized(qinv(T0.q))));
        for (int k = 0; k < 3; ++k) sh.e[k] = T1.r[k] - T0.r[k];
        sh.e[3] = 2 * dq.x; sh.e[4] = 2 * dq.y; sh.e[5] = 2 * dq.z;
        double Q[9], Qo[9];
        quatPlusMat3(dq, Q);
        quatOplusMat3(dq, Qo);
        for (int a = 0; a < 3; ++a) {
          sh.F[a * 12 + a] = -1.0;
          sh.F[a * 12 + 6 + a] = 1.0;
          for (int b = 0; b < 3; ++b) {
            sh.F[(3 + a) * 12 + 3 + b] = -Q[a * 3 + b];
            sh.F[(3 + a) * 12 + 6 + 3 + b] = Qo[a * 3 + b];
          }
        }
      } else if (fac.kind == F_SONAR) {  // SonarError.cpp:118-183 (reference sign/anchor quirks kept)
        const TF Tx = makeTF(x0);
        const double range = fac.meas[0], heading = fac.meas[1];
        const double d[3] = {Tx.r[0] - fac.meas[2], Tx.r[1] - fac.meas[3], Tx.r[2] - fac.meas[4]};
        sh.e[0] = range - sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const TF Tso = makeTF(fac.aux);
        // T_WSo = T_WS * T_SSo ; point = T_WSo * (range cos h, range sin h, 0)
        const Vec3 rso = rotate(Tx.C, Vec3{Tso.r[0], Tso.r[1], Tso.r[2]});
        const Quat qwso = qnormalized(qmul(Tx.q, Tso.q));
        const Mat3 Cwso = quatToR(qwso);
        const Vec3 pp = rotate(Cwso, Vec3{range * cos(heading), range * sin(heading), 0.0});
        const double sp[3] = {pp.x + rso.x + Tx.r[0], pp.y + rso.y + Tx.r[1], pp.z + rso.z + Tx.r[2]};
        for (int a = 0; a < 3; ++a) sh.F[a] = (Tx.r[a] - sp[a]) / range;
      } else if (fac.kind == F_DEPTH) {  // DepthError.cpp:75-139
        sh.e[0] = x0[2] - (-1 * fac.meas[0] + fac.meas[1]);
        sh.F[2] = 1.0;
      }
    }
  }
  __syncthreads();
  // r = W e ; J = W F.  Thread (a = t & 15, c = t >> 4, + 16): row a, columns c and c + 16 -- no integer division
  {
    const int a = t & 15, c0 = t >> 4;
    if (fac.kind == F_IMU) {
      // m = 15, ncols = 30 at compile time: the row of W is read once, every LDS access has an immediate offset (with the
      // run-time bounds below this was ~120 instructions per output)
      if (a < 15) {
        double w[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) w[k] = sh.W[a * 15 + k];
        if (c0 == 0) {
          double sr = 0;
#pragma unroll
          for (int k = 0; k < 15; ++k) sr += w[k] * sh.e[k];
          lin.r[a] = sr;
          sh.rw[a] = sr;
        }
        double s0 = 0, s1 = 0;
        const bool second = c0 + 16 < 30;
        const double* Fc = sh.F + c0;
#pragma unroll
        for (int k = 0; k < 15; ++k) { s0 += w[k] * Fc[k * 30]; s1 += w[k] * Fc[k * 30 + (second ? 16 : 0)]; }
        lin.J[a * 30 + c0] = s0;
        if (second) lin.J[a * 30 + c0 + 16] = s1;
      }
    } else if (a < m) {
      if (c0 == 0) {
        double sr = 0;
        for (int k = 0; k < m; ++k) sr += sh.W[a * m + k] * sh.e[k];
        lin.r[a] = sr;
        sh.rw[a] = sr;
      }
      for (int c = c0; c < ncols; c += 16) {
        double sj = 0;
        for (int k = 0; k < m; ++k) sj += sh.W[a * m + k] * sh.F[k * ncols + c];
        lin.J[a * ncols + c] = sj;
      }
    }
  }
  ldsBarrier();   // sh.rw only: the stores of r and J above need not have landed
  // cost partial: 0.5 |r|^2 (the residuals sit in the first 16 lanes: one DPP row sum)
  if (t < 64) {
    const double rv = ((t & 15) < m && t < 16) ? sh.rw[t & 15] : 0.0;
    const double c = rowSum16(rv * rv);
    if (t == 0) cstore(p.partial + (size_t)PS_COST_FACTORS * kMaxPartials + f, 0.5 * c);
  }
  if (t >= 192 && t < 196) { lin.off[t - 192] = tblOff; lin.dim[t - 192] = tblDim; }
  if (t == 196) { lin.m = m; lin.ncols = ncols; }
  IMU_TICK(qe3);
  IMU_ACC(10, qe0, qe3, t == 0 && fac.kind == F_IMU);
}

__global__ __launch_bounds__(256) void k_eval_factors(DeviceProblem p, int cand, int costBlocksA) {
  __shared__ FactorShared sh;
  evalFactorBlock(p, cand, blockIdx.x, sh);
  // the factor block that finishes last also sums the cost (reprojection partials were written by the previous
  // kernel of the stream) -- saves the separate reduction launch
  if (costBlocksA >= 0) {
    __shared__ int lastFlag;
    if (lastBlockDone(&p.tickets[TK_EVAL], &lastFlag)) {
      reduceCost(p, costBlocksA, (int)gridDim.x, sh.P);
      if (threadIdx.x == 0) p.tickets[TK_EVAL] = 0;
    }
  }
}

void launchImuPropagation(const DevImu* im, const uint32_t* T, const double* M, double* io, double* jac, double* cov,
                          int* used, hipStream_t s) {
  hipLaunchKernelGGL(k_imu_propagation, dim3(1), dim3(256), 0, s, im, T, M, io, jac, cov, used);
}

void launchEvalFactors(const DeviceProblem& p, bool cand, hipStream_t s, bool sumCost) {
  if (p.F == 0) return;
  hipLaunchKernelGGL(k_eval_factors, dim3(p.F), dim3(256), 0, s, p, cand ? 1 : 0,
                     sumCost ? (p.N > 0 ? evalGrid(p.N) : 0) : -1);
}
// which kernel of evaluateAll sums the cost: 2 = prior evaluation, 1 = factor evaluation, 0 = separate launch
int costSummedBy(const DeviceProblem& p) {
  if (p.priorM > 0) return 2;   // (priorM is zero on the ranks that do not own the prior)
  return p.F > 0 ? 1 : 0;
}

// ================================================================ K3: marginalisation prior (H-space)
// cost = 0.5 c0 + bp^T dchi + 0.5 dchi^T Ht dchi ;  grad (lin space) = bp + Ht dchi
// Ceres multiplies the ambient Jacobian (J_min * lift(x_lin)) by PlusJacobian(x): for a pose block the
// effective tangent map is M = blockdiag(I3, oplus(q_cur * q_lin^-1)[0:3,0:3]) (MarginalizationError.cpp:798-844).
__device__ __forceinline__ void priorEvalBlock(const DeviceProblem& p, int cand, double* red) {
  const int t = threadIdx.x, m = p.priorM;
  double* priorDchi = cand ? p.priorDchiC : p.priorDchi;
  double* priorGrad = cand ? p.priorGradC : p.priorGrad;
  for (int b = t; b < p.priorBlocks; b += blockDim.x) {
    const PriorBlock& pb = p.priorBlk[b];
    double* M3 = (cand ? p.priorM3C : p.priorM3) + 9 * b;
    for (int k = 0; k < 9; ++k) M3[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (pb.mdim == 0) continue;
    const double* x = blockPtr(p, cand != 0, pb.kind, pb.slot);
    if (pb.kind == B_SB) {
      for (int k = 0; k < 9; ++k) priorDchi[pb.ord + k] = x[k] - pb.lin[k];
    } else {
      double d[6];
      poseMinus(x, pb.lin, d);
      for (int k = 0; k < 6; ++k) priorDchi[pb.ord + k] = d[k];
      // PlusJacobian normalises q (Transformation ctor); lift uses the raw linearisation quaternion
      const Quat qc = qnormalized(Quat{x[3], x[4], x[5], x[6]});
      const Quat ql = Quat{-pb.lin[3], -pb.lin[4], -pb.lin[5], pb.lin[6]};
      // 2*oplus(q_lin^-1)[0:3,:] * 0.5*oplus(q_cur)[:,0:3]
      double A[16], B[16], C[16];
      quatOplusMat4(ql, A);
      quatOplusMat4(qc, B);
      mm4(A, B, C);
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) M3[a * 3 + c] = C[a * 4 + c];
    }
  }
  __syncthreads();
  // grad = bp + Ht dchi (thread per row), cost
  double c = 0;
  for (int i = t; i < m; i += blockDim.x) {
    double s = 0;
    const double* row = p.priorH + (size_t)i * m;
    for (int k = 0; k < m; ++k) s += row[k] * priorDchi[k];
    priorGrad[i] = p.priorBp[i] + s;
    c += priorDchi[i] * (p.priorBp[i] + 0.5 * s);
  }
  const double tot = blockSum(c, red);
  if (t == 0) cstore(&p.scal->costPrior, 0.5 * (*p.priorC0) + tot);
}
__global__ __launch_bounds__(256) void k_prior_eval(DeviceProblem p, int cand, int costBlocksA) {
  __shared__ double red[72];
  priorEvalBlock(p, cand, red);
  if (costBlocksA >= 0) {  // last evaluation kernel of the stream: sum the total cost here
    __syncthreads();
    reduceCost(p, costBlocksA, p.F, red);
  }
}

// reprojBlocks: number of reprojection cost partials to sum (-1: the stand-alone evaluation's grid)
void launchEvalPrior(const DeviceProblem& p, bool cand, hipStream_t s, bool sumCost, int reprojBlocks) {
  if (p.priorM == 0 || !p.ownsCamera) return;
  const int nA = reprojBlocks >= 0 ? reprojBlocks : (p.N > 0 ? evalGrid(p.N) : 0);
  hipLaunchKernelGGL(k_prior_eval, dim3(1), dim3(256), 0, s, p, cand ? 1 : 0, sumCost ? nA : -1);
}

// Fused evaluation for the trust-region loop: blocks [0, F) evaluate the small factors (IMU re-preintegration can
// take tens of microseconds), blocks [F, F + nR) the reprojection residuals with 256 observations each, side by side
// in one launch, plus one block for the marginalisation prior (hasPrior); the block that finishes last sums the cost
// (sumCost) and publishes the scalars.
template <bool WITH_EXT>
__device__ __forceinline__ void k_eval_all_body(const DeviceProblem& p, int cand, int nR, int sumCost, int hasPrior) {
  __shared__ FactorShared sh;
  const int F = (int)gridDim.x - nR - hasPrior;
  if ((int)blockIdx.x < F) {
    SVIN_ARGS(SA(p.factors), SA(p.imus), SA(p.linCand), SA(p.linCur), SA(p.poseC), SA(p.pose), SA(p.sbC), SA(p.sb), SA(p.extC), SA(p.ext),
              SA(p.poseOff), SA(p.sbOff), SA(p.extOff), SA(p.partial), SA(p.imuT), SA(p.imuMeas));
  } else {
    SVIN_ARGS(SA(p.poseC), SA(p.pose), SA(p.extC), SA(p.ext), SA(p.lm), SA(p.lmC), SA(p.cams), SA(p.obsUv), SA(p.obsW), SA(p.obsIdx),
              SA(p.obsLm), SA(p.rCand), SA(p.JpCand), SA(p.JlCand), SA(p.partial), SA(p.scal), SA(p.vL), SA(p.yL), SA(p.lmPtr), SA(p.N),
              SA(p.nPose), SA(p.nExt), SA(p.nCam));
  }
  TRACE(16);
  if ((int)blockIdx.x < F) {
    evalFactorBlock(p, cand, blockIdx.x, sh);
    TRACE(17);
  } else if ((int)blockIdx.x == F + nR) {
    priorEvalBlock(p, cand, reinterpret_cast<double*>(&sh));
  } else {
    double* smem = reinterpret_cast<double*>(&sh);  // poses / extrinsics / cameras staged in the same LDS
    const bool defer = cand && p.lmDeferred;
    LmDefer df;
    if (defer) {
      df.on = 1;
      df.cg = p.scal->spareA0; df.cn = p.scal->spareA1;   // the dogleg coefficients of the fused step (k_post_solve)
      df.vL = p.vL; df.yL = p.yL; df.lmPtr = p.lmPtr; df.lmC = p.lmC;
      df.stepPartial = p.partial + (size_t)PS_STEP * kMaxPartials;
      df.xPartial = p.partial + (size_t)PS_XNORM * kMaxPartials;
    }
    evalReprojBlock<true, WITH_EXT>(blockIdx.x - F, smem + 48, smem, p.N, p.nPose, p.nExt, p.nCam, cand ? p.poseC : p.pose,
                                    cand ? p.extC : p.ext, defer ? p.lm : (cand ? p.lmC : p.lm), p.cams, p.obsUv, p.obsW, p.obsIdx, p.obsLm,
                                    cand ? p.rCand : p.rCur, cand ? p.JpCand : p.JpCur, cand ? p.JlCand : p.JlCur,
                                    cand ? p.JeCand : p.JeCur, p.partial + (size_t)PS_COST_REPROJ * kMaxPartials, (size_t)p.N,
                                    df, p.lmPrior);
  }
  TRACE(18);
  if (sumCost) {
    __shared__ int lastFlag;
    __shared__ double red4[72];
    if (lastBlockDoneLight(&p.tickets[TK_EVAL], &lastFlag)) {   // (cost partials and costPrior are cstore()d)
      TRACE(19);
      reduceCost(p, nR, F, red4, (cand && p.lmDeferred) ? nR : 0);
      if (threadIdx.x == 0) p.tickets[TK_EVAL] = 0;
      TRACE(20);
    }
  }
}
template <bool WITH_EXT>
__global__ __launch_bounds__(256) void k_eval_all(DeviceProblem p, int cand, int nR, int sumCost, int hasPrior) { k_eval_all_body<WITH_EXT>(p, cand, nR, sumCost, hasPrior); }
// The slot of this block's window.  The table is written by the host's copy before the launch and by nothing afterwards, so it is
// read through the CONSTANT address space like the kernel arguments it replaces: every field stays a scalar load the compiler may
// repeat wherever it likes.  Through a plain global pointer the loads behind the first barrier go through the vector memory path
// into VGPRs (k_eval_all_batch: 407 registers and 976 bytes of scratch where k_eval_all has 348 and none).
__device__ __forceinline__ const BatchSlot& batchSlot(const BatchSlot* slots) {
  typedef const BatchSlot __attribute__((address_space(4))) * ConstantSlot;
  return *(const BatchSlot*)(ConstantSlot)(slots + blockIdx.y);
}
// Batched form (blockIdx.y = the window of the batch, its problem from the slot table), as TWO launches: k_eval_all sizes every
// block for the factor blocks' 131 KB of LDS (one workgroup per CU) and for the registers of the IMU chain; with B windows in the
// grid the reprojection blocks are what fills the chip, so they get a kernel of their own (staging area only, registers of
// evalReprojBlock alone) and the factor / prior blocks follow with the cost sum.  Same device functions, same partial slots,
// same summation order as k_eval_all: a window's numbers do not change.
template <bool WITH_EXT>
__device__ __forceinline__ void evalReprojSplitBody(const DeviceProblem& p, int cand, double* smem) {
  SVIN_ARGS(SA(p.poseC), SA(p.pose), SA(p.extC), SA(p.ext), SA(p.lm), SA(p.lmC), SA(p.cams), SA(p.obsUv), SA(p.obsW), SA(p.obsIdx),
            SA(p.obsLm), SA(p.rCand), SA(p.JpCand), SA(p.JlCand), SA(p.partial), SA(p.scal), SA(p.vL), SA(p.yL), SA(p.lmPtr), SA(p.N),
            SA(p.nPose), SA(p.nExt), SA(p.nCam));
  const bool defer = cand && p.lmDeferred;
  LmDefer df;
  if (defer) {
    df.on = 1;
    df.cg = p.scal->spareA0; df.cn = p.scal->spareA1;
    df.vL = p.vL; df.yL = p.yL; df.lmPtr = p.lmPtr; df.lmC = p.lmC;
    df.stepPartial = p.partial + (size_t)PS_STEP * kMaxPartials;
    df.xPartial = p.partial + (size_t)PS_XNORM * kMaxPartials;
  }
  evalReprojBlock<true, WITH_EXT>(blockIdx.x, smem + 48, smem, p.N, p.nPose, p.nExt, p.nCam, cand ? p.poseC : p.pose,
                                  cand ? p.extC : p.ext, defer ? p.lm : (cand ? p.lmC : p.lm), p.cams, p.obsUv, p.obsW, p.obsIdx, p.obsLm,
                                  cand ? p.rCand : p.rCur, cand ? p.JpCand : p.JpCur, cand ? p.JlCand : p.JlCur,
                                  cand ? p.JeCand : p.JeCur, p.partial + (size_t)PS_COST_REPROJ * kMaxPartials, (size_t)p.N,
                                  df, p.lmPrior);
}
// blocks [0, F): the small factors, block F (hasPrior): the marginalisation prior; the block that finishes last sums the cost
__device__ __forceinline__ void evalRestSplitBody(const DeviceProblem& p, int cand, int nR, int hasPrior, FactorShared& sh) {
  const int F = (int)gridDim.x - hasPrior;
  if ((int)blockIdx.x < F) {
    SVIN_ARGS(SA(p.factors), SA(p.imus), SA(p.linCand), SA(p.linCur), SA(p.poseC), SA(p.pose), SA(p.sbC), SA(p.sb), SA(p.extC), SA(p.ext),
              SA(p.poseOff), SA(p.sbOff), SA(p.extOff), SA(p.partial), SA(p.imuT), SA(p.imuMeas));
    evalFactorBlock(p, cand, blockIdx.x, sh);
  } else {
    priorEvalBlock(p, cand, reinterpret_cast<double*>(&sh));
  }
  __shared__ int lastFlag;
  __shared__ double red4[72];
  if (lastBlockDoneLight(&p.tickets[TK_EVAL], &lastFlag)) {   // (cost partials and costPrior are cstore()d; the reprojection partials: previous launch)
    reduceCost(p, nR, F, red4, (cand && p.lmDeferred) ? nR : 0);
    if (threadIdx.x == 0) p.tickets[TK_EVAL] = 0;
  }
}
template <bool WITH_EXT>
__global__ __launch_bounds__(256) void k_eval_reproj_batch(const BatchSlot* __restrict__ slots, int cand) {
  extern __shared__ double smem[];
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchEval)) return;
  evalReprojSplitBody<WITH_EXT>(sl.p, cand, smem);
}
__global__ __launch_bounds__(256) void k_eval_rest_batch(const BatchSlot* __restrict__ slots, int cand, int nR, int hasPrior) {
  __shared__ FactorShared sh;
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchEval)) return;
  evalRestSplitBody(sl.p, cand, nR, hasPrior, sh);
}
// The same two launches for ONE window whose evaluation has more blocks than the chip has CUs (wide windows: 1 954 reprojection
// blocks at 500 000 observations): in k_eval_all every block carries the factor blocks' LDS, one workgroup per CU.
template <bool WITH_EXT>
__global__ __launch_bounds__(256) void k_eval_reproj_split(DeviceProblem p, int cand) {
  extern __shared__ double smem[];
  evalReprojSplitBody<WITH_EXT>(p, cand, smem);
}
__global__ __launch_bounds__(256) void k_eval_rest_split(DeviceProblem p, int cand, int nR, int hasPrior) {
  __shared__ FactorShared sh;
  evalRestSplitBody(p, cand, nR, hasPrior, sh);
}

constexpr int kEvalSplitBlocks = 512;   // more evaluation blocks than this: reprojection blocks in a launch of their own (launchEvalAll)
// fused evaluation possible: factors and observations present, camera-owning rank, staging area fits
bool canFuseEvaluation(const DeviceProblem& p) {
  const size_t stage = (size_t)(p.nPose + p.nExt) * 7 * 8 + (size_t)p.nCam * sizeof(CameraModel) + 64;
  return p.F > 0 && p.N > 0 && stage <= sizeof(FactorShared) && (p.N + 255) / 256 + p.F <= kMaxPartials;
}
static size_t evalSplitStageBytes(const DeviceProblem& p) {
  return (size_t)48 * 8 + (size_t)(p.nPose + p.nExt) * 7 * 8 + (size_t)p.nCam * sizeof(CameraModel) + 64;
}
void launchEvalAll(const DeviceProblem& p, bool cand, bool sumCost, hipStream_t s) {
  const int nR = (p.N + 255) / 256, pri = p.priorM > 0 ? 1 : 0;
  if (sumCost && p.F + nR + pri > kEvalSplitBlocks && !optOn(kOptNoEvalSplit)) {
    const size_t stage = evalSplitStageBytes(p);
    // (measured, round 6: the factor blocks on the side stream beside the reprojection blocks, one cost ticket for both launches --
    //  0.665 ms per iteration against 0.65 one after the other: both kernels slow down side by side by more than the overlap gains.
    //  Also measured: the factor blocks FIRST with an event behind them, so that the speculative build's side chain starts beside the
    //  reprojection blocks instead of behind them -- the side chain then ends before the main one needs it, but the reprojection
    //  launch takes 41 us instead of 27 next to it: 0.65 again.  A side stream of the highest priority changes nothing either: the
    //  one workgroup of k_sb_factor / the few of k_sb_forward run 4-5 x slower next to k_schur_rows whatever the queue says.)
    if (p.anyExtVariable) hipLaunchKernelGGL(k_eval_reproj_split<true>, dim3(nR), dim3(256), stage, s, p, cand ? 1 : 0);
    else hipLaunchKernelGGL(k_eval_reproj_split<false>, dim3(nR), dim3(256), stage, s, p, cand ? 1 : 0);
    hipLaunchKernelGGL(k_eval_rest_split, dim3(p.F + pri), dim3(256), 0, s, p, cand ? 1 : 0, nR, pri);
    return;
  }
  if (p.anyExtVariable)
    hipLaunchKernelGGL(k_eval_all<true>, dim3(p.F + nR + pri), dim3(256), 0, s, p, cand ? 1 : 0, nR, sumCost ? 1 : 0, pri);
  else
    hipLaunchKernelGGL(k_eval_all<false>, dim3(p.F + nR + pri), dim3(256), 0, s, p, cand ? 1 : 0, nR, sumCost ? 1 : 0, pri);
}

// row i of the prior -> (reduced-system row, or -1) with the 3x3 rotation map applied on the fly
__device__ __forceinline__ int priorFindBlock(const DeviceProblem& p, int row) {
  int b = 0;
  for (int k = 0; k < p.priorBlocks; ++k)
    if (p.priorBlk[k].mdim > 0 && row >= p.priorBlk[k].ord && row < p.priorBlk[k].ord + p.priorBlk[k].mdim) b = k;
  return b;
}
// (M^T X M)(i,j) for prior rows i,j: rows/cols 3..5 of pose blocks mix through M3
__device__ __forceinline__ double priorHeff(const DeviceProblem& p, int i, int j) {
  const int m = p.priorM;
  const int bi = priorFindBlock(p, i), bj = priorFindBlock(p, j);
  const PriorBlock& Bi = p.priorBlk[bi];
  const PriorBlock& Bj = p.priorBlk[bj];
  const int li = i - Bi.ord, lj = j - Bj.ord;
  const bool ri = (Bi.kind != B_SB) && li >= 3, rj = (Bj.kind != B_SB) && lj >= 3;
  double s = 0;
  const int i0 = ri ? Bi.ord + 3 : i, ni = ri ? 3 : 1;
  const int j0 = rj ? Bj.ord + 3 : j, nj = rj ? 3 : 1;
  for (int a = 0; a < ni; ++a) {
    const double wa = ri ? p.priorM3[9 * bi + a * 3 + (li - 3)] : 1.0;  // M[a][li-3] -> (M^T)[li-3][a]
    for (int b = 0; b < nj; ++b) {
      const double wb = rj ? p.priorM3[9 * bj + b * 3 + (lj - 3)] : 1.0;
      s += wa * p.priorH[(size_t)(i0 + a) * m + j0 + b] * wb;
    }
  }
  return s;
}
__device__ __forceinline__ int priorRowToReduced(const DeviceProblem& p, int row) {
  const PriorBlock& B = p.priorBlk[priorFindBlock(p, row)];
  const int off = blockOff(p, B.kind, B.slot);
  return off < 0 ? -1 : off + (row - B.ord);
}
__device__ void priorAccumulateBlock(const DeviceProblem& p, int block) {
  const int m = p.priorM;
  const int idx = block * blockDim.x + threadIdx.x;
  if (idx >= m * m) return;
  const int i = idx / m, j = idx % m;
  const int ri = priorRowToReduced(p, i), rj = priorRowToReduced(p, j);
  if (ri < 0 || rj < 0) return;
  const double h = priorHeff(p, i, j);
  atomicAdd(&p.S[(size_t)ri * p.ldS + rj], h);
  if (i == j) {
    atomicAdd(&p.hC[ri], h);
    // gradient: (M^T grad)(i)
    const int bi = priorFindBlock(p, i);
    const PriorBlock& Bi = p.priorBlk[bi];
    const int li = i - Bi.ord;
    double g;
    if (Bi.kind != B_SB && li >= 3) {
      g = 0;
      for (int a = 0; a < 3; ++a) g += p.priorM3[9 * bi + a * 3 + (li - 3)] * p.priorGrad[Bi.ord + 3 + a];
    } else {
      g = p.priorGrad[i];
    }
    atomicAdd(&p.gRed[ri], g);
    atomicAdd(&p.gFull[ri], g);
  }
}
__global__ __launch_bounds__(256) void k_prior_accumulate(DeviceProblem p) { priorAccumulateBlock(p, blockIdx.x); }
static int priorAccBlocks(const DeviceProblem& p) { return (p.priorM > 0 && p.ownsCamera) ? (p.priorM * p.priorM + 255) / 256 : 0; }
// ================================================================ K5: normal equations + landmark Schur complement
// generic small factors: J^T J into S (both triangles), J^T r into gRed/gFull, column norms into hC
__device__ void factorsAccumulate(const DeviceProblem& p, int f, int* colRow) {
  const FactorLin& lin = p.linCur[f];
  const int m = lin.m, nc = lin.ncols;
  // column -> reduced row map (colRow: 30 ints of LDS)
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
    for (int k = 0; k < m; ++k) s += lin.J[k * nc + a] * lin.J[k * nc + b];
    atomicAdd(&p.S[(size_t)ra * p.ldS + rb], s);
    if (a == b) {
      atomicAdd(&p.hC[ra], s);
      double g = 0;
      for (int k = 0; k < m; ++k) g += lin.J[k * nc + a] * lin.r[k];
      atomicAdd(&p.gRed[ra], g);
      atomicAdd(&p.gFull[ra], g);
    }
  }
}

constexpr int kStage = 34;  // doubles staged per observation: Jl 6, Jp 12, Je 12, offP, offE (as double), pad

// Blocks [0, nSchurBlocks) run the landmark Schur complement; the next nFacBlocks accumulate one small factor each
// (J^T J straight into S with atomics), the rest the marginalisation prior (256 entries of M^T Ht M each), so that the
// independent parts of the build share one launch.
template <bool USE_LDS, bool WITH_EXT>
__global__ __launch_bounds__(256) void k_schur(DeviceProblem p, double mu, int initScale, int nSchurBlocks, int nFacBlocks) {
  extern __shared__ double smem[];
  if ((int)blockIdx.x >= nSchurBlocks) {
    const int e = blockIdx.x - nSchurBlocks;
    if (e < nFacBlocks) factorsAccumulate(p, e, reinterpret_cast<int*>(smem));
    else priorAccumulateBlock(p, e - nFacBlocks);
    return;
  }
#ifdef SVIN_SCHUR_TIMING
  const long long qs0 = __builtin_readcyclecounter();
  long long qPass1 = 0, qVinv = 0, qPass2 = 0;
#endif
  const int dC = p.dC;
  const int ld = USE_LDS ? dC : dC;  // slab leading dimension
  double* accS;
  double* accV;  // gRed | gFull | hC (3*dC)
  double* stage;
  if (USE_LDS) {
    accS = smem;
    accV = smem + (size_t)dC * dC;
    stage = accV + 3 * dC;
    for (int i = threadIdx.x; i < dC * dC + 3 * dC; i += blockDim.x) smem[i] = 0;
    __syncthreads();
  } else {
    accS = p.slabs;                       // single global slab, atomics
    accV = p.slabs + (size_t)dC * dC;
    stage = smem;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* stg = stage + (size_t)wave * 64 * kStage;
  const size_t N = (size_t)p.N;
  const double* r = p.rCur;
  const double* Jp = p.JpCur;
  const double* Jl = p.JlCur;
  const double* Je = p.JeCur;
#ifdef SVIN_SCHUR_TIMING
  const long long qs1 = __builtin_readcyclecounter();
#endif
  const int gw = blockIdx.x * 4 + wave, nw = nSchurBlocks * 4;
  for (int l = gw; l < p.L; l += nw) {
    const int start = p.lmPtr[l], n = p.lmPtr[l + 1] - start;
#ifdef SVIN_SCHUR_TIMING
    const long long qa = __builtin_readcyclecounter();
#endif
    // ---- pass 1: V = sum Jl^T Jl, bl = sum Jl^T r (wave reduction)
    double v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (int i = lane; i < n; i += 64) {
      const size_t o = start + i;
      const double r0 = r[o], r1 = r[N + o];
      const double a0 = Jl[o], a1 = Jl[N + o], a2 = Jl[2 * N + o], c0 = Jl[3 * N + o], c1 = Jl[4 * N + o], c2 = Jl[5 * N + o];
      v00 += a0 * a0 + c0 * c0; v01 += a0 * a1 + c0 * c1; v02 += a0 * a2 + c0 * c2;
      v11 += a1 * a1 + c1 * c1; v12 += a1 * a2 + c1 * c2; v22 += a2 * a2 + c2 * c2;
      b0 += a0 * r0 + c0 * r1; b1 += a1 * r0 + c1 * r1; b2 += a2 * r0 + c2 * r1;
    }
    v00 = waveSum(v00); v01 = waveSum(v01); v02 = waveSum(v02); v11 = waveSum(v11); v12 = waveSum(v12);
    v22 = waveSum(v22); b0 = waveSum(b0); b1 = waveSum(b1); b2 = waveSum(b2);
#ifdef SVIN_SCHUR_TIMING
    const long long qb = __builtin_readcyclecounter();
    qPass1 += qb - qa;
#endif
    // trust-region metric for the landmark columns (Jacobi scaling fixed at iteration 0)
    double sc0, sc1, sc2;
    if (initScale) {
      sc0 = 1.0 / (1.0 + sqrt(v00)); sc1 = 1.0 / (1.0 + sqrt(v11)); sc2 = 1.0 / (1.0 + sqrt(v22));
      if (lane == 0) { p.scaleL[3 * l] = sc0; p.scaleL[3 * l + 1] = sc1; p.scaleL[3 * l + 2] = sc2; }
    } else {
      sc0 = p.scaleL[3 * l]; sc1 = p.scaleL[3 * l + 1]; sc2 = p.scaleL[3 * l + 2];
    }
    const double ht0 = fmin(fmax(v00 * sc0 * sc0, 1e-6), 1e32) / (sc0 * sc0);
    const double ht1 = fmin(fmax(v11 * sc1 * sc1, 1e-6), 1e32) / (sc1 * sc1);
    const double ht2 = fmin(fmax(v22 * sc2 * sc2, 1e-6), 1e32) / (sc2 * sc2);
    // (V + mu*htil)^-1 through its Cholesky factor
    const double d00 = v00 + mu * ht0, d11 = v11 + mu * ht1, d22 = v22 + mu * ht2;
    bool bad = !(d00 > 0);
    const double l00 = sqrt(d00);
    const double l10 = v01 / l00, l20 = v02 / l00;
    const double t11 = d11 - l10 * l10;
    bad = bad || !(t11 > 0);
    const double l11 = sqrt(t11);
    const double l21 = (v12 - l20 * l10) / l11;
    const double t22 = d22 - l20 * l20 - l21 * l21;
    bad = bad || !(t22 > 0);
    const double l22 = sqrt(t22);
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    // Vinv = Linv^T Linv
    const double w00 = i00 * i00 + i10 * i10 + i20 * i20, w01 = i10 * i11 + i20 * i21, w02 = i20 * i22;
    const double w11 = i11 * i11 + i21 * i21, w12 = i21 * i22, w22 = i22 * i22;
    if (lane == 0) {
      if (bad) atomicOr(&p.scal->cholFail, 1);
      double* vi = p.Vinv + 6 * (size_t)l;
      vi[0] = w00; vi[1] = w01; vi[2] = w02; vi[3] = w11; vi[4] = w12; vi[5] = w22;
      p.bl[3 * l] = b0; p.bl[3 * l + 1] = b1; p.bl[3 * l + 2] = b2;
      p.hL[3 * l] = ht0; p.hL[3 * l + 1] = ht1; p.hL[3 * l + 2] = ht2;
    }
    const double vb0 = w00 * b0 + w01 * b1 + w02 * b2, vb1 = w01 * b0 + w11 * b1 + w12 * b2, vb2 = w02 * b0 + w12 * b1 + w22 * b2;
#ifdef SVIN_SCHUR_TIMING
    const long long qc = __builtin_readcyclecounter();
    qVinv += qc - qb;
#endif
    // ---- pass 2: pairwise blocks  Jc_i^T (delta_ij I - Jl_i Vinv Jl_j^T) Jc_j
    for (int ci = 0; ci < n; ci += 64) {
      const int i = ci + lane;
      const bool act = i < n;
      const size_t o = start + (act ? i : 0);
      double jp[12], je[12], jl[6], ri[2];
      int offP = -1, offE = -1;
      if (act) {
        const uint32_t idx = p.obsIdx[o];
        offP = p.poseOff[idx & 0xfff];
        if (WITH_EXT) offE = p.extOff[(idx >> 12) & 0xfff];
        ri[0] = r[o]; ri[1] = r[N + o];
#pragma unroll
        for (int k = 0; k < 6; ++k) jl[k] = Jl[k * N + o];
#pragma unroll
        for (int k = 0; k < 12; ++k) jp[k] = Jp[k * N + o];
        if (WITH_EXT) {
#pragma unroll
          for (int k = 0; k < 12; ++k) je[k] = Je[k * N + o];
        }
      }
      // Y = Jl Vinv (2x3), z = r - Jl Vinv bl
      double Y[6], z[2];
      if (act) {
        for (int a = 0; a < 2; ++a) {
          Y[a * 3 + 0] = jl[a * 3] * w00 + jl[a * 3 + 1] * w01 + jl[a * 3 + 2] * w02;
          Y[a * 3 + 1] = jl[a * 3] * w01 + jl[a * 3 + 1] * w11 + jl[a * 3 + 2] * w12;
          Y[a * 3 + 2] = jl[a * 3] * w02 + jl[a * 3 + 1] * w12 + jl[a * 3 + 2] * w22;
          z[a] = ri[a] - (jl[a * 3] * vb0 + jl[a * 3 + 1] * vb1 + jl[a * 3 + 2] * vb2);
        }
        if (offP >= 0) {
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            atomicAdd(&accV[offP + a], jp[a] * z[0] + jp[6 + a] * z[1]);
            atomicAdd(&accV[dC + offP + a], jp[a] * ri[0] + jp[6 + a] * ri[1]);
            atomicAdd(&accV[2 * dC + offP + a], jp[a] * jp[a] + jp[6 + a] * jp[6 + a]);
          }
        }
        if (WITH_EXT && offE >= 0) {
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            atomicAdd(&accV[offE + a], je[a] * z[0] + je[6 + a] * z[1]);
            atomicAdd(&accV[dC + offE + a], je[a] * ri[0] + je[6 + a] * ri[1]);
            atomicAdd(&accV[2 * dC + offE + a], je[a] * je[a] + je[6 + a] * je[6 + a]);
          }
        }
      }
      for (int cj = 0; cj < n; cj += 64) {
        // stage chunk cj (this wave only)
        waveSync();
        {
          const int j = cj + lane;
          if (j < n) {
            const size_t oj = start + j;
            double* sj = stg + (size_t)lane * kStage;
            if (cj == ci) {
#pragma unroll
              for (int k = 0; k < 6; ++k) sj[k] = jl[k];
#pragma unroll
              for (int k = 0; k < 12; ++k) sj[6 + k] = jp[k];
              if (WITH_EXT) {
#pragma unroll
                for (int k = 0; k < 12; ++k) sj[18 + k] = je[k];
              }
              sj[30] = (double)offP; sj[31] = (double)offE;
            } else {
              const uint32_t idx = p.obsIdx[oj];
#pragma unroll
              for (int k = 0; k < 6; ++k) sj[k] = Jl[k * N + oj];
#pragma unroll
              for (int k = 0; k < 12; ++k) sj[6 + k] = Jp[k * N + oj];
              if (WITH_EXT) {
#pragma unroll
                for (int k = 0; k < 12; ++k) sj[18 + k] = Je[k * N + oj];
              }
              sj[30] = (double)p.poseOff[idx & 0xfff];
              sj[31] = WITH_EXT ? (double)p.extOff[(idx >> 12) & 0xfff] : -1.0;
            }
          }
        }
        waveSync();
        const int nj = min(64, n - cj);
        for (int jj = 0; jj < nj; ++jj) {
          const double* sj = stg + (size_t)jj * kStage;
          const int offPj = (int)sj[30], offEj = (int)sj[31];
          if (!act) continue;
          // K = delta - Y Jl_j^T
          const double e = (cj + jj == i) ? 1.0 : 0.0;
          const double K00 = e - (Y[0] * sj[0] + Y[1] * sj[1] + Y[2] * sj[2]);
          const double K01 = -(Y[0] * sj[3] + Y[1] * sj[4] + Y[2] * sj[5]);
          const double K10 = -(Y[3] * sj[0] + Y[4] * sj[1] + Y[5] * sj[2]);
          const double K11 = e - (Y[3] * sj[3] + Y[4] * sj[4] + Y[5] * sj[5]);
          // camera-side blocks of observation j: pose (sj+6), ext (sj+18)
          auto addBlock = [&](const double* ji, int offA, const double* jj_, int offB) {
            double t0[6], t1[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
              t0[b] = K00 * jj_[b] + K01 * jj_[6 + b];
              t1[b] = K10 * jj_[b] + K11 * jj_[6 + b];
            }
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int b = 0; b < 6; ++b) atomicAdd(&accS[(size_t)(offA + a) * ld + offB + b], ji[a] * t0[b] + ji[6 + a] * t1[b]);
          };
          if (offP >= 0 && offPj >= 0 && offP <= offPj) addBlock(jp, offP, sj + 6, offPj);
          if (WITH_EXT) {
            if (offP >= 0 && offEj >= 0) addBlock(jp, offP, sj + 18, offEj);          // poses precede extrinsics
            if (offE >= 0 && offEj >= 0 && offE <= offEj) addBlock(je, offE, sj + 18, offEj);
          }
        }
      }
    }
  }
#ifdef SVIN_SCHUR_TIMING
  const long long qs2 = __builtin_readcyclecounter();
#endif
  if (USE_LDS) {
    __syncthreads();
    double* slab = p.slabs + (size_t)blockIdx.x * ((size_t)dC * dC + 3 * dC);
    for (int i = threadIdx.x; i < dC * dC + 3 * dC; i += blockDim.x) slab[i] = smem[i];
  }
#ifdef SVIN_SCHUR_TIMING
  const long long qs3 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double* dbg = p.partial + (size_t)15 * 4096 + 16;
    dbg[0] += (double)(qs1 - qs0); dbg[1] += (double)qPass1; dbg[2] += (double)qVinv; dbg[3] += (double)(qs2 - qs1); dbg[4] += (double)(qs3 - qs2);
  }
#endif
}

// ---------------------------------------------------------------- dense Schur complement for narrow windows
// For windows of up to ~20 poses nearly every landmark couples most poses, and the landmark elimination is a
// Gram matrix:
//     S_cam = A - G G^T,   A = blockdiag_p( sum_{o in p} Jp_o^T Jp_o ),   G = [G_1 .. G_L],
//     G_l = (sum_{i in l} Jp_i^T Jl_i) L_l^-T  (dC x 3),   (V_l + mu*htil) = L_l L_l^T.
// Two augmented rows carry the gradients through the same product:
//     row dC   : sum Jp^T r  -  G c,  c_l = L_l^-1 b_l   = reduced gradient
//     row dC+1 : sum Jp^T r                              = full camera gradient
// Chunk blocks take 16 landmarks at a time, 16 lanes per landmark: V, b, L^-1, the columns of G into an LDS tile
// and the 6x6 pose blocks of A (plus Jp^T r) into per-wave LDS copies; then every wave adds the A blocks of the
// chunk into the accumulator tiles it owns and subtracts G G^T with v_mfma_f64_16x16x4_f64.  A and G G^T of the
// same 16 landmarks meet in the accumulator back to back, so their cancellation happens at the scale of one
// chunk (the rounding behaviour of the pairwise kernel).  Tiles stay in registers across chunks; one private slab
// per block, summed by k_reduce_slabs -- no atomics on S.  Extra blocks of the launch accumulate the small factors.
constexpr int kDenseLm = 16;              // landmarks per chunk
constexpr int kDenseK = 3 * kDenseLm;     // columns of G per chunk
constexpr int kDenseLd = kDenseK + 1;     // odd leading dimension keeps the MFMA operand reads off the same banks
constexpr int kPoseAcc = 28;              // per pose: 21 (upper 6x6) + 6 (Jp^T r) + pad

// index of (a, c), a <= c, in the packed upper triangle of a 6x6 block
__device__ __forceinline__ int sym6(int a, int c) { return a * 6 - a * (a - 1) / 2 + (c - a); }

// MAXT: accumulator tiles per wave (9 on 4 waves: up to 8 tile rows, dC <= 126; 10 / 17 on 8 waves: up to 12 / 16 tile rows,
// dC <= 190 / 254).
// A_MFMA: the A part goes through the same MFMA path as G (U = [.. Jc_o^T ..] in batches of obsBatch observations, two
// columns each, U U^T added to the tiles; the augmented rows carry r) instead of per-wave block copies.  Required with
// variable extrinsics (their Jacobians add rows to G, and A gets pose-extrinsics cross blocks) and used for the
// 34-tile variant (the block-copy merge does not fit the register file next to 34 accumulator tiles).
__host__ __device__ constexpr int denseObsBatch(int rows) { return rows <= 128 ? 32 : (rows <= 192 ? 16 : 8); }
// NW: waves per workgroup.  The landmark part always runs on the first 256 threads (16 lanes x 16 landmarks); with NW = 8 the
// tiles are dealt over twice as many waves (clear, merge, products and slab stores take half as long per wave, and 10 tiles
// per wave stay in registers where 20 spill).
template <int MAXT, bool A_MFMA, int NW>
__device__ __forceinline__ void k_schur_dense_body(const DeviceProblem& p, double mu, int initScale, int nChunkBlocks, int nFacBlocks) {
  static_assert(A_MFMA || NW == 4, "the per-wave block copies of A exist for four waves");
  extern __shared__ double smem[];
  const int t = threadIdx.x, b = blockIdx.x;
  if (b >= nChunkBlocks) {
    const int e = b - nChunkBlocks;
    if (e < nFacBlocks) factorsAccumulate(p, e, reinterpret_cast<int*>(smem));
    else priorAccumulateBlock(p, e - nFacBlocks);
    return;
  }
  SVIN_ARGS(SA(p.lmPtr), SA(p.poseOff), SA(p.rCur), SA(p.JlCur), SA(p.JpCur), SA(p.obsIdx), SA(p.scaleL), SA(p.Vinv), SA(p.bl), SA(p.hL),
            SA(p.slabs), SA(p.scal), SA(p.L), SA(p.N), SA(p.dC), SA(p.nPose), SA(p.anyExtVariable), SA(p.aBlocks), SA(p.dCPose));
  const size_t N = (size_t)p.N;
  const bool WITH_EXT = A_MFMA && p.anyExtVariable != 0;
  const int dC = p.dC, nP = dC / 6;          // reduced 6-blocks (poses, then variable extrinsics)
  const int nTr = (dC + 2 + 15) / 16, rows = 16 * nTr;
  const int obsBatch = denseObsBatch(rows), ldU = 2 * obsBatch + 1;
  double* Gt = smem;                          // rows x kDenseLd
  double* Aw = smem + (size_t)rows * kDenseLd;  // !A_MFMA: 4 waves x nP x kPoseAcc;  A_MFMA: U tile, rows x ldU
  double* Ut = Aw;
  // A_MFMA with p.aBlocks: the A part is accumulated block-wise with LDS atomics instead (one shared copy): diagonal
  // 6x6 blocks + Jc^T r per reduced block (Ash, kPoseAcc each), then the extrinsics x pose cross blocks (36 each)
  const bool useBlocks = A_MFMA && p.aBlocks != 0;
  const int nPB = p.dCPose / 6, nEB = nP - nPB;
  double* Ash = Aw;
  double* Across = Aw + (size_t)nP * kPoseAcc;
  const size_t extraLds = useBlocks ? (size_t)nP * kPoseAcc + (size_t)nEB * nPB * 36
                                    : (A_MFMA ? (size_t)rows * ldU : (size_t)4 * nP * kPoseAcc);
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, grp = t >> 4, gl = t & 15;   // scalar: the tile tests below are SALU
  double* Amine = Aw + (size_t)wave * nP * kPoseAcc;
  // accumulator tiles (I >= J) owned by this wave: tile index tl = wave, wave + 4, ...
  constexpr int kMaxTiles = MAXT;
  d4_t acc[kMaxTiles];
#pragma unroll
  for (int k = 0; k < kMaxTiles; ++k) acc[k] = d4_t{0, 0, 0, 0};
  const int nTiles = nTr * (nTr + 1) / 2;
  double hcAcc = 0;                           // thread c < dC: column norm hC[c]
#ifdef SVIN_SCHUR_TIMING
  const long long qd0 = __builtin_readcyclecounter();
  long long qdA = 0, qdL = 0, qdG = 0;
#endif
  // the pose -> row map of the reduced system rides on the same round trip as the first chunk's observation ranges (it was
  // a dependent global load per observation)
  // (kDensePoseCap poses at most: the host only takes this kernel for windows that narrow; a pointer that may be LDS or
  // global would turn every access into a FLAT instruction)
  __shared__ int sPoseOff[kDensePoseCap];
  for (int i = t; i < p.nPose && i < kDensePoseCap; i += blockDim.x) sPoseOff[i] = p.poseOff[i];
  const int* poseOffS = sPoseOff;
  {
    const int nz = rows * kDenseLd + (int)extraLds;   // 16-byte stores: half the LDS instructions of the clear
    for (int i = t; i < (nz >> 1); i += blockDim.x) reinterpret_cast<double2*>(smem)[i] = double2{0.0, 0.0};
    if ((nz & 1) && t == 0) smem[nz - 1] = 0.0;
  }
  __syncthreads();
#ifdef SVIN_SCHUR_TIMING
  const long long qd1 = __builtin_readcyclecounter();
#endif
  // acc(I,J) += sign * T_I T_J^T over nK4 steps of 4 columns of the LDS tile T (leading dimension ld)
  // tileMask: bit I set = tile row I of T holds non-zeros (a product of two tile rows needs both)
  auto rankUpdate = [&](const double* T, int ld, int nK4, double sign, unsigned tileMask) {
#pragma unroll
    for (int k = 0; k < kMaxTiles; ++k) {  // compile-time k: the accumulators stay in registers
      const int tl = wave + NW * k;
      if (tl < nTiles) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= tl) ++I;
        const int J = tl - I * (I + 1) / 2;
        if (!((tileMask >> I) & 1u) || !((tileMask >> J) & 1u)) continue;
        const double* A = T + (size_t)(16 * J + (lane & 15)) * ld + (lane >> 4);   // rows: the smaller tile index (upper triangle)
        const double* B = T + (size_t)(16 * I + (lane & 15)) * ld + (lane >> 4);
        d4_t c = acc[k];
        for (int q = 0; q < nK4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * A[4 * q], B[4 * q], c, 0, 0, 0);
        acc[k] = c;
      }
    }
  };
  for (int chunk = b; chunk * kDenseLm < p.L; chunk += nChunkBlocks) {
#ifdef SVIN_SCHUR_TIMING
    const long long qc0 = __builtin_readcyclecounter();
#endif
    if (A_MFMA && !useBlocks) {
      // ---- A part on MFMA: the chunk's observations are contiguous (landmark-major CSR); batches of obsBatch
      const int l0 = chunk * kDenseLm, l1 = min(p.L, l0 + kDenseLm);
      const int oBeg = p.lmPtr[l0], oEnd = p.lmPtr[l1];
      // the chunk's observations are taken pose by pose (p.obsOrder): a batch then touches the rows of one or two
      // poses and their extrinsics, and only the tiles between touched tile rows are multiplied
      __shared__ unsigned touched;
      const unsigned gradBits = (1u << (dC >> 4)) | (1u << ((dC + 1) >> 4));
      if (t == 0) touched = gradBits;
      __syncthreads();
      // observation index and row offsets of up to kMetaCap observations at a time: resolved once (three dependent
      // loads) instead of in every fill and clear pass of every batch
      constexpr int kMetaCap = 256;
      __shared__ int metaO[kMetaCap], metaP[kMetaCap], metaE[kMetaCap];
      for (int g0 = oBeg; g0 < oEnd; g0 += kMetaCap) {
        const int gn = min(kMetaCap, oEnd - g0);
        for (int j = t; j < gn; j += blockDim.x) {
          const int o = p.obsOrder ? p.obsOrder[g0 + j] : g0 + j;
          const uint32_t idx = p.obsIdx[o];
          metaO[j] = o;
          metaP[j] = p.poseOff[idx & 0xfff];
          metaE[j] = WITH_EXT ? p.extOff[(idx >> 12) & 0xfff] : -1;
        }
        __syncthreads();
        for (int ob = 0; ob < gn; ob += obsBatch) {
          const int nb = min(obsBatch, gn - ob);
          // entry e = k * nb + j: component k (0..11 Jp, 12..23 Je, 24..25 r) of observation ob + j of the group
          auto forEntries = [&](bool clear) {
            for (int e = t; e < 26 * nb; e += blockDim.x) {
              const int k = e / nb, j = e - k * nb;
              const size_t o = (size_t)metaO[ob + j];
              if (k < 24) {
                const bool isP = k < 12;
                if (!isP && !WITH_EXT) continue;
                const int kk = isP ? k : k - 12;
                const int off = isP ? metaP[ob + j] : metaE[ob + j];
                if (off >= 0 && !clear && kk == 0) atomicOr(&touched, (1u << (off >> 4)) | (1u << ((off + 5) >> 4)));
                if (off >= 0)
                  Ut[(size_t)(off + (kk % 6)) * ldU + 2 * j + kk / 6] = clear ? 0.0 : (isP ? p.JpCur : p.JeCur)[kk * N + o];
              } else {
                const double r = clear ? 0.0 : p.rCur[(k - 24) * N + o];
                Ut[(size_t)dC * ldU + 2 * j + (k - 24)] = r;
                Ut[(size_t)(dC + 1) * ldU + 2 * j + (k - 24)] = r;
              }
            }
          };
          forEntries(false);
          __syncthreads();
          if (t < dC) {  // column norms hC = diag(U U^T)
            double s2 = 0;
            for (int c = 0; c < 2 * nb; ++c) { const double u = Ut[(size_t)t * ldU + c]; s2 += u * u; }
            hcAcc += s2;
          }
          rankUpdate(Ut, ldU, (2 * nb + 3) / 4, 1.0, touched);
          __syncthreads();
          forEntries(true);  // clear exactly what was written
          if (t == 0) touched = gradBits;
          __syncthreads();
        }
      }
    }
#ifdef SVIN_SCHUR_TIMING
    const long long qc1 = __builtin_readcyclecounter();
    qdA += qc1 - qc0;
#endif
    double lmStore[12];
    bool lmStoreBad = false;
#pragma unroll
    for (int k = 0; k < 12; ++k) lmStore[k] = 0.0;
    const int l = (t < 256) ? chunk * kDenseLm + grp : p.L;
    if (l < p.L) {
      const int start = p.lmPtr[l], n = p.lmPtr[l + 1] - start;
      // everything this lane needs of its first observation (a landmark rarely has more than 16) is requested in ONE round
      // trip: residual, landmark Jacobian, packed indices and the pose Jacobian of the second pass below
      const bool has0 = gl < n;
      const size_t o0 = (size_t)start + (has0 ? gl : 0);
      double pr0[2] = {0, 0}, pjl[6] = {0, 0, 0, 0, 0, 0}, pjp[12];
      uint32_t pidx = 0;
#pragma unroll
      for (int k = 0; k < 12; ++k) pjp[k] = 0;
      if (has0) {
        pidx = p.obsIdx[o0];
        pr0[0] = p.rCur[o0]; pr0[1] = p.rCur[N + o0];
#pragma unroll
        for (int k = 0; k < 6; ++k) pjl[k] = p.JlCur[k * N + o0];
#pragma unroll
        for (int k = 0; k < 12; ++k) pjp[k] = p.JpCur[k * N + o0];
      }
      // V = sum Jl^T Jl, b = sum Jl^T r over the landmark's observations (16 lanes)
      double v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, b0 = 0, b1 = 0, b2 = 0;
      for (int i = gl; i < n; i += 16) {
        const size_t o = (size_t)start + i;
        const bool first = i == gl;
        const double r0 = first ? pr0[0] : p.rCur[o], r1 = first ? pr0[1] : p.rCur[N + o];
        const double a0 = first ? pjl[0] : p.JlCur[o], a1 = first ? pjl[1] : p.JlCur[N + o], a2 = first ? pjl[2] : p.JlCur[2 * N + o];
        const double c0 = first ? pjl[3] : p.JlCur[3 * N + o], c1 = first ? pjl[4] : p.JlCur[4 * N + o], c2 = first ? pjl[5] : p.JlCur[5 * N + o];
        v00 += a0 * a0 + c0 * c0; v01 += a0 * a1 + c0 * c1; v02 += a0 * a2 + c0 * c2;
        v11 += a1 * a1 + c1 * c1; v12 += a1 * a2 + c1 * c2; v22 += a2 * a2 + c2 * c2;
        b0 += a0 * r0 + c0 * r1; b1 += a1 * r0 + c1 * r1; b2 += a2 * r0 + c2 * r1;
      }
      v00 = rowSum16(v00); v01 = rowSum16(v01); v02 = rowSum16(v02); v11 = rowSum16(v11); v12 = rowSum16(v12); v22 = rowSum16(v22);
      b0 = rowSum16(b0); b1 = rowSum16(b1); b2 = rowSum16(b2);
      // trust-region metric for the landmark columns (Jacobi scaling fixed at iteration 0)
      double sc0, sc1, sc2;
      if (initScale) {
        sc0 = 1.0 / (1.0 + sqrt(v00)); sc1 = 1.0 / (1.0 + sqrt(v11)); sc2 = 1.0 / (1.0 + sqrt(v22));
        if (gl == 0) { p.scaleL[3 * l] = sc0; p.scaleL[3 * l + 1] = sc1; p.scaleL[3 * l + 2] = sc2; }
      } else {
        sc0 = p.scaleL[3 * l]; sc1 = p.scaleL[3 * l + 1]; sc2 = p.scaleL[3 * l + 2];
      }
      const double ht0 = fmin(fmax(v00 * sc0 * sc0, 1e-6), 1e32) / (sc0 * sc0);
      const double ht1 = fmin(fmax(v11 * sc1 * sc1, 1e-6), 1e32) / (sc1 * sc1);
      const double ht2 = fmin(fmax(v22 * sc2 * sc2, 1e-6), 1e32) / (sc2 * sc2);
      // (V + mu*htil) = L L^T; only L^-1 is needed, so the factor runs on reciprocal square roots
      const double d00 = v00 + mu * ht0, d11 = v11 + mu * ht1, d22 = v22 + mu * ht2;
      bool bad = !(d00 > 0);
      const double i00 = rsqrtNewton(bad ? 1.0 : d00);          // 1/l00
      const double l10 = v01 * i00, l20 = v02 * i00;
      const double t11 = d11 - l10 * l10;
      bad = bad || !(t11 > 0);
      const double i11 = rsqrtNewton(t11 > 0 ? t11 : 1.0);
      const double l21 = (v12 - l20 * l10) * i11;
      const double t22 = d22 - l20 * l20 - l21 * l21;
      bad = bad || !(t22 > 0);
      const double i22 = rsqrtNewton(t22 > 0 ? t22 : 1.0);
      const double i10 = -l10 * i00 * i11;
      const double i21 = -l21 * i11 * i22;
      const double i20 = -(l20 * i00 + l21 * i10) * i22;
      if (gl == 0) {
        // (the landmark's quantities for the later kernels are stored after the MFMA part of the chunk: a global store ahead
        // of the barrier below would hold the workgroup until it has completed)
        lmStore[0] = i00 * i00 + i10 * i10 + i20 * i20; lmStore[1] = i10 * i11 + i20 * i21; lmStore[2] = i20 * i22;   // Vinv = Linv^T Linv
        lmStore[3] = i11 * i11 + i21 * i21; lmStore[4] = i21 * i22; lmStore[5] = i22 * i22;
        lmStore[6] = b0; lmStore[7] = b1; lmStore[8] = b2; lmStore[9] = ht0; lmStore[10] = ht1; lmStore[11] = ht2;
        lmStoreBad = bad;
        // augmented row dC: c = Linv b  (row dC+1 stays zero under G)
        double* cr = Gt + (size_t)dC * kDenseLd + 3 * grp;
        cr[0] = i00 * b0; cr[1] = i10 * b0 + i11 * b1; cr[2] = i20 * b0 + i21 * b1 + i22 * b2;
      }
      // per observation: columns of G = (Jp^T Jl) Linv^T at the pose's rows (two cameras of one pose meet
      // there), and the pose block of A with Jp^T r into this wave's copy
      for (int i = gl; i < n; i += 16) {
        const size_t o = (size_t)start + i;
        const bool first = i == gl;
        const uint32_t idx = first ? pidx : p.obsIdx[o];
        const int offP = poseOffS[idx & 0xfff];
        const int offE = WITH_EXT ? p.extOff[(idx >> 12) & 0xfff] : -1;
        if (offP < 0 && offE < 0) continue;
        const double a0 = first ? pjl[0] : p.JlCur[o], a1 = first ? pjl[1] : p.JlCur[N + o], a2 = first ? pjl[2] : p.JlCur[2 * N + o];
        const double c0 = first ? pjl[3] : p.JlCur[3 * N + o], c1 = first ? pjl[4] : p.JlCur[4 * N + o], c2 = first ? pjl[5] : p.JlCur[5 * N + o];
        // one 6-block of camera-side Jacobian: rows of G (and, with fixed extrinsics, the block of A + Jc^T r)
        auto addBlock = [&](const double* J, int off) {
          double jc[12];
#pragma unroll
          for (int k = 0; k < 12; ++k) jc[k] = (first && J == p.JpCur) ? pjp[k] : J[k * N + o];
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            const double j0 = jc[a], j1 = jc[6 + a];
            const double e0 = j0 * a0 + j1 * c0, e1 = j0 * a1 + j1 * c1, e2 = j0 * a2 + j1 * c2;
            double* g = Gt + (size_t)(off + a) * kDenseLd + 3 * grp;
            atomicAdd(&g[0], e0 * i00);
            atomicAdd(&g[1], e0 * i10 + e1 * i11);
            atomicAdd(&g[2], e0 * i20 + e1 * i21 + e2 * i22);
            if (!A_MFMA) {
              const double r0 = first ? pr0[0] : p.rCur[o], r1 = first ? pr0[1] : p.rCur[N + o];
              double* ap = Amine + (size_t)(off / 6) * kPoseAcc;
#pragma unroll
              for (int c = a; c < 6; ++c) atomicAdd(&ap[sym6(a, c)], j0 * jc[c] + j1 * jc[6 + c]);
              atomicAdd(&ap[21 + a], j0 * r0 + j1 * r1);
            }
          }
        };
        if (offP >= 0) addBlock(p.JpCur, offP);
        if (WITH_EXT && offE >= 0) addBlock(p.JeCur, offE);
        if (useBlocks) {
          // A = sum Jc^T Jc of this observation: its pose block, its extrinsics block, their cross block, Jc^T r
          const double r0 = p.rCur[o], r1 = p.rCur[N + o];
          double jp[12], je[12];
#pragma unroll
          for (int k = 0; k < 12; ++k) { jp[k] = offP >= 0 ? p.JpCur[k * N + o] : 0.0; je[k] = offE >= 0 ? p.JeCur[k * N + o] : 0.0; }
          auto diagBlock = [&](const double* j, int off) {
            double* ap = Ash + (size_t)(off / 6) * kPoseAcc;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
              for (int c = a; c < 6; ++c) atomicAdd(&ap[sym6(a, c)], j[a] * j[c] + j[6 + a] * j[6 + c]);
              atomicAdd(&ap[21 + a], j[a] * r0 + j[6 + a] * r1);
            }
          };
          if (offP >= 0) diagBlock(jp, offP);
          if (offE >= 0) diagBlock(je, offE);
          if (offP >= 0 && offE >= 0) {
            double* cr = Across + ((size_t)((offE - p.dCPose) / 6) * nPB + offP / 6) * 36;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int c = 0; c < 6; ++c) atomicAdd(&cr[a * 6 + c], je[a] * jp[c] + je[6 + a] * jp[6 + c]);
          }
        }
      }
    }
    __syncthreads();
#ifdef SVIN_SCHUR_TIMING
    const long long qc2 = __builtin_readcyclecounter();
    qdL += qc2 - qc1;
#endif
    // ---- acc(I,J) += A_chunk (pose-diagonal blocks, gradient rows), then acc(I,J) -= G_I G_J^T (12 k-steps)
    if (!A_MFMA && t < dC) {
      const int ps = t / 6, a = t % 6;
      double s = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += Aw[((size_t)w * nP + ps) * kPoseAcc + sym6(a, a)];
      hcAcc += s;
    }
    if (useBlocks && t < dC) hcAcc += Ash[(size_t)(t / 6) * kPoseAcc + sym6(t % 6, t % 6)];
#pragma unroll
    for (int k = 0; k < kMaxTiles; ++k) {  // compile-time k: the accumulators stay in registers
      const int tl = wave + NW * k;
      if (tl < nTiles) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= tl) ++I;
        const int J = tl - I * (I + 1) / 2;
        // the wave owns the UPPER-triangle tile (R, C) = (J, I), R <= C: rows from tile R, columns from tile C -- the slab
        // rows are then written contiguously.  Only tiles near the diagonal (6x6 blocks straddle at most two tiles), the
        // tile column of the two gradient columns and the extrinsics x pose rectangle receive anything from A: the
        // tests are wave-uniform, every other tile goes straight to the products
        const int R = J, C = I;
        d4_t c = acc[k];
        const bool gradTile = C == (dC >> 4) || C == ((dC + 1) >> 4);
        const bool diagTile = C - R <= 1;
        const bool crossTile = useBlocks && 16 * C + 15 >= p.dCPose && 16 * R < p.dCPose;
        if (gradTile || diagTile || crossTile) {
          const int cc = 16 * C + (lane & 15);
          const int c6 = cc / 6, ce = cc - 6 * c6;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int r = 16 * R + (lane >> 4) + 4 * rg;
            const int r6 = r / 6, ra = r - 6 * r6;
            int idx = -1;
            if (r < dC) {
              if (cc < dC) { if (r6 == c6) idx = sym6(min(ra, ce), max(ra, ce)); }
              else if (cc <= dC + 1) idx = 21 + ra;   // both gradient columns start from Jc^T r
            }
            if (!A_MFMA && idx >= 0) {
              double s = 0;
#pragma unroll
              for (int w = 0; w < 4; ++w) s += Aw[((size_t)w * nP + r6) * kPoseAcc + idx];
              c[rg] += s;
            }
            if (useBlocks) {
              if (idx >= 0) {   // diagonal block / gradient columns
                c[rg] += Ash[(size_t)r6 * kPoseAcc + idx];
              } else if (crossTile && cc < dC && r < dC) {   // extrinsics x pose cross block, stored [extrinsics row][pose column]
                const bool rExt = r >= p.dCPose, cExt = cc >= p.dCPose;
                if (rExt != cExt) {
                  const int eb = (rExt ? r6 : c6) - nPB, pb = rExt ? c6 : r6;
                  const int ea = rExt ? ra : ce, pe = rExt ? ce : ra;
                  c[rg] += Across[((size_t)eb * nPB + pb) * 36 + ea * 6 + pe];
                }
              }
            }
          }
        }
        const double* A = Gt + (size_t)(16 * R + (lane & 15)) * kDenseLd + (lane >> 4);
        const double* B = Gt + (size_t)(16 * C + (lane & 15)) * kDenseLd + (lane >> 4);
#pragma unroll
        for (int q = 0; q < kDenseK / 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[4 * q], B[4 * q], c, 0, 0, 0);
        acc[k] = c;
      }
    }
    if (l < p.L && gl == 0) {   // the deferred per-landmark stores of this chunk
      if (lmStoreBad) atomicOr(&p.scal->cholFail, 1);
      double* vi = p.Vinv + 6 * (size_t)l;
#pragma unroll
      for (int k = 0; k < 6; ++k) vi[k] = lmStore[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) { p.bl[3 * l + k] = lmStore[6 + k]; p.hL[3 * l + k] = lmStore[9 + k]; }
    }
    if ((chunk + nChunkBlocks) * kDenseLm < p.L) {   // another chunk follows (wide problems only): clear the staging tiles for it
      __syncthreads();
      for (int i = t; i < rows * kDenseLd + (A_MFMA ? (useBlocks ? (int)extraLds : 0) : 4 * nP * kPoseAcc); i += blockDim.x) smem[i] = 0.0;
      __syncthreads();
    }
#ifdef SVIN_SCHUR_TIMING
    qdG += __builtin_readcyclecounter() - qc2;
#endif
  }
#ifdef SVIN_SCHUR_TIMING
  const long long qd2 = __builtin_readcyclecounter();
#endif
  // ---- private slab: [S (dC x dC) | gRed | gFull | hC]
  double* slab = p.slabs + (size_t)b * ((size_t)dC * dC + 3 * dC);
  if (t < dC) slab[(size_t)dC * dC + 2 * dC + t] = hcAcc;
#pragma unroll
  for (int k = 0; k < kMaxTiles; ++k) {
    const int tl = wave + NW * k;
    if (tl < nTiles) {
      int I = 0;
      while ((I + 1) * (I + 2) / 2 <= tl) ++I;
      const int J = tl - I * (I + 1) / 2;
      const int c = 16 * I + (lane & 15), c6 = c / 6;   // tile (R, C) = (J, I) of the upper triangle: 16 contiguous doubles per row
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 16 * J + (lane >> 4) + 4 * rg;
        const double v = acc[k][rg];
        if (r < dC && c < dC) {
          // k_reduce_slabs only reads the upper block triangle (6x6 blocks, diagonal blocks full)
          if (I != J || r / 6 <= c6) slab[(size_t)r * dC + c] = v;
          if (I != J && r / 6 == c6) slab[(size_t)c * dC + r] = v;   // a diagonal 6x6 block that straddles two tiles
        } else if (r < dC && c <= dC + 1) {
          slab[(size_t)dC * dC + (c - dC) * dC + r] = v;  // reduced gradient / full camera gradient
        }
      }
    }
  }
#ifdef SVIN_SCHUR_TIMING
  if (b == 0 && t == 0) {
    double* dbg = p.partial + (size_t)15 * 4096 + 16;
    dbg[0] += (double)(qd1 - qd0); dbg[1] += (double)qdA; dbg[2] += (double)qdL; dbg[3] += (double)qdG;
    dbg[4] += (double)(__builtin_readcyclecounter() - qd2);
  }
#endif
}
template <int MAXT, bool A_MFMA, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_schur_dense(DeviceProblem p, double mu, int initScale, int nChunkBlocks, int nFacBlocks) { k_schur_dense_body<MAXT, A_MFMA, NW>(p, mu, initScale, nChunkBlocks, nFacBlocks); }
// (batched form: blockIdx.y = the window of the batch, its problem and trust-region scalars from the slot table)
template <int MAXT, bool A_MFMA, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? SVIN_BATCH_OCC_SCHUR : 1) void k_schur_dense_batch(const BatchSlot* __restrict__ slots, int nChunkBlocks, int nFacBlocks) {
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchFull)) return;
  k_schur_dense_body<MAXT, A_MFMA, NW>(sl.p, sl.mu, sl.initScale, nChunkBlocks, nFacBlocks);
}


// ---------------------------------------------------------------- Gram-matrix Schur complement for WIDE windows
// dC > 254 rows do not fit one block's accumulator tiles, so the camera matrix is cut into panels of kPanelRows = 96
// rows (16 pose blocks, 6 MFMA tile rows) and every workgroup owns one panel pair (I >= J) for a list of landmark
// chunks whose observations touch both panels (host-built work list: in a sliding window a landmark is seen from a
// few consecutive frames, so most chunks touch one or two panels).  Per chunk the block rebuilds V, b, L^-1 (cheap),
// writes the rows of G that fall into panel I and panel J into two LDS tiles and subtracts G_I G_J^T on MFMA; diagonal
// pairs also add the 6x6 pose blocks of A (per-wave LDS copies) and collect the gradient / column-norm vectors.  One
// slab (96 x 96 + 3 x 96) per workgroup, summed per pair by k_reduce_panel_slabs.  Fixed extrinsics only (wide
// windows with variable extrinsics keep the pairwise kernel).
constexpr int kPanelRows = 96;
constexpr int kPanelSlab = kPanelRows * kPanelRows + 3 * kPanelRows;
static_assert(kPanelChunksPerBlock * 16 <= 256, "one thread per landmark of a workgroup's chunks stages its observation range");

#ifdef SVIN_SCHUR_TIMING
__device__ int g_panelCount;
#endif
// Two workgroups per CU (round 4): the three phases of a chunk -- landmark part (VALU), tile products (MFMA), clearing -- run one
// after the other inside a workgroup, and with ONE workgroup of four waves per CU (92 KB of LDS, 414 registers) nothing ever ran
// beside them: matrix pipes 16 % busy, VALU 19 %, LDS 5 % (profiles/r03_solver_pmc.txt).  A second resident workgroup fills
// those gaps without any hand-written producer / consumer protocol, so the footprint is cut to fit two: the per-wave pose
// blocks of A exist for diagonal pairs only and take the place of the second G tile there, the staged pose map holds 256 poses,
// and the register budget is 256 per lane (__launch_bounds__(256, 2)).
// Per-landmark quantities of a wide window, ONCE per build: V_l, b_l, the metric, L_l^-1 of the damped block and c_l = L_l^-1 b_l.
// A chunk of 16 landmarks is worked on by every panel pair its observations touch (three to six of them): each pair used to
// repeat this pass over all observations of the chunk, now it reads nine doubles per landmark.  16 lanes per landmark.
__global__ __launch_bounds__(256) void k_panels_landmarks(DeviceProblem p, double mu, int initScale) {
  const int t = threadIdx.x, grp = t >> 4, gl = t & 15;
  const int l = blockIdx.x * 16 + grp;
  if (l >= p.L) return;
  const size_t N = (size_t)p.N;
  const int start = p.lmPtr[l], n = p.lmPtr[l + 1] - start;
  double v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, b0 = 0, b1 = 0, b2 = 0;
  for (int i = gl; i < n; i += 16) {
    const size_t o = (size_t)start + i;
    const double r0 = p.rCur[o], r1 = p.rCur[N + o];
    const double a0 = p.JlCur[o], a1 = p.JlCur[N + o], a2 = p.JlCur[2 * N + o];
    const double c0 = p.JlCur[3 * N + o], c1 = p.JlCur[4 * N + o], c2 = p.JlCur[5 * N + o];
    v00 += a0 * a0 + c0 * c0; v01 += a0 * a1 + c0 * c1; v02 += a0 * a2 + c0 * c2;
    v11 += a1 * a1 + c1 * c1; v12 += a1 * a2 + c1 * c2; v22 += a2 * a2 + c2 * c2;
    b0 += a0 * r0 + c0 * r1; b1 += a1 * r0 + c1 * r1; b2 += a2 * r0 + c2 * r1;
  }
  v00 = rowSum16(v00); v01 = rowSum16(v01); v02 = rowSum16(v02); v11 = rowSum16(v11); v12 = rowSum16(v12); v22 = rowSum16(v22);
  b0 = rowSum16(b0); b1 = rowSum16(b1); b2 = rowSum16(b2);
  if (gl != 0) return;
  double sc0, sc1, sc2;
  if (initScale) {
    sc0 = 1.0 / (1.0 + sqrt(v00)); sc1 = 1.0 / (1.0 + sqrt(v11)); sc2 = 1.0 / (1.0 + sqrt(v22));
    p.scaleL[3 * l] = sc0; p.scaleL[3 * l + 1] = sc1; p.scaleL[3 * l + 2] = sc2;
  } else {
    sc0 = p.scaleL[3 * l]; sc1 = p.scaleL[3 * l + 1]; sc2 = p.scaleL[3 * l + 2];
  }
  const double ht0 = fmin(fmax(v00 * sc0 * sc0, 1e-6), 1e32) / (sc0 * sc0);
  const double ht1 = fmin(fmax(v11 * sc1 * sc1, 1e-6), 1e32) / (sc1 * sc1);
  const double ht2 = fmin(fmax(v22 * sc2 * sc2, 1e-6), 1e32) / (sc2 * sc2);
  const double d00 = v00 + mu * ht0, d11 = v11 + mu * ht1, d22 = v22 + mu * ht2;
  bool bad = !(d00 > 0);
  const double i00 = rsqrtNewton(bad ? 1.0 : d00);
  const double l10 = v01 * i00, l20 = v02 * i00;
  const double t11 = d11 - l10 * l10;
  bad = bad || !(t11 > 0);
  const double i11 = rsqrtNewton(t11 > 0 ? t11 : 1.0);
  const double l21 = (v12 - l20 * l10) * i11;
  const double t22 = d22 - l20 * l20 - l21 * l21;
  bad = bad || !(t22 > 0);
  const double i22 = rsqrtNewton(t22 > 0 ? t22 : 1.0);
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  if (bad) atomicOr(&p.scal->cholFail, 1);
  double* vi = p.Vinv + 6 * (size_t)l;
  vi[0] = i00 * i00 + i10 * i10 + i20 * i20; vi[1] = i10 * i11 + i20 * i21; vi[2] = i20 * i22;
  vi[3] = i11 * i11 + i21 * i21; vi[4] = i21 * i22; vi[5] = i22 * i22;
  p.bl[3 * l] = b0; p.bl[3 * l + 1] = b1; p.bl[3 * l + 2] = b2;
  p.hL[3 * l] = ht0; p.hL[3 * l + 1] = ht1; p.hL[3 * l + 2] = ht2;
  double* f = p.lmFactor + 9 * (size_t)l;
  f[0] = i00; f[1] = i10; f[2] = i11; f[3] = i20; f[4] = i21; f[5] = i22;
  f[6] = i00 * b0; f[7] = i10 * b0 + i11 * b1; f[8] = i20 * b0 + i21 * b1 + i22 * b2;
}

template <int kWavesPerSimd, bool kPrefetch, bool kPre>
__global__ __launch_bounds__(256, kWavesPerSimd) void k_schur_panels(DeviceProblem p, double mu, int initScale, int nWorkBlocks, int nFacBlocks) {
  extern __shared__ double smem[];
  const int t = threadIdx.x, b = blockIdx.x;
  (void)nWorkBlocks; (void)nFacBlocks;   // (the small factors and the prior have a launch of their own, k_factors_only: inlined here
                                         // their register needs decided this kernel's allocation)
  const int4 work = p.panelWork[b];  // x = I, y = J, z = first entry of panelChunks, w = number of chunks
  const int pI = work.x, pJ = work.y;
  const bool diag = pI == pJ;
  const int r0I = kPanelRows * pI, r0J = kPanelRows * pJ;
  const size_t N = (size_t)p.N;
  constexpr int nPB = kPanelRows / 6;              // pose blocks per panel
  double* GtI = smem;                               // kPanelRows x kDenseLd
  double* GtJ = diag ? GtI : smem + (size_t)kPanelRows * kDenseLd;
  double* Aw = smem + (size_t)kPanelRows * kDenseLd;        // diagonal pairs: 4 waves x nPB x kPoseAcc, where GtJ is otherwise
  double* cvec = diag ? Aw + (size_t)4 * nPB * kPoseAcc : smem + (size_t)2 * kPanelRows * kDenseLd;   // kDenseK: c_l = L_l^-1 b_l
  const int ldsDoubles = diag ? kPanelRows * kDenseLd + 4 * nPB * kPoseAcc + kDenseK : 2 * kPanelRows * kDenseLd + kDenseK;   // (even)
  const int wave = t >> 6, lane = t & 63, grp = t >> 4, gl = t & 15;
  double* Amine = Aw + (size_t)wave * nPB * kPoseAcc;
  constexpr int kMaxTiles = 9;                      // 6 x 6 tiles over 4 waves
  d4_t acc[kMaxTiles];
#pragma unroll
  for (int k = 0; k < kMaxTiles; ++k) acc[k] = d4_t{0, 0, 0, 0};
  double gRedAcc = 0, gFullAcc = 0, hcAcc = 0;      // thread r < kPanelRows (diagonal pairs)
  // A chunk used to cost five dependent round trips to HBM (chunk id -> observation range -> residual and landmark Jacobian ->
  // packed index -> pose row -> pose Jacobian) with ONE workgroup per CU to hide them (32 us per chunk, most of it waiting).
  // Now the block's chunk ids and observation ranges and the pose -> row map are staged in LDS up front, and everything a lane
  // needs of its first observation of chunk ci + 1 (a landmark rarely has more than 16) is requested while chunk ci is being
  // worked on.
  constexpr int kPoseStage = 256;
  __shared__ int sPoseOff[kPoseStage];
  __shared__ int sStart[kPanelChunksPerBlock * kDenseLm], sCount[kPanelChunksPerBlock * kDenseLm], sLm[kPanelChunksPerBlock * kDenseLm];
  // tile rows (16 rows each) of the two panels that the current chunk writes to: a tile whose row or column block got
  // nothing is skipped (16 landmarks see a dozen consecutive frames, not all sixteen of a panel)
  __shared__ unsigned sTouched[2];
  // ... and, per tile row, which of the twelve 4-column steps of the product hold anything: a landmark's three columns of G are
  // non-zero only in the rows of the poses that see it, so for most (tile row, tile column, step) triples one operand is all
  // zero -- the counters showed 20 executed MFMA flops per algorithmic one (profiles/r04_config4_mfma.json, before this mask)
  __shared__ unsigned sKMask[2][6];
  if (t < 2) sTouched[t] = 0u;
  if (t < 12) sKMask[t / 6][t % 6] = 0u;
  const bool stagedPose = p.nPose <= kPoseStage;
  if (stagedPose)
    for (int i = t; i < p.nPose; i += blockDim.x) sPoseOff[i] = p.poseOff[i];
  if (t < kPanelChunksPerBlock * kDenseLm) {
    const int ci = t >> 4, gq = t & 15;
    int st = 0, cnt = 0, lq = -1;
    if (ci < work.w) {
      const int lm = p.panelChunks[work.z + ci] * kDenseLm + gq;
      if (lm < p.L) { lq = lm; st = p.lmPtr[lm]; cnt = p.lmPtr[lm + 1] - st; }
    }
    sStart[t] = st; sCount[t] = cnt; sLm[t] = lq;
  }
  for (int i = t; i < ldsDoubles; i += blockDim.x) smem[i] = 0.0;
  __syncthreads();
  // first observation of this lane in a chunk: index, residual, landmark and pose Jacobian, and the landmark's column scales
  uint32_t preIdx = 0;
  // (the pose Jacobian is not among them: with two workgroups per CU its latency is covered by the other workgroup, and the 48
  // registers of a prefetched and a current copy are what kept the kernel at one wave per SIMD)
  double preR[2] = {0, 0}, preJl[6] = {0, 0, 0, 0, 0, 0}, preSc[3] = {1, 1, 1};
  auto fetch = [&](int ci) {
    if (!kPrefetch) return;
    const int st = sStart[ci * kDenseLm + grp], cnt = sCount[ci * kDenseLm + grp], lq = sLm[ci * kDenseLm + grp];
    if (gl < cnt) {
      const size_t o = (size_t)st + gl;
      preIdx = p.obsIdx[o];
      preR[0] = p.rCur[o]; preR[1] = p.rCur[N + o];
#pragma unroll
      for (int k = 0; k < 6; ++k) preJl[k] = p.JlCur[k * N + o];
    }
    if (!initScale && lq >= 0) { preSc[0] = p.scaleL[3 * lq]; preSc[1] = p.scaleL[3 * lq + 1]; preSc[2] = p.scaleL[3 * lq + 2]; }
  };
#ifdef SVIN_SCHUR_TIMING
  long long pT[6] = {0, 0, 0, 0, 0, 0}, pq0 = __builtin_readcyclecounter(), pq1;
#define PNT(i) do { pq1 = __builtin_readcyclecounter(); pT[i] += pq1 - pq0; pq0 = pq1; } while (0)
#else
#define PNT(i) do { } while (0)
#endif
  if (work.w > 0) fetch(0);
  PNT(0);
  for (int ci = 0; ci < work.w; ++ci) {
    const int l = sLm[ci * kDenseLm + grp];
    const int start = sStart[ci * kDenseLm + grp], n = sCount[ci * kDenseLm + grp];
    const uint32_t curIdx = preIdx;
    double curR[2] = {preR[0], preR[1]}, curJl[6], curSc[3] = {preSc[0], preSc[1], preSc[2]};
#pragma unroll
    for (int k = 0; k < 6; ++k) curJl[k] = preJl[k];
    if (ci + 1 < work.w) fetch(ci + 1);
    if (l >= 0) {
     double i00, i10, i11, i20, i21, i22;
     if (kPre) {   // (k_panels_landmarks has been there)
      const double* f = p.lmFactor + 9 * (size_t)l;
      i00 = f[0]; i10 = f[1]; i11 = f[2]; i20 = f[3]; i21 = f[4]; i22 = f[5];
      if (gl < 3) cvec[3 * grp + gl] = f[6 + gl];
     } else {
      double v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, b0 = 0, b1 = 0, b2 = 0;
      for (int i = gl; i < n; i += 16) {
        const size_t o = (size_t)start + i;
        const bool first = kPrefetch && i == gl;
        const double r0 = first ? curR[0] : p.rCur[o], r1 = first ? curR[1] : p.rCur[N + o];
        const double a0 = first ? curJl[0] : p.JlCur[o], a1 = first ? curJl[1] : p.JlCur[N + o], a2 = first ? curJl[2] : p.JlCur[2 * N + o];
        const double c0 = first ? curJl[3] : p.JlCur[3 * N + o], c1 = first ? curJl[4] : p.JlCur[4 * N + o], c2 = first ? curJl[5] : p.JlCur[5 * N + o];
        v00 += a0 * a0 + c0 * c0; v01 += a0 * a1 + c0 * c1; v02 += a0 * a2 + c0 * c2;
        v11 += a1 * a1 + c1 * c1; v12 += a1 * a2 + c1 * c2; v22 += a2 * a2 + c2 * c2;
        b0 += a0 * r0 + c0 * r1; b1 += a1 * r0 + c1 * r1; b2 += a2 * r0 + c2 * r1;
      }
      v00 = rowSum16(v00); v01 = rowSum16(v01); v02 = rowSum16(v02); v11 = rowSum16(v11); v12 = rowSum16(v12); v22 = rowSum16(v22);
      b0 = rowSum16(b0); b1 = rowSum16(b1); b2 = rowSum16(b2);
      double sc0, sc1, sc2;
      if (initScale) {
        sc0 = 1.0 / (1.0 + sqrt(v00)); sc1 = 1.0 / (1.0 + sqrt(v11)); sc2 = 1.0 / (1.0 + sqrt(v22));
        if (gl == 0) { p.scaleL[3 * l] = sc0; p.scaleL[3 * l + 1] = sc1; p.scaleL[3 * l + 2] = sc2; }
      } else {
        if (kPrefetch) { sc0 = curSc[0]; sc1 = curSc[1]; sc2 = curSc[2]; }
        else { sc0 = p.scaleL[3 * l]; sc1 = p.scaleL[3 * l + 1]; sc2 = p.scaleL[3 * l + 2]; }
      }
      const double ht0 = fmin(fmax(v00 * sc0 * sc0, 1e-6), 1e32) / (sc0 * sc0);
      const double ht1 = fmin(fmax(v11 * sc1 * sc1, 1e-6), 1e32) / (sc1 * sc1);
      const double ht2 = fmin(fmax(v22 * sc2 * sc2, 1e-6), 1e32) / (sc2 * sc2);
      const double d00 = v00 + mu * ht0, d11 = v11 + mu * ht1, d22 = v22 + mu * ht2;
      bool bad = !(d00 > 0);
      i00 = rsqrtNewton(bad ? 1.0 : d00);
      const double l10 = v01 * i00, l20 = v02 * i00;
      const double t11 = d11 - l10 * l10;
      bad = bad || !(t11 > 0);
      i11 = rsqrtNewton(t11 > 0 ? t11 : 1.0);
      const double l21 = (v12 - l20 * l10) * i11;
      const double t22 = d22 - l20 * l20 - l21 * l21;
      bad = bad || !(t22 > 0);
      i22 = rsqrtNewton(t22 > 0 ? t22 : 1.0);
      i10 = -l10 * i00 * i11;
      i21 = -l21 * i11 * i22;
      i20 = -(l20 * i00 + l21 * i10) * i22;
      if (gl == 0) {
        // every pair that sees this chunk computes the same per-landmark quantities; all of them store (same values)
        if (bad) atomicOr(&p.scal->cholFail, 1);
        double* vi = p.Vinv + 6 * (size_t)l;
        vi[0] = i00 * i00 + i10 * i10 + i20 * i20; vi[1] = i10 * i11 + i20 * i21; vi[2] = i20 * i22;
        vi[3] = i11 * i11 + i21 * i21; vi[4] = i21 * i22; vi[5] = i22 * i22;
        p.bl[3 * l] = b0; p.bl[3 * l + 1] = b1; p.bl[3 * l + 2] = b2;
        p.hL[3 * l] = ht0; p.hL[3 * l + 1] = ht1; p.hL[3 * l + 2] = ht2;
        double* cr = cvec + 3 * grp;
        cr[0] = i00 * b0; cr[1] = i10 * b0 + i11 * b1; cr[2] = i20 * b0 + i21 * b1 + i22 * b2;
      }
     }
      for (int i = gl; i < n; i += 16) {
        const size_t o = (size_t)start + i;
        const bool first = kPrefetch && i == gl;
        const int pi = (int)((first ? curIdx : p.obsIdx[o]) & 0xfff);
        int offP;
        if (stagedPose) offP = sPoseOff[pi]; else offP = p.poseOff[pi];
        if (offP < 0) continue;
        const bool inI = offP >= r0I && offP < r0I + kPanelRows;
        const bool inJ = !diag && offP >= r0J && offP < r0J + kPanelRows;
        if (!inI && !inJ) continue;
        const double a0 = first ? curJl[0] : p.JlCur[o], a1 = first ? curJl[1] : p.JlCur[N + o], a2 = first ? curJl[2] : p.JlCur[2 * N + o];
        const double c0 = first ? curJl[3] : p.JlCur[3 * N + o], c1 = first ? curJl[4] : p.JlCur[4 * N + o], c2 = first ? curJl[5] : p.JlCur[5 * N + o];
        double jc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) jc[k] = p.JpCur[k * N + o];
        double* Gt = inI ? GtI : GtJ;
        const int rl = offP - (inI ? r0I : r0J);
        atomicOr(&sTouched[inI ? 0 : 1], (1u << (rl >> 4)) | (1u << ((rl + 5) >> 4)));
        {
          const unsigned kbits = (1u << ((3 * grp) >> 2)) | (1u << ((3 * grp + 2) >> 2));   // the steps columns 3 grp .. 3 grp + 2 fall into
          unsigned* km = sKMask[inI ? 0 : 1];
          atomicOr(&km[rl >> 4], kbits);
          if (((rl + 5) >> 4) != (rl >> 4)) atomicOr(&km[(rl + 5) >> 4], kbits);
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double j0 = jc[a], j1 = jc[6 + a];
          const double e0 = j0 * a0 + j1 * c0, e1 = j0 * a1 + j1 * c1, e2 = j0 * a2 + j1 * c2;
          double* g = Gt + (size_t)(rl + a) * kDenseLd + 3 * grp;
          atomicAdd(&g[0], e0 * i00);
          atomicAdd(&g[1], e0 * i10 + e1 * i11);
          atomicAdd(&g[2], e0 * i20 + e1 * i21 + e2 * i22);
          if (diag) {
            const double r0 = first ? curR[0] : p.rCur[o], r1 = first ? curR[1] : p.rCur[N + o];
            double* ap = Amine + (size_t)(rl / 6) * kPoseAcc;
#pragma unroll
            for (int c = a; c < 6; ++c) atomicAdd(&ap[sym6(a, c)], j0 * jc[c] + j1 * jc[6 + c]);
            atomicAdd(&ap[21 + a], j0 * r0 + j1 * r1);
          }
        }
      }
    }
    PNT(1);
    __syncthreads();
    PNT(2);
    if (diag && t < kPanelRows) {  // gradient and column norms of this panel's rows
      const int ps = t / 6, a = t % 6;
      double hc = 0, gf = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        hc += Aw[((size_t)w * nPB + ps) * kPoseAcc + sym6(a, a)];
        gf += Aw[((size_t)w * nPB + ps) * kPoseAcc + 21 + a];
      }
      double gc = 0;
      for (int k = 0; k < kDenseK; ++k) gc += GtI[(size_t)t * kDenseLd + k] * cvec[k];
      hcAcc += hc; gFullAcc += gf; gRedAcc += gf - gc;
    }
    const unsigned touchedI = __builtin_amdgcn_readfirstlane(sTouched[0]);
    const unsigned touchedJ = diag ? touchedI : __builtin_amdgcn_readfirstlane(sTouched[1]);
#pragma unroll
    for (int k = 0; k < kMaxTiles; ++k) {
      const int tl = wave + 4 * k;           // 36 tiles: tile row ti (panel I), tile column tj (panel J)
      const int ti = tl / 6, tj = tl % 6;
      if (!((touchedI >> ti) & 1u) || !((touchedJ >> tj) & 1u)) continue;   // (wave-uniform)
      if (diag && ti < tj) continue;   // a diagonal pair is symmetric: k_reduce_panel_slabs mirrors its lower tiles
      d4_t c = acc[k];
      if (diag && ti - tj <= 1) {   // 6x6 blocks of A straddle at most two neighbouring tiles
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 16 * ti + (lane >> 4) + 4 * rg, cc = 16 * tj + (lane & 15);
          if (r / 6 == cc / 6) {
            const int a = r % 6, e = cc % 6, idx = sym6(min(a, e), max(a, e)), ps = cc / 6;
            double s = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += Aw[((size_t)w * nPB + ps) * kPoseAcc + idx];
            c[rg] += s;
          }
        }
      }
      const double* A = GtI + (size_t)(16 * ti + (lane & 15)) * kDenseLd + (lane >> 4);
      const double* B = GtJ + (size_t)(16 * tj + (lane & 15)) * kDenseLd + (lane >> 4);
      const unsigned steps = __builtin_amdgcn_readfirstlane(sKMask[0][ti]) & __builtin_amdgcn_readfirstlane(sKMask[diag ? 0 : 1][tj]);
#pragma unroll
      for (int q = 0; q < kDenseK / 4; ++q)
        if ((steps >> q) & 1u) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[4 * q], B[4 * q], c, 0, 0, 0);   // (wave-uniform)
      acc[k] = c;
    }
    PNT(3);
    __syncthreads();
    // (clearing only the tile rows the chunk wrote to was measured: the index arithmetic costs more than the stores it saves)
    for (int i = t; i < (ldsDoubles >> 1); i += blockDim.x) reinterpret_cast<double2*>(smem)[i] = double2{0.0, 0.0};   // (ldsDoubles is even)
    if (t < 2) sTouched[t] = 0u;
    if (t < 12) sKMask[t / 6][t % 6] = 0u;
    __syncthreads();
    PNT(4);
  }
  double* slab = p.slabs + (size_t)b * kPanelSlab;
#pragma unroll
  for (int k = 0; k < kMaxTiles; ++k) {
    const int tl = wave + 4 * k, ti = tl / 6, tj = tl % 6;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) slab[(size_t)(16 * ti + (lane >> 4) + 4 * rg) * kPanelRows + 16 * tj + (lane & 15)] = acc[k][rg];
  }
  if (t < kPanelRows) {
    double* v = slab + kPanelRows * kPanelRows;
    v[t] = gRedAcc; v[kPanelRows + t] = gFullAcc; v[2 * kPanelRows + t] = hcAcc;
  }
#ifdef SVIN_SCHUR_TIMING
  PNT(5);
  if ((b == 3 || b == 400) && (t == 0 || t == 64) && atomicAdd(&g_panelCount, 1) < 4)
    printf("[panels block %d wave %d, %d chunks, diag %d] prologue %lld  landmark part %lld  wait %lld  vectors+tiles %lld  clear %lld  slab %lld\n", b, t >> 6,
           work.w, (int)diag, pT[0], pT[1], pT[2], pT[3], pT[4], pT[5]);
#endif
#undef PNT
}

// ---------------------------------------------------------------- wide windows, round 6: the Schur complement by BLOCK PAIRS
// The tile form above writes the columns of G of 16 landmarks into 96 x 48 LDS tiles and multiplies whole 16 x 16 tiles: on the
// bench window of configs[3] a landmark sees 9.2 poses scattered over a span of 28, its 55 rows live in ~8 tiles of 16, and 6.3
// executed MFMA flops per algorithmic one is what no landmark order gets below (profiles/r05_config4_mfma.json).  Here the unit of
// work is what the algebra is made of:  S = A - sum_l sum_(a, b in poses(l)) E_la E_lb^T,  E_la = (sum_obs Jp^T Jl) L_l^-T  (6 x 3),
// A = blockdiag_pose sum_obs Jp^T Jp.  A SLOT is a (landmark, distinct variable pose) pair; slots, their observation lists and the
// work list are structure, built by Window::pack once per window.
//  * k_panels_landmarks (once per build, as for the tile form): V_l, b_l, the metric, L_l^-1, c_l.  k_blocks_slots, one thread per
//    slot: E_la and E_la c_l as a RECORD of 24 doubles rec[8 k + row] -- k = 0..2 the column of E, rows 0..5 the pose's tangent
//    directions, rows 6, 7 of the three columns the six entries of E_la c_l -- and the slot's share of its pose's block of A and of
//    Jp^T r (27 numbers), added to the workgroup's per-pose accumulators in LDS (ds_add_f64) and written out as one partial per
//    workgroup; k_blocks_pose_reduce sums the partials into S, gFull, gRed, hC.
//  * k_schur_rows: one workgroup (eight waves, two workgroups per CU) per panel pair (I, J) and list of up to 256 ENTRIES (a
//    landmark with slots in both panels), worked through in BATCHES of up to 360 records: the records of the batch's slots are
//    staged in LDS by all threads (stride 26 doubles: 24, a zero -- the K = 3 padding of every operand -- and a pad), then every
//    slot pair is ONE v_mfma_f64_4x4x4_4b_f64: its four independent 4 x 4 x 4 blocks are the four quadrants of the 8 x 8 padding
//    of a 6 x 6 block product with K = 3 padded to 4 (operand / result lane layout measured with tools/ubench/mfma_f64_4x4.hip:
//    A lane 16 k + 4 b + i, B lane 16 k + 4 b + j, D lane 16 i + 4 b + j; 16 cycles per instruction = the 16x16x4 form's flop
//    rate).  512 executed flops per 216 algorithmic ones, whatever the landmarks see.  The 6 x 6 blocks of the pair ACCUMULATE IN
//    REGISTERS: a wave owns two of the sixteen block rows (the host deals the rows by their pair counts), i.e. 32 accumulators
//    of two VGPRs, and its pairs arrive as host-built PAIR WORDS (A record | B record << 9 | pose block in J << 18 | "new A" << 24)
//    sorted by row and A record: B is read for every pair, A once per run, the accumulator is picked through the VGPR index
//    register.  No
//    atomics, no accumulator image in LDS, a fixed summation order (the result is bit-reproducible), one barrier pair per batch;
//    the second workgroup of the CU computes while this one waits for its records.  Diagonal pairs collect sum_l E_la c_l from
//    the operand lanes that hold it.  At the end the accumulators go through LDS into the slab, which has the tile form's
//    layout: k_reduce_panel_slabs is unchanged.
//    (Measured on the way, profiles/r06_schur_blocks_history.txt: accumulators as an LDS image fed by ds_add_f64, a wave per
//    entry with private staging buffers -- 582 us with four dependent global round trips per entry, 399 with the work list in LDS
//    and a register prefetch, 234-275 with pair words: bound by the LDS, whose ds_add_f64 takes 8 cycles per instruction whatever
//    the number of active lanes and whose queue, kept full by twelve waves, turned every dependent LDS or scalar-memory wait of an
//    entry's prologue into ~700 cycles.)
constexpr int kBlkPanelPoses = kPanelRows / 6;   // 16
constexpr int kBlkRecChunks = kBlkRec / 2;       // 16-byte pieces of a slot record (9)
constexpr int kBlkStride = 20;                   // doubles per record staged in LDS: 18, then a pair nothing ever writes (zero: the K = 3 padding of every operand)
constexpr int kBlkStrideChunks = kBlkStride / 2;
constexpr int kBlkPoseLd = 35;                   // per pose block of the slot pass's accumulators: odd, so that the sixteen slots of a landmark hit different banks
constexpr int kBlkPart = 34;                     // doubles per pose block of a partial: 21 (A) + 6 (Jp^T r) + 6 (sum E c) + pad

// one THREAD per slot, 1024 slots per workgroup (k_panels_landmarks has left L_l^-1 and c_l in lmFactor): W = sum over the slot's
// observations of Jp^T Jl (6 x 3), E = W L^-T, E c -> the slot's record; the pose's share of A and Jp^T r -> the workgroup's
// accumulators.  (The first version did this inside the per-landmark pass, 16 lanes per landmark and four landmarks per group
// one after the other: six dependent global round trips per landmark and 4-way address conflicts of the LDS atomics -- 119 us.)
__global__ __launch_bounds__(256) void k_blocks_slots(DeviceProblem p, int nCopies) {
  extern __shared__ double sA[];   // nCopies x per pose block: 21 (upper 6 x 6 of sum Jp^T Jp) + 6 (Jp^T r) + pad
  const int t = threadIdx.x;
  const int nBlk = p.dC / 6;
  for (int i = t; i < nCopies * nBlk * kBlkPoseLd; i += 256) sA[i] = 0.0;
  __syncthreads();
  // (consecutive slots are the poses of one landmark, then of the next: the sixteen lanes of a group rarely meet in a pose,
  // the four groups of a wave often do -- a copy per group)
  double* sMine = sA + (size_t)((t >> 4) & (nCopies - 1)) * nBlk * kBlkPoseLd;
  const size_t N = (size_t)p.N;
  // the index chain of ALL the thread's slots first (observation range, pose block, landmark, first observation): three dependent
  // global round trips once per workgroup instead of once per trip -- the kernel has ~1 workgroup per CU and was latency all the way
  // (68 -> 57 us)
  constexpr int kTrips = kBlkSlotsPerWorkgroup / 256;
  int q0s[kTrips], q1s[kTrips], blks[kTrips], lms[kTrips], o0s[kTrips];
#pragma unroll
  for (int j = 0; j < kTrips; ++j) {
    const int sl = blockIdx.x * kBlkSlotsPerWorkgroup + 256 * j + t;
    const bool has = sl < p.nSlots;
    q0s[j] = has ? p.slotObsPtr[sl] : 0; q1s[j] = has ? p.slotObsPtr[sl + 1] : 0;
    blks[j] = has ? (int)p.slotBlk[sl] : 0; lms[j] = has ? p.slotLm[sl] : 0;
  }
#pragma unroll
  for (int j = 0; j < kTrips; ++j) o0s[j] = q0s[j] < q1s[j] ? p.slotObs[q0s[j]] : 0;
#pragma unroll
  for (int j = 0; j < kTrips; ++j) {
    const int sl = blockIdx.x * kBlkSlotsPerWorkgroup + 256 * j + t;
    if (sl >= p.nSlots) break;
    const int q0 = q0s[j], q1 = q1s[j];
    const int blk = blks[j];
    const double* f = p.lmFactor + 9 * (size_t)lms[j];
    const double i00 = f[0], i10 = f[1], i11 = f[2], i20 = f[3], i21 = f[4], i22 = f[5], cv0 = f[6], cv1 = f[7], cv2 = f[8];
    double W[6][3], U[21], gF[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) { W[a][0] = W[a][1] = W[a][2] = 0.0; gF[a] = 0.0; }
#pragma unroll
    for (int a = 0; a < 21; ++a) U[a] = 0.0;
    for (int q = q0; q < q1; ++q) {
      const size_t o = (size_t)(q == q0 ? o0s[j] : p.slotObs[q]);
      const double a0 = p.JlCur[o], a1 = p.JlCur[N + o], a2 = p.JlCur[2 * N + o];
      const double c0 = p.JlCur[3 * N + o], c1 = p.JlCur[4 * N + o], c2 = p.JlCur[5 * N + o];
      const double r0 = p.rCur[o], r1 = p.rCur[N + o];
      double jc[12];
#pragma unroll
      for (int a = 0; a < 12; ++a) jc[a] = p.JpCur[a * N + o];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double j0 = jc[a], j1 = jc[6 + a];
        W[a][0] += j0 * a0 + j1 * c0; W[a][1] += j0 * a1 + j1 * c1; W[a][2] += j0 * a2 + j1 * c2;
#pragma unroll
        for (int c = a; c < 6; ++c) U[sym6(a, c)] += j0 * jc[c] + j1 * jc[6 + c];
        gF[a] += j0 * r0 + j1 * r1;
      }
    }
    double* ap = sMine + (size_t)blk * kBlkPoseLd;
#ifndef SVIN_SLOTS_NOATOM   // (timing experiments only: wrong results)
#pragma unroll
    for (int a = 0; a < 21; ++a) atomicAdd(&ap[a], U[a]);
#pragma unroll
    for (int a = 0; a < 6; ++a) atomicAdd(&ap[21 + a], gF[a]);
#else
    if (U[0] + gF[0] == 1.2345) ap[0] = 1.0;
#endif
    double rec[kBlkRec];
    double ec[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double e0 = W[a][0] * i00, e1 = W[a][0] * i10 + W[a][1] * i11, e2 = W[a][0] * i20 + W[a][1] * i21 + W[a][2] * i22;
      rec[a] = e0; rec[6 + a] = e1; rec[12 + a] = e2;
      ec[a] = e0 * cv0 + e1 * cv1 + e2 * cv2;
    }
#ifndef SVIN_SLOTS_NOATOM
#pragma unroll
    for (int a = 0; a < 6; ++a) atomicAdd(&ap[27 + a], ec[a]);   // sum_l E_la c_l: the reduced gradient's share
#endif
    double2* out = reinterpret_cast<double2*>(p.slotRec + (size_t)sl * kBlkRec);   // (records are 144 bytes: 16-byte aligned)
#ifndef SVIN_SLOTS_NOSTORE
#pragma unroll
    for (int q = 0; q < kBlkRec / 2; ++q) out[q] = double2{rec[2 * q], rec[2 * q + 1]};
#else
    { double sum = 0; for (int q = 0; q < kBlkRec; ++q) sum += rec[q]; if (sum == 1.2345) out[0] = double2{sum, sum}; }
#endif
  }
  __syncthreads();
  double* part = p.blkPartial + (size_t)blockIdx.x * nBlk * kBlkPart;
  for (int i = t; i < nBlk * kBlkPart; i += 256) {
    const int blk = i / kBlkPart, e = i - blk * kBlkPart;
    double v = 0;
    if (e < 33)
      for (int c = 0; c < nCopies; ++c) v += sA[((size_t)c * nBlk + blk) * kBlkPoseLd + e];
    part[i] = v;
  }
}

// S (pose diagonal blocks), gFull, gRed, hC += the per-pose sums of k_blocks_slots' partials; one workgroup per pose block,
// 32 lanes per partition of the partials (fixed order: deterministic)
__global__ __launch_bounds__(1024) void k_blocks_pose_reduce(DeviceProblem p, int nPartials) {
  __shared__ double part[1024];
  const int t = threadIdx.x, e = t & 63, q = t >> 6, blk = blockIdx.x;
  const int nBlk = p.dC / 6;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (e < kBlkPart) {
    const int per = (nPartials + 15) / 16;
    const int k1 = min(nPartials, (q + 1) * per);
    const double* src = p.blkPartial + (size_t)blk * kBlkPart + e;
    const size_t stride = (size_t)nBlk * kBlkPart;
    int k = q * per;
    for (; k + 3 < k1; k += 4) { s0 += src[k * stride]; s1 += src[(k + 1) * stride]; s2 += src[(k + 2) * stride]; s3 += src[(k + 3) * stride]; }
    for (; k < k1; ++k) s0 += src[k * stride];
  }
  part[t] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q != 0 || e >= 33) return;
  double v = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) v += part[64 * k + e];
  if (e < 21) {
    int a = 0;
    while (sym6(a + 1, a + 1) <= e) ++a;   // row of the upper-triangle entry
    const int c = a + (e - sym6(a, a));
    const int r0 = 6 * blk + a, c0 = 6 * blk + c;
    p.S[(size_t)r0 * p.ldS + c0] += v;
    if (a != c) p.S[(size_t)c0 * p.ldS + r0] += v;
    else p.hC[r0] += v;
  } else if (e < 27) {
    const int r0 = 6 * blk + (e - 21);
    p.gFull[r0] += v; p.gRed[r0] += v;
  } else {
    p.gRed[6 * blk + (e - 27)] -= v;   // gRed = Jp^T r - sum_l E_la c_l
  }
}

typedef double d16_t __attribute__((ext_vector_type(16)));
// EIGHT pair words of one block row per statement.  Words 2 j / 2 j + 1 share their A record (the host pads every run of an A
// record to an even length and a row's words to whole eights), so a quad is six operand reads and four products, each into the
// accumulator its word names: the VGPR INDEX MODE reaches the C / D operands of v_mfma_f64_4x4x4_4b_f64 (tools/ubench/
// mfma_gpridx.hip: results right, also for back-to-back products into one accumulator with five wait states between them; 4.2
// cycles per pair and CU with 16 waves, against 6.2 for the indexed moves the compiler puts around the product).  The accumulators
// of a row are pinned to the registers the text names; the index mode writes M0, which the compiler uses for the LDS-DMA base:
// saved and restored.  Words 2 j / 2 j + 1 never meet in an accumulator (one landmark, two poses; a padding word names the next
// one), words 2 j + 1 / 2 j + 2 may: four wait states between their products.  The index of the next product is set behind the
// current one (s_set_gpr_idx_idx; a product has its registers when it issues) and a wait state before it is used.
// The reads run one quad AHEAD of the products: the statement first requests words 4..7 (operand set B), multiplies words 0..3
// (set A, requested by the previous statement -- or by SVIN_ROWS_FIRST), requests the NEXT statement's words 0..3 into set A,
// multiplies words 4..7.  Set A is in flight across the loop's back edge: between two statements the loop only computes scalar
// offsets, and nothing may touch the twelve registers of set A (tools/dbg/check_rows_asm.py looks at the disassembly).  lgkmcnt
// counts the reads in order (a scalar load of the compiler's in between only makes a wait longer): a quad's products start when
// at most the six reads of the quad behind it are in flight.  One address register: a ds_read has taken its address when the
// next instruction issues.  (The kernel is bound by instruction issue, not by the LDS or the matrix pipe -- 155-173 cycles per
// word and wave with ~14 instructions per word, whether the reads ran ahead or not: the word format and this statement are what
// nine instructions per word look like.)
#define SVIN_ROWS_RD(DST, OFF, LANE) "v_add_u32 %[t], " OFF ", " LANE "\n\tds_read_b64 " DST ", %[t]\n\t"
#define SVIN_ROWS_QUAD(FIRST, I0, I1, I2, I3, A0, B0, B1, A1, B2, B3)                                                        \
  "s_waitcnt lgkmcnt(6)\n\ts_set_gpr_idx_on " I0 ", 0xc\n\t"                                                                 \
  "v_mfma_f64_4x4x4_4b_f64 " FIRST ", " A0 ", " B0 ", " FIRST "\n\ts_set_gpr_idx_idx " I1 "\n\ts_nop 0\n\t"                   \
  "v_mfma_f64_4x4x4_4b_f64 " FIRST ", " A0 ", " B1 ", " FIRST "\n\ts_set_gpr_idx_idx " I2 "\n\ts_nop 2\n\t"                   \
  "v_mfma_f64_4x4x4_4b_f64 " FIRST ", " A1 ", " B2 ", " FIRST "\n\ts_set_gpr_idx_idx " I3 "\n\ts_nop 0\n\t"                   \
  "v_mfma_f64_4x4x4_4b_f64 " FIRST ", " A1 ", " B3 ", " FIRST "\n\ts_set_gpr_idx_off\n\t"
#define SVIN_ROWS_FIRST(W)                                                                                                   \
  asm volatile(SVIN_ROWS_RD("%[a0]", "%[sa0]", "%[lA]") SVIN_ROWS_RD("%[b0]", "%[sb0]", "%[lB]") SVIN_ROWS_RD("%[b1]", "%[sb1]", "%[lB]") \
               SVIN_ROWS_RD("%[a1]", "%[sa1]", "%[lA]") SVIN_ROWS_RD("%[b2]", "%[sb2]", "%[lB]") SVIN_ROWS_RD("%[b3]", "%[sb3]", "%[lB]") \
               : [a0] "=&v"(opA[0]), [b0] "=&v"(opA[1]), [b1] "=&v"(opA[2]), [a1] "=&v"(opA[3]), [b2] "=&v"(opA[4]), [b3] "=&v"(opA[5]), [t] "=&v"(adr) \
               : [sa0] "s"(offA(W[0])), [sb0] "s"(offB(W[0])), [sb1] "s"(offB(W[1])), [sa1] "s"(offA(W[2])),                 \
                 [sb2] "s"(offB(W[2])), [sb3] "s"(offB(W[3])), [lA] "v"(oA), [lB] "v"(oB))
#define SVIN_ROWS_EIGHT(ACC, TUPLE, FIRST, WA, WB, WN)                                                                       \
  asm volatile(SVIN_ROWS_RD("%[xa0]", "%[sxa0]", "%[lA]") SVIN_ROWS_RD("%[xb0]", "%[sxb0]", "%[lB]") SVIN_ROWS_RD("%[xb1]", "%[sxb1]", "%[lB]") \
               SVIN_ROWS_RD("%[xa1]", "%[sxa1]", "%[lA]") SVIN_ROWS_RD("%[xb2]", "%[sxb2]", "%[lB]") SVIN_ROWS_RD("%[xb3]", "%[sxb3]", "%[lB]") \
               "s_mov_b32 %[ms], m0\n\t"                                                                                     \
               SVIN_ROWS_QUAD(FIRST, "%[i0]", "%[i1]", "%[i2]", "%[i3]", "%[a0]", "%[b0]", "%[b1]", "%[a1]", "%[b2]", "%[b3]") \
               SVIN_ROWS_RD("%[a0]", "%[sa0]", "%[lA]") SVIN_ROWS_RD("%[b0]", "%[sb0]", "%[lB]") SVIN_ROWS_RD("%[b1]", "%[sb1]", "%[lB]") \
               SVIN_ROWS_RD("%[a1]", "%[sa1]", "%[lA]") SVIN_ROWS_RD("%[b2]", "%[sb2]", "%[lB]") SVIN_ROWS_RD("%[b3]", "%[sb3]", "%[lB]") \
               SVIN_ROWS_QUAD(FIRST, "%[i4]", "%[i5]", "%[i6]", "%[i7]", "%[xa0]", "%[xb0]", "%[xb1]", "%[xa1]", "%[xb2]", "%[xb3]") \
               "s_mov_b32 m0, %[ms]\n\t"                                                                                     \
               "s_nop 2"                                                                                                     \
               : "+{" TUPLE "}"(ACC), [ms] "=&s"(m0Save), [t] "=&v"(adr), [a0] "+v"(opA[0]), [b0] "+v"(opA[1]), [b1] "+v"(opA[2]), \
                 [a1] "+v"(opA[3]), [b2] "+v"(opA[4]), [b3] "+v"(opA[5]), [xa0] "=&v"(opB[0]), [xb0] "=&v"(opB[1]),           \
                 [xb1] "=&v"(opB[2]), [xa1] "=&v"(opB[3]), [xb2] "=&v"(opB[4]), [xb3] "=&v"(opB[5])                           \
               : [sxa0] "s"(offA(WB[0])), [sxb0] "s"(offB(WB[0])), [sxb1] "s"(offB(WB[1])), [sxa1] "s"(offA(WB[2])),         \
                 [sxb2] "s"(offB(WB[2])), [sxb3] "s"(offB(WB[3])), [sa0] "s"(offA(WN[0])), [sb0] "s"(offB(WN[0])),           \
                 [sb1] "s"(offB(WN[1])), [sa1] "s"(offA(WN[2])), [sb2] "s"(offB(WN[2])), [sb3] "s"(offB(WN[3])),             \
                 [i0] "s"(WA[0]), [i1] "s"(WA[1]), [i2] "s"(WA[2]), [i3] "s"(WA[3]),                                         \
                 [i4] "s"(WB[0]), [i5] "s"(WB[1]), [i6] "s"(WB[2]), [i7] "s"(WB[3]), [lA] "v"(oA), [lB] "v"(oB))
// one block row's words q0 .. q1 (whole eights)
#define SVIN_ROWS_ROW(ACC, TUPLE, FIRST, Q0, Q1)                                                                             \
  do {                                                                                                                       \
    const int q0_ = (Q0), q1_ = (Q1);                                                                                        \
    if (q0_ < q1_) {                                                                                                         \
      unsigned wa[4], wb[4], wn[4];                                                                                          \
      fetchWords(q0_, q1_, wa);                                                                                              \
      SVIN_ROWS_FIRST(wa);                                                                                                   \
      for (int q = q0_; q < q1_; q += 8) {                                                                                   \
        fetchWords(q + 4, q1_, wb);                                                                                          \
        fetchWords(q + 8, q1_, wn);                                                                                          \
        SVIN_ROWS_EIGHT(ACC, TUPLE, FIRST, wa, wb, wn);                                                                      \
        wa[0] = wn[0]; wa[1] = wn[1]; wa[2] = wn[2]; wa[3] = wn[3];                                                          \
      }                                                                                                                      \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(opA[0]), "+v"(opA[1]), "+v"(opA[2]), "+v"(opA[3]), "+v"(opA[4]), "+v"(opA[5])); /* (the reads past the row's end: of the zero record) */ \
    }                                                                                                                        \
  } while (0)

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_schur_rows(DeviceProblem p) {
  extern __shared__ double smem[];   // two record buffers of kBlkBatchRecs x kBlkStride doubles; at the end the 96 x 96 image of the slab
  const int t = threadIdx.x, b = blockIdx.x, wave = t >> 6, lane = t & 63;
  const int4 work = p.panelWork[b];  // x = I, y = J, z = first batch, w = number of batches
  const int pI = work.x, pJ = work.y, nb = work.w;
  const bool diag = pI == pJ;
  constexpr int nPB = kBlkPanelPoses;
  constexpr int kBuf = kBlkBatchRecs * kBlkStride;   // doubles per buffer
  // the two block rows this wave owns (255: none)
  int row0, row1;
  {
    const int4 own = p.blkOwn[b];
    const unsigned words[4] = {(unsigned)own.x, (unsigned)own.y, (unsigned)own.z, (unsigned)own.w};
    const unsigned wd = words[wave >> 1] >> (16 * (wave & 1));
    row0 = (int)(wd & 0xffu); row1 = (int)((wd >> 8) & 0xffu);
  }
  // (the zero pair behind every record and the last record of either buffer are never written again)
  for (int i = t; i < kBuf; i += 64 * NW) reinterpret_cast<double2*>(smem)[i] = double2{0.0, 0.0};
  // operand lanes (A: 16 k + 4 b + i holds row 4 (b >> 1) + i of the slot in I; B: 16 k + 4 b + j holds row 4 (b & 1) + j of the
  // slot in J, column k of E; k = 3 is the padding of K and reads the zero pair; rows 6 and 7 read what follows the column --
  // finite numbers that only reach rows / columns 6 and 7 of the result, which nobody stores) and result lanes (16 i + 4 b + j)
  const int kk = lane >> 4, bq = (lane >> 2) & 3, ij = lane & 3;
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) double*)smem;
  const unsigned laneA = ldsBase + 8u * (unsigned)(kk < 3 ? 6 * kk + 4 * (bq >> 1) + ij : kBlkRec);
  const unsigned laneB = ldsBase + 8u * (unsigned)(kk < 3 ? 6 * kk + 4 * (bq & 1) + ij : kBlkRec);
  const int dRow = 4 * (bq >> 1) + kk, dCol = 4 * (bq & 1) + ij;
  const bool dValid = dRow < 6 && dCol < 6;
  const int dOff = dRow * 6 + dCol;
  // The accumulators: one 6 x 6 block (in its 8 x 8 padding: one double per lane) per pose block of panel J, for either owned row
  d16_t acc0, acc1;
#pragma unroll
  for (int k = 0; k < 16; ++k) { acc0[k] = 0.0; acc1[k] = 0.0; }
  const int wv = __builtin_amdgcn_readfirstlane(wave);
#ifdef SVIN_BLOCKS_TIMING
  long long bT[4] = {0, 0, 0, 0}, bq0 = __builtin_readcyclecounter(), bq1;
  int bPairs = 0;
#define BNT(i) do { bq1 = __builtin_readcyclecounter(); bT[i] += bq1 - bq0; bq0 = bq1; } while (0)
#else
#define BNT(i) do { } while (0)
#endif
  // Records travel global memory -> LDS on the DMA path (global_load_lds_dwordx4: no registers, no ds_write pass), nine 16-byte
  // pieces per record into ten slots of the buffer: thread t of trip j owns slot 512 j + t of the buffer, i.e. piece (512 j + t)
  // % 10 of record (512 j + t) / 10, and stays idle for piece 9.  Two buffers: batch i + 1 streams in while the pairs of batch
  // i are worked through; the slot numbers of a batch are fetched one batch earlier, its descriptor two batches earlier, its
  // pair words one batch earlier -- no dependent global round trip inside the loop.
  constexpr int kTrips = ((kBlkBatchRecs - 1) * kBlkStrideChunks + 64 * NW - 1) / (64 * NW);
  // (the host pads both tables with three empty batches: the loads below run past the workgroup's last batch unconditionally --
  // a conditional load merges with a default value, and the copy behind that merge waits for the load where it is issued)
  auto loadBatch = [&](int bi) __attribute__((always_inline)) -> int2 { return p.blkBatch[work.z + bi]; };
  auto loadTab = [&](int bi) __attribute__((always_inline)) -> int4 { return p.blkWaveTab[(work.z + bi) * NW + wv]; };
  int sl[kTrips];   // slot of the record this thread fetches a piece of in trip j of the next batch to be requested; -1: none
  // (thread number behind an empty asm: what is derived from it is recomputed where it is used -- hoisted out of the batch loop
  // it was ten more registers than the kernel has, and a spilled address is a scratch load in front of every request)
  auto loadSlots = [&](int2 bd) __attribute__((always_inline)) {
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int j = 0; j < kTrips; ++j) {
      const int c = tt + 64 * NW * j, rec = c / kBlkStrideChunks, part = c - rec * kBlkStrideChunks;
      sl[j] = (rec < bd.y && part < kBlkRecChunks) ? p.blkRecSlot[bd.x + rec] : -1;
    }
  };
  auto requestRecords = [&](int buf) __attribute__((always_inline)) {
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int j = 0; j < kTrips; ++j) {
      const int c = tt + 64 * NW * j, rec = c / kBlkStrideChunks, part = c - rec * kBlkStrideChunks;
      if (sl[j] >= 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.slotRec + (size_t)sl[j] * kBlkRec + 2 * part),
                                         (__attribute__((address_space(3))) void*)(smem + (size_t)buf * kBuf + 2 * (64 * NW * j + 64 * wv)), 16, 0, 0);
    }
  };
  auto loadWords = [&](const int4& wt, unsigned& W0, unsigned& W1) __attribute__((always_inline)) {
    W0 = 0u; W1 = 0u;
    if (lane < wt.y + wt.z) W0 = p.blkPairs[wt.x + lane];
    if (lane + 64 < wt.y + wt.z) W1 = p.blkPairs[wt.x + 64 + lane];
  };
  int2 bdA = loadBatch(2);
  int4 wtC = loadTab(0), wtN = loadTab(1);
  loadSlots(loadBatch(0));
  unsigned Wn0, Wn1;
  loadWords(wtC, Wn0, Wn1);
  __syncthreads();   // (the buffers are cleared)
  requestRecords(0);
  loadSlots(loadBatch(1));
  for (int bi = 0; bi < nb; ++bi) {
    // Everything requested so far has arrived (the prefetched values are operands of the statement: the compiler waits for them
    // here, knows them complete from here on, and puts no wait of its own behind the requests below -- such a wait would cover
    // the records just requested as well), and every wave is done with the pairs of batch bi - 1 (the other buffer).
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2]), "+v"(sl[3]), "+v"(sl[4]), "+v"(Wn0), "+v"(Wn1), "+v"(bdA.x), "+v"(bdA.y), "+v"(wtN.x),
                   "+v"(wtN.y), "+v"(wtN.z)
                 :
                 : "memory");
    static_assert(kTrips == 5, "the operand list above names the five trips");
    __syncthreads();
    BNT(0);
    const unsigned Wc0 = Wn0, Wc1 = Wn1;
    const int n0 = __builtin_amdgcn_readfirstlane(wtC.y), n1 = __builtin_amdgcn_readfirstlane(wtC.z);
    if (bi + 1 < nb) requestRecords((bi + 1) & 1);
    loadSlots(bdA);                 // batch bi + 2
    loadWords(wtN, Wn0, Wn1);       // batch bi + 1
    wtC = wtN;
    bdA = loadBatch(bi + 3); wtN = loadTab(bi + 2);
    BNT(1);
#ifdef SVIN_BLOCKS_TIMING
    bPairs += n0 + n1;
#endif
    const unsigned oA = laneA + (unsigned)((bi & 1) * kBuf * 8), oB = laneB + (unsigned)((bi & 1) * kBuf * 8);
    double opA[6], opB[6];   // (the operands of two quads)
    unsigned adr, m0Save;    // (the address register of the reads)
    // four words from q on as scalars; past the row's end a word of the zero record
    auto fetchWords = [&](int q, int qEnd, unsigned (&w)[4]) __attribute__((always_inline)) {
      const unsigned wsrc = q < 64 ? Wc0 : Wc1;   // (wave-uniform)
      constexpr unsigned kZero = ((unsigned)(kBlkBatchRecs - 1) << 24) | ((unsigned)((kBlkBatchRecs - 1) * kBlkStride * 8) << 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = q < qEnd ? (unsigned)__builtin_amdgcn_readlane((int)wsrc, (q & 63) + u) : (kZero | (unsigned)(2 * (u & 1)));
    };
    // pair word (host: Window::pack): twice the accumulator's number in bits 0-7 (what the index mode takes from a scalar register: its
    // low byte), the B record's byte offset in bits 8-23, the A record's number in bits 24-31
    auto offA = [](unsigned w) __attribute__((always_inline)) -> unsigned { return (w >> 24) * (unsigned)(kBlkStride * 8); };
    auto offB = [](unsigned w) __attribute__((always_inline)) -> unsigned { return (w >> 8) & 0xffffu; };
    SVIN_ROWS_ROW(acc0, "v[64:95]", "v[64:65]", 0, n0);
    SVIN_ROWS_ROW(acc1, "v[96:127]", "v[96:97]", n0, n0 + n1);
    BNT(2);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // (the last products have written their accumulators)
#ifdef SVIN_BLOCKS_TIMING
  if ((b == 3 || b == 200 || b == 500) && lane == 0 && (wave == 0 || wave == 5))
    printf("[rows block %d (%d, %d) wave %d: %d batches, %d pair words] wait + barrier %lld  requests %lld  pairs %lld\n", b, pI, pJ, wave,
           nb, bPairs, bT[0], bT[1], bT[2]);
#endif
#undef BNT
  __syncthreads();
  // ---- the accumulators into an image of the pair's 16 x 16 blocks (36 doubles each), then the slab
  double* img = smem;
  for (int i = t; i < nPB * nPB * 36 / 2; i += 64 * NW) reinterpret_cast<double2*>(smem)[i] = double2{0.0, 0.0};
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // (a heavy block row may have a second accumulator set on another wave -- Window::pack --: the image is zero, the adds of the
    //  two sets commute)
    if (row0 < nPB && dValid) atomicAdd(&img[(row0 * nPB + k) * 36 + dOff], acc0[k]);
    if (row1 < nPB && dValid) atomicAdd(&img[(row1 * nPB + k) * 36 + dOff], acc1[k]);
  }
  __syncthreads();
  double* slab = p.slabs + (size_t)b * kPanelSlab;
  for (int e = t; e < kPanelRows * kPanelRows; e += 64 * NW) {
    const int r = e / kPanelRows, c = e - r * kPanelRows;
    const int pa = r / 6, ra = r - 6 * pa, pb = c / 6, cb = c - 6 * pb;
    double v;
    if (diag && pa < pb) v = img[(pb * nPB + pa) * 36 + cb * 6 + ra];   // (mirror of the block below the diagonal)
    else v = img[(pa * nPB + pb) * 36 + ra * 6 + cb];
    slab[e] = -v;   // S = A - G G^T: the blocks were accumulated with a plus sign
  }
  if (t < kPanelRows) {   // (the vectors -- gRed, gFull, hC -- and the blocks of A come from k_blocks_pose_reduce)
    double* v = slab + kPanelRows * kPanelRows;
    v[t] = 0.0; v[kPanelRows + t] = 0.0; v[2 * kPanelRows + t] = 0.0;
  }
}
#undef SVIN_ROWS_ROW
#undef SVIN_ROWS_EIGHT
#undef SVIN_ROWS_FIRST
#undef SVIN_ROWS_QUAD
#undef SVIN_ROWS_RD


// sums the slabs of every panel pair (fixed order) into S (both triangles) and, for diagonal pairs, the vectors
__global__ __launch_bounds__(256) void k_reduce_panel_slabs(DeviceProblem p) {
  // 16 entries x 16 partitions of the pair's slabs per workgroup, like k_reduce_slabs (a diagonal pair of a wide window has
  // 100-200 slabs: one thread per entry walking all of them was 50 dependent rounds of loads, 35 us)
  __shared__ double part[256];
  const int pair = blockIdx.y;
  const int e16 = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + e16;
  const int b0 = p.panelPairPtr[pair], b1 = p.panelPairPtr[pair + 1];
  if (b0 == b1) return;
  const int pI = p.panelWork[b0].x, pJ = p.panelWork[b0].y;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (e < kPanelSlab) {
    const int per = (b1 - b0 + 15) / 16;
    int k = b0 + q * per;
    const int k1 = min(b1, k + per);
    for (; k + 3 < k1; k += 4) {
      s0 += p.slabs[(size_t)k * kPanelSlab + e];
      s1 += p.slabs[(size_t)(k + 1) * kPanelSlab + e];
      s2 += p.slabs[(size_t)(k + 2) * kPanelSlab + e];
      s3 += p.slabs[(size_t)(k + 3) * kPanelSlab + e];
    }
    for (; k < k1; ++k) s0 += p.slabs[(size_t)k * kPanelSlab + e];
  }
  part[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q != 0 || e >= kPanelSlab) return;
  double s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += part[16 * k + e16];
  if (e < kPanelRows * kPanelRows) {
    const int r = kPanelRows * pI + e / kPanelRows, c = kPanelRows * pJ + e % kPanelRows;
    // diagonal pairs carry their lower tiles only (16 x 16 tiles, the diagonal tiles full)
    const int ti = (e / kPanelRows) >> 4, tj = (e % kPanelRows) >> 4;
    if (r < p.dC && c < p.dC && (pI != pJ || ti >= tj)) {
      p.S[(size_t)r * p.ldS + c] += s;
      if (pI != pJ || ti > tj) p.S[(size_t)c * p.ldS + r] += s;
    }
  } else if (pI == pJ) {
    const int v = e - kPanelRows * kPanelRows, which = v / kPanelRows, r = kPanelRows * pI + v % kPanelRows;
    if (r < p.dC) {
      if (which == 0) p.gRed[r] += s;
      else if (which == 1) p.gFull[r] += s;
      else p.hC[r] += s;
    }
  }
}

// S += reduce(slabs) (block-upper data mirrored), vectors += reduce(slab vectors)
// 16 entries x 16 slab-partitions per 256-thread block; fixed summation order -> deterministic
constexpr int kSlabParts = 16;
__device__ __forceinline__ void k_reduce_slabs_body(const DeviceProblem& p) {
  __shared__ double part[256];
  SVIN_ARGS(SA(p.slabs), SA(p.S), SA(p.gRed), SA(p.gFull), SA(p.hC), SA(p.nSlabs), SA(p.dC), SA(p.ldS));
  const int dC = p.dC;
  const size_t slabSize = (size_t)dC * dC + 3 * dC;
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + e;
  const int total = dC * dC + 3 * dC;
  size_t src = 0;
  int rr = 0, cc = 0;
  bool work = false;
  if (idx < dC * dC) {
    // the slabs hold the upper block triangle (6x6 blocks, diagonal blocks full): only those entries are read --
    // row-wise, coalesced -- and mirrored into the lower triangle on the way out
    rr = idx / dC; cc = idx % dC;
    work = (rr / 6) <= (cc / 6);
    src = (size_t)rr * dC + cc;
  } else if (idx < total) {
    work = true;
    src = (size_t)idx;
  }
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (work) {
    const int per = (p.nSlabs + kSlabParts - 1) / kSlabParts;
    const int k0 = q * per, k1 = min(p.nSlabs, k0 + per);
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s0 += p.slabs[(size_t)k * slabSize + src];
      s1 += p.slabs[(size_t)(k + 1) * slabSize + src];
      s2 += p.slabs[(size_t)(k + 2) * slabSize + src];
      s3 += p.slabs[(size_t)(k + 3) * slabSize + src];
    }
    for (; k < k1; ++k) s0 += p.slabs[(size_t)k * slabSize + src];
  }
  part[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && work) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < kSlabParts; ++k) s += part[16 * k + e];
    if (idx < dC * dC) {
      p.S[(size_t)rr * p.ldS + cc] += s;
      if ((rr / 6) < (cc / 6)) p.S[(size_t)cc * p.ldS + rr] += s;
    } else {
      const int v = idx - dC * dC;
      if (v < dC) p.gRed[v] += s;
      else if (v < 2 * dC) p.gFull[v - dC] += s;
      else p.hC[v - 2 * dC] += s;
    }
  }
}
__global__ __launch_bounds__(256) void k_reduce_slabs(DeviceProblem p) { k_reduce_slabs_body(p); }
// (batched form: blockIdx.y = the window of the batch, its problem and trust-region scalars from the slot table)
__global__ __launch_bounds__(256) void k_reduce_slabs_batch(const BatchSlot* __restrict__ slots) {
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchFull)) return;
  k_reduce_slabs_body(sl.p);
}

// camera-column metric (Jacobi scaling fixed at iteration 0): returns the damping mu*htil for row i
__device__ __forceinline__ double finalizeRow(const DeviceProblem& p, int i, double mu, int initScale) {
  double sc;
  if (initScale) { sc = 1.0 / (1.0 + sqrt(p.hC[i])); p.scaleC[i] = sc; }
  else sc = p.scaleC[i];
  const double ht = fmin(fmax(p.hC[i] * sc * sc, 1e-6), 1e32) / (sc * sc);
  p.htilC[i] = ht;
  return mu * ht;
}
// stand-alone version for the inspection hooks (the trust-region loop fuses this into the solver's load)
__global__ void k_finalize_diag(DeviceProblem p, double mu, int initScale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.d) return;
  p.S[(size_t)i * p.ldS + i] += finalizeRow(p, i, mu, initScale);
}

__global__ void k_zero_build(DeviceProblem p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int d = p.d;
  if (i < p.ldS * d) p.S[i] = 0.0;
  if (i < d) { p.gRed[i] = 0.0; p.gFull[i] = 0.0; p.hC[i] = 0.0; }
  if (i == 0) p.scal->cholFail = 0;
}

void launchBuildNormalEquations(const DeviceProblem& p, double mu, bool initScale, hipStream_t s) {
  launchAccumulateNormalEquations(p, mu, initScale, s);
  launchFinalizeNormalEquations(p, mu, initScale, s);
}
void launchFinalizeNormalEquations(const DeviceProblem& p, double mu, bool initScale, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize_diag, dim3((p.d + 255) / 256), dim3(256), 0, s, p, mu, initScale ? 1 : 0);
}
__global__ __launch_bounds__(256) void k_factors_only(DeviceProblem p, int nFacBlocks) {
  __shared__ int colRow[30];
  if ((int)blockIdx.x < nFacBlocks) factorsAccumulate(p, blockIdx.x, colRow);
  else priorAccumulateBlock(p, blockIdx.x - nFacBlocks);
}
// one workgroup per (segment, 16 KB slice): 16-byte copies from the staged block to the arrays' own allocations
__global__ __launch_bounds__(256) void k_scatter_staged(const unsigned char* block, int nSeg) {
  const StageSegment* segs = reinterpret_cast<const StageSegment*>(block);
  for (int sIdx = blockIdx.y; sIdx < nSeg; sIdx += gridDim.y) {
    const StageSegment sg = segs[sIdx];
    uint4* dst = reinterpret_cast<uint4*>(sg.dst);
    const size_t n16 = sg.bytes / 16;
    if (sg.srcOff == kStageClear) {   // a clear riding along (no bytes in the block): saves a fill launch per region
      for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = uint4{0u, 0u, 0u, 0u};
      continue;
    }
    const uint4* src = reinterpret_cast<const uint4*>(block + sg.srcOff);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  }
}
void launchScatterStaged(const void* block, int nSeg, hipStream_t s) {
  if (nSeg <= 0) return;
  hipLaunchKernelGGL(k_scatter_staged, dim3(32, std::min(nSeg, 64)), dim3(256), 0, s, reinterpret_cast<const unsigned char*>(block), nSeg);
}
__global__ __launch_bounds__(256) void k_gather_staged(unsigned char* block, GatherArgs a) {
  const int sIdx = blockIdx.y;
  if (sIdx >= a.n) return;
  const uint4* src = reinterpret_cast<const uint4*>(a.src[sIdx]);
  uint4* dst = reinterpret_cast<uint4*>(block + a.off[sIdx]);
  const size_t n16 = a.bytes[sIdx] / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launchGatherStaged(void* block, const GatherArgs& a, hipStream_t s) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_gather_staged, dim3(16, a.n), dim3(256), 0, s, reinterpret_cast<unsigned char*>(block), a);
}
void launchZeroBuild(const DeviceProblem& p, hipStream_t s) {
  hipLaunchKernelGGL(k_zero_build, dim3((p.ldS * p.d + 255) / 256), dim3(256), 0, s, p);
}
// zeroFirst = false: the accumulators are already clear (pack() clears them, and k_post_solve clears them again
// for the next linearisation of the trust-region loop)
// the dense Gram-matrix form (k_schur_dense): variant, LDS size and DeviceProblem::aBlocks for a window's geometry
struct DenseSchurPlan { int nTr; bool aMfma, aBlocks; size_t ldsBytes; };
static DenseSchurPlan denseSchurPlan(const DeviceProblem& p) {
  const int dC = p.dC;
  DenseSchurPlan q;
  q.nTr = (dC + 2 + 15) / 16;
  const int rows = 16 * q.nTr;
  q.aMfma = p.anyExtVariable || q.nTr > 8;
  // A on the MFMA path needs a fill / barrier / clear round per batch of observations; when the blocks of A fit LDS
  // next to G they are accumulated directly (LDS atomics) and merged into the accumulator tiles once per chunk
  const int nPB = p.dCPose / 6, nEB = dC / 6 - nPB;
  const size_t blocksExtra = (size_t)(dC / 6) * kPoseAcc + (size_t)nEB * nPB * 36;
  const bool forceU = optOn(kOptSchurAMfma);
  q.aBlocks = q.aMfma && !forceU && ((size_t)rows * kDenseLd + blocksExtra) * 8 <= 150 * 1024;
  const size_t extra = q.aBlocks ? blocksExtra : (q.aMfma ? (size_t)rows * (2 * denseObsBatch(rows) + 1) : (size_t)4 * (dC / 6) * kPoseAcc);
  q.ldsBytes = ((size_t)rows * kDenseLd + extra) * 8;
  return q;
}
int schurDenseABlocks(const DeviceProblem& p) { return (p.L > 0 && p.N > 0 && p.dC > 0 && p.schurDense && denseSchurPlan(p).aBlocks) ? 1 : 0; }
static bool sbEarlyActive(const DeviceProblem& p);                                          // (behind the chain kernels, below)
static void launchSbEarly(const DeviceProblem& p, hipStream_t side, double mu, bool initScale);
void launchAccumulateNormalEquations(const DeviceProblem& p, double mu, bool initScale, hipStream_t s, bool zeroFirst) {
  const int dC = p.dC;
  if (zeroFirst) launchZeroBuild(p, s);
  const int nFac = p.F;   // this rank's factors
  const int nPri = priorAccBlocks(p);  // the prior rides along as extra blocks of the same launch
  if (p.L > 0 && p.N > 0 && dC > 0 && p.schurDense) {
    const DenseSchurPlan plan = denseSchurPlan(p);
    const int nTr = plan.nTr;
    const bool aMfma = plan.aMfma;
    const size_t ldsBytes = plan.ldsBytes;
    const dim3 grid(p.nSlabs + nFac + nPri);
    DeviceProblem pb = p;
    pb.aBlocks = plan.aBlocks ? 1 : 0;
#define LAUNCH(MAXT, E, NWV)                                                                                        \
  do {                                                                                                              \
    ensureDynamicLds((const void*)k_schur_dense<MAXT, E, NWV>, ldsBytes); \
    hipLaunchKernelGGL((k_schur_dense<MAXT, E, NWV>), grid, dim3(64 * NWV), ldsBytes, s, pb, mu, initScale ? 1 : 0, p.nSlabs, \
                       nFac);                                                                                       \
  } while (0)
    if (nTr > 12) LAUNCH(17, true, 8);        // <= 16 tile rows: 136 tiles over 8 waves
    else if (nTr > 8) LAUNCH(10, true, 8);    // <= 12 tile rows: 78 tiles
    else if (aMfma) LAUNCH(9, true, 4);
    else LAUNCH(9, false, 4);
#undef LAUNCH
  } else if (p.L > 0 && p.N > 0 && dC > 0 && p.schurPanels) {
    // off-diagonal pairs: two G tiles + c; diagonal pairs: one G tile + the per-wave pose blocks + c (smaller)
    const size_t ldsBytes = ((size_t)2 * kPanelRows * kDenseLd + kDenseK) * 8;
    static_assert(2 * kPanelRows * kDenseLd >= kPanelRows * kDenseLd + 4 * (kPanelRows / 6) * kPoseAcc, "diagonal pairs fit the same allocation");
    // Round 4 (config #4, build of the normal equations 1.01 ms -> 0.67 ms, A/B in one gpurun call): two workgroups per CU instead
    // of one (0.79: 86 spilled registers with the next chunk's first observation prefetched, 0.73 without the prefetch and without
    // spills), per-landmark quantities from k_panels_landmarks instead of once per panel pair (0.67).
    // (Measured for the DENSE form too, configs[2]: factors and chain on the side stream beside k_schur_dense<10, true, 8> -- 0.209 ms
    //  per iteration against 0.191.  A launch's workgroups are dealt round robin over the eight XCDs; 250 chunk workgroups fill two
    //  XCDs completely (32 CUs each, one workgroup of 512 threads x 256 registers per CU), and the side kernels' workgroups dealt
    //  to those XCDs wait for the chunk kernel to end -- k_factors_only then ends 10 us AFTER k_schur_dense instead of inside it.
    //  Concurrency across streams needs free room in EVERY XCD; the wide form below gains only what no longer sits between the
    //  slab sum and the solve.)
    // Round 6: the block-pair form (k_blocks_slots + k_schur_rows) is the default; pack() decides (DeviceProblem::schurBlocks,
    // its slot tables) -- SVIN_PANELS_OLD=1 at pack() time keeps the round-4 / 5 tile form, whose work list holds fewer chunks per
    // workgroup.
    // p.sideLane (one GPU, wide window; DeviceProblem): the small factors -- and behind them the factorisation and the forward
    // substitution of the speed / bias chain, which read nothing the landmarks contribute to (S_ss, S_sk and g_s come from the IMU
    // factors and the prior alone) -- run on a side stream beside the landmark elimination: 22 + 35 + 29 us off the iteration's
    // critical path.  The pose blocks of S receive the factors' atomic adds AND the plain read-modify-writes of
    // k_blocks_pose_reduce: the latter waits for the former (event `mid`).
    SideLane* lane = nullptr;
    const bool early = sbEarlyActive(p);
    if (early) {
      lane = &sideLaneOf(s);
      HIP_LAUNCH_OK(hipEventRecord(lane->fork, s));
      HIP_LAUNCH_OK(hipStreamWaitEvent(lane->side, lane->fork, 0));
      hipLaunchKernelGGL(k_factors_only, dim3(nFac + nPri), dim3(256), 0, lane->side, p, nFac);
      HIP_LAUNCH_OK(hipEventRecord(lane->mid, lane->side));
      launchSbEarly(p, lane->side, mu, initScale);
      HIP_LAUNCH_OK(hipEventRecord(lane->join, lane->side));
    }
    if (p.schurBlocks) {
      constexpr int NW = kBlkWaves;
      const size_t ldsBlk = (size_t)std::max(2 * kBlkBatchRecs * kBlkStride, kBlkPanelPoses * kBlkPanelPoses * 36) * 8;   // (two record buffers / slab image)
      // the slot pass keeps up to four copies of its per-pose accumulators (one per 16-lane group of a wave: the four landmarks a
      // wave works on see the same poses, and four lanes adding to one address serialise) -- as many as 64 KB hold
      int nCopies = 4;
#ifndef SVIN_SLOT_COPIES_KB
#define SVIN_SLOT_COPIES_KB 76
#endif
      while (nCopies > 1 && (size_t)nCopies * (dC / 6) * kBlkPoseLd * 8 > SVIN_SLOT_COPIES_KB * 1024) nCopies >>= 1;
      const size_t ldsLm = (size_t)nCopies * (dC / 6) * kBlkPoseLd * 8;
      const int nSlotBlocks = (p.nSlots + kBlkSlotsPerWorkgroup - 1) / kBlkSlotsPerWorkgroup;
      hipLaunchKernelGGL(k_panels_landmarks, dim3((p.L + 15) / 16), dim3(256), 0, s, p, mu, initScale ? 1 : 0);
      ensureDynamicLds((const void*)k_blocks_slots, ldsLm);
      if (nSlotBlocks > 0) hipLaunchKernelGGL(k_blocks_slots, dim3(nSlotBlocks), dim3(256), ldsLm, s, p, nCopies);
      ensureDynamicLds((const void*)k_schur_rows<NW>, ldsBlk);
      if (p.nPanelBlocks > 0) hipLaunchKernelGGL((k_schur_rows<NW>), dim3(p.nPanelBlocks), dim3(64 * NW), ldsBlk, s, p);
      if (early) HIP_LAUNCH_OK(hipStreamWaitEvent(s, lane->mid, 0));
      if (nSlotBlocks > 0) hipLaunchKernelGGL(k_blocks_pose_reduce, dim3(dC / 6), dim3(1024), 0, s, p, nSlotBlocks);
    } else {
      hipLaunchKernelGGL(k_panels_landmarks, dim3((p.L + 15) / 16), dim3(256), 0, s, p, mu, initScale ? 1 : 0);
      ensureDynamicLds((const void*)k_schur_panels<2, false, true>, ldsBytes);
      hipLaunchKernelGGL((k_schur_panels<2, false, true>), dim3(p.nPanelBlocks), dim3(256), ldsBytes, s, p, mu, initScale ? 1 : 0, p.nPanelBlocks, nFac);
    }
    if (!early && nFac + nPri > 0) hipLaunchKernelGGL(k_factors_only, dim3(nFac + nPri), dim3(256), 0, s, p, nFac);
    hipLaunchKernelGGL(k_reduce_panel_slabs, dim3((kPanelSlab + 15) / 16, p.nPanelPairs), dim3(256), 0, s, p);
    // The join comes LAST (the slab sum touches pose rows only, the chain's kernels read speed / bias rows only) but inside this
    // function: once it returns, everything it enqueued is ordered on `s` (a later k_zero_build must not meet a chain still reading S).
    // The chain is what the main stream waits for: k_sb_factor (145 KB of LDS) and k_sb_forward (161 KB per workgroup) need EMPTY CUs
    // and get them only when k_schur_rows (two workgroups of 77 KB per CU, refilled from a queue of ~1 000) runs out -- they end with
    // it however early they are launched, whatever the stream's or the waves' priority (both measured).
    if (early) HIP_LAUNCH_OK(hipStreamWaitEvent(s, lane->join, 0));
    return;
  } else if (p.L > 0 && p.N > 0 && dC > 0) {
    const size_t accBytes = ((size_t)dC * dC + 3 * dC) * 8;
    const size_t stageBytes = (size_t)4 * 64 * kStage * 8;
    const bool useLds = accBytes + stageBytes <= 150 * 1024;
    if (useLds) {
      const int grid = p.nSlabs;
#define LAUNCH(E)                                                                                                   \
  do {                                                                                                              \
    ensureDynamicLds((const void*)k_schur<true, E>, accBytes + stageBytes);                                        \
    hipLaunchKernelGGL((k_schur<true, E>), dim3(grid + nFac + nPri), dim3(256), accBytes + stageBytes, s, p, mu,   \
                       initScale ? 1 : 0, grid, nFac);                                                              \
  } while (0)
      if (p.anyExtVariable) LAUNCH(true); else LAUNCH(false);
#undef LAUNCH
    } else {
      (void)hipMemsetAsync(p.slabs, 0, accBytes, s);
      const int grid = min((p.L + 3) / 4, 2048);
      DeviceProblem q = p;
#define LAUNCH(E)                                                                                                   \
  do {                                                                                                              \
    ensureDynamicLds((const void*)k_schur<false, E>, stageBytes); \
    hipLaunchKernelGGL((k_schur<false, E>), dim3(grid + nFac + nPri), dim3(256), stageBytes, s, q, mu,             \
                       initScale ? 1 : 0, grid, nFac);                                                              \
  } while (0)
      if (p.anyExtVariable) LAUNCH(true); else LAUNCH(false);
#undef LAUNCH
    }
  } else if (nFac + nPri > 0) {
    hipLaunchKernelGGL(k_factors_only, dim3(nFac + nPri), dim3(256), 0, s, p, nFac);
  }
  if (p.L > 0 && p.N > 0 && dC > 0) {
    const size_t accBytes = ((size_t)dC * dC + 3 * dC) * 8;
    const bool useLds = accBytes + (size_t)4 * 64 * kStage * 8 <= 150 * 1024;
    DeviceProblem q = p;
    if (!useLds && !p.schurDense) q.nSlabs = 1;
    const int n = dC * dC + 3 * dC;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((n + 15) / 16), dim3(256), 0, s, q);
  }
}

// ================================================================ K6: reduced system solve
// v_mfma_f64_16x16x4_f64 operand layout used throughout: A/B one f64 per lane, A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// C/D: col=l&15, row=(l>>4)+4*reg.  The matrix is padded to a multiple of 16 with an identity tail so that every tile is full.
// LDS-resident variant for dpad <= 176: the lower triangle lives in LDS as 16x17 tiles (tile (I,J), I>=J at
// index I(I+1)/2+J), so the whole factorisation and both triangular solves run at LDS latency.
constexpr int kTile = 16 * kPanelLd;  // doubles per tile
__device__ __forceinline__ double* tileAt(double* base, int I, int J) { return base + (size_t)(I * (I + 1) / 2 + J) * kTile; }

constexpr int kCholLdsThreads = 512;  // 8 waves (16 waves measured slower: LDS pressure, the diagonal block is the critical path)
constexpr int kCholFlagInts = 48;
#ifdef SVIN_CHOL_TIMING
constexpr int kCholBorderOff = 240;   // (doubles behind the flags: the timing build's stamp buffer comes first)
#else
constexpr int kCholBorderOff = 0;
#endif
// Flags of the barrier-free factorisation (LDS ints, monotonic counters, written by exactly one wave each):
//   fl[0]       pivotDone  number of diagonal tiles whose factor D(kb) and 1/L_ii are in LDS
//   fl[1 + I]   xReady[I]  number of block columns for which the panel tile X(I, .) of tile row I is stored
//   fl[13 + I]  rowUpd[I]  number of block columns applied to every tile of tile row I
//   fl[27]      ySteps     solution blocks stored by the backward substitution; fl[28 + kb] farDone[kb] (see there)
//   fl[26]      a bounded spin gave up (a bug, not a numerical event: reported through cholFail bit 4 -- kCholFailSync, Window::solve throws on it)
// Everything the flags guard lives in LDS, and the DS operations of one wave execute in issue order: "data stores, then flag
// store" on the writer and "flag load, then data loads" on the reader are ordered by the hardware.  The compiler is held to
// that order by memory clobbers; a release / acquire fence pair would add an s_waitcnt (one LDS round trip) on either side.
__device__ __forceinline__ void cholFlagSet(int* f, int v, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
// kSleep: s_sleep units (64 clocks) between polls -- 1 on the waves whose wait is on the critical path, more for the wave
// that shares wave 0's SIMD (every poll of a waiting wave takes issue slots from the pivot routine)
template <int kSleep = 1>
__device__ __forceinline__ int cholFlagWait(int* f, int v, int* bail) {   // returns the number of polls that failed
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
    __builtin_amdgcn_s_sleep(kSleep);
    if (++spins > (1 << 18)) { __hip_atomic_store(bail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
  }
  asm volatile("" ::: "memory");
  return spins;
}
// all of f[0 .. n) >= v, polled with one LDS read per round (lane j reads f[j])
template <int kSleep = 1>
__device__ __forceinline__ int cholFlagWaitAll(int* f, int n, int v, int lane, int* bail) {
  int spins = 0;
  while (true) {
    const int x = (lane < n) ? __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : v;
    if (__all(x >= v)) break;
    __builtin_amdgcn_s_sleep(kSleep);
    if (++spins > (1 << 18)) { __hip_atomic_store(bail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
  }
  asm volatile("" ::: "memory");
  return spins;
}
// Round 3: the factorisation has NO workgroup barrier between the load and the backward substitution.  Round 2 ran two
// phases per block column with a barrier after each (P: panel solves, D: pivot tile on wave 0 next to the trailing update of
// the others), so wave 0 -- the serial chain of the whole solve, 150 pivots -- waited twice per column for the slowest
// worker (in-kernel timeline, profiles/r03_chol_timeline_before.txt: 10 x (1.4 k + 4.8 k) cycles of 90 k).  Now every tile
// row I >= 2 has an OWNER wave for the whole factorisation, and the waves meet through the counters above:
//   wave 0      per column kb: pivot tile kb out of its registers -> pivotDone; panel solve of the tile right below it
//               (tile row kb+1, waits for rowUpd[kb+1]) -> xReady[kb+1]; update of the next diagonal tile in registers.
//               Its loop never waits for more than the look-ahead row.
//   owners      (waves 1-3, 5-7; rows dealt largest first, serpentine, so the short early rows -- the look-ahead rows -- sit
//               on the lightly loaded waves) per column: panel solves of their rows once pivotDone allows, then the
//               trailing update of each row as soon as the panel tiles xReady[kb+1 .. I-1] it multiplies with exist.
//               They run as far behind wave 0 as the dependencies allow; the imbalance of the first columns (nine tile
//               products in row 9 against a 3 k-cycle pivot tile) is absorbed instead of stalling the chain.
//   wave 4      shares SIMD 0 with wave 0 and stays off the matrix pipe: forward substitution of the right-hand side.
// Load phase: wave 0 loads ONLY tile (0, 0), straight into the accumulator layout, and factorises it while the other
// waves are still waiting for theirs; the damping is added by the lanes that hold the diagonal entries (no second
// barrier); gFull and the damped diagonal metric stay in registers / LDS, so the kernel ends with stores only.
// (two waves per SIMD by construction: the register budget is 256 VGPRs, which keeps a wave's ~45 loads in ONE batch)
// kBorder (d = 177 .. 180: the stereo_rig_v2 sliding window with four speed / bias blocks is 180): the system is `border` <= 4 rows
// larger than the eleven tile rows LDS holds.  Those rows are eliminated FIRST, while the tiles are loaded: with the border block
// C = Lc Lc^T (damped like every diagonal entry), B the border's rows under the main block and V = Lc^-1 B (4 x 176),
//   A' = A - V^T V  (one 16x16x4 product per tile: lane (g, c) supplies -V[g][16 I + c] and V[g][16 J + c]; the same products in the
//                    same order on either side of the diagonal, so diagonal tiles stay exactly symmetric),
//   g' = g1 - B^T C^-1 g2,   y2 = C^-1 (g2 - B y1)  after the main solve.
// Any elimination order is stable for a positive definite matrix; the kernel below this prologue is the d = 176 solver unchanged.
// The instantiation without border is the code of rounds 3-5 (every border statement sits under if constexpr).
// kBorder = 2 (d = 181 .. 200: the stereo_rig_v2 sliding window is 198): up to 24 border rows.  The border block is factorised and
// V = Lc^-1 B, q = C^-1 g2, Lc^-1 are written to `bscr` by k_chol_border_prepare, one small launch ahead (a 22 x 22 Cholesky per lane
// is not an option); here every tile takes up to six products with V read from there (L2), the right-hand side g1 - B^T q, and at the
// end y2 = q - Lc^-T (V y1).
constexpr int kBorderMaxRows = 24, kBorderMaxQ = kBorderMaxRows / 4, kBorderLdV = 176;
// layout of the border scratch (doubles): V [32 x 176] | Lc^-1 [32 x 32, row-major, lower] | q [32] | g1 - B^T q [176] | tile-column mask [16]
constexpr int kBorderMP = 32, kBorderOffLinv = kBorderMP * kBorderLdV, kBorderOffQ = kBorderOffLinv + kBorderMP * kBorderMP,
              kBorderOffG = kBorderOffQ + kBorderMP /* g1 - B^T q, 176 */, kBorderOffMask = kBorderOffG + kBorderLdV /* 16: tile column J of V is not zero */,
              kBorderScratchDoubles = kBorderOffMask + 16;
template <int kBorder>
__device__ __forceinline__ void k_chol_solve_lds_body(const DeviceProblem& p, int dpad, double mu, int initScale, int fuseFinalize, int border, const double* bscr) {
  extern __shared__ double smem[];
  SVIN_ARGS(SA(p.S), SA(p.gRed), SA(p.gFull), SA(p.hC), SA(p.scaleC), SA(p.htilC), SA(p.yC), SA(p.vC), SA(p.scal), SA(p.d), SA(p.ldS),
            SA(p.sPadded), SA(dpad), SA(mu), SA(initScale), SA(fuseFinalize));
  const int t = threadIdx.x, d = kBorder != 0 ? p.d - border : p.d, nT = dpad / 16;   // d: rows of the main block
  // the wave index through v_readfirstlane: tile indices and LDS tile addresses become scalar (SALU) arithmetic
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, nW = kCholLdsThreads / 64;
  const int nTilesAll = nT * (nT + 1) / 2;
  double* tiles = smem;
  double* rhs = smem + (size_t)nTilesAll * kTile;  // dpad
  double* dinv = rhs + dpad;                       // dpad
  double* htil = dinv + dpad;                      // dpad: damped diagonal metric (the steepest-descent scaling)
  int* fl = reinterpret_cast<int*>(htil + dpad);   // kCholFlagInts
  int* bail = fl + 26;
  const int g = lane >> 4, c = lane & 15;
#ifdef SVIN_CHOL_TIMING
  const long long ql0 = __builtin_readcyclecounter();
  long long qWait = 0, qBusy = 0;
// time spent waiting on flags, in failed polls (~150-200 cycles each: s_sleep 1 + an LDS read); a cycle stamp on either side of
// every wait costs more than most waits (s_memtime + its s_waitcnt: ~250 cycles)
#define CHOL_SPINS(x) qWait += (x)
#define CHOL_NOTE(which, kb, val) do { if (lane == 0) stampBuf[(which) * 12 + (kb)] = (double)(val); } while (0)
// raw stamp `which` of block column kb, relative to the kernel start (last launch wins)
// (into LDS, copied out at the end: a global store per stamp makes the next acquire fence wait for it -- ~1 k cycles)
  double* stampBuf = reinterpret_cast<double*>(fl + kCholFlagInts);
  if (t < 240) stampBuf[t] = 0;
#define CHOL_STAMP(which, kb) do { if (lane == 0) stampBuf[(which) * 12 + (kb)] = (double)(__builtin_readcyclecounter() - ql0); } while (0)
#ifdef SVIN_CHOL_TIMING_FINE
#define CHOL_STAMP_FINE(which, kb) CHOL_STAMP(which, kb)
#else
#define CHOL_STAMP_FINE(which, kb)
#endif
#else
#define CHOL_STAMP_FINE(which, kb)
#define CHOL_SPINS(x) (void)(x)
#define CHOL_NOTE(which, kb, val) (void)(val)
#define CHOL_STAMP(which, kb)
#endif
  const int ldS = p.ldS ? p.ldS : d;
  // right-hand side, full gradient, and (without the fused metric) the metric of an earlier pass: one element per thread
  double rhsMine = (t < d) ? p.gRed[t] : 0.0;
  // ---- border prologue: every lane factorises the (at most) 4 x 4 border block itself (uniform values)
  double lgB[4] = {0, 0, 0, 0};            // row g of Lc^-1 (what turns a column of B into this lane row's entry of V)
  double LiB[4][4], qB[4] = {0, 0, 0, 0};  // Lc^-1 (lower), q = C^-1 g2
  double htB[4] = {1, 1, 1, 1}, scB[4] = {1, 1, 1, 1};
  double bEnd[3] = {0, 0, 0};              // row `wave` of B at columns lane, lane + 64, lane + 128 (the product B y1 at the end)
  auto bAt = [&](int a, int col) { return (a < border) ? p.S[(size_t)(d + a) * ldS + col] : 0.0; };
  const int nQ = (border + 3) >> 2;   // (kBorder == 2) products per tile
  // (kBorder == 2) this lane's entries of V for the tile column whose first matrix column is col0: V[4 q + g][col0 + c], q < nQ
  // (unconditional: the scratch holds kBorderMP rows, zero beyond the border's -- a predicate per load is an exec-mask branch per load)
  unsigned vMask = 0;   // (kBorder == 2) bit J: tile column J of V is not identically zero
  if constexpr (kBorder == 2) {
#pragma unroll
    for (int J = 0; J < 11; ++J) vMask |= (bscr[kBorderOffMask + J] != 0.0) ? (1u << J) : 0u;
    vMask = __builtin_amdgcn_readfirstlane(vMask);
  }
  const double* vLane = bscr + (kBorder == 2 ? g * kBorderLdV + c : 0);
  auto vLoad = [&](int col0, double (&v)[kBorderMaxQ]) {
#pragma unroll
    for (int q = 0; q < kBorderMaxQ; ++q) v[q] = vLane[q * 4 * kBorderLdV + col0];
  };
  double vEnd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // (kBorder == 2) rows wave, wave + 8, wave + 16 of V at columns lane + 64 k: for V y1 at the end
  if constexpr (kBorder == 2) {
    if (t < d) rhsMine = bscr[kBorderOffG + t];   // g' = g1 - B^T q, formed by k_chol_border_prepare
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        vEnd[kk][k] = (wave + 8 * kk < border && lane + 64 * k < d) ? bscr[(size_t)(wave + 8 * kk) * kBorderLdV + lane + 64 * k] : 0.0;
  }
  if constexpr (kBorder == 1) {
    double Cb[4][4], g2[4], bcol[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) Cb[a][b] = (a < border) ? p.S[(size_t)(d + a) * ldS + d + b] : (a == b ? 1.0 : 0.0);
      g2[a] = (a < border) ? p.gRed[d + a] : 0.0;
      bcol[a] = (t < d) ? bAt(a, t) : 0.0;
      const int ia = d + min(a, border - 1);
      const double hc = fuseFinalize ? p.hC[ia] : 0.0;
      double sc = (fuseFinalize && !initScale) ? p.scaleC[ia] : 1.0;
      double ht = fuseFinalize ? 0.0 : p.htilC[ia];
      if (fuseFinalize) {   // the same metric and damping dampDiag gives a diagonal entry of the main block
        if (initScale) sc = 1.0 / (1.0 + sqrt(hc));
        ht = fmin(fmax(hc * sc * sc, 1e-6), 1e32) / (sc * sc);
        if (a < border) Cb[a][a] += mu * ht;
      }
      htB[a] = ht; scB[a] = sc;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) bEnd[k] = (wave < 4 && lane + 64 * k < d) ? bAt(wave, lane + 64 * k) : 0.0;
    // Lc (lower) in place, then its inverse by forward substitution on the unit vectors
    bool bad = false;
    double rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double dj = Cb[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) dj = __builtin_fma(-Cb[j][k], Cb[j][k], dj);
      bad = bad || !(dj > 0);
      const double lj = sqrt(dj > 0 ? dj : 1.0);
      Cb[j][j] = lj;
      rd[j] = 1.0 / lj;
#pragma unroll
      for (int i = j + 1; i < 4; ++i) {
        double v = Cb[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) v = __builtin_fma(-Cb[i][k], Cb[j][k], v);
        Cb[i][j] = v * rd[j];
      }
    }
    if (bad && t == 0) atomicOr(&p.scal->cholFail, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // column j of Lc^-1
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < j) { LiB[i][j] = 0.0; continue; }
        double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = j; k < i; ++k) v = __builtin_fma(-Cb[i][k], LiB[k][j], v);
        LiB[i][j] = v * rd[i];
      }
    }
    double u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i] = 0.0;
#pragma unroll
      for (int k = 0; k <= i; ++k) u[i] = __builtin_fma(LiB[i][k], g2[k], u[i]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      qB[k] = 0.0;
#pragma unroll
      for (int i = k; i < 4; ++i) qB[k] = __builtin_fma(LiB[i][k], u[i], qB[k]);
      rhsMine = __builtin_fma(-bcol[k], qB[k], rhsMine);   // g' = g1 - B^T q
      lgB[k] = selectByRow(lane >> 4, LiB[0][k], LiB[1][k], LiB[2][k], LiB[3][k]);
    }
  }
  // entry of V this lane supplies to a tile product, from its own lane row's entry of the column of B: V[g][col] = sum_b Lc^-1[g][b] B[b][col]
  auto vOf = [&](double bMine) {
    double P[4];
    allGatherRows(bMine, P);
    return __builtin_fma(lgB[3], P[3], __builtin_fma(lgB[2], P[2], __builtin_fma(lgB[1], P[1], lgB[0] * P[0])));
  };
  const double gFullMine = (t < d) ? p.gFull[t] : 0.0;
  const double htilOld = (!fuseFinalize && t < d) ? p.htilC[t] : 1.0;
  if (t < kCholFlagInts) fl[t] = 0;
  // one tile in the accumulator layout (lane (g, c), register rg = entry (g + 4 rg, c)); identity padding; the lower
  // triangle is read, diagonal tiles come out fully symmetric (the MFMA trailing update preserves that)
  // split in two so that a wave can request ALL its tiles before it looks at the first value: tileRequest issues the
  // loads (always a valid element of the lower triangle), tileSelect substitutes the identity padding afterwards.
  // p.sPadded (the window's own S): the buffer has dpad rows of ldS >= dpad doubles with zeros beyond d, so an OFF-diagonal
  // tile is a plain 16 x 16 block -- uniform base + one lane offset, no clamps, no select (the address arithmetic of the
  // clamped form, ~14 VALU instructions per element on 7 waves, was what the load phase spent its time on).
  const int laneOff = g * ldS + c;
  auto tileRequest = [&](int I, int J, double (&x)[4]) {
    if (p.sPadded && I != J) {
      const double* base = p.S + (size_t)(16 * I) * ldS + 16 * J;   // wave-uniform
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) x[rg] = base[(size_t)(4 * rg) * ldS + laneOff];
      return;
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int gi = 16 * I + g + 4 * rg, gj = 16 * J + c;
      const int ci = min(max(gi, gj), d - 1), cj = min(min(gi, gj), d - 1);
      x[rg] = p.S[(size_t)ci * ldS + cj];
    }
  };
  auto tileSelect = [&](int I, int J, double (&x)[4]) {
    if (p.sPadded && I != J) return;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int gi = 16 * I + g + 4 * rg, gj = 16 * J + c;
      x[rg] = (gi < d && gj < d) ? x[rg] : ((gi == gj) ? 1.0 : 0.0);
    }
  };
  // A lane holds at most one diagonal entry of a diagonal tile: row 16 I + c in register (c - g) / 4.  The metric inputs of
  // that row (hC, scaleC) are requested together with the tile; `dampDiag` then adds mu * htil to the entry and keeps htil.
  const bool diagLane = ((c - g) & 3) == 0 && c >= g;
  const int diagReg = (c - g) >> 2;
  // (the metric's global stores are handed back to the caller: waves 1-7 issue them AFTER the load barrier, whose vmcnt(0) --
  //  for the LDS-DMA pieces -- would otherwise wait for them too)
  struct MetricOut { int i; double sc, ht; };
  auto dampDiag = [&](int I, bool valid, double hc, double scIn, double (&v)[4]) {
    const int i = 16 * I + c;
    double damp = 0.0;
    MetricOut out{-1, 0.0, 0.0};
    if (valid && diagLane && i < d) {
      double sc = scIn;
      if (initScale) sc = 1.0 / (1.0 + sqrt(hc));
      const double ht = fmin(fmax(hc * sc * sc, 1e-6), 1e32) / (sc * sc);
      htil[i] = ht;
      damp = mu * ht;
      out = MetricOut{i, sc, ht};
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) v[rg] += (rg == diagReg) ? damp : 0.0;
    return out;
  };
  auto storeMetric = [&](const MetricOut& m) {
    if (m.i >= 0) {
      if (initScale) p.scaleC[m.i] = m.sc;
      p.htilC[m.i] = m.ht;
    }
  };
  // (rhsMine / htilOld go to LDS right before the load barrier: storing them here would wait for their loads before the
  //  first tile load is issued)
  const int lrow = g * kPanelLd + c;  // accumulator layout: + 4 rg kPanelLd
  const int lop = c * kPanelLd + g;   // operand layout: + 4 q
  // X^T = L^-1 A^T: with the operands in this order the product comes out TRANSPOSED in the accumulator layout, which is
  // X in the operand layout (lane (row, g) register r = X[row][4r + g]) -- exactly what the trailing update reads
  auto panelSolve = [&](double* A, const double* D, int k0) {
    // all eight operands requested before the first product (one LDS latency instead of four on a dependent chain); the
    // pivot tile holds L^-T with zeros below the diagonal, so row kk of it IS column kk of L^-1 -- no select
    double a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = A[lop + 4 * q];
      b[q] = D[(4 * q + g) * kPanelLd + c];
    }
    d4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b[q], a[q], acc, 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) A[lop + 4 * rg] = acc[rg];
    return acc;
  };
  // tile (I, J) -= X(I, kb) X(J, kb)^T for J = kb+1 .. I (the trailing update of one tile row).  TWO tiles at a time on
  // independent accumulators: the four products of one tile are a dependent chain (64 cycles each on the matrix pipe), a
  // second chain fills the gaps -- with two owner waves per SIMD the pipe stays busy -- and the operands of the next pair
  // are in flight meanwhile.  The A operand X(I, kb) is read once per row, B / C tiles are walked by pointer increments.
  auto updateRow = [&](int I, int kb, int skip) {   // tiles (I, kb+1+skip .. I)
    const double* A = tileAt(tiles, I, kb);
    double a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = -A[lop + 4 * q];
      asm volatile("" : "+v"(a[q]));   // negated once, kept (the compiler re-derives -x in front of every product otherwise)
    }
    double* Cb = tileAt(tiles, I, kb + 1 + skip);
    const double* B = tileAt(tiles, kb + 1 + skip, kb);
    int rowTiles = kb + 2 + skip;  // tiles in the block row of B's tile: the next row's tile (., kb) lies that many tiles on
    const int nTl = I - kb - skip;  // tiles in this row segment
    // The body is BRANCH-FREE over a pair of tiles: every pass requests the next pair and runs eight products, whether the
    // tiles exist or not (a row segment of odd length computes one product chain on whatever lies behind the row -- LDS reads
    // past the tiles are harmless -- and does not store it).  With the second tile and the prefetch under wave-uniform
    // branches the compiler met every product at a control-flow join: worst-case s_nop 9 in front of it and s_waitcnt
    // lgkmcnt(0/1) on the prefetch just issued -- 940 cycles per tile where the shared matrix pipe allows 560.
    const double* Cf = Cb;   // fetch cursor (tile about to be requested); Cb stays on the pair being computed
    auto fetch = [&](double (&bq)[4], d4_t& cq) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) cq[rg] = Cf[lrow + 4 * rg * kPanelLd];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = B[lop + 4 * q];
      B += (size_t)rowTiles * kTile;
      ++rowTiles;
      Cf += kTile;
    };
    auto products = [&](const double (&p0)[4], d4_t& q0, const double (&p1)[4], d4_t& q1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], p0[q], q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], p1[q], q1, 0, 0, 0);
      }
    };
    auto store = [&](const d4_t& q0, const d4_t& q1, bool two) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) Cb[lrow + 4 * rg * kPanelLd] = q0[rg];
      if (two) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Cb[kTile + lrow + 4 * rg * kPanelLd] = q1[rg];
      }
      Cb += 2 * kTile;
    };
    // two register sets take turns (no rotation copies): set X is computed while set Y is in flight
    double bx0[4], bx1[4], by0[4], by1[4];
    d4_t cx0, cx1, cy0, cy1;
    fetch(bx0, cx0);
    fetch(bx1, cx1);
    for (int J = 0; J < nTl; J += 4) {
      fetch(by0, cy0);
      fetch(by1, cy1);
      products(bx0, cx0, bx1, cx1);
      store(cx0, cx1, J + 1 < nTl);
      if (J + 2 >= nTl) break;
      fetch(bx0, cx0);
      fetch(bx1, cx1);
      products(by0, cy0, by1, cy1);
      store(cy0, cy1, J + 3 < nTl);
    }
  };
  // tile (I, kb+1) -= X(I, kb) X(kb+1, kb)^T alone: the one tile of row I that the NEXT column's panel solve needs
  auto updateOne = [&](int I, int kb) {
    const double* A = tileAt(tiles, I, kb);
    const double* B = tileAt(tiles, kb + 1, kb);
    double* C = tileAt(tiles, I, kb + 1);
    double a[4], b[4];
    d4_t cq;
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = -A[lop + 4 * q]; b[q] = B[lop + 4 * q]; }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) cq[rg] = C[lrow + 4 * rg * kPanelLd];
#pragma unroll
    for (int q = 0; q < 4; ++q) cq = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], cq, 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[lrow + 4 * rg * kPanelLd] = cq[rg];
  };
  if (wave == 0) {
    // ------------------------------------------------------------------------------------------ the serial chain
    double v0[4];
    tileRequest(0, 0, v0);
    const int i0 = min(c, d - 1);
    const double hc0 = fuseFinalize ? p.hC[i0] : 0.0, sc0 = (fuseFinalize && !initScale) ? p.scaleC[i0] : 1.0;
    tileSelect(0, 0, v0);
    if (fuseFinalize) storeMetric(dampDiag(0, true, hc0, sc0, v0));
    d4_t accD = {v0[0], v0[1], v0[2], v0[3]};
    if constexpr (kBorder == 1) {
      const double v = vOf(bAt(g, c));
      accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, accD, 0, 0, 0);
    }
    if constexpr (kBorder == 2) {
      double v[kBorderMaxQ];
      vLoad(0, v);
#pragma unroll
      for (int q = 0; q < kBorderMaxQ; ++q)
        if (q < nQ && (vMask & 1u)) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-v[q], v[q], accD, 0, 0, 0);
    }
#ifdef SVIN_CHOL_TIMING
    long long pivotCycles = 0;
#endif
    for (int kb = 0; kb < nT; ++kb) {
      const int k0 = 16 * kb;
      double* D = tileAt(tiles, kb, kb);
      CHOL_STAMP(0, kb);
#ifdef SVIN_CHOL_TIMING
      cholDiag16Acc<true>(accD, D, dinv + k0, lane, &p.scal->cholFail, &pivotCycles);
#else
      cholDiag16Acc<true>(accD, D, dinv + k0, lane, &p.scal->cholFail);
#endif
      CHOL_STAMP_FINE(1, kb);
      cholFlagSet(fl + 0, kb + 1, lane);
      CHOL_STAMP_FINE(2, kb);
      if (kb == 0) {
        if (t < dpad) { rhs[t] = rhsMine; if (!fuseFinalize) htil[t] = htilOld; }
        ldsBarrier();   // the other waves have stored their tiles (their only barrier before the backward substitution)
#ifdef SVIN_CHOL_TIMING
        if (t == 0) p.partial[(size_t)15 * 4096 + 1] += (double)(__builtin_readcyclecounter() - ql0);
#endif
      }
      if (kb + 1 < nT) {
        CHOL_STAMP_FINE(3, kb);
        if (kb > 0) { const int sp = cholFlagWait(fl + 13 + kb + 1, kb, bail); CHOL_SPINS(sp); CHOL_NOTE(3, kb, sp); }
        CHOL_STAMP_FINE(4, kb);
        double* A = tileAt(tiles, kb + 1, kb);
        const double* Cb = tileAt(tiles, kb + 1, kb + 1);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) accD[rg] = Cb[lrow + 4 * rg * kPanelLd];
        const d4_t xT = panelSolve(A, D, k0);
        CHOL_STAMP_FINE(5, kb);
        // the first product of the diagonal update is on the matrix pipe while the stores of X drain and the flag goes out
        accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-xT[0], xT[0], accD, 0, 0, 0);
        cholFlagSet(fl + 1 + kb + 1, kb + 1, lane);
        CHOL_STAMP_FINE(6, kb);
#pragma unroll
        for (int q = 1; q < 4; ++q) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-xT[q], xT[q], accD, 0, 0, 0);
      }
    }
#ifdef SVIN_CHOL_TIMING
    if (lane == 0) g_cholDbg[0] += (double)pivotCycles;
#endif
  } else {
    // ------------------------------------------------------------------------------------------ load (waves 1-7)
    // The wave next to wave 0 loads nothing: the two share SIMD 0's VALU, and while wave 0 factorises its first tile the address
    // arithmetic of a loader there took twice as long as anywhere else (it reached the load barrier 1.4-2.5 k cycles after the
    // other six) and slowed that first pivot tile down as well.
    const int ldr = wave < nW / 2 ? wave - 1 : wave - 2;   // loader index 0 .. 5 (waves 1-3, 5-7)
    if (wave != nW / 2) {
      // Diagonal tiles 1 .. nT-1: tiles 1 + ldr and 7 + ldr (nT <= 11); off-diagonal tiles dealt round robin, at most
      // ten per loader (55 at nT = 11).  Tile indices are scalar and computed first, so that what follows is ONE batch of
      // loads without a branch in it (out-of-range slots re-read a valid tile and are not stored).
      constexpr int kMaxOff = 10, kLoaders = 6;
      const int nOff = nT * (nT - 1) / 2;
      int dI[2], oI[kMaxOff], oJ[kMaxOff];
      dI[0] = min(1 + ldr, nT - 1);
      dI[1] = min(7 + ldr, nT - 1);
      const bool dmaOff = kBorder != 0 || p.sPadded != 0;   // (the border variants are only launched on a padded S)
#pragma unroll
      for (int it = 0; it < kMaxOff; ++it) {
        if (dmaOff) { oI[it] = 1; oJ[it] = 0; continue; }   // (the DMA path walks whole tile rows: no index search)
        const int e = min(ldr + kLoaders * it, max(nOff - 1, 0));
        // I (I - 1) / 2 <= e < I (I + 1) / 2 without a loop (ten independent scalar compares; a search loop costs a
        // dependent multiply per step and these indices gate the very first loads): e <= 54 at nT = 11
        const int I = 1 + (e >= 1) + (e >= 3) + (e >= 6) + (e >= 10) + (e >= 15) + (e >= 21) + (e >= 28) + (e >= 36) + (e >= 45);
        oI[it] = __builtin_amdgcn_readfirstlane(I);
        oJ[it] = __builtin_amdgcn_readfirstlane(e - I * (I - 1) / 2);
      }
      double vd[2][4], hcv[2], scv[2], vo[kMaxOff][4];
      MetricOut metricOut[2] = {{-1, 0.0, 0.0}, {-1, 0.0, 0.0}};
      // border: this lane row's entries of B at the columns of the tiles this wave loads -- its two diagonal tiles, its two DMA rows
      // (rowA, rowB below) and every tile column J < 10 those rows cross; requested with the tiles, turned into V after the batch
      constexpr int kBorderCols = 10;
      double bDiag[2] = {0, 0}, bRow[2] = {0, 0}, bJ[kBorderCols];
      // (kBorder == 2) every entry of V this wave will need, requested with its tiles: from one launch to the next V crosses the XCDs'
      // L2s, ~2 us per round trip -- requested tile by tile behind the DMA the update pass took 8 us
      double vDiag[2][kBorderMaxQ], vRow[2][kBorderMaxQ], vAll[kBorderCols][kBorderMaxQ];
      if constexpr (kBorder == 2) {
        vLoad(16 * dI[0], vDiag[0]);
        vLoad(16 * dI[1], vDiag[1]);
        vLoad(16 * min(1 + ldr, nT - 1), vRow[0]);
        vLoad(16 * (nT - 1 - ldr), vRow[1]);
#pragma unroll
        for (int J = 0; J < kBorderCols; ++J) vLoad(16 * J, vAll[J]);
      }
      if constexpr (kBorder == 1) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) bDiag[sl] = bAt(g, 16 * dI[sl] + c);
        bRow[0] = bAt(g, 16 * min(1 + ldr, nT - 1) + c);
        bRow[1] = bAt(g, 16 * (nT - 1 - ldr) + c);
#pragma unroll
        for (int J = 0; J < kBorderCols; ++J) bJ[J] = bAt(g, 16 * J + c);
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        tileRequest(dI[sl], dI[sl], vd[sl]);
        const int i = min(16 * dI[sl] + c, d - 1);
        hcv[sl] = fuseFinalize ? p.hC[i] : 0.0;
        scv[sl] = (fuseFinalize && !initScale) ? p.scaleC[i] : 1.0;
      }
      // Off-diagonal tiles of a padded S go through the LDS-DMA path (global_load_lds_dwordx4: 16 bytes per lane, 1 KB of
      // CONSECUTIVE LDS per wave instruction, no register round trip and no ds_write pass).  The 16 x 17 tile is 136 pairs
      // of doubles; pair p sits in row (2p) / 17 at column (2p) % 17 and is two consecutive doubles of S as well -- also
      // across the padding column: (row, 16) | (row + 1, 0) is fetched from one double to the left of the next row's first
      // element (what lands in the padding is never read).  Three instructions per tile, the third on eight lanes.
      if (dmaOff) {
        const double* laneSrc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int off = 2 * (lane + 64 * k), row = (off * 241) >> 12, col = off - 17 * row;   // off / 17 for off < 272
          laneSrc[k] = p.S + ((col == 16) ? (row + 1) * ldS - 1 : row * ldS + col);
        }
        // whole tile rows, the longest with the shortest (rows 1 + ldr and nT - 1 - ldr: 10-11 tiles per loader at nT = 10, 11):
        // along a row the tiles are 16 columns apart in S and adjacent in LDS -- two pointer increments per tile
        auto dmaRow = [&](int I) {
          size_t so = (size_t)(16 * I) * ldS;
          char* dst = reinterpret_cast<char*>(tileAt(tiles, I, 0));
          for (int J = 0; J < I; ++J) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(laneSrc[0] + so),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(laneSrc[1] + so),
                                             (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
            if (lane < 8)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(laneSrc[2] + so),
                                               (__attribute__((address_space(3))) void*)(dst + 2048), 16, 0, 0);
            so += 16;
            dst += kTile * 8;
          }
        };
        const int rowA = 1 + ldr, rowB = nT - 1 - ldr;
        if (rowB > rowA) dmaRow(rowB);
        if (rowA <= rowB && rowA < nT) dmaRow(rowA);
      } else {
#pragma unroll
        for (int it = 0; it < kMaxOff; ++it) tileRequest(oI[it], oJ[it], vo[it]);
      }
      if (wave == 1) CHOL_STAMP_FINE(8, 0);   // requests issued
      __builtin_amdgcn_sched_barrier(0);   // nothing that consumes a loaded value moves above this line: one batch of ~44 loads
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) tileSelect(dI[sl], dI[sl], vd[sl]);
      if (!dmaOff) {
#pragma unroll
        for (int it = 0; it < kMaxOff; ++it) tileSelect(oI[it], oJ[it], vo[it]);
      }
#ifdef SVIN_CHOL_TIMING
      if (wave == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHOL_STAMP_FINE(8, 1); }   // all values have arrived
#endif
      if constexpr (kBorder == 2) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          d4_t a = {vd[sl][0], vd[sl][1], vd[sl][2], vd[sl][3]};
#pragma unroll
          for (int q = 0; q < kBorderMaxQ; ++q)
            if (q < nQ && ((vMask >> dI[sl]) & 1u)) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-vDiag[sl][q], vDiag[sl][q], a, 0, 0, 0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) vd[sl][rg] = a[rg];
        }
      }
      if constexpr (kBorder == 1) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const double v = vOf(bDiag[sl]);
          d4_t a = {vd[sl][0], vd[sl][1], vd[sl][2], vd[sl][3]};
          a = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, a, 0, 0, 0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) vd[sl][rg] = a[rg];
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bool valid = (sl == 0 ? 1 + ldr : 7 + ldr) < nT;
        if (fuseFinalize) metricOut[sl] = dampDiag(dI[sl], valid, hcv[sl], scv[sl], vd[sl]);
        if (valid) {
          double* dst = tileAt(tiles, dI[sl], dI[sl]);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) dst[lrow + 4 * rg * kPanelLd] = vd[sl][rg];
        }
      }
      if (!dmaOff) {
#pragma unroll
        for (int it = 0; it < kMaxOff; ++it) {
          if (ldr + kLoaders * it < nOff) {
            double* dst = tileAt(tiles, oI[it], oJ[it]);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) dst[lrow + 4 * rg * kPanelLd] = vo[it][rg];
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces of this wave have landed
      if constexpr (kBorder == 2) {
        // the off-diagonal tiles this wave brought in (its two DMA rows): tile (I, J) -= V_I^T V_J, in place; V_J of the next tile is
        // requested before the products of this one
        const int rowA = 1 + ldr, rowB = nT - 1 - ldr;
        const bool hasB = rowB > rowA, hasA = rowA <= rowB && rowA < nT;          // rows this wave brought in
        const bool actB = hasB && ((vMask >> rowB) & 1u), actA = hasA && ((vMask >> rowA) & 1u);   // ... whose tile column of V is not zero
        const int nJ = actB ? rowB : (actA ? rowA : 0);
        double* TB = tileAt(tiles, actB ? rowB : 1, 0);
        double* TA = tileAt(tiles, actA ? rowA : 1, 0);
        // one pass over the tile columns serves both rows: V_J is read once, the two tiles' products are independent chains
        auto step = [&](int J, const double (&vJ)[kBorderMaxQ]) {
          const bool doA = actA && J < rowA;
          d4_t xb = {0, 0, 0, 0}, xa = {0, 0, 0, 0};
          if (actB) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) xb[rg] = TB[lrow + 4 * rg * kPanelLd];
          }
          if (doA) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) xa[rg] = TA[lrow + 4 * rg * kPanelLd];
          }
          // (one 16x16x4 product occupies the SIMD's matrix pipe for 64 cycles and two loader waves share a SIMD: a product that is
          //  not needed is not issued -- with both chains unconditional the pass took 8 us)
          if (doA && actB) {
#pragma unroll
            for (int q = 0; q < kBorderMaxQ; ++q) {
              if (q < nQ) {
                xb = __builtin_amdgcn_mfma_f64_16x16x4f64(-vRow[1][q], vJ[q], xb, 0, 0, 0);
                xa = __builtin_amdgcn_mfma_f64_16x16x4f64(-vRow[0][q], vJ[q], xa, 0, 0, 0);
              }
            }
          } else if (actB) {
#pragma unroll
            for (int q = 0; q < kBorderMaxQ; ++q)
              if (q < nQ) xb = __builtin_amdgcn_mfma_f64_16x16x4f64(-vRow[1][q], vJ[q], xb, 0, 0, 0);
          } else if (doA) {
#pragma unroll
            for (int q = 0; q < kBorderMaxQ; ++q)
              if (q < nQ) xa = __builtin_amdgcn_mfma_f64_16x16x4f64(-vRow[0][q], vJ[q], xa, 0, 0, 0);
          }
          if (actB) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) TB[lrow + 4 * rg * kPanelLd] = xb[rg];
          }
          if (doA) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) TA[lrow + 4 * rg * kPanelLd] = xa[rg];
          }
          TB += kTile;
          TA += kTile;
        };
#pragma unroll
        for (int J = 0; J < kBorderCols; ++J) {
          if (J < nJ) {
            if ((vMask >> J) & 1u) step(J, vAll[J]);
            else { TB += kTile; TA += kTile; }
          }
        }
      }
      if constexpr (kBorder == 1) {
        // the off-diagonal tiles this wave brought in (its two DMA rows): tile (I, J) -= V_I^T V_J, in place
        double vJ[kBorderCols];
#pragma unroll
        for (int J = 0; J < kBorderCols; ++J) vJ[J] = vOf(bJ[J]);
        auto borderRow = [&](int I, double vI) {
          double* T = tileAt(tiles, I, 0);
#pragma unroll
          for (int J = 0; J < kBorderCols; ++J) {
            if (J < I) {
              d4_t a;
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) a[rg] = T[lrow + 4 * rg * kPanelLd];
              a = __builtin_amdgcn_mfma_f64_16x16x4f64(-vI, vJ[J], a, 0, 0, 0);
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) T[lrow + 4 * rg * kPanelLd] = a[rg];
            }
            T += kTile;
          }
        };
        const int rowA = 1 + ldr, rowB = nT - 1 - ldr;
        const double vA = vOf(bRow[0]), vB = vOf(bRow[1]);
        if (rowB > rowA) borderRow(rowB, vB);
        if (rowA <= rowB && rowA < nT) borderRow(rowA, vA);
      }
      storeMetric(metricOut[0]);
      storeMetric(metricOut[1]);
    }
    if (wave == 1) CHOL_STAMP_FINE(8, 2);   // tiles stored to LDS
    if (t < dpad) { rhs[t] = rhsMine; if (!fuseFinalize) htil[t] = htilOld; }
    ldsBarrier();   // LDS only: the global stores of the metric (scaleC / htilC) need not have landed
    if (wave == 1) CHOL_STAMP_FINE(8, 3);   // past the load barrier
    const int quiet = nW / 2;  // shares SIMD 0 with wave 0
    {
      // ------------------------------------------------------------------ tile-row owners (and the forward substitution)
      // Rows nT-1 .. nT-6: ONE row per owner wave (1-3, 5-7).  The rows above those (2 .. nT-7: rows 2 and 3 at nT = 10) go to the
      // wave that shares SIMD 0 with wave 0, ahead of its forward substitution: they are the look-ahead rows of the first
      // columns, and on an owner wave they queued behind that wave's other row -- wave 0 polled 36 + 15 times (~9 k cycles)
      // for rows 3 and 4 while their owner finished four more tiles of the column before.  SIMD 0's matrix pipe carries wave
      // 0's sixteen products per column; ten more chains in the first two columns cost it less than those waits.
      const bool isQuiet = wave == quiet;
      // waves w and w + 4 share a SIMD: the longest row goes with the shortest (9 | 4, 8 | 5, 7 | 6 at nT = 10) -- two long rows on
      // one matrix pipe starve each other (their look-ahead turns came 5 k cycles late), two short ones leave it idle
      const int nWork = nW - 2, widx = wave < quiet ? wave - 1 : nWork + quiet - wave;
      const int rowMaxQ = nT - 1 - nWork;   // last row of the quiet wave (none if < 2)
      auto owns = [&](int I) { return isQuiet ? (I >= 2 && I <= rowMaxQ) : (I > rowMaxQ && nT - 1 - I == widx); };
      const int kbEnd = isQuiet ? rowMaxQ - 1 : nT - 2;   // columns kb < kbEnd have work for this wave
      // Per column kb and own row I (ascending, the look-ahead row kb+2 first):  panel tile X(I, kb);  then tile (I, kb+1) ALONE
      // -- it needs wave 0's panel tile only -- and, if pivot kb+1 is out already (rows far from the diagonal run behind wave 0),
      // the panel tile X(I, kb+1) right away: the other rows wait for panel tiles, not for finished rows, and an owner that
      // first completes its whole row (up to nine products) holds all of them up;  then the rest of the row.
      unsigned early = 0;   // bit I: X(I, kb+1) was solved ahead of column kb+1
      for (int kb = 0; kb < kbEnd; ++kb) {
        const int k0 = 16 * kb;
        const double* D = tileAt(tiles, kb, kb);
        bool waited = false;
        const unsigned done = early;
        early = 0;
        if (((done >> (kb + 2)) & 1) && owns(kb + 2)) {   // look-ahead row whose panel tile is there already
          CHOL_SPINS(cholFlagWait(fl + 1 + kb + 1, kb + 1, bail));
          updateRow(kb + 2, kb, 0);
          cholFlagSet(fl + 13 + kb + 2, kb + 1, lane);
          CHOL_STAMP_FINE(7, kb);
        }
        for (int I = kb + 2; I < nT; ++I) {
          if (!owns(I) || ((done >> I) & 1)) continue;
          if (!waited) CHOL_SPINS(cholFlagWait(fl + 0, kb + 1, bail)); waited = true;
          panelSolve(tileAt(tiles, I, kb), D, k0);
          cholFlagSet(fl + 1 + I, kb + 1, lane);
          if (I == kb + 2) {   // the look-ahead row: wave 0 waits for its two tiles next
            CHOL_SPINS(cholFlagWait(fl + 1 + kb + 1, kb + 1, bail));
            updateRow(I, kb, 0);
            cholFlagSet(fl + 13 + I, kb + 1, lane);
            CHOL_STAMP_FINE(7, kb);
          }
        }
        for (int I = kb + 3; I < nT; ++I) {
          if (!owns(I)) continue;
          CHOL_SPINS(cholFlagWait(fl + 1 + kb + 1, kb + 1, bail));
          updateOne(I, kb);
          if (__hip_atomic_load(fl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= kb + 2) {
            asm volatile("" ::: "memory");
            panelSolve(tileAt(tiles, I, kb + 1), tileAt(tiles, kb + 1, kb + 1), k0 + 16);
            cholFlagSet(fl + 1 + I, kb + 2, lane);
            early |= 1u << I;
          }
          {
            // the rest needs the panel tiles of rows kb+2 .. I-1 (row I is mine)
            CHOL_SPINS(cholFlagWaitAll(fl + 1 + kb + 2, I - kb - 2, kb + 1, lane, bail));
            updateRow(I, kb, 1);
          }
          cholFlagSet(fl + 13 + I, kb + 1, lane);
        }
      }
      if (isQuiet) {
        // ------------------------------------------------------------------------------------ forward substitution
        for (int kb = 0; kb < nT; ++kb) {
          const int k0 = 16 * kb;
          const double* D = tileAt(tiles, kb, kb);
          CHOL_SPINS(cholFlagWait<8>(fl + 0, kb + 1, bail));
          double yv = 0;   // y'_kb = L_kb^-1 rhs_kb: column c of the stored L^-T (zeros below its diagonal)
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) yv = __builtin_fma(D[cc * kPanelLd + c], rhs[k0 + cc], yv);
          waveSync();
          if (lane < 16) rhs[k0 + lane] = yv;
          waveSync();
          if (kb + 1 < nT) {
            CHOL_SPINS(cholFlagWaitAll<8>(fl + 1 + kb + 1, nT - kb - 1, kb + 1, lane, bail));
            for (int i = k0 + 16 + lane; i < dpad; i += 64) {
              const double* row = tileAt(tiles, i >> 4, kb) + (i & 15) * kPanelLd;
              double sacc = 0;
#pragma unroll
              for (int k = 0; k < 16; ++k) sacc += row[k] * rhs[k0 + k];
              rhs[i] -= sacc;
            }
            waveSync();
          }
        }
      }
    }
  }
#ifdef SVIN_CHOL_TIMING
  const long long qf = __builtin_readcyclecounter();
  if (lane == 0) { p.partial[(size_t)15 * 4096 + 32 + wave] += (double)(qf - ql0); p.partial[(size_t)15 * 4096 + 40 + wave] += (double)qWait; }
#endif
  __syncthreads();   // factor, 1/L_ii and the forward-substituted right-hand side are complete
  if (t == 0 && *bail) atomicOr(&p.scal->cholFail, 4);
  // (kBorder == 2) what the last step needs from global memory, requested now by the lanes of wave 1 that will take it: column a of
  // Lc^-1, q, the gradient and the metric of border row a
  double liCol[kBorderMaxRows], qEnd = 0.0, gEnd = 0.0, hEnd = 1.0;
  const int aEnd = t - 64;
  if constexpr (kBorder == 2) {
    if (aEnd >= 0 && aEnd < border) {
#pragma unroll
      for (int i = 0; i < kBorderMaxRows; ++i) liCol[i] = (i >= aEnd && i < border) ? bscr[kBorderOffLinv + i * kBorderMP + aEnd] : 0.0;
      qEnd = bscr[kBorderOffQ + aEnd];
      gEnd = p.gFull[d + aEnd];
      hEnd = p.htilC[d + aEnd];
    }
  }
#ifdef SVIN_CHOL_TIMING
  long long q5 = __builtin_readcyclecounter();
  if (t == 0) p.partial[(size_t)15 * 4096 + 3] += (double)(q5 - ql0);
#endif
  // Backward substitution L^T y = y' without a barrier.  Wave 0 runs the chain entirely in registers: a 16-vector that is
  // REPLICATED over the columns of an MFMA result (accumulator layout: lane (g, c) register r = v[g + 4 r]) is exactly the B
  // operand of the next 16x16x4 product, so  rv = y'_kb - L(kb+1, kb)^T y_kb+1  and  y_kb = L_kb^-T rv  are two chains of
  // four products with no LDS round trip and no cross-lane step between them (the round-2 form -- every lane a 16-term
  // dot product, two LDS round trips and a barrier per block -- took 1.4 k cycles per block).  The contributions of the
  // blocks further down (j >= kb + 2) are taken off the chain: block kb has a helper wave that adds L(j, kb)^T y_j as the
  // y_j appear (four FMAs per lane, summed over the lane rows once at the end) and hands y'_kb over through farDone[kb].
  int* ySteps = fl + 27;    // number of solution blocks wave 0 has stored (from the bottom)
  int* farDone = fl + 28;   // [kb] = 1: rhs block kb holds y'_kb minus the contributions of the blocks j >= kb + 2
  // Handshakes without fences: the DS operations of one wave execute in issue order, so "data stores, then flag store" on the
  // writer and "flag load, then data loads" in ONE batch on the reader (repeated until the flag shows) need no s_waitcnt in
  // between -- a release / acquire pair costs two LDS round trips per hand-over, and wave 0 is a single in-order
  // instruction stream: whatever it waits for is on the chain.  The compiler is held to the order by memory clobbers.
#define LDS_ORDER() asm volatile("" ::: "memory")
  // (plain LDS loads behind a clobber are re-issued every time; a volatile access through the generic pointer turns into
  //  FLAT loads with system scope and a wait after each)
  if (wave == 0) {
    // One iteration = store y_kb+1, compute y_kb: two dependent 16 x 16 matrix-vector products, t = L(kb+1, kb)^T y_kb+1 and
    // y_kb = L_kb^-T (y'_kb - t).  On the VALU: the vector lives REPLICATED over the four lane rows (lane (g, c) holds v[c]),
    // colToRowForm hands lane (g, c) the entries v[4 q + g] with seven DPP moves, four FMAs per lane and the sum over the lane
    // rows (v_permlane swaps) give the product in the replicated form again -- 388 cycles per step against 632 for two chains
    // of four v_mfma_f64_16x16x4 on a column-replicated operand (64 cycles of matrix pipe each: tools/ubench/matvec_chain.hip),
    // and no LDS round trip on the chain.  Every LDS request of an iteration goes out before its arithmetic starts.
    auto diagOps = [&](const double* D, double (&oD)[4]) {   // L_kb^-1[4q + g][c]: the stored L^-T read along its rows
#pragma unroll
      for (int q = 0; q < 4; ++q) oD[q] = D[lop + 4 * q];
    };
    auto belowOps = [&](const double* Lb, double (&oL)[4]) {   // L(16 (kb+1) + 4q + g, 16 kb + c) out of tile (kb+1, kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) oL[q] = Lb[lrow + 4 * q * kPanelLd];
    };
    auto matVec = [&](const double (&m)[4], double v) {   // sum_k m[k][c] v[k], replicated
      double vq[4];
      colToRowForm(v, vq);
      double acc = m[0] * vq[0];
#pragma unroll
      for (int q = 1; q < 4; ++q) acc = __builtin_fma(m[q], vq[q], acc);
      return sumLaneRows(acc);
    };
    double aD[4], aL[4] = {0, 0, 0, 0};
    double y;
    int lateSpins = 0;
    // tile (kb+1, kb) sits right before the pivot tile (kb+1, kb+1), the pivot tile (kb, kb) kb + 2 tiles before that
    const double* Dn = tileAt(tiles, nT - 1, nT - 1);   // pivot tile of the block whose operands are requested next
    {
      double lD[4];
      diagOps(Dn, lD);
      const double rv = rhs[16 * (nT - 1) + c];
      if (nT > 1) { belowOps(Dn - kTile, aL); Dn -= (size_t)nT * kTile; diagOps(Dn, aD); }
      y = matVec(lD, rv);
    }
    CHOL_STAMP(15, 0);   // first block solved, loop entry
#pragma unroll 2
    for (int kb = nT - 2; kb >= 0; --kb) {
      const int k0 = 16 * kb;
      CHOL_STAMP_FINE(14, kb);
      if (g == 0) rhs[k0 + 16 + c] = y;
      LDS_ORDER();
      if (lane == 0) __hip_atomic_store(ySteps, nT - 1 - kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      LDS_ORDER();
      const bool far = kb + 2 < nT;
      int f = far ? __hip_atomic_load(farDone + kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 1;
      LDS_ORDER();
      double rr = rhs[k0 + c];
      double nD[4] = {0, 0, 0, 0}, nL[4] = {0, 0, 0, 0};
      if (kb > 0) { belowOps(Dn - kTile, nL); Dn -= (size_t)(kb + 1) * kTile; diagOps(Dn, nD); }
      __builtin_amdgcn_sched_barrier(0);
      CHOL_STAMP_FINE(17, kb);
      const double tv = matVec(aL, y);
      CHOL_STAMP_FINE(18, kb);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(f));   // the flag is looked at HERE, not where it was requested
      int spins = 0;
      while (f < 1) {   // the helper is late: ask again (flag first, data behind it)
        LDS_ORDER();
        f = __hip_atomic_load(farDone + kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        LDS_ORDER();
        rr = rhs[k0 + c];
        if (++spins > (1 << 18)) { __hip_atomic_store(bail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
      }
      lateSpins += spins;
      CHOL_STAMP_FINE(19, kb);
      y = matVec(aD, rr - tv);
#pragma unroll
      for (int q = 0; q < 4; ++q) { aD[q] = nD[q]; aL[q] = nL[q]; }
    }
    CHOL_STAMP(16, 0);   // loop exit
    if (g == 0) rhs[c] = y;
#ifdef SVIN_CHOL_TIMING
    if (lane == 0) p.partial[(size_t)15 * 4096 + 5] += (double)lateSpins;
#endif
  } else {
    // target blocks of this wave: wave - 1 and wave + 6 (nT <= 11: blocks 0 .. nT-3 have contributions from further down)
    const int tgt0 = wave - 1, tgt1 = wave + 6;
    const bool on0 = tgt0 + 2 < nT, on1 = tgt1 + 2 < nT;
    double s0 = 0, s1 = 0;
    const double rhs0 = on0 ? rhs[16 * tgt0 + c] : 0.0, rhs1 = on1 ? rhs[16 * tgt1 + c] : 0.0;   // final since the barrier above
    if (on0) {
      for (int j = nT - 1; j >= tgt0 + 2; --j) {
        const bool use1 = on1 && j >= tgt1 + 2;
        const double* L0 = tileAt(tiles, j, tgt0);
        const double* L1 = tileAt(tiles, j, use1 ? tgt1 : tgt0);
        double l0[4], l1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { l0[q] = L0[lrow + 4 * q * kPanelLd]; l1[q] = L1[lrow + 4 * q * kPanelLd]; }
        // the flag alone is polled (one broadcast read per round): seven waves re-reading y_j with every poll kept the LDS
        // pipe ~80 % busy and wave 0's own LDS traffic queued behind them
        int spins = 0;
        while (__hip_atomic_load(ySteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nT - j) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 18)) { __hip_atomic_store(bail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
        }
        LDS_ORDER();
        double yj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yj[q] = rhs[16 * j + 4 * q + g];
        if (use1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) s1 = __builtin_fma(l1[q], yj[q], s1);
          if (j == tgt1 + 2) {
            const double tot = sumLaneRows(s1);
            if (g == 0) rhs[16 * tgt1 + c] = rhs1 - tot;
            LDS_ORDER();
            if (lane == 0) __hip_atomic_store(farDone + tgt1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s0 = __builtin_fma(l0[q], yj[q], s0);
      }
      const double tot = sumLaneRows(s0);
      if (g == 0) rhs[16 * tgt0 + c] = rhs0 - tot;
      LDS_ORDER();
      if (lane == 0) __hip_atomic_store(farDone + tgt0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
#undef LDS_ORDER
  __syncthreads();
#ifdef SVIN_CHOL_TIMING
  if (t == 0) p.partial[(size_t)15 * 4096 + 4] += (double)(__builtin_readcyclecounter() - q5);
  if (t < 240) p.partial[(size_t)15 * 4096 + 64 + t] = stampBuf[t];
#endif
  if (t < d) { p.yC[t] = rhs[t]; p.vC[t] = gFullMine / htil[t]; }  // Gauss-Newton solution + steepest-descent direction
  if constexpr (kBorder == 2) {
    // y2 = q - Lc^-T (V y1): the rows of V y1 dealt over the waves, then one thread per border row
    double* bord = reinterpret_cast<double*>(fl + kCholFlagInts) + kCholBorderOff;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc = __builtin_fma(vEnd[kk][k], (lane + 64 * k < d) ? rhs[lane + 64 * k] : 0.0, acc);
      acc = waveSum(acc);
      if (lane == 0 && wave + 8 * kk < border) bord[wave + 8 * kk] = acc;
    }
    ldsBarrier();
    if (aEnd >= 0 && aEnd < border) {
      double z = 0;
#pragma unroll
      for (int i = 0; i < kBorderMaxRows; ++i) z = __builtin_fma(liCol[i], (i < border) ? bord[i] : 0.0, z);   // (Lc^-T s)[a]
      p.yC[d + aEnd] = qEnd - z;
      p.vC[d + aEnd] = gEnd / hEnd;   // (the metric of the border rows: written by k_chol_border_prepare, or older)
    }
  }
  if constexpr (kBorder == 1) {
    // y2 = q - C^-1 (B y1): row a of B y1 on wave a, then one thread per border row
    double* bord = reinterpret_cast<double*>(fl + kCholFlagInts) + kCholBorderOff;
    if (wave < 4) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc = __builtin_fma(bEnd[k], (lane + 64 * k < d) ? rhs[lane + 64 * k] : 0.0, acc);
      acc = waveSum(acc);
      if (lane == 0) bord[wave] = acc;
    }
    ldsBarrier();
    if (t < border) {
      double w[4], z = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[i] = 0.0;
#pragma unroll
        for (int k = 0; k <= i; ++k) w[i] = __builtin_fma(LiB[i][k], bord[k], w[i]);
      }
      const double q = selectByRow(t, qB[0], qB[1], qB[2], qB[3]);
      const double ht = selectByRow(t, htB[0], htB[1], htB[2], htB[3]), sc = selectByRow(t, scB[0], scB[1], scB[2], scB[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) z = __builtin_fma(selectByRow(t, LiB[i][0], LiB[i][1], LiB[i][2], LiB[i][3]), w[i], z);   // (Lc^-T w)[t]
      p.yC[d + t] = q - z;
      p.vC[d + t] = p.gFull[d + t] / ht;
      if (fuseFinalize) {
        if (initScale) p.scaleC[d + t] = sc;
        p.htilC[d + t] = ht;
      }
    }
  }
#undef CHOL_SPINS
#undef CHOL_NOTE
#undef CHOL_STAMP
#undef CHOL_STAMP_FINE
}
template <int kBorder>
__global__ __launch_bounds__(kCholLdsThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chol_solve_lds(DeviceProblem p, int dpad, double mu, int initScale, int fuseFinalize, int border, const double* bscr) { k_chol_solve_lds_body<kBorder>(p, dpad, mu, initScale, fuseFinalize, border, bscr); }
// (batched form: blockIdx.y = the window of the batch, its problem and trust-region scalars from the slot table)
template <int kBorder>
__global__ __launch_bounds__(kCholLdsThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chol_solve_lds_batch(const BatchSlot* __restrict__ slots, int dpad, int fuseFinalize, int border, const double* bscr) {
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchFull)) return;
  k_chol_solve_lds_body<kBorder>(sl.p, dpad, sl.mu, sl.initScale, fuseFinalize, border, bscr);
}


// ================================================================ K6': reduced systems beyond the LDS-resident solver
// Blocked Cholesky over 64x64 blocks on many workgroups (d > 272, wide windows).  The right-hand side rides along as one
// extra row block below the matrix, so the forward substitution falls out of the block solves; k_big_back finishes with
// the backward substitution.
constexpr int kNB = 64;
constexpr int kBigTileLd = kPanelLd;                 // 16 x 17 LDS tiles
constexpr int kBigBlockLds = 16 * 16 * kBigTileLd;   // a 64x64 block as 4x4 tiles

// global (row-major, leading dimension ld) 64x64 block <-> 4x4 LDS tiles, 256 threads,
// through agent-scope relaxed atomics (sc1: the access itself is coherent across the XCDs' L2s), for blocks
// handed between workgroups of one launch without an L2 write-back / invalidate
__device__ __forceinline__ void loadBlock64Coherent(const double* g, int ld, double* tiles) {
  for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
    const int r = e >> 6, c = e & 63;
    tiles[((r >> 4) * 4 + (c >> 4)) * (16 * kBigTileLd) + (r & 15) * kBigTileLd + (c & 15)] =
        __hip_atomic_load(g + (size_t)r * ld + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void storeBlock64Coherent(double* g, int ld, const double* tiles) {
  for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
    const int r = e >> 6, c = e & 63;
    __hip_atomic_store(g + (size_t)r * ld + c, tiles[((r >> 4) * 4 + (c >> 4)) * (16 * kBigTileLd) + (r & 15) * kBigTileLd + (c & 15)],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// in-LDS factorisation of a 64x64 SPD block (4x4 tiles, diagonal tiles fully symmetric) by 4 waves:
// lower tiles <- L, strict upper triangle of the diagonal tiles <- L_tt^-T, dinv <- 1/L_ii.
// Per tile column, as in k_chol_solve_lds: phase P = panel solves (X^T = L^-1 A^T on MFMA, so that X lands in the operand
// layout); wave 0 takes the tile right below the diagonal and from it updates the NEXT diagonal tile in its registers;
// phase D = wave 0 factorises that tile out of its registers while waves 1-3 update the other trailing tiles.  Two barriers
// per tile column and no LDS round trip of the pivot tile (it was three barriers with the pivot tile through LDS:
// 25.3 k -> see DESIGN.md for the cycles per 64-column step).
__device__ void factor64(double* T, double* dinv, int* failFlag) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  auto tile = [&](int I, int J) { return T + (I * 4 + J) * (16 * kBigTileLd); };
  const int lrow = (lane >> 4) * kBigTileLd + (lane & 15);   // accumulator layout: + 4 rg ld
  const int lop = (lane & 15) * kBigTileLd + (lane >> 4);    // operand layout: + 4 q
  if (wave == 0) cholDiag16Reg(tile(0, 0), dinv, lane, failFlag);
  __syncthreads();
  for (int kb = 0; kb < 4; ++kb) {
    const double* D = tile(kb, kb);
    const int nR = 3 - kb;   // tiles below the diagonal in this tile column
    auto panelSolve = [&](double* A) {
      d4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = 4 * q + (lane >> 4), jj = lane & 15;
        const double a = A[lop + 4 * q];
        const double b = (jj > kk) ? D[kk * kBigTileLd + jj] : ((jj == kk) ? dinv[16 * kb + kk] : 0.0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) A[lop + 4 * rg] = acc[rg];
      return acc;
    };
    d4_t accD = {0, 0, 0, 0};
    if (wave == 0) {
      if (nR > 0) {
        const double* Cb = tile(kb + 1, kb + 1);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) accD[rg] = Cb[lrow + 4 * rg * kBigTileLd];
        const d4_t xT = panelSolve(tile(kb + 1, kb));
#pragma unroll
        for (int q = 0; q < 4; ++q) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-xT[q], xT[q], accD, 0, 0, 0);
      }
    } else if (wave < nR) {
      panelSolve(tile(kb + 1 + wave, kb));
    }
    __syncthreads();
    if (wave == 0) {
      if (nR > 0) cholDiag16Acc(accD, tile(kb + 1, kb + 1), dinv + 16 * (kb + 1), lane, failFlag);
    } else {
      // trailing tiles (I, J), kb + 1 <= J <= I < 4 without the next diagonal tile, dealt to waves 1-3 (both triangles of
      // the diagonal tiles: MFMA computes the full tile)
      int cnt = 0;
      for (int I = kb + 2; I < 4; ++I)
        for (int J = kb + 1; J <= I; ++J, ++cnt) {
          if (cnt % 3 != wave - 1) continue;
          double* Cb = tile(I, J);
          const double* A = tile(I, kb);
          const double* B = tile(J, kb);
          d4_t acc;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) acc[rg] = Cb[lrow + 4 * rg * kBigTileLd];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[lop + 4 * q], B[lop + 4 * q], acc, 0, 0, 0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) Cb[lrow + 4 * rg * kBigTileLd] = acc[rg];
        }
    }
    __syncthreads();
  }
}

// X = A L^-T for a 64x64 slab in LDS tiles (At, in place) against a factorised diagonal block (Dt, dinv) from
// factor64: wave w owns row tile w; block forward substitution over the tile columns j.  Everything between the first read
// and the last write of a row stays in registers: the tiles are taken TRANSPOSED into the accumulator layout, which is the B
// operand layout of a K = 16 product, so X_j^T = L_jj^-1 (A_j^T - sum_{i<j} L_ji X_i^T) chains from accumulator to operand
// without a trip through LDS (it was a store, a wave barrier and a reload per tile column: 6.3 k cycles per slab).
__device__ __forceinline__ void slabSolve64(const double* Dt, double* At, const double* dinv) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int lop = (lane & 15) * kBigTileLd + (lane >> 4);   // (row l & 15, column l >> 4): + 4 q
  auto dt = [&](int I, int J) { return Dt + (I * 4 + J) * (16 * kBigTileLd); };
  auto at = [&](int I, int J) { return At + (I * 4 + J) * (16 * kBigTileLd); };
  d4_t X[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d4_t T;
    const double* A = at(wave, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) T[r] = A[lop + 4 * r];   // A_j^T: lane (g, c) register r = A[c][g + 4r]
#pragma unroll
    for (int i = 0; i < j; ++i) {
      const double* Lji = dt(j, i);
#pragma unroll
      for (int q = 0; q < 4; ++q) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lji[lop + 4 * q], X[i][q], T, 0, 0, 0);
    }
    const double* D = dt(j, j);
    d4_t out = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = 4 * q + (lane >> 4), jj = lane & 15;
      const double b = (jj > kk) ? D[kk * kBigTileLd + jj] : ((jj == kk) ? dinv[16 * j + kk] : 0.0);
      out = __builtin_amdgcn_mfma_f64_16x16x4f64(b, T[q], out, 0, 0, 0);
    }
    X[j] = out;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double* A = at(wave, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) A[lop + 4 * r] = X[j][r];
  }
}

// M = p.cholL: (dpad + kNB) x dpad row-major; rows [dpad, dpad + kNB) hold the right-hand side in their first row
__global__ __launch_bounds__(256) void k_big_load(DeviceProblem p, int dpad, double mu, int initScale, int fuseFinalize, int* ready,
                                                  int nReady) {
  const int d = p.d;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nReady; i += gridDim.x * blockDim.x) ready[i] = 0;
  const size_t total = (size_t)(dpad + kNB) * dpad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gi = (int)(idx / dpad), gj = (int)(idx - (size_t)gi * dpad);
    double x = 0.0;
    if (gi < dpad) {
      x = (gi == gj) ? 1.0 : 0.0;
      if (gi < d && gj < d) {
        x = p.S[(size_t)max(gi, gj) * (p.ldS ? p.ldS : d) + min(gi, gj)];
        if (gi == gj && fuseFinalize) x += finalizeRow(p, gi, mu, initScale);
      }
    } else if (gi == dpad && gj < d) {
      x = p.gRed[gj];
    }
    p.cholL[idx] = x;
  }
}

// Left-looking over 64x64 blocks in ONE launch: task (I, J), I >= J (I = nb is the right-hand-side row block), owns
// block (I, J): C = A(I,J) - sum_{k<J} X(I,k) X(J,k)^T on MFMA (C in registers), then potrf (I == J, factor64) or
// X = C L_JJ^-T (slabSolve64).  Tasks are numbered column by column and dealt round-robin to <= 256 co-resident
// workgroups, each working through its tasks in increasing order; a task only depends on tasks with a smaller
// number, which are either finished, running elsewhere or earlier in the same workgroup, so the waits cannot cycle.
// Finished blocks cross XCD L2s: they are written and read with agent-scope relaxed atomics (sc1 accesses, coherent by
// themselves), the writer waits for its stores to complete before ready[I][J] is set, the reader polls ready[I][J]
// before it loads -- no L2 write-back / invalidate (buffer_wbl2 / buffer_inv cost ~10 us per hand-over here).  Every wait is bounded (kSpinMax polls): a stuck wait raises cholFail bit 8 (kCholFailSync: Window::solve throws) instead of hanging.
// (k_big_chol_chain below; its predecessors -- a launch pair per 64-wide panel, then plain block tasks without the
// critical-path workgroup -- are in the history of this file.)
constexpr int kSpinMax = 1 << 20;
// Workgroups of one solve: all must be co-resident (1 per CU: 104 KB of LDS each).  120 leaves room for a second
// solve of another handle / stream on the same GPU (2 x 120 <= 256 CUs); roots with more than kPersistWideTasks
// block tasks (> ~2000 unknowns) take the whole chip.  Beyond that the bounded waits turn a would-be deadlock into
// a flagged failure.
constexpr int kPersistMaxGrid = 120, kPersistWideGrid = 256, kPersistWideTasks = 512;
__device__ __forceinline__ bool pollReady(const int* f) {
  for (int it = 0; it < kSpinMax; ++it) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    __builtin_amdgcn_s_sleep(4);
  }
  return false;
}
// ---------------------------------------------------------------- tile Cholesky with a critical-path workgroup
// Same left-looking block tasks, but the two blocks per column that sit on the critical path -- the diagonal block
// (J, J) and the block below it (J+1, J) -- belong to ONE workgroup (the chain) that keeps L_JJ and X(J+1, J) in LDS
// from one column to the next: no hand-over on the critical path (a hand-over of a 32 KB block costs ~5 us,
// tools/ubench/handoff.hip).  The helpers do everything else:
//   H(I, J), I >= J + 2   the full block task of k_big_chol_tasks
//   PD(J), PS(J), J >= 2  blocks (J, J) and (J+1, J) minus the updates k <= J - 2, stored back in place and flagged,
//                         so that the chain only applies the last update (k = J - 1) itself
// Helper tasks are numbered stage by stage (stage s: H(., s), then PD(s+2), PS(s+2)) and dealt round-robin; chain
// step J needs helper stages <= J - 1, helper stage s needs chain steps <= s: no cycles, bounded waits as before.
struct TileLds { double *A, *B, *Dt, *dinv; int* seen; };
__device__ __forceinline__ void tileAccLoad(d4_t acc[4], const double* blk, int ld, bool coherent) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const double* q = blk + (size_t)(16 * wave + (lane >> 4) + 4 * rg) * ld + 16 * tj + (lane & 15);
      acc[tj][rg] = coherent ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
}
__device__ __forceinline__ void tileAccToLds(const d4_t acc[4], double* tiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      tiles[(wave * 4 + tj) * (16 * kBigTileLd) + ((lane >> 4) + 4 * rg) * kBigTileLd + (lane & 15)] = acc[tj][rg];
}
// acc -= X_a(row tile of this wave) X_b^T, both 64x64 blocks as LDS tiles
__device__ __forceinline__ void tileMfmaSub(d4_t acc[4], const double* Xa, const double* Xb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const double* A = Xa + (wave * 4 + kt) * (16 * kBigTileLd);
      const double* B = Xb + (tj * 4 + kt) * (16 * kBigTileLd);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[(lane & 15) * kBigTileLd + 4 * q + (lane >> 4)],
                                                       B[(lane & 15) * kBigTileLd + 4 * q + (lane >> 4)], acc[tj], 0, 0, 0);
    }
}
__device__ __forceinline__ void tileWait(const int* flag, bool& gaveUp, int* fail) {
  if (threadIdx.x == 0 && !gaveUp && !pollReady(flag)) { atomicOr(fail, 8); gaveUp = true; }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void tilePublish(int* flag) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this thread's coherent stores have completed (vmcnt 0)
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// acc -= sum_{k < kEnd} X(I,k) X(J,k)^T, blocks taken as they become ready
__device__ __forceinline__ void tileAccumulate(d4_t acc[4], const TileLds& L, double* M, int dpad, const int* ready, int nb,
                                               int I, int J, int kEnd, bool& gaveUp, int* fail) {
  const int tid = threadIdx.x;
  if (tid == 0) L.seen[0] = kEnd;
  __syncthreads();
  int firstMissing = kEnd;
  for (int k = tid; k < kEnd; k += blockDim.x) {   // one parallel look: usually all but the last columns are there
    const bool ok = __hip_atomic_load(ready + I * nb + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
                    __hip_atomic_load(ready + J * nb + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (!ok) firstMissing = min(firstMissing, k);
  }
  if (firstMissing < kEnd) atomicMin(L.seen, firstMissing);
  __syncthreads();
  const int kSafe = L.seen[0];
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  __syncthreads();
  for (int k = 0; k < kEnd; ++k) {
    if (k >= kSafe) {
      if (tid == 0 && !gaveUp) {
        const bool ok = pollReady(ready + I * nb + k) && pollReady(ready + J * nb + k);
        if (!ok) { atomicOr(fail, 8); gaveUp = true; }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    loadBlock64Coherent(M + (size_t)(kNB * I) * dpad + kNB * k, dpad, L.A);
    if (I != J) loadBlock64Coherent(M + (size_t)(kNB * J) * dpad + kNB * k, dpad, L.B);
    __syncthreads();
    tileMfmaSub(acc, L.A, (I == J) ? L.A : L.B);
    __syncthreads();
  }
}
#ifdef SVIN_CHAIN_TIMING
__device__ int g_chainCount;
#endif
__global__ __launch_bounds__(256) void k_big_chol_chain(DeviceProblem p, int dpad, double* dinvG, double* diagF, int* ready) {
  extern __shared__ double smem[];
  TileLds L;
  L.A = smem; L.B = smem + kBigBlockLds; L.Dt = smem + 2 * kBigBlockLds; L.dinv = L.Dt + kBigBlockLds;
  L.seen = reinterpret_cast<int*>(L.dinv + kNB);
  const int tid = threadIdx.x;
  const int nb = dpad / kNB;
  double* M = p.cholL;
  int* pd = ready + (nb + 1) * nb;   // PD(J) / PS(J) done
  int* ps = pd + nb;
  int* fail = &p.scal->cholFail;
  bool gaveUp = false;
  d4_t acc[4];
  if (blockIdx.x == 0) {
    // The critical-path workgroup.  Per block column it used to pay five exposed round trips (three block loads, two
    // store drains ahead of a flag); now everything a step reads from other workgroups is requested in ONE go at its top
    // (the helpers run ahead, their flags are normally long set), and a flag is raised where the wait for its stores is
    // free: X(J, J-1) of the previous step together with that batch of loads (the diagonal factor is still flagged at once:
    // deferring it past the update of the block below made the helpers late for the next step).
    double* S = L.B;    // X(J, J-1) from the previous column
    double* W = L.A;    // work block
    d4_t accB[4];
    double wreg[16];
#ifdef SVIN_CHAIN_TIMING
    long long cT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c0 = __builtin_readcyclecounter(), c1;
#define CHT(i) do { c1 = __builtin_readcyclecounter(); cT[i] += c1 - c0; c0 = c1; } while (0)
#else
#define CHT(i) do { } while (0)
#endif
    for (int J = 0; J < nb; ++J) {
      // flags of this step's inputs: PD(J), PS(J) (blocks minus the updates k <= J - 2) and X(J+1, J-1)
      if (J >= 1) {
        if (tid == 0 && !gaveUp) {
          bool ok = pollReady(ready + (J + 1) * nb + (J - 1));
          if (J >= 2) ok = ok && pollReady(pd + J) && pollReady(ps + J);
          if (!ok) { atomicOr(fail, 8); gaveUp = true; }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      CHT(0);
      tileAccLoad(acc, M + (size_t)(kNB * J) * dpad + kNB * J, dpad, true);
      tileAccLoad(accB, M + (size_t)(kNB * (J + 1)) * dpad + kNB * J, dpad, true);   // (row block J + 1 <= nb: the last one is the right-hand side)
      if (J >= 1) {
        const double* g = M + (size_t)(kNB * (J + 1)) * dpad + kNB * (J - 1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int e = tid + 256 * m;
          wreg[m] = __hip_atomic_load(g + (size_t)(e >> 6) * dpad + (e & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // X(J, J-1) of the previous step: its stores drain together with the loads above
        tilePublish(ready + J * nb + (J - 1));
      }
      // diagonal block
      CHT(1);
      if (J >= 1) tileMfmaSub(acc, S, S);
      tileAccToLds(acc, L.Dt);
      __syncthreads();
      CHT(2);
      factor64(L.Dt, L.dinv, fail);
      CHT(3);
      storeBlock64Coherent(diagF + (size_t)(kNB * J) * kNB, kNB, L.Dt);
      if (tid < kNB) __hip_atomic_store(dinvG + kNB * J + tid, L.dinv[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the block below it (its operand goes to LDS first: the store drain of the flag below then overlaps something)
      if (J >= 1) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int e = tid + 256 * m, r = e >> 6, c = e & 63;
          W[((r >> 4) * 4 + (c >> 4)) * (16 * kBigTileLd) + (r & 15) * kBigTileLd + (c & 15)] = wreg[m];
        }
      }
      tilePublish(ready + J * nb + J);   // right away: the helpers' column-J solves sit on the path to the next step's inputs
      CHT(4);
      if (J >= 1) {
        tileMfmaSub(accB, W, S);
        __syncthreads();
      }
      tileAccToLds(accB, W);
      __syncthreads();
      CHT(5);
      slabSolve64(L.Dt, W, L.dinv);
      __syncthreads();
      CHT(6);
      storeBlock64Coherent(M + (size_t)(kNB * (J + 1)) * dpad + kNB * J, dpad, W);
      CHT(7);
      double* t = S; S = W; W = t;
    }
    tilePublish(ready + nb * nb + (nb - 1));
#ifdef SVIN_CHAIN_TIMING
    if (tid == 0 && atomicAdd(&g_chainCount, 1) == 6)
      printf("[chain nb %d] flags %lld  loads+publish %lld  mfmaSS %lld  factor64 %lld  storeD+publish %lld  W+mfma %lld  slabSolve %lld  storeX %lld\n",
             nb, cT[0], cT[1], cT[2], cT[3], cT[4], cT[5], cT[6], cT[7]);
#endif
#undef CHT
    return;
  }
  // helpers
  int s = 0, stageStart = 0;
  auto stageCount = [&](int st) { return max(nb - st - 1, 0) + ((st + 2 <= nb - 1) ? 2 : 0); };
  int nTasks = 0;
  for (int st = 0; st < nb; ++st) nTasks += stageCount(st);
  for (int task = blockIdx.x - 1; task < nTasks; task += gridDim.x - 1) {
    while (task >= stageStart + stageCount(s)) { stageStart += stageCount(s); ++s; }
    const int li = task - stageStart, nH = max(nb - s - 1, 0);
    if (li < nH) {   // H(I, s)
      const int I = s + 2 + li, J = s;
      tileAccLoad(acc, M + (size_t)(kNB * I) * dpad + kNB * J, dpad, false);
      tileAccumulate(acc, L, M, dpad, ready, nb, I, J, J, gaveUp, fail);
      tileWait(ready + J * nb + J, gaveUp, fail);
      loadBlock64Coherent(diagF + (size_t)(kNB * J) * kNB, kNB, L.Dt);
      if (tid < kNB) L.dinv[tid] = __hip_atomic_load(dinvG + kNB * J + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tileAccToLds(acc, L.A);
      __syncthreads();
      slabSolve64(L.Dt, L.A, L.dinv);
      __syncthreads();
      storeBlock64Coherent(M + (size_t)(kNB * I) * dpad + kNB * J, dpad, L.A);
      tilePublish(ready + I * nb + J);
    } else {         // PD(J) / PS(J): everything but the last update, back in place
      const int J = s + 2, I = J + (li - nH);
      tileAccLoad(acc, M + (size_t)(kNB * I) * dpad + kNB * J, dpad, false);
      tileAccumulate(acc, L, M, dpad, ready, nb, I, J, J - 1, gaveUp, fail);
      tileAccToLds(acc, L.A);
      __syncthreads();
      storeBlock64Coherent(M + (size_t)(kNB * I) * dpad + kNB * J, dpad, L.A);
      tilePublish((li - nH == 0 ? pd : ps) + J);
    }
  }
}

// backward substitution L^T y = y' (y' = first row of the right-hand-side block) in super-panels of kBackSpan columns,
// from the last to the first.  For the columns [c0, c1) of a super-panel the rows below it (i >= c1, already solved)
// are a plain matrix-vector product spread over many workgroups (k_big_back_gemv: partial sums per 512-row chunk,
// kept in the unused rows 1.. of the right-hand-side block, summed in fixed order); the triangle inside the
// super-panel is one workgroup (k_big_back).
constexpr int kBackSpan = 512;
__global__ __launch_bounds__(512) void k_big_back_gemv(DeviceProblem p, int dpad, int c0, int c1) {
  __shared__ double part[8 * 64];
  double* M = p.cholL;
  const double* y = M + (size_t)dpad * dpad;
  const int t = threadIdx.x, c = t & 63, stripe = t >> 6;
  const int col = c0 + 64 * blockIdx.x + c;
  const int i0 = c1 + kBackSpan * blockIdx.y, i1 = min(dpad, i0 + kBackSpan);
  double s = 0;
  for (int i = i0 + stripe; i < i1; i += 8) s += M[(size_t)i * dpad + col] * y[i];
  part[stripe * 64 + c] = s;
  __syncthreads();
  if (t < 64) {
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += part[k * 64 + t];
    M[(size_t)(dpad + 1 + blockIdx.y) * dpad + col] = acc;
  }
}
__global__ __launch_bounds__(512) void k_big_back(DeviceProblem p, int dpad, int c0, int c1, int nChunks, const double* dinvG,
                                                  const double* diagF) {
  extern __shared__ double smem[];
  double* y = smem - c0;            // y[c0 .. c1) lives in smem[0 .. c1 - c0)
  double* part = smem + (c1 - c0);  // 8 x 64 partial sums
  double* Fl = part + 8 * 64;       // this panel's factorised diagonal block (64 x 65: conflict-free columns) + 1/L_ii
  double* dl = Fl + kNB * (kNB + 1);
  constexpr int kFld = kNB + 1;
  const int t = threadIdx.x, d = p.d;
  double* M = p.cholL;
  for (int i = c0 + t; i < c1; i += blockDim.x) {
    double v = M[(size_t)dpad * dpad + i];
    for (int k = 0; k < nChunks; ++k) v -= M[(size_t)(dpad + 1 + k) * dpad + i];
    y[i] = v;
  }
  __syncthreads();
  for (int k0 = c1 - kNB; k0 >= c0; k0 -= kNB) {
    // the panel's diagonal factor goes to LDS while the sums below run (it is read 4 x 2 times, serially, afterwards)
    for (int e = t; e < kNB * kNB; e += blockDim.x) Fl[(e >> 6) * kFld + (e & 63)] = diagF[(size_t)k0 * kNB + e];
    if (t < kNB) dl[t] = dinvG[k0 + t];
    // s_c = sum_{k0 + 64 <= i < c1} L[i][k0 + c] y[i]: 8 row stripes x 64 columns
    {
      const int c = t & 63, stripe = t >> 6;
      double s0 = 0, s1 = 0;
      int i = k0 + kNB + stripe;
      for (; i + 8 < c1; i += 16) {
        s0 += M[(size_t)i * dpad + k0 + c] * y[i];
        s1 += M[(size_t)(i + 8) * dpad + k0 + c] * y[i + 8];
      }
      if (i < c1) s0 += M[(size_t)i * dpad + k0 + c] * y[i];
      part[stripe * 64 + c] = s0 + s1;
    }
    __syncthreads();
    if (t < kNB) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += part[k * 64 + t];
      y[k0 + t] -= s;
    }
    __syncthreads();
    // 64x64 triangular solve L_kk^T y_k = rhs with the stored tile inverses, tile by tile from the last
    if (t < 64) {
      const int lane = t;
      for (int tt = 3; tt >= 0; --tt) {
        const int b0 = k0 + 16 * tt;
        const int li = lane & 15;
        // y_t = L_tt^-T rhs_t : (L^-T)[li][r] = Linv[r][li], stored at tile(tt,tt)[li][r] for r > li
        double yv = y[b0 + li] * dl[16 * tt + li];
#pragma unroll
        for (int r = 1; r < 16; ++r) {
          const double term = Fl[(16 * tt + li) * kFld + 16 * tt + r] * y[b0 + r];
          yv += (r > li) ? term : 0.0;
        }
        waveSync();
        if (lane < 16) y[b0 + lane] = yv;
        waveSync();
        // remove this tile's contribution from the earlier tiles of the panel: rhs_u -= L[b0 + k][u] y[b0 + k]
        for (int u = lane; u < 16 * tt; u += 64) {
          double s = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) s += Fl[(16 * tt + k) * kFld + u] * y[b0 + k];
          y[k0 + u] -= s;
        }
        waveSync();
      }
    }
    __syncthreads();
  }
  for (int i = c0 + t; i < c1; i += blockDim.x) {
    M[(size_t)dpad * dpad + i] = y[i];   // the solved part of y, read by the super-panels before this one
    if (i < d) { p.yC[i] = y[i]; p.vC[i] = p.gFull[i] / p.htilC[i]; }  // + steepest-descent direction
  }
}

// ================================================================ K6''': the speed / bias chain eliminated ahead of the blocked Cholesky
// In a wide window the reduced system is [poses + extrinsics (dC rows, dense) | speed / bias blocks (9 rows each)], and the
// speed / bias part is a CHAIN: block b couples with b - 1 and b + 1 only (the IMU factors), whatever the landmarks do to the
// pose part.  At configs[3] (64 key frames) that is 576 of the 960 unknowns -- nine of the fifteen 64-column steps of
// k_big_chol_chain, each ~23 us of critical path, spent on a block-tridiagonal matrix.  Eliminating the chain first leaves a
// dC x dC system for the blocked Cholesky:
//     S_ss = L L^T,   Y = L^-1 [S_sk | g_s],   S_kk' = S_kk - Y_k^T Y_k,   g_k' = g_k - Y_k^T y_g,   x_s = L^-T (y_g - Y_k x_k)
// A chain factorised end to end is 64 dependent 9x9 steps (that is what lost at d = 150 in round 2).  Here the elimination
// order is cyclic reduction: level by level the blocks b = s (mod 2 s), s = 1, 2, 4, ..., whose two neighbours b - s, b + s
// survive the level; all blocks of a level are independent, so the chain costs log2(n) + 1 dependent block steps.  Per block
// the factor is three 9x9 matrices: G = L_bb^-1, F_lo = G S(b, b-s), F_hi = G S(b+s, b)^T (S = the matrix as that level sees
// it) -- L's two off-diagonal blocks are F_lo^T and F_hi^T.
//   k_sb_factor    one workgroup: the chain's factor (levels in LDS, one barrier-separated pass per level)
//   k_sb_forward   8 columns of [S_sk | g_s] per workgroup, the level sweep in LDS: Y
//   k_sb_load      replaces k_big_load: M = S_kk - Y^T Y on MFMA (one 16x16 tile per workgroup, K split over its 4 waves),
//                  right-hand side row, identity padding, flags
//   k_sb_back      after k_big_back: t = y_g - Y_k x_k over many workgroups, the last one to finish walks the levels backwards
struct SbElimArgs {
  int n;        // chain blocks: rows dC + 9 b of the reduced system
  int dK;       // kept unknowns (= dC)
  int dp;       // dK rounded up to 64: the blocked solver's matrix is (dp + 64) x dp
  int ldY;      // leading dimension of Y: dK + 1 columns (the last one is the right-hand side), rounded up to 16
  int rowsY;    // 9 n rounded up to 4 (the pad rows are zero)
  double* Lf;   // n records of kSbRec doubles
  double* Y;
  double* tvec; // 9 n
  int* counter; // last-workgroup ticket of k_sb_back
  int compact;  // the kept system goes to Sout / gOut (a single-workgroup solver takes it from there) instead of the blocked solver's matrix
  int ldOut;    // leading dimension (= rows) of Sout: dK rounded up to 16, zero beyond dK
  double *Sout, *gOut;
};
constexpr int kSbRec = 264, kSbG = 0, kSbFlo = 88, kSbFhi = 176;   // 9x9 row-major each (16-byte aligned starts)
constexpr int kSbMaxChain = 64;                                    // k_sb_factor keeps the whole chain in LDS

// 1 / sqrt(s) to the last bits: the hardware estimate (v_rsq_f64, ~2^-26) refined by a third-order and a second-order step
__device__ __forceinline__ double rsqrtRefined(double s) {
  const double y0 = __builtin_amdgcn_rsq(s);
  const double e0 = fma(-s * y0, y0, 1.0);
  const double y1 = fma(y0 * e0, fma(e0, 0.375, 0.5), y0);
  const double e1 = fma(-s * y1, y1, 1.0);
  return fma(0.5 * y1, e1, y1);
}
// 9x9 SPD block (row-major in LDS) -> its Cholesky factor, packed lower triangle L[i (i + 1) / 2 + j] with the RECIPROCAL diagonal
// (1 / L_ii at i (i + 3) / 2), one thread, all in registers
__device__ __forceinline__ bool cholPacked9(const double* Din, double* Lout) {
  double L[45];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = Din[i * 9 + j];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    double s = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
    if (!(s > 0.0)) { ok = false; s = 1.0; }
    const double rs = rsqrtRefined(s);
    L[j * (j + 1) / 2 + j] = rs;
#pragma unroll
    for (int i = j + 1; i < 9; ++i) {
      double v = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      L[i * (i + 1) / 2 + j] = v * rs;
    }
  }
#pragma unroll
  for (int q = 0; q < 45; ++q) Lout[q] = L[q];
  return ok;
}
// column j of G = L^-1 from the packed factor: G_jj = 1 / L_jj, G_ij = -(1 / L_ii) sum_{j <= k < i} L_ik G_kj
__device__ __forceinline__ void inverseColumn9(const double* Lp, int j, double* Gout) {
  double X[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) s += Lp[i * (i + 1) / 2 + k] * X[k];   // (X_k = 0 above the diagonal, k < j: no test, no branch around a load)
    X[i] = (i == j) ? Lp[i * (i + 3) / 2] : -Lp[i * (i + 3) / 2] * s;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) Gout[i * 9 + j] = X[i];
}

constexpr int kSbLdsRec = 243;   // records in LDS: G | F_lo | F_hi back to back (an odd stride: eight blocks per wave, eight banks apart)
constexpr int kSbFactorThreads = 576;   // 9 x 64: a level of 32 eliminations has 576 column tasks and 576 row tasks
__global__ __launch_bounds__(kSbFactorThreads) void k_sb_factor(DeviceProblem p, SbElimArgs a, double mu, int initScale, int fuseFinalize) {
  extern __shared__ double smem[];
  const int n = a.n, t = threadIdx.x, nT = blockDim.x;
  double* D = smem;                         // [n][81] diagonal blocks as the current level sees them
  double* C = D + n * 81;                   // [n][81] coupling of block b with its lower active neighbour: rows b, columns b - s
  double* G = C + n * 81;                   // [(n + 1) / 2][81]   this level's factors
  double* F = G + ((n + 1) / 2) * 81;       // [(n + 1) / 2][2][81]
  const int ld = p.ldS ? p.ldS : p.d, r0 = a.dK;
  if (t == 0) *a.counter = 0;
#ifdef SVIN_SB_TIMING
  long long qT[5] = {0, 0, 0, 0, 0}, q0 = __builtin_readcyclecounter(), q1;
  int lvl = 0;
#define SBT(i) do { q1 = __builtin_readcyclecounter(); qT[i] += q1 - q0; if (t == 0 && i > 0) printf("   level %d phase %d: %lld\n", lvl, i, q1 - q0); q0 = __builtin_readcyclecounter(); } while (0)
#else
#define SBT(i) do { } while (0)
#endif
  {
    // all loads of a thread are issued before the first one is used (a load and a store per round would pay the L2 latency
    // nine times over)
    constexpr int kRounds = (kSbMaxChain * 81 + kSbFactorThreads - 1) / kSbFactorThreads;
    double dv[kRounds], cv[kRounds];
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int e = t + k * kSbFactorThreads;
      dv[k] = 0.0; cv[k] = 0.0;
      if (e < n * 81) {
        const int b = e / 81, i = (e % 81) / 9, j = e % 9;
        const int gi = r0 + 9 * b + i, gj = r0 + 9 * b + j;
        dv[k] = p.S[(size_t)max(gi, gj) * ld + min(gi, gj)];
        if (b > 0) cv[k] = p.S[(size_t)gi * ld + (gj - 9)];
      }
    }
    static_assert(9 * kSbMaxChain <= kSbFactorThreads, "one diagonal entry per thread");
    double damp0 = 0.0;
    if (fuseFinalize && t < 9 * n) damp0 = finalizeRow(p, r0 + t, mu, initScale);
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int e = t + k * kSbFactorThreads;
      if (e < n * 81) { D[e] = dv[k]; C[e] = cv[k]; }
    }
    __syncthreads();
    if (t < 9 * n) D[(t / 9) * 81 + (t % 9) * 10] += damp0;
  }
  __syncthreads();
  SBT(0);
  // (the barriers below are LDS-only: the records' global stores drain behind them -- a __syncthreads would wait for every one)
  for (int s = 1;; s *= 2) {
    const bool last = s >= n;                                   // block 0 alone is left
    const int nE = last ? 1 : (n - s + 2 * s - 1) / (2 * s);    // eliminated now: b = s + 2 s e < n
    if (t < nE) {   // the factor of D[b], packed, into the (still free) F slot of this elimination
      const int b = last ? 0 : s + 2 * s * t;
      if (!cholPacked9(D + b * 81, F + t * 162)) atomicOr(&p.scal->cholFail, 1);
    }
    ldsBarrier();
    SBT(1);
    if (t < nE * 9) inverseColumn9(F + (t / 9) * 162, t % 9, G + (t / 9) * 81);
    ldsBarrier();
    SBT(4);
    // F_lo = G C[b], F_hi = G C[b + s]^T on MFMA (9x9 blocks in 16x16x4 tiles, K = 12: three instructions a product, one
    // elimination per wave at a time -- six operand reads a lane instead of the 54 of a column per thread); G and F also go to
    // the record of block b.  Operand lane l: row / column l & 15, k = 4 q + (l >> 4); accumulator register rg: row (l >> 4) + 4 rg.
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, nW = nT >> 6, li = lane & 15, lk = lane >> 4;
    for (int e = wave; e < nE; e += nW) {
      const int b = last ? 0 : s + 2 * s * e;
      const double* Ge = G + e * 81;
      double* rec = a.Lf + (size_t)b * kSbRec;
      for (int q = lane; q < 81; q += 64) rec[kSbG + q] = Ge[q];
      if (last) break;
      const bool hasHi = b + s < n;
      const double* Clo = C + b * 81;
      const double* Chi = C + (hasHi ? b + s : b) * 81;
      d4_t lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int k = 4 * q + lk;
        const bool ok = k < 9 && li < 9;
        const double one = ok ? 1.0 : 0.0;
        const double g = Ge[ok ? li * 9 + k : 0] * one;
        const double cl = Clo[ok ? k * 9 + li : 0] * one;
        const double ch = Chi[ok ? li * 9 + k : 0] * (hasHi ? one : 0.0);
        lo = __builtin_amdgcn_mfma_f64_16x16x4f64(g, cl, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f64_16x16x4f64(g, ch, hi, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = lk + 4 * rg;
        if (i < 9 && li < 9) {
          F[(e * 2) * 81 + i * 9 + li] = lo[rg];
          F[(e * 2 + 1) * 81 + i * 9 + li] = hi[rg];
          rec[kSbFlo + i * 9 + li] = lo[rg];
          rec[kSbFhi + i * 9 + li] = hi[rg];
        }
      }
    }
    ldsBarrier();
    SBT(2);
    if (last) break;
    // the survivors m = 0 (mod 2 s) collect: D[m] -= F_hi(m - s)^T F_hi(m - s) + F_lo(m + s)^T F_lo(m + s), and their new
    // lower neighbour is m - 2 s: C[m] = -F_hi(m - s)^T F_lo(m - s).  The same tiles: a survivor per wave at a time, D[m] in the
    // accumulator, nine MFMA.  (Per-thread versions -- an entry, then a row of an output per thread -- were bound by their LDS
    // reads: 2 and 1.1 per multiply-add, 7 200 and 6 000 cycles a level.)
    const int nR = (n + 2 * s - 1) / (2 * s);
    for (int e = wave; e < nR; e += nW) {
      const int m = 2 * s * e;
      const bool below = m >= 2 * s, above = m + s < n;             // eliminated neighbours m - s and m + s
      const double* Fh = F + ((below ? (m - 2 * s) / (2 * s) : 0) * 2 + 1) * 81;   // F_hi(m - s); F_lo(m - s) sits right before it
      const double* Fa = F + ((above ? m / (2 * s) : 0) * 2) * 81;                 // F_lo(m + s)
      double* Dm = D + m * 81;
      d4_t accD, accC = {0, 0, 0, 0};
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = lk + 4 * rg;
        const bool ok = i < 9 && li < 9;
        accD[rg] = Dm[ok ? i * 9 + li : 0] * (ok ? 1.0 : 0.0);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int k = 4 * q + lk;
        const bool ok = k < 9 && li < 9;
        const int idx = ok ? k * 9 + li : 0;
        const double fh = Fh[idx] * ((ok && below) ? 1.0 : 0.0);
        const double fs = (Fh - 81)[idx] * ((ok && below) ? 1.0 : 0.0);
        const double fa = Fa[idx] * ((ok && above) ? 1.0 : 0.0);
        accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-fh, fh, accD, 0, 0, 0);
        accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-fa, fa, accD, 0, 0, 0);
        accC = __builtin_amdgcn_mfma_f64_16x16x4f64(-fh, fs, accC, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = lk + 4 * rg;
        if (i < 9 && li < 9) {
          Dm[i * 9 + li] = accD[rg];
          if (below) C[m * 81 + i * 9 + li] = accC[rg];
        }
      }
    }
    ldsBarrier();
    SBT(3);
#ifdef SVIN_SB_TIMING
    ++lvl;
#endif
  }
#ifdef SVIN_SB_TIMING
  if (t == 0) printf("[k_sb_factor n %d] load %lld  cholesky (all levels) %lld  inverse %lld  F %lld  survivors %lld\n", n, qT[0], qT[1], qT[4], qT[2], qT[3]);
#endif
#undef SBT
}

constexpr int kSbCols = 8;   // records in LDS: G | F_lo | F_hi back to back (an odd stride: eight blocks per wave, eight banks apart)
__global__ __launch_bounds__(256) void k_sb_forward(DeviceProblem p, SbElimArgs a) {
  extern __shared__ double smem[];
  double* w = smem;                                   // [rowsY][8]
  double* R = smem + (size_t)a.rowsY * kSbCols;       // [n][243]: every workgroup sweeps all levels, so the records come to LDS once
  const int t = threadIdx.x, c = t & (kSbCols - 1), q = t / kSbCols, nQ = blockDim.x / kSbCols, n = a.n;
  const int col = blockIdx.x * kSbCols + c;
  const int ld = p.ldS ? p.ldS : p.d;
  for (int base = 0; base < n * kSbLdsRec; base += 16 * 256) {   // sixteen loads in flight per thread
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = base + k * 256 + t;
      const int b = e / kSbLdsRec, r = e % kSbLdsRec;
      v[k] = (e < n * kSbLdsRec) ? a.Lf[(size_t)b * kSbRec + (r < 81 ? r : (r < 162 ? kSbFlo + r - 81 : kSbFhi + r - 162))] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = base + k * 256 + t;
      if (e < n * kSbLdsRec) R[e] = v[k];
    }
  }
  {
    // this workgroup's columns of [S_sk | g_s]: six rows in flight per thread (a load and an LDS store per trip pays the L2
    // latency rowsY / 32 = 18 times in a row)
    const double* src = (col < a.dK) ? p.S + (size_t)a.dK * ld + col : p.gRed + a.dK;
    const size_t stride = (col < a.dK) ? (size_t)ld : 1;
    const double keep = (col <= a.dK) ? 1.0 : 0.0;
    for (int r0 = q; r0 < a.rowsY; r0 += 6 * nQ) {
      double v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int r = r0 + k * nQ;
        v[k] = src[(size_t)min(r, 9 * n - 1) * stride];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int r = r0 + k * nQ;
        if (r < a.rowsY) w[r * kSbCols + c] = (r < 9 * n) ? v[k] * keep : 0.0;
      }
    }
  }
  __syncthreads();
  auto solveBlock = [&](int b) {   // w_b <- G_b w_b
    const double* Gb = R + b * kSbLdsRec;
    double v[9], y[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = w[(9 * b + k) * kSbCols + c];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k <= i; ++k) acc += Gb[i * 9 + k] * v[k];
      y[i] = acc;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) w[(9 * b + k) * kSbCols + c] = y[k];
  };
  for (int s = 1; s < n; s *= 2) {
    const int nE = (n - s + 2 * s - 1) / (2 * s);
    for (int e = q; e < nE; e += nQ) solveBlock(s + 2 * s * e);
    __syncthreads();
    const int nR = (n + 2 * s - 1) / (2 * s);
    for (int e = q; e < nR; e += nQ) {   // w_m -= F_hi(m - s)^T y_{m-s} + F_lo(m + s)^T y_{m+s}
      const int m = 2 * s * e;
      double acc[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[j] = 0.0;
      // (a missing neighbour reads the other one's record against a zero vector: the same code in every lane)
      const bool below = m >= s, above = m + s < n;
      const int bB = below ? m - s : m + s, bA = above ? m + s : m - s;
      const double wB = below ? 1.0 : 0.0, wA = above ? 1.0 : 0.0;
      double y[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) y[i] = w[(9 * bB + i) * kSbCols + c] * wB;
      {
        const double* Fh = R + bB * kSbLdsRec + 162;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
          for (int j = 0; j < 9; ++j) acc[j] += Fh[i * 9 + j] * y[i];
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) y[i] = w[(9 * bA + i) * kSbCols + c] * wA;
      {
        const double* Fl = R + bA * kSbLdsRec + 81;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
          for (int j = 0; j < 9; ++j) acc[j] += Fl[i * 9 + j] * y[i];
      }
#pragma unroll
      for (int j = 0; j < 9; ++j) w[(9 * m + j) * kSbCols + c] -= acc[j];
    }
    __syncthreads();
  }
  if (q == 0) solveBlock(0);
  __syncthreads();
  if (col < a.ldY)
    for (int r = q; r < a.rowsY; r += nQ) a.Y[(size_t)r * a.ldY + col] = w[r * kSbCols + c];
}

__global__ __launch_bounds__(256) void k_sb_load(DeviceProblem p, SbElimArgs a, double mu, int initScale, int fuseFinalize, int* ready,
                                                 int nReady) {
  __shared__ double red[3 * 256];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int dK = a.dK, dp = a.dp, nT = (dK + 15) / 16, nTiles = nT * (nT + 1) / 2, nRhs = (dK + 15) / 16;
  const int ld = p.ldS ? p.ldS : p.d;
  double* M = a.compact ? a.Sout : p.cholL;
  const int ldM = a.compact ? a.ldOut : dp;
  for (int i = blockIdx.x * blockDim.x + t; i < nReady; i += gridDim.x * blockDim.x) ready[i] = 0;
  // everything outside the tiles and the right-hand side row: identity padding, zero scratch rows (the compact matrix is all tiles)
  const size_t total = a.compact ? 0 : (size_t)(dp + kNB) * dp;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + t; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int gi = (int)(idx / dp), gj = (int)(idx - (size_t)gi * dp);
    if (gi < 16 * nT && gj < 16 * nT) continue;
    if (gi == dp && gj < dK) continue;
    M[idx] = (gi < dp && gi == gj) ? 1.0 : 0.0;
  }
  if ((int)blockIdx.x < nTiles) {
    int I = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) ++I;
    while (I * (I + 1) / 2 > (int)blockIdx.x) --I;
    const int J = blockIdx.x - I * (I + 1) / 2;
    d4_t acc = {0, 0, 0, 0};
    // wave 0 finishes the tile: its S entries and the metric of its diagonal entries are requested before the products
    double sPre[4] = {0, 0, 0, 0}, dampPre[4] = {0, 0, 0, 0};
    if (wave == 0) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int gi = 16 * I + (lane >> 4) + 4 * rg, gj = 16 * J + (lane & 15);
        sPre[rg] = p.S[(size_t)min(max(gi, gj), dK - 1) * ld + min(min(gi, gj), dK - 1)];
        if (gi == gj && gi < dK && fuseFinalize) dampPre[rg] = finalizeRow(p, gi, mu, initScale);
      }
    }
    const int steps = a.rowsY / 4, s0 = steps * wave / 4, s1 = steps * (wave + 1) / 4;
    const double* ya = a.Y + (size_t)(lane >> 4) * a.ldY + 16 * I + (lane & 15);
    const double* yb = a.Y + (size_t)(lane >> 4) * a.ldY + 16 * J + (lane & 15);
    const size_t stride = (size_t)4 * a.ldY;
    ya += s0 * stride; yb += s0 * stride;
    int st = s0;
    for (; st + 12 <= s1; st += 12, ya += 12 * stride, yb += 12 * stride) {   // 24 loads in flight per wave: the loop is L2 latency, not MFMA
      double av[12], bv[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) { av[k] = ya[k * stride]; bv[k] = yb[k * stride]; }
#pragma unroll
      for (int k = 0; k < 12; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], bv[k], acc, 0, 0, 0);
    }
    for (; st < s1; ++st, ya += stride, yb += stride) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[0], yb[0], acc, 0, 0, 0);
    if (wave > 0) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) red[(wave - 1) * 256 + rg * 64 + lane] = acc[rg];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const double yy = ((acc[rg] + red[rg * 64 + lane]) + red[256 + rg * 64 + lane]) + red[512 + rg * 64 + lane];
        const int gi = 16 * I + (lane >> 4) + 4 * rg, gj = 16 * J + (lane & 15);
        if (I == J && gj > gi) continue;   // the diagonal tiles are mirrored from their lower triangle
        double x = (gi == gj && !a.compact) ? 1.0 : 0.0;   // (the single-workgroup solvers pad with the identity themselves)
        if (gi < dK && gj < dK) x = sPre[rg] - yy + dampPre[rg];
        M[(size_t)gi * ldM + gj] = x;
        M[(size_t)gj * ldM + gi] = x;
      }
    }
  } else if ((int)blockIdx.x < nTiles + nRhs) {
    // g_k' = g_k - Y_k^T y_g: 16 columns per workgroup, 16 row groups of 9 n / 16 rows, six rows in flight
    const int cl = t & 15, rr = t >> 4, j = (blockIdx.x - nTiles) * 16 + cl, jc = min(j, dK - 1), rows = 9 * a.n;
    double sum = 0.0;
    for (int r0 = rr; r0 < rows; r0 += 6 * 16) {
      double yv[6], gv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) { const int r = min(r0 + 16 * k, rows - 1); yv[k] = a.Y[(size_t)r * a.ldY + jc]; gv[k] = a.Y[(size_t)r * a.ldY + dK]; }
#pragma unroll
      for (int k = 0; k < 6; ++k) sum += (r0 + 16 * k < rows) ? yv[k] * gv[k] : 0.0;
    }
    red[t] = sum;
    __syncthreads();
    if (t < 16 && j < dK) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += red[k * 16 + t];
      (a.compact ? a.gOut : M + (size_t)dp * dp)[j] = p.gRed[j] - tot;
    }
  }
}

__global__ __launch_bounds__(256) void k_sb_back(DeviceProblem p, SbElimArgs a) {
  extern __shared__ double smem[];   // the last workgroup: all records (n x kSbRec), x (9 n), u (9 n)
  __shared__ int isLast;
  const int t = threadIdx.x, n = a.n;
  {
    const int r = blockIdx.x * 16 + (t >> 4), cl = t & 15;
    double s = 0.0;
    if (r < 9 * n) {
      const double* yr = a.Y + (size_t)r * a.ldY;
      for (int j0 = cl; j0 < a.dK; j0 += 8 * 16) {
        double yv[8], xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int j = min(j0 + 16 * k, a.dK - 1); yv[k] = yr[j]; xv[k] = p.yC[j]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (j0 + 16 * k < a.dK) ? yv[k] * xv[k] : 0.0;
      }
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 16);
    if (cl == 0 && r < 9 * n) __hip_atomic_store(a.tvec + r, a.Y[(size_t)r * a.ldY + a.dK] - s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (t == 0) isLast = (atomicAdd(a.counter, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!isLast) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  double* rec = smem;
  double* x = rec + (size_t)n * kSbRec;
  double* u = x + 9 * n;
  {
    const double2* src = reinterpret_cast<const double2*>(a.Lf);
    double2* dst = reinterpret_cast<double2*>(rec);
    const int total = n * kSbRec / 2;
    for (int base = t; base < total; base += 12 * 256) {   // twelve 16-byte loads in flight per thread
      double2 v[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] = src[min(base + k * 256, total - 1)];
#pragma unroll
      for (int k = 0; k < 12; ++k)
        if (base + k * 256 < total) dst[base + k * 256] = v[k];
    }
  }
  for (int r = t; r < 9 * n; r += blockDim.x) x[r] = __hip_atomic_load(a.tvec + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  // L^T x = t from the last eliminated block to the first: x_b = G_b^T (t_b - F_lo x_{b-s} - F_hi x_{b+s})
  auto applyGt = [&](int b, int i, const double* rhs) {
    const double* Gb = rec + (size_t)b * kSbRec + kSbG;
    double acc = 0.0;
    for (int k = i; k < 9; ++k) acc += Gb[k * 9 + i] * rhs[k];
    return acc;
  };
  if (t < 9) u[t] = x[t];
  __syncthreads();
  if (t < 9) x[t] = applyGt(0, t, u);
  __syncthreads();
  int sTop = 1;
  while (2 * sTop < n) sTop *= 2;
  for (int s = sTop; s >= 1; s >>= 1) {
    const int nE = (n - s + 2 * s - 1) / (2 * s);
    for (int task = t; task < 9 * nE; task += blockDim.x) {
      const int e = task / 9, i = task % 9, b = s + 2 * s * e;
      const double* Fl = rec + (size_t)b * kSbRec + kSbFlo + i * 9;
      const double* Fh = rec + (size_t)b * kSbRec + kSbFhi + i * 9;
      double acc = x[9 * b + i];
#pragma unroll
      for (int j = 0; j < 9; ++j) acc -= Fl[j] * x[9 * (b - s) + j];
      if (b + s < n) {
#pragma unroll
        for (int j = 0; j < 9; ++j) acc -= Fh[j] * x[9 * (b + s) + j];
      }
      u[task] = acc;
    }
    __syncthreads();
    for (int task = t; task < 9 * nE; task += blockDim.x) {
      const int e = task / 9, i = task % 9, b = s + 2 * s * e;
      x[9 * b + i] = applyGt(b, i, u + 9 * e);
    }
    __syncthreads();
  }
  for (int r = t; r < 9 * n; r += blockDim.x) {
    const int gi = a.dK + r;
    p.yC[gi] = x[r];
    p.vC[gi] = p.gFull[gi] / p.htilC[gi];
  }
}

// ================================================================ K6'': left-looking LDS Cholesky, 176 < dpad <= 272
// The lower triangle of a 17 x 17-tile system (306 KB) does not fit LDS, but a left-looking factorisation never needs
// all of it at once: when block column k is finished, what later columns still read are the tiles L(I, j), I > k,
// j <= k -- at most (nT - 1 - k)(k + 1) <= 72 tiles (147 KB).  Every tile (I, j) lives from its panel solve (step j)
// to the last update of block column I (step I - 1), and three families share the slots of the rectangle
// rows h.., columns 0..h-1 (h = ceil(nT / 2)):
//     I <  h           -> slot of rectangle tile (h + j, I)    (born at step I, when (I, .) has just died)
//     I >= h, j <  h   -> its own slot
//     I >= h, j >= h   -> slot of rectangle tile (j, I - h)    (row j dies before step j's panel solve)
// (tests/test_ll_schedule.py replays the schedule on the CPU with this slot map and checks that no live tile is overwritten.)  Finished tiles are also
// written through to global memory for the backward substitution.
//   wave 0          the serial chain: diagonal tile k out of its registers (cholDiag16Acc), then forward substitution
//                   of the right-hand side block k, then the last update of diagonal tile k + 1
//   waves 1-7       own the tiles of one block column at a time, TRANSPOSED in MFMA accumulators: C(I,c)^T.  An
//                   accumulator-layout tile is directly the B operand of a K = 16 product (register q = rows 4q + g),
//                   so the panel solve X^T = L_cc^-1 C^T runs out of registers, and X^T in the accumulator layout is X
//                   in the operand layout -- it goes to its LDS slot (XOR-swizzled, unpadded) and to global memory.
//                   While wave 0 factorises diagonal tile k they apply the updates j < k to block column k + 1
//                   (look-ahead); after the panel solve of column k only the update j = k is left.
//                   (wave 4 shares its SIMD with wave 0; the chain has slack -- the look-ahead is the longer phase)
// Two LDS-only barriers per block column.  The backward substitution stages L back from global memory in batches of
// whole tile rows (bottom rows first) and sweeps them row by row.
constexpr int kLLThreads = 512;
#ifdef SVIN_LL_TIMING
__device__ int g_llCount;
#endif
__host__ __device__ constexpr int llHalf(int nT) { return (nT + 1) / 2; }
__host__ __device__ constexpr int llSlots(int nT) { return (nT - llHalf(nT)) * llHalf(nT); }
__host__ __device__ constexpr size_t llLdsDoubles(int nT) { return (size_t)llSlots(nT) * 256 + 2 * 256 + 16 * kPanelLd + 16 + 2 * 16 * nT; }
// (branch-free on purpose: as a chain of conditionals the compiler turns every slot lookup of the update loops into three
// scalar branches)
__host__ __device__ __forceinline__ int llSlot(int I, int j, int h) {
  const int low = (I < h) ? 1 : 0, left = (j < h) ? 1 : 0;
  const int sA = j * h + I, sB = (I - h) * h + j, sC = (j - h) * h + (I - h);
  return low * sA + (1 - low) * (left * sB + (1 - left) * sC);
}

// the slot map as the kernel uses it, for the CPU replay of the schedule (tests/test_ll_schedule.py)
extern "C" int svin_debug_ll_slot(int I, int j, int nT) { return llSlot(I, j, llHalf(nT)); }
extern "C" int svin_debug_ll_slots(int nT) { return llSlots(nT); }
// T_u -= A B_u^T for NV tiles at once (independent accumulators: their MFMAs interleave), DIAG: also Td -= A A^T
// (separate references, not arrays: the accumulators have to stay in registers)
#define SVIN_MFMA_SUB(T, x, y) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-(x), (y), T, 0, 0, 0)
template <int NV, bool DIAG>
__device__ __forceinline__ void llUpdate(const double (&a)[4], const double (&b0)[4], const double (&b1)[4], const double (&b2)[4],
                                         d4_t& T0, d4_t& T1, d4_t& T2, d4_t& Td) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (DIAG) SVIN_MFMA_SUB(Td, a[q], a[q]);
    if (NV > 0) SVIN_MFMA_SUB(T0, a[q], b0[q]);
    if (NV > 1) SVIN_MFMA_SUB(T1, a[q], b1[q]);
    if (NV > 2) SVIN_MFMA_SUB(T2, a[q], b2[q]);
  }
}
// n tiles; rows: how many of the first two are real rows (the last worker's third tile is the diagonal one)
__device__ __forceinline__ void llUpdateN(int n, int rows, const double (&a)[4], const double (&b0)[4], const double (&b1)[4],
                                          const double (&b2)[4], d4_t& T0, d4_t& T1, d4_t& T2) {
  // one short block per tile (uniform branches): variants with interleaved accumulators cost more registers than the
  // kernel has (their merge points keep copies of every accumulator alive)
  d4_t none = {0, 0, 0, 0};
  if (rows > 0) llUpdate<1, false>(a, b0, b0, b0, T0, T0, T0, none);
  if (rows > 1) llUpdate<1, false>(a, b1, b1, b1, T1, T1, T1, none);
  if (n > 2) llUpdate<1, false>(a, b2, b2, b2, T2, T2, T2, none);
}
// X_u^T = Linv T_u for NV tiles
template <int NV>
__device__ __forceinline__ void llPanel(const double (&li4)[4], d4_t& T0, d4_t& T1, d4_t& T2) {
  d4_t x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0}, x2 = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (NV > 0) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(li4[q], T0[q], x0, 0, 0, 0);
    if (NV > 1) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(li4[q], T1[q], x1, 0, 0, 0);
    if (NV > 2) x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(li4[q], T2[q], x2, 0, 0, 0);
  }
  if (NV > 0) T0 = x0;
  if (NV > 1) T1 = x1;
  if (NV > 2) T2 = x2;
}

__global__ __launch_bounds__(kLLThreads) void k_chol_solve_ll(DeviceProblem p, int dpad, double mu, int initScale, int fuseFinalize) {
  extern __shared__ double smem[];
  SVIN_ARGS(SA(p.S), SA(p.gRed), SA(p.gFull), SA(p.hC), SA(p.scaleC), SA(p.htilC), SA(p.yC), SA(p.vC), SA(p.scal), SA(p.d), SA(p.ldS),
            SA(p.cholL));
  const int t = threadIdx.x, d = p.d, nT = dpad / 16, h = llHalf(nT);
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int g = lane >> 4, cc = lane & 15;
  const int ldS = p.ldS ? p.ldS : d;
  double* slots = smem;                                  // llSlots(nT) unpadded, swizzled 16 x 16 tiles
  double* H1 = slots + (size_t)llSlots(nT) * 256;        // look-ahead result of the next diagonal tile, two buffers, [r][lane]
  double* diagBuf = H1 + 512;                            // factorised diagonal tile (16 x kPanelLd), as cholDiag16Acc leaves it
  double* dinv16 = diagBuf + 16 * kPanelLd;
  double* rhs = dinv16 + 16;                             // dpad
  double* damp = rhs + dpad;                             // dpad: mu * htil per row (added to the diagonal tiles as they are taken up)
  double* Lg = p.cholL;                                  // tile (I, j), j <= I, at (I (I + 1) / 2 + j) * 256, row-major 16 x 16
  double* dinvG = Lg + (size_t)(nT * (nT + 1) / 2) * 256;  // dpad
  int* fail = &p.scal->cholFail;
  const int lopS = cc * 16 + (g ^ cc);                   // swizzled operand-layout offset of (row cc, column g): ^ 4q for column 4q + g
  const int widx = wave - 1;                             // workers: waves 1 .. 7 -> 0 .. 6
  const bool worker = wave != 0;
  constexpr int nWork = 7, kTurns = 3;                   // (17 - 1) rows at most over 7 workers
  // the look-ahead of diagonal tile (c, c) rides with the first worker that has one row less than the others in block column c
  // (the rows are dealt round-robin: worker (nT - c - 1) mod 7; at most two rows there, its third entry is free)
  auto diagOwner = [&](int c) { return widx == (nT - c - 1) % nWork; };
  // rows of block column c owned by this worker: c + 1 + widx + 6 u, u < rowsOf(c)
  auto rowsOf = [&](int c) { const int n = nT - (c + 1 + widx); return n <= 0 ? 0 : min(kTurns, (n + nWork - 1) / nWork); };

  // S tile (R, C), R <= C, in the accumulator layout: lane (g, cc) register r = S[16 R + g + 4 r][16 C + cc], identity padding;
  // read from the lower triangle like the other loaders
  auto loadS = [&](int R, int C) {
    d4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 16 * R + g + 4 * r, gj = 16 * C + cc;
      const int ci = min(max(gi, gj), d - 1), cj = min(min(gi, gj), d - 1);
      const double x = p.S[(size_t)ci * ldS + cj];
      v[r] = (gi < d && gj < d) ? x : ((gi == gj) ? 1.0 : 0.0);
    }
    return v;
  };
  // the same without the padding select: the value is not touched before maskS (a select right behind the load would expose
  // its whole latency where the tile is only being prefetched)
  const int laneS = cc * ldS + g;
  auto loadSRaw = [&](int R, int C) {
    d4_t v;
    if (R < C && 16 * C + 15 < d) {   // interior off-diagonal tile: one scalar base, one lane offset, no clamping (and fewer
                                      // registers); diagonal tiles go through min / max: only the lower triangle of S is valid for every caller
      const double* base = p.S + (size_t)(16 * C) * ldS + 16 * R;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = base[laneS + 4 * r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = 16 * R + g + 4 * r, gj = 16 * C + cc;
        const int ci = min(max(gi, gj), d - 1), cj = min(min(gi, gj), d - 1);
        v[r] = p.S[(size_t)ci * ldS + cj];
      }
    }
    return v;
  };
  auto maskS = [&](d4_t v, int R, int C) {
    if (16 * C + 15 < d) return v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 16 * R + g + 4 * r, gj = 16 * C + cc;
      v[r] = (gi < d && gj < d) ? v[r] : ((gi == gj) ? 1.0 : 0.0);
    }
    return v;
  };
  auto slotAt = [&](int I, int j) { return slots + (size_t)llSlot(I, j, h) * 256; };
  auto readOp = [&](const double* tile, double (&o)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = tile[lopS ^ (4 * q)];
  };

#ifdef SVIN_LL_TIMING
  long long qT[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long q0 = __builtin_readcyclecounter(), q1;
#define LLT(i) do { q1 = __builtin_readcyclecounter(); qT[i] += q1 - q0; q0 = q1; } while (0)
#else
#define LLT(i) do { } while (0)
#endif
  // ---- prologue: damping / right-hand side, the first two block columns
  // per wave three tiles of the current block column (Tc), of the next one (Tn) and the prefetched S tiles of the one after
  // (Tp).  The third entry of the worker with the fewest rows holds the next DIAGONAL tile.  Wave 0 keeps the
  // diagonal tile it is factorising in Tc[0] (the register file is the scarce resource of this kernel: every spill reload
  // is a vmcnt(0) wait behind the write-through stores and the prefetches)
  d4_t Tc[kTurns], Tn[kTurns], Tp[kTurns];
#pragma unroll
  for (int u = 0; u < kTurns; ++u) { Tc[u] = d4_t{0, 0, 0, 0}; Tn[u] = d4_t{0, 0, 0, 0}; Tp[u] = d4_t{0, 0, 0, 0}; }
  d4_t& accD = Tc[0];
  if (wave == 0) accD = loadSRaw(0, 0);
  if (worker) {
#pragma unroll
    for (int u = 0; u < kTurns; ++u) {
      const int I0 = 1 + widx + nWork * u, I1 = 2 + widx + nWork * u;
      if (I0 < nT) Tc[u] = loadSRaw(0, I0);
      if (I1 < nT) Tp[u] = loadSRaw(1, I1);
    }
    if (nT > 1 && diagOwner(1)) Tp[2] = loadSRaw(1, 1);
#pragma unroll
    for (int u = 0; u < kTurns; ++u) Tc[u] = maskS(Tc[u], 0, 1 + widx + nWork * u);
  }
  if (wave == 0) accD = maskS(accD, 0, 0);
  if (t < dpad) {
    const double dmp = (fuseFinalize && t < d) ? finalizeRow(p, t, mu, initScale) : 0.0;
    rhs[t] = (t < d) ? p.gRed[t] : 0.0;
    damp[t] = dmp;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (g + 4 * r == cc) accD[r] += damp[cc];
  }
  LLT(0);

  for (int k = 0; k < nT; ++k) {
    const int k0 = 16 * k;
    // ================= phase FD(k): last update (j = k - 1) of block column k, then look-ahead of block column k + 1
    if (wave == 0) {
      if (k >= 1) {
        const double* Hb = H1 + (k & 1) * 256;
        const double dmp = damp[k0 + cc];
#pragma unroll
        for (int r = 0; r < 4; ++r) accD[r] = Hb[r * 64 + lane] + ((g + 4 * r == cc) ? dmp : 0.0);
        double x[4];
        readOp(slotAt(k, k - 1), x);
#pragma unroll
        for (int q = 0; q < 4; ++q) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[q], x[q], accD, 0, 0, 0);
      }
      LLT(6);
      cholDiag16Acc<true>(accD, diagBuf, dinv16, lane, fail);   // diagBuf <- L_kk^-T (zeros below the diagonal)
      LLT(7);
    } else if (worker) {
      if (k >= 1) {
        // the operands of the last update C(I, k)^T -= L(k, k-1) L(I, k-1)^T are requested first; the forward substitution
        // rhs_I -= X(I, k-1) y_{k-1} (out of the registers of the panel solve: Tc = X^T, lane (g, cc) register r =
        // X[cc][g + 4r]) runs while they are on their way
        const int nv = rowsOf(k), nvPrev = rowsOf(k - 1);
        double a[4], b0[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0}, b2[4] = {0, 0, 0, 0};
        if (nv > 0) {
          readOp(slotAt(k, k - 1), a);
          readOp(slotAt(k + 1 + widx, k - 1), b0);
          if (nv > 1) readOp(slotAt(k + 1 + widx + nWork, k - 1), b1);
          if (nv > 2) readOp(slotAt(k + 1 + widx + 2 * nWork, k - 1), b2);
        }
        double yk[4], rOld[kTurns];
#pragma unroll
        for (int r = 0; r < 4; ++r) yk[r] = rhs[k0 - 16 + g + 4 * r];
#pragma unroll
        for (int u = 0; u < kTurns; ++u) rOld[u] = (u < nvPrev) ? rhs[16 * (k + widx + nWork * u) + cc] : 0.0;
#pragma unroll
        for (int u = 0; u < kTurns; ++u) {
          if (u < nvPrev) {
            const double s = Tc[u][0] * yk[0] + Tc[u][1] * yk[1] + Tc[u][2] * yk[2] + Tc[u][3] * yk[3];
            double sg[4];
            allGatherRows(s, sg);   // the four lane rows' partial sums (v_permlane swaps, no LDS round trip)
            if (g == 0) rhs[16 * (k + widx + nWork * u) + cc] = rOld[u] - ((sg[0] + sg[1]) + (sg[2] + sg[3]));
          }
        }
        if (nv > 0) llUpdateN(nv, nv, a, b0, b1, b2, Tn[0], Tn[1], Tn[2]);
#pragma unroll
        for (int u = 0; u < kTurns; ++u) Tc[u] = Tn[u];
      }
      LLT(1);
      const int c = k + 1;   // look-ahead column
      if (c < nT) {
        const bool diagW = diagOwner(c);
#pragma unroll
        for (int u = 0; u < kTurns; ++u) Tn[u] = maskS(Tp[u], c, (diagW && u == 2) ? c : c + 1 + widx + nWork * u);
        // the block column after it is requested now: one step of latency cover
        if (c + 1 < nT) {
#pragma unroll
          for (int u = 0; u < kTurns; ++u) {
            const int I = c + 2 + widx + nWork * u;
            if (I < nT) Tp[u] = loadSRaw(c + 1, I);
          }
          if (diagOwner(c + 1)) Tp[2] = loadSRaw(c + 1, c + 1);
        }
        LLT(6);
        const int nv = rowsOf(c);
        if (nv > 0 || diagW) {
          // updates j < k, tile by tile: one accumulator takes the whole chain of products while the operands of step j + 1
          // are already in flight (two operand sets of one tile pair fit the register file, two sets of all tiles do not)
          auto chain = [&](d4_t& T, int I, bool isDiag) {
            double a0[4], b0[4], a1[4], b1[4];
            readOp(slotAt(c, 0), a0);
            if (!isDiag) readOp(slotAt(I, 0), b0);
            for (int j = 0; j < k; j += 2) {
              if (j + 1 < k) {
                readOp(slotAt(c, j + 1), a1);
                if (!isDiag) readOp(slotAt(I, j + 1), b1);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) SVIN_MFMA_SUB(T, a0[q], isDiag ? a0[q] : b0[q]);
              if (j + 1 < k) {
                if (j + 2 < k) {
                  readOp(slotAt(c, j + 2), a0);
                  if (!isDiag) readOp(slotAt(I, j + 2), b0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) SVIN_MFMA_SUB(T, a1[q], isDiag ? a1[q] : b1[q]);
              }
            }
          };
          if (k > 0) {
            if (nv > 0) chain(Tn[0], c + 1 + widx, false);
            if (nv > 1) chain(Tn[1], c + 1 + widx + nWork, false);
            if (diagW) chain(Tn[2], c, true);
            else if (nv > 2) chain(Tn[2], c + 1 + widx + 2 * nWork, false);
          }
        }
        LLT(7);
        if (diagW) {
          double* Hb = H1 + (c & 1) * 256;
#pragma unroll
          for (int r = 0; r < 4; ++r) Hb[r * 64 + lane] = Tn[2][r];
        }
      }
    }
    LLT(1);
    ldsBarrier();
    LLT(2);
    // ================= phase P(k): panel solve of block column k; wave 0: y_k and the write-through of the diagonal tile
    if (wave == 0) {
      const int li = cc;
      double yv = 0;
#pragma unroll
      for (int c = 0; c < 16; ++c) yv = __builtin_fma(diagBuf[c * kPanelLd + li], rhs[k0 + c], yv);   // row li of L_kk^-1
      double dv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) { const int e = lane + 64 * m; dv[m] = diagBuf[(e & 15) * kPanelLd + (e >> 4)]; }   // transposed: L_kk^-1 row-major, as the backward solve reads it
      const double di = dinv16[cc];
      waveSync();
      if (lane < 16) rhs[k0 + lane] = yv;
      double* Dg = Lg + (size_t)(k * (k + 1) / 2 + k) * 256;
#pragma unroll
      for (int m = 0; m < 4; ++m) Dg[lane + 64 * m] = dv[m];
      if (lane < 16) dinvG[k0 + lane] = di;
    } else if (worker) {
      double li4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = 4 * q + g;
        li4[q] = diagBuf[kk * kPanelLd + cc];   // L_kk^-1[cc][kk]
      }
      const int nv = rowsOf(k);
      if (nv > 0) llPanel<1>(li4, Tc[0], Tc[1], Tc[2]);
      if (nv > 1) llPanel<1>(li4, Tc[1], Tc[1], Tc[2]);
      if (nv > 2) llPanel<1>(li4, Tc[2], Tc[1], Tc[2]);
#pragma unroll
      for (int u = 0; u < kTurns; ++u) {
        const int I = k + 1 + widx + nWork * u;
        if (u < nv) {
          double* sl = slotAt(I, k);
          double* Xg = Lg + (size_t)(I * (I + 1) / 2 + k) * 256 + cc * 16 + g;
#pragma unroll
          for (int r = 0; r < 4; ++r) { sl[lopS ^ (4 * r)] = Tc[u][r]; Xg[4 * r] = Tc[u][r]; }
        }
      }
    }
    LLT(3);
    ldsBarrier();
    LLT(4);
  }

  // ================= backward substitution L^T x = y: tile rows staged back from global memory, bottom rows first
  __syncthreads();   // the write-through stores have completed
  {
    double* stage = smem;
    const int cap = (int)(((size_t)llSlots(nT) * 256 + 512 + 16 * kPanelLd + 16) / 256) - (dpad + 255) / 256 - 1;
    double* dinvAll = stage + (size_t)cap * 256;   // dpad doubles
    // Round 3: the tile rows come back through the LDS-DMA path (global_load_lds_dwordx4: 1 KB per wave instruction, no
    // register round trip, no ds_write pass) into TWO half-size buffers: while one batch of whole tile rows is solved and swept,
    // the next one streams in.  (Through registers -- 38 doubles per thread per batch, then 38 ds_writes -- staging cost 7.5 k
    // cycles per batch, more than the solves of the batch.)
    const int capHalf = cap / 2;   // >= nT tiles: the longest tile row fits (cap >= 2 nT for 12 <= nT <= 17)
    auto batchLo = [&](int top) {
      int lo = top, used = top + 1;
      while (lo > 0 && used + lo <= capHalf) { used += lo; --lo; }   // rows lo .. top, row i = i + 1 tiles (diagonal included)
      return lo;
    };
    // global order is row-major over (i, j): a batch is one contiguous range of tiles, copied linearly
    auto request = [&](int lo, int top, double* buf) {
      const size_t e0 = (size_t)(lo * (lo + 1) / 2) * 256;
      const int nChunks = ((top + 1) * (top + 2) / 2 - lo * (lo + 1) / 2) * 2;   // 1 KB each
      const char* src = reinterpret_cast<const char*>(Lg + e0) + lane * 16;
      for (int ch = wave; ch < nChunks; ch += kLLThreads / 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)ch * 1024),
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(buf) + (size_t)ch * 1024), 16, 0, 0);
    };
    // Round 3: no barrier per tile row.  Per batch (tile rows lo .. top staged in LDS) wave 0 runs the chain -- per row two
    // dependent 16 x 16 matrix-vector products on the VALU, the vector replicated over the lane rows (colToRowForm /
    // sumLaneRows, as in k_chol_solve_lds: ~450 cycles per row against ~2.8 k for a 16-term dot product per lane, a sweep and
    // two barriers) -- and takes the contribution of row i + 1 to block i itself; waves 1-7 own the columns (thread t - 64 =
    // column) and sweep row j over the columns left of block j - 1 as soon as y_j is flagged.  Hand-overs through two kinds
    // of monotonic LDS counters (progress value of row j: nT - j), no fences (DS operations of a wave execute in order).
    int* yCount = reinterpret_cast<int*>(damp);   // damp is dead after the factorisation
    int* sweepCount = yCount + 1;                 // [wave]
    int* bailB = yCount + 9;
    if (t < 16) yCount[t] = 0;
#ifdef SVIN_LL_TIMING
    long long bsStage = 0, bsChain = 0, bsTail = 0, bsSpins = 0, bsMark = __builtin_readcyclecounter();
#define BST(acc) do { const long long n_ = __builtin_readcyclecounter(); acc += n_ - bsMark; bsMark = n_; } while (0)
#else
#define BST(acc) do { } while (0)
#endif
    int top = nT - 1;
    double swAcc[3] = {0, 0, 0};
    // (buffer addresses by arithmetic on `stage`: picked out of an array of pointers they lose their LDS address space and every
    //  tile read turns into a FLAT load)
    int cur = 0;
    request(batchLo(top), top, stage);
    (void)dinvAll;
    (void)dinvG;
    while (top >= 0) {
      const int lo = batchLo(top);
      const int firstTile = lo * (lo + 1) / 2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the batch have landed
      ldsBarrier();                                       // ... and everybody else's; the other buffer is free
      if (lo > 0) request(batchLo(lo - 1), lo - 1, stage + (size_t)((cur ^ 1) * capHalf) * 256);
      BST(bsStage);
      double* stageB = stage + (size_t)(cur * capHalf) * 256;
      auto tileB = [&](int i, int j) { return stageB + (size_t)(i * (i + 1) / 2 - firstTile + j) * 256; };
      if (wave == 0) {
        // row-major unpadded tiles: entry (4q + g, c) of a tile sits at lane + 64 q -- both products read their matrix like that
        auto ops = [&](const double* tile, double (&o)[4]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = tile[lane + 64 * q];
        };
        auto matVec = [&](const double (&m)[4], double v) {   // sum_k m[k][c] v[k], replicated
          double vq[4];
          colToRowForm(v, vq);
          double acc = m[0] * vq[0];
#pragma unroll
          for (int q = 1; q < 4; ++q) acc = __builtin_fma(m[q], vq[q], acc);
          return sumLaneRows(acc);
        };
        // "the sweeps of rows top .. `row` have reached the columns of `block`" + the block itself, requested together (flag
        // first: DS operations execute in order) and looked at after the product that does not depend on them
        auto sweepFlag = [&](int block) { return __hip_atomic_load(sweepCount + 1 + (block % 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        double aD[4], aL[4] = {0, 0, 0, 0};
        ops(tileB(top, top), aD);
        double y = 0;
        for (int i = top; i >= lo; --i) {
          double nD[4] = {0, 0, 0, 0}, nL[4] = {0, 0, 0, 0};
          if (i < top) {   // y_{i+1} out first: the sweepers start on it while this row is solved
            if (g == 0) rhs[16 * (i + 1) + cc] = y;
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(yCount, nT - (i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
          }
          const int need = (i + 2 <= top) ? nT - (i + 2) : 0;
          int f = sweepFlag(i);
          asm volatile("" ::: "memory");
          double rr = rhs[16 * i + cc];
          if (i > lo) { ops(tileB(i - 1, i - 1), nD); ops(tileB(i, i - 1), nL); }
          __builtin_amdgcn_sched_barrier(0);
          const double tv = (i < top) ? matVec(aL, y) : 0.0;
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" : "+v"(f));
          int spins = 0;
          while (f < need) {
            asm volatile("" ::: "memory");
            f = sweepFlag(i);
            asm volatile("" ::: "memory");
            rr = rhs[16 * i + cc];
#ifdef SVIN_LL_TIMING
            ++bsSpins;
#endif
            if (++spins > (1 << 18)) { __hip_atomic_store(bailB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
          }
          y = matVec(aD, rr - tv);
#pragma unroll
          for (int q = 0; q < 4; ++q) { aD[q] = nD[q]; aL[q] = nL[q]; }
        }
        if (g == 0) rhs[16 * lo + cc] = y;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(yCount, nT - lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        if (lo > 0) {   // the contribution of row lo to block lo - 1 (its tile leaves LDS with this batch)
          double bL[4];
          ops(tileB(lo, lo - 1), bL);
          const double tv = matVec(bL, y);
          if (lo + 1 <= top) {
            int spins = 0;
            while (sweepFlag(lo - 1) < nT - (lo + 1)) {
              if (++spins > (1 << 18)) { __hip_atomic_store(bailB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
            }
            asm volatile("" ::: "memory");
          }
          if (g == 0) rhs[16 * (lo - 1) + cc] -= tv;
        }
      } else {
        // Sweepers: wave w folds the rows further down into the blocks b = w - 1, w + 6, w + 13 (lane (g, c): four FMAs per row
        // and block on the tile read as it lies, entry (4q + g, c) at lane + 64 q -- conflict-free; a thread per COLUMN read
        // its tile 4-way conflicted, sixteen times per row, and seven such waves kept the LDS pipe busier than the chain could
        // bear).  The sum over the lane rows waits until the block's last contribution (row b + 2), across batches.
        for (int j = top; j >= lo; --j) {
          int spins = 0;
          while (__hip_atomic_load(yCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nT - j) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 18)) { __hip_atomic_store(bailB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
          }
          asm volatile("" ::: "memory");
          double yq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) yq[q] = rhs[16 * j + 4 * q + g];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int bt = wave - 1 + 7 * u;
            if (bt + 2 <= j) {
              const double* tl = tileB(j, bt);
#pragma unroll
              for (int q = 0; q < 4; ++q) swAcc[u] = __builtin_fma(tl[lane + 64 * q], yq[q], swAcc[u]);
              if (bt + 2 == j) {
                const double tot = sumLaneRows(swAcc[u]);
                if (g == 0) rhs[16 * bt + cc] -= tot;
              }
            }
          }
          asm volatile("" ::: "memory");
          if (lane == 0) __hip_atomic_store(sweepCount + wave, nT - j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
        }
      }
      if (wave == 0) BST(bsChain);
      BST(bsTail);
      cur ^= 1;
      top = lo - 1;
    }
    ldsBarrier();
#ifdef SVIN_LL_TIMING
    if (t == 0 && g_llCount == 12) printf("[ll backsub wave 0] staging + barrier %lld  chain %lld  waiting for the sweepers at the batch end %lld  late polls %lld\n", bsStage, bsChain, bsTail, bsSpins);
#endif
#undef BST
    if (t == 0 && *bailB) atomicOr(fail, 4);
  }
  for (int i = t; i < d; i += blockDim.x) { p.yC[i] = rhs[i]; p.vC[i] = p.gFull[i] / p.htilC[i]; }  // + steepest-descent direction
#ifdef SVIN_LL_TIMING
  LLT(5);
  __shared__ int llPrint;
  if (t == 0) llPrint = atomicAdd(&g_llCount, 1);
  __syncthreads();
  if (llPrint == 12 && lane == 0)
    printf("[ll wave %d] prologue %lld  FD(F/rest) %lld (prefetch|w0 final) %lld (lookahead|w0 diag) %lld  waitB1 %lld  P %lld  waitB2 %lld  backsub %lld\n", wave, qT[0], qT[1], qT[6], qT[7], qT[2], qT[3], qT[4], qT[5]);
#endif
#undef LLT
}

// ---- border block of the LDS-resident solver's kBorder = 2 variant: one workgroup of four waves, everything on 16 x 16 tiles
// (tile16.hpp).  C = S[dM .., dM ..] (damped like every diagonal entry of the system; the metric of its rows is stored here),
// C = Lc Lc^T, Lc^-1, q = C^-1 g2, V = Lc^-1 B with B = S[dM .., 0 .. dM) -- into `scr` (layout: kBorderOff*).  A pivot that is not
// positive raises cholFail like any other pivot of the system.
constexpr int kBorderPrepThreads = 256;
__global__ __launch_bounds__(kBorderPrepThreads) void k_chol_border_prepare(DeviceProblem p, int dM, double mu, int initScale, int fuseFinalize, int m, double* scr) {
  constexpr int MP = kBorderMP, ld = MP + 1, nTb = MP / 16;
  __shared__ double sA[MP * ld];              // the border block, then Lc (lower tiles) / Lc^-1 transposed (upper tiles)
  __shared__ double sD[nTb * 16 * kPanelLd];  // diagonal scratch tiles
  __shared__ double sDinv[MP];
  __shared__ double sLi[MP * ld];             // Lc^-1, dense lower
  __shared__ double sU[MP], sG[MP], sQ[MP];
  __shared__ int sFail;
  const int t = threadIdx.x, ldS = p.ldS ? p.ldS : p.d;
  const int wave = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
  lds_f64* A = tileToLds(sA);
  if (t == 0) sFail = 0;
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 0] = (double)wall_clock64();
#endif
  // B in the B-operand layout of the products that form V, requested before anything else (it does not depend on the factor): column
  // tiles J = wave, wave + 4, wave + 8 of the 11, lane (g, c) holds B[4 q + g][16 J + c]
  double bOp[3][kBorderMaxQ], g1v[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int col = 16 * (wave + 4 * kk) + c;
    const bool valid = col < dM;
#pragma unroll
    for (int q = 0; q < kBorderMaxQ; ++q) bOp[kk][q] = (valid && 4 * q + g < m) ? p.S[(size_t)(dM + 4 * q + g) * ldS + col] : 0.0;
    g1v[kk] = valid ? p.gRed[col] : 0.0;
  }
  for (int idx = t; idx < MP * ld; idx += kBorderPrepThreads) {
    const int r = idx / ld, cc = idx - r * ld;
    double v = (r == cc) ? 1.0 : 0.0;
    if (r < m && cc < m) {
      v = p.S[(size_t)(dM + max(r, cc)) * ldS + dM + min(r, cc)];
      if (r == cc && fuseFinalize) {   // the same metric and damping dampDiag gives a diagonal entry of the main block
        const double hc = p.hC[dM + r];
        const double sc = initScale ? 1.0 / (1.0 + sqrt(hc)) : p.scaleC[dM + r];
        const double ht = fmin(fmax(hc * sc * sc, 1e-6), 1e32) / (sc * sc);
        v += mu * ht;
        if (initScale) p.scaleC[dM + r] = sc;
        p.htilC[dM + r] = ht;
      }
    }
    A[idx] = v;
  }
  if (t < MP) sG[t] = (t < m) ? p.gRed[dM + t] : 0.0;
  __syncthreads();
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 1] = (double)wall_clock64();
#endif
  tileCholFactor<kBorderPrepThreads / 64>(A, ld, nTb, sD, sDinv, &sFail);
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 2] = (double)wall_clock64();
#endif
  (void)tileCholInverse<kBorderPrepThreads / 64>(A, ld, nTb, MP, tileToLds(sD), tileToLds(sDinv));
  __syncthreads();
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 3] = (double)wall_clock64();
#endif
  if (t == 0 && sFail) atomicOr(&p.scal->cholFail, 1);
  // Lc^-1 dense (LDS + scratch): element (i, j), j <= i
  for (int idx = t; idx < MP * MP; idx += kBorderPrepThreads) {
    const int i = idx / MP, j = idx - i * MP;
    double v = 0.0;
    if (j <= i) v = ((j >> 4) < (i >> 4)) ? (double)A[j * ld + i] : tileLinvAt(tileToLds(sD), tileToLds(sDinv), i >> 4, i & 15, j & 15);
    sLi[i * ld + j] = v;
    scr[kBorderOffLinv + idx] = v;
  }
  __syncthreads();
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 4] = (double)wall_clock64();
#endif
  auto sum8 = [](double v) {   // over the 8 lanes of an aligned group
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
  };
  {   // u = Lc^-1 g2, 8 lanes per row (MP rows x 8 = the workgroup)
    const int i = t >> 3, sub = t & 7;
    double u = 0.0;
    for (int j = sub; j <= i; j += 8) u = __builtin_fma(sLi[i * ld + j], sG[j], u);
    u = sum8(u);
    if (sub == 0) sU[i] = u;
  }
  __syncthreads();
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 5] = (double)wall_clock64();
#endif
  {   // q = Lc^-T u
    const int a = t >> 3, sub = t & 7;
    double q = 0.0;
    for (int i = a + sub; i < MP; i += 8) q = __builtin_fma(sLi[i * ld + a], sU[i], q);
    q = sum8(q);
    if (sub == 0) { scr[kBorderOffQ + a] = q; sQ[a] = q; }
  }
  __syncthreads();
#ifdef SVIN_BORDER_TIMING
  if (t == 0) scr[kBorderScratchDoubles + 6] = (double)wall_clock64();
#endif
  // V = Lc^-1 B on v_mfma_f64_16x16x4: per column tile two row tiles (rows 24 .. 31 come out zero); g1 - B^T q beside it
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int J = wave + 4 * kk;
    if (16 * J >= kBorderLdV) continue;   // (wave-uniform)
    d4_t v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < kBorderMaxQ; ++q) {
      const int k = 4 * q + g;
      v0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sLi[c * ld + k], bOp[kk][q], v0, 0, 0, 0);          // (Lc^-1 is lower triangular: zeros above)
      v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sLi[(16 + c) * ld + k], bOp[kk][q], v1, 0, 0, 0);
      part = __builtin_fma(bOp[kk][q], sQ[k], part);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      scr[(size_t)(g + 4 * r) * kBorderLdV + 16 * J + c] = v0[r];
      scr[(size_t)(16 + g + 4 * r) * kBorderLdV + 16 * J + c] = v1[r];
    }
    part = sumLaneRows(part);
    if (g == 0) scr[kBorderOffG + 16 * J + c] = g1v[kk] - part;
    // The border's rows couple with few columns of the main block (speed / bias blocks: the poses their IMU factors touch), and a
    // column of B that is exactly zero is an exactly zero column of V: the solver skips the tiles of such tile columns
    bool nz = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) nz = nz || v0[r] != 0.0 || v1[r] != 0.0;
    const bool any = __any(nz) != 0;
    if (lane == 0) scr[kBorderOffMask + J] = any ? 1.0 : 0.0;
  }
#ifdef SVIN_BORDER_TIMING
  __syncthreads();
  if (t == 0) scr[kBorderScratchDoubles + 7] = (double)wall_clock64();
#endif
}

// LDS bytes of k_chol_solve_lds for nT tile rows (+ kBorderMP doubles of the border variants)
static size_t cholLdsBytes(int nT) {
#ifdef SVIN_CHOL_TIMING
  return ((size_t)nT * (nT + 1) / 2 * kTile + 3 * 16 * nT) * 8 + kCholFlagInts * 4 + 240 * 8 + kBorderMP * 8;
#else
  return ((size_t)nT * (nT + 1) / 2 * kTile + 3 * 16 * nT) * 8 + kCholFlagInts * 4 + kBorderMP * 8;
#endif
}
constexpr int kCholLdsMaxTiles = 11;   // tile rows of the largest system LDS holds (156 KB)
// rows beyond the LDS-resident solver's eleven tile rows that it eliminates while loading (k_chol_solve_lds<true>); needs the window's padded S
static int cholBorderRows(int d, bool padded) {
  const int m = d - 16 * kCholLdsMaxTiles;
  return (padded && m >= 1 && m <= kBorderMaxRows && !optOn(kOptNoLdsBorder)) ? m : 0;
}
static int solverClass(int d, bool padded = false) {   // 0 = LDS-resident, 1 = left-looking in one workgroup, 2 = blocked over many workgroups
  const int nT = (d + 15) / 16;
  if (cholLdsBytes(nT) <= 156 * 1024 || cholBorderRows(d, padded) > 0) return 0;
  if (nT >= 12 && nT <= 17 && !optOn(kOptNoLL)) return 1;
  return 2;
}
// Whether (and where in p.cholL) the speed / bias chain is eliminated ahead of the dense solve: 0 = no, 1 = the kept rows go
// to the blocked solver's matrix (k_sb_load writes it instead of k_big_load), 2 = the kept system is small enough for one of
// the single-workgroup solvers: k_sb_load writes it as a compact padded matrix S' + right-hand side g', which that solver
// takes through a DeviceProblem view.  A system the LDS-resident solver takes whole is left alone (14 us at d = 150: the
// elimination's four launches cost more), and so is a chain of fewer than 8 blocks ahead of the blocked solver.
static int planSbElimination(const DeviceProblem& p, SbElimArgs& a) {
  if (optOn(kOptNoSbElim) || p.sbChain < 2 || p.sbChain > kSbMaxChain || p.dC < 16 || p.dC + 9 * p.sbChain != p.d) return 0;
  if (solverClass(p.d, p.sPadded != 0) == 0) return 0;
  // Measured (tools/sb_elim_time.py, reduced solve with / without): d = 180 66 / 64 us, 240: 73 / 91, 270: 85 / 113, 360: 99 / 183,
  // 600: 185 / 313, 960: 289 / 476 -- the four launches cost ~45 us before they gain anything, so short chains stay with the
  // dense solvers.  (Tried and dropped: eliminating only the last blocks of a chain in ONE fused launch so that a system a few
  // rows over the LDS-resident solver's limit drops into it: 39 + 35 + 10 us against the left-looking solver's 70.  d = 177 .. 180
  // is now the LDS-resident solver's own border variant; the stereo_rig_v2 sliding window is d = 198.)
  const int mode = solverClass(p.dC) == 2 ? 1 : 2;
  // (the compact form needed 16 blocks until the end of round 5, when the kept rows went to the left-looking solver more often than
  //  not; with the LDS-resident solver taking up to 200 rows a chain of 8 pays: config #3 -- chain of 10, 180 kept rows -- 112.6 -> 83.6 us,
  //  d = 210 / 225 (chains of 14 / 15) 80.9 -> 65.2 / 90.2 -> 65.6)
  if (p.sbChain < 8) return 0;
  a.n = p.sbChain; a.dK = p.dC;
  a.dp = ((a.dK + kNB - 1) / kNB) * kNB;
  a.ldY = ((a.dK + 1 + 15) / 16) * 16;
  a.rowsY = ((9 * a.n + 3) / 4) * 4;
  a.compact = mode == 2 ? 1 : 0;
  size_t off0;
  if (mode == 1) {
    const size_t nb = a.dp / kNB;
    off0 = (size_t)(a.dp + kNB) * a.dp + a.dp + (size_t)a.dp * kNB + ((nb + 3) * nb + 1) / 2 + 2;
    off0 = (off0 + 1) & ~(size_t)1;
    a.Sout = nullptr; a.gOut = nullptr; a.ldOut = 0;
  } else {
    const size_t dpadK = ((size_t)a.dK + 15) / 16 * 16;
    a.ldOut = (int)dpadK;
    a.Sout = p.cholL + dpadK * dpadK;      // behind the kept solver's own spill / write-through area
    a.gOut = a.Sout + dpadK * dpadK;
    off0 = 2 * dpadK * dpadK + dpadK;
  }
  a.Lf = p.cholL + off0;
  a.Y = a.Lf + (size_t)a.n * kSbRec;
  a.tvec = a.Y + (size_t)a.rowsY * a.ldY;
  a.counter = reinterpret_cast<int*>(a.tvec + a.rowsY);
  return off0 + (size_t)a.n * kSbRec + (size_t)a.rowsY * a.ldY + a.rowsY + 2 <= solveReducedScratchDoubles(p.d, true) ? mode : 0;
}
// the dense solve of p's system (or, with `sb`, of the kept rows the chain elimination left in the blocked solver's matrix)
static void launchSolveDense(const DeviceProblem& p, hipStream_t s, double mu, bool initScale, bool fuseFinalize, const SbElimArgs* sb) {
  const int dpad = ((p.d + 15) / 16) * 16;
  const int nT = dpad / 16;
  const int cls = solverClass(p.d, p.sPadded != 0);
  const int border = cholBorderRows(p.d, p.sPadded != 0);
  if (cls == 0 && border > 4) {
    // the border block factorised and V, q, Lc^-1 written to the solver's global scratch by a small launch of its own
    const size_t ldsBytes = cholLdsBytes(kCholLdsMaxTiles);
    hipLaunchKernelGGL(k_chol_border_prepare, dim3(1), dim3(kBorderPrepThreads), 0, s, p, 16 * kCholLdsMaxTiles, mu, initScale ? 1 : 0,
                       fuseFinalize ? 1 : 0, border, p.cholL);
    ensureDynamicLds((const void*)k_chol_solve_lds<2>, ldsBytes);
    hipLaunchKernelGGL(k_chol_solve_lds<2>, dim3(1), dim3(kCholLdsThreads), ldsBytes, s, p, 16 * kCholLdsMaxTiles, mu, initScale ? 1 : 0,
                       fuseFinalize ? 1 : 0, border, (const double*)p.cholL);
  } else if (cls == 0 && border > 0) {
    const size_t ldsBytes = cholLdsBytes(kCholLdsMaxTiles);
    ensureDynamicLds((const void*)k_chol_solve_lds<1>, ldsBytes);
    hipLaunchKernelGGL(k_chol_solve_lds<1>, dim3(1), dim3(kCholLdsThreads), ldsBytes, s, p, 16 * kCholLdsMaxTiles, mu, initScale ? 1 : 0,
                       fuseFinalize ? 1 : 0, border, (const double*)nullptr);
  } else if (cls == 0) {
    const size_t ldsBytes = cholLdsBytes(nT);
    ensureDynamicLds((const void*)k_chol_solve_lds<0>, ldsBytes);
    hipLaunchKernelGGL(k_chol_solve_lds<0>, dim3(1), dim3(kCholLdsThreads), ldsBytes, s, p, dpad, mu, initScale ? 1 : 0,
                       fuseFinalize ? 1 : 0, 0, (const double*)nullptr);
  } else if (cls == 1) {
    // one workgroup, left-looking: at most 72 live tiles in LDS, finished tiles written through to p.cholL
    const size_t ldsLL = llLdsDoubles(nT) * 8;
    ensureDynamicLds((const void*)k_chol_solve_ll, ldsLL);
    hipLaunchKernelGGL(k_chol_solve_ll, dim3(1), dim3(kLLThreads), ldsLL, s, p, dpad, mu, initScale ? 1 : 0, fuseFinalize ? 1 : 0);
  } else {
    // multi-workgroup blocked factorisation, 64-wide panels; p.cholL holds (dpad64 + 64) x dpad64 doubles, its tail
    // the 1/L_ii vector.  Behind a chain elimination the matrix is already there (k_sb_load) and only spans the kept rows.
    const int dp = sb ? sb->dp : ((p.d + kNB - 1) / kNB) * kNB;
    double* dinvG = p.cholL + (size_t)(dp + kNB) * dp;
    double* diagF = dinvG + dp;   // per panel the factorised 64x64 diagonal block (dp x 64)
    const int nb = dp / kNB;
    int* ready = reinterpret_cast<int*>(diagF + (size_t)dp * kNB);   // (nb + 3) x nb block flags
    if (sb) {
      const int nTk = (sb->dK + 15) / 16;
      hipLaunchKernelGGL(k_sb_load, dim3(nTk * (nTk + 1) / 2 + (sb->dK + 15) / 16), dim3(256), 0, s, p, *sb, mu, initScale ? 1 : 0,
                         fuseFinalize ? 1 : 0, ready, (nb + 3) * nb);
    } else {
      hipLaunchKernelGGL(k_big_load, dim3(256), dim3(256), 0, s, p, dp, mu, initScale ? 1 : 0, fuseFinalize ? 1 : 0, ready,
                         (nb + 3) * nb);
    }
    {
      const size_t ldsTasks = ((size_t)3 * kBigBlockLds + kNB + 2) * 8;
      ensureDynamicLds((const void*)k_big_chol_chain, ldsTasks);
      int nHelperTasks = 0;
      for (int st = 0; st < nb; ++st) nHelperTasks += std::max(nb - st - 1, 0) + ((st + 2 <= nb - 1) ? 2 : 0);
      hipLaunchKernelGGL(k_big_chol_chain,
                         dim3(1 + std::max(1, std::min(nHelperTasks, (nHelperTasks > kPersistWideTasks ? kPersistWideGrid : kPersistMaxGrid) - 1))),
                         dim3(256), ldsTasks, s, p, dp,
                         dinvG, diagF, ready);
    }
    const size_t ldsBack = ((size_t)kBackSpan + 8 * 64 + kNB * (kNB + 1) + kNB) * 8;
    for (int c1 = dp; c1 > 0; c1 -= kBackSpan) {
      const int c0 = std::max(0, c1 - kBackSpan);
      const int nChunks = (dp - c1 + kBackSpan - 1) / kBackSpan;   // <= 63: the spare rows of the rhs block
      if (nChunks > 0) hipLaunchKernelGGL(k_big_back_gemv, dim3((c1 - c0) / kNB, nChunks), dim3(512), 0, s, p, dp, c0, c1);
      hipLaunchKernelGGL(k_big_back, dim3(1), dim3(512), ldsBack, s, p, dp, c0, c1, nChunks, (const double*)dinvG,
                         (const double*)diagF);
    }
  }
}
// A refused launch (LDS limit, attribute failure) would leave a stale y_C / v_C behind that the trust-region step consumes silently
// (ADVICE r5): the error of any launch of the solve is read back here -- hipGetLastError is thread-local and costs no synchronisation.
static void checkSolverLaunches() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("reduced-system solver launch: ") + hipGetErrorString(e));
}
static void launchSolveReducedUnchecked(const DeviceProblem& p, hipStream_t s, double mu, bool initScale, bool fuseFinalize);
// Reduced pose manifolds (PoseManifold3d / 4d / 2d, PoseManifold.cpp:173-466; Map::resetParameterization, Map.cpp:513-543).  Their
// Plus() is the 6-DoF one with some components of delta held at zero, so Ceres sees the 6-column local Jacobian with those columns
// dropped.  Dropping columns of J drops the same rows and columns of J^T J and of every Schur complement built from it, and the
// landmark blocks V_l do not contain pose columns at all -- so the kernels build the 6-DoF system as always and the locked rows are
// struck out of the finished reduced system: row and column zero, unit diagonal, zero gradient, zero column norm (scale 1).  The
// Gauss-Newton step and the gradient are then zero in the locked directions, which is all the dogleg step, the landmark
// back-substitution (W^T x), the model-cost products (J d) and the retraction ever see of them.  One workgroup per locked row.
__global__ __launch_bounds__(256) void k_lock_rows(DeviceProblem p) {
  const int i = p.lockedRows[blockIdx.x];
  const int d = p.d, ld = p.ldS ? p.ldS : d;
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    p.S[(size_t)i * ld + j] = (j == i) ? 1.0 : 0.0;
    if (j != i) p.S[(size_t)j * ld + i] = 0.0;
  }
  if (threadIdx.x == 0) { p.gRed[i] = 0.0; p.gFull[i] = 0.0; p.hC[i] = 0.0; }
}
void launchSolveReduced(const DeviceProblem& p, hipStream_t s, double mu, bool initScale, bool fuseFinalize) {
  (void)hipGetLastError();   // (a polling hipEventQuery / hipStreamQuery of another library -- RCCL -- leaves hipErrorNotReady behind: not ours)
  if (p.nLocked > 0) hipLaunchKernelGGL(k_lock_rows, dim3(p.nLocked), dim3(256), 0, s, p);
  launchSolveReducedUnchecked(p, s, mu, initScale, fuseFinalize);
  checkSolverLaunches();
}
// factorisation and forward substitution of the speed / bias chain (the first two launches of the reduced solve with the chain
// eliminated); fuseFinalize: the chain's rows get their metric and damping here
static void launchSbChainFactor(const DeviceProblem& p, const SbElimArgs& sb, hipStream_t s, double mu, bool initScale, bool fuseFinalize) {
  const size_t ldsFactor = ((size_t)sb.n * 162 + (size_t)((sb.n + 1) / 2) * 243) * 8;
  ensureDynamicLds((const void*)k_sb_factor, ldsFactor);
  hipLaunchKernelGGL(k_sb_factor, dim3(1), dim3(kSbFactorThreads), ldsFactor, s, p, sb, mu, initScale ? 1 : 0, fuseFinalize ? 1 : 0);
  const size_t ldsFwd = ((size_t)sb.rowsY * kSbCols + (size_t)sb.n * kSbLdsRec) * 8;
  ensureDynamicLds((const void*)k_sb_forward, ldsFwd);
  hipLaunchKernelGGL(k_sb_forward, dim3(sb.ldY / kSbCols), dim3(256), ldsFwd, s, p, sb);
}
// ONE predicate for the build (which then launches the chain's kernels) and the solve (which then skips them)
static bool sbEarlyActive(const DeviceProblem& p) {
  if (!(p.sideLane != 0 && p.L > 0 && p.N > 0 && p.dC > 0 && !p.schurDense && p.schurPanels && p.schurBlocks && p.nLocked == 0)) return false;
  if (p.F + priorAccBlocks(p) <= 0) return false;
  SbElimArgs sb;
  return planSbElimination(p, sb) != 0;
}
static void launchSbEarly(const DeviceProblem& p, hipStream_t side, double mu, bool initScale) {
  SbElimArgs sb;
  if (planSbElimination(p, sb)) launchSbChainFactor(p, sb, side, mu, initScale, /*fuseFinalize=*/true);
}
static void launchSolveReducedUnchecked(const DeviceProblem& p, hipStream_t s, double mu, bool initScale, bool fuseFinalize) {
  SbElimArgs sb;
  const int elim = planSbElimination(p, sb);
  if (!elim) { launchSolveDense(p, s, mu, initScale, fuseFinalize, nullptr); return; }
  // (p.sideLane: the build of this iteration has run them already, with this mu and initScale -- launchAccumulateNormalEquations)
  if (p.sideLane != 0 && sbEarlyActive(p) && !fuseFinalize) throw std::logic_error("sbEarly without the fused finalisation");
  if (!sbEarlyActive(p)) launchSbChainFactor(p, sb, s, mu, initScale, fuseFinalize);
  if (elim == 1) {
    launchSolveDense(p, s, mu, initScale, fuseFinalize, &sb);
  } else {
    const int nTk = (sb.dK + 15) / 16;
    hipLaunchKernelGGL(k_sb_load, dim3(nTk * (nTk + 1) / 2 + (sb.dK + 15) / 16), dim3(256), 0, s, p, sb, mu, initScale ? 1 : 0,
                       fuseFinalize ? 1 : 0, (int*)nullptr, 0);
    DeviceProblem q = p;   // the kept system as a problem of its own: rows 0 .. dK of every vector are the kept rows
    q.d = sb.dK; q.S = sb.Sout; q.ldS = sb.ldOut; q.sPadded = 1; q.gRed = sb.gOut; q.sbChain = 0;
    launchSolveDense(q, s, 0.0, false, false, nullptr);   // (metric and damping are in S' already)
  }
  const size_t ldsBackSb = ((size_t)sb.n * kSbRec + 18 * (size_t)sb.n) * 8;
  ensureDynamicLds((const void*)k_sb_back, ldsBackSb);
  hipLaunchKernelGGL(k_sb_back, dim3((9 * sb.n + 15) / 16), dim3(256), ldsBackSb, s, p, sb);
}

// ================================================================ post-solve pass and dogleg step
// traditional dogleg (ceres dogleg_strategy.cc) expressed on the un-scaled vectors:
//   delta_i = cg * g_i/htil_i + cn * (-y_i)
// J*delta is never formed: |J delta|^2 and (J delta).r follow from group B by linearity.
struct DoglegCoeff { double cg, cn, stepNorm, jdSq, jdDotR; };
__device__ __forceinline__ DoglegCoeff doglegCoefficients(double gHatSq, double jgSq, double gnHatSq, double gDotGn,
                                                          double jySq, double jvDotJy, double jvDotR, double jyDotR,
                                                          double radius) {
  const double gnorm = sqrt(gHatSq), gnnorm = sqrt(gnHatSq);
  const double alpha = gHatSq / jgSq;
  DoglegCoeff c;
  if (gnnorm <= radius) { c.cg = 0; c.cn = 1; c.stepNorm = gnnorm; }
  else if (gnorm * alpha >= radius) { c.cg = -(radius / gnorm); c.cn = 0; c.stepNorm = radius; }
  else {
    const double b_dot_a = -alpha * gDotGn;
    const double a_sq = (alpha * gnorm) * (alpha * gnorm);
    const double b_minus_a_sq = a_sq - 2 * b_dot_a + gnnorm * gnnorm;
    const double cc = b_dot_a - a_sq;
    const double dd = sqrt(cc * cc + b_minus_a_sq * (radius * radius - a_sq));
    const double beta = (cc <= 0) ? (dd - cc) / b_minus_a_sq : (radius * radius - a_sq) / (dd + cc);
    c.cg = -alpha * (1.0 - beta);
    c.cn = beta;
    c.stepNorm = sqrt(fmax(c.cg * c.cg * gHatSq + 2 * c.cg * c.cn * gDotGn + c.cn * c.cn * gnHatSq, 0.0));
  }
  c.jdSq = c.cg * c.cg * jgSq - 2.0 * c.cg * c.cn * jvDotJy + c.cn * c.cn * jySq;
  c.jdDotR = c.cg * jvDotR - c.cn * jyDotR;
  return c;
}
// candidate = x [+] delta for item i (variable blocks first, then landmarks); acc += |x - x_cand|^2, |x|^2
// poseOplus (dmath.hpp) with the device exponential and one reciprocal per normalisation
__device__ __forceinline__ void poseOplusDev(const double* x, const double* delta, double* xo) {
  xo[0] = x[0] + delta[0]; xo[1] = x[1] + delta[1]; xo[2] = x[2] + delta[2];
  const double n0 = 1.0 / sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  const Quat q = {x[3] * n0, x[4] * n0, x[5] * n0, x[6] * n0};
  const Quat qn = qmul(deltaQDev(delta[3], delta[4], delta[5]), q);
  const double n1 = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
  xo[3] = qn.x * n1; xo[4] = qn.y * n1; xo[5] = qn.z * n1; xo[6] = qn.w * n1;
}
// One item of the retraction x_cand = x [+] (cg v - cn y).  Everything is loaded before the first store: the candidate
// arrays may alias the inputs as far as the compiler knows, and a store between two loads turns them into serial
// memory round trips (9 us for the 22 blocks of a 10-keyframe window before this was written this way).
// value-level retraction of one parameter block (7 doubles for a pose / extrinsics block, 9 for speed and biases)
__device__ __forceinline__ void retractBlockValues(bool isSb, bool has, bool count, const double* x, const double* v, const double* y,
                                                   double cg, double cn, double* xo, double* acc) {
  if (!isSb) {
    if (has) {
      double dl[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) dl[k] = cg * v[k] - cn * y[k];
      poseOplusDev(x, dl, xo);
      if (count) {
#pragma unroll
        for (int k = 0; k < 7; ++k) { acc[0] += (x[k] - xo[k]) * (x[k] - xo[k]); acc[1] += x[k] * x[k]; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 7; ++k) xo[k] = x[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      xo[k] = has ? x[k] + (cg * v[k] - cn * y[k]) : x[k];
      if (has && count) { acc[0] += (x[k] - xo[k]) * (x[k] - xo[k]); acc[1] += x[k] * x[k]; }
    }
  }
}
__device__ __forceinline__ void retractItem(const DeviceProblem& p, int i, double cg, double cn, double* acc) {
  const int nBlk = p.nPose + p.nExt + p.nSb;
  if (i < nBlk) {
    const bool isSb = i >= p.nPose + p.nExt, isPose = i < p.nPose;
    const int slot = isSb ? i - p.nPose - p.nExt : (isPose ? i : i - p.nPose);
    const int len = isSb ? 9 : 7, nd = isSb ? 9 : 6;
    const double* xp = isSb ? p.sb + (size_t)slot * 9 : (isPose ? p.pose : p.ext) + (size_t)slot * 7;
    double* xc = isSb ? p.sbC + (size_t)slot * 9 : (isPose ? p.poseC : p.extC) + (size_t)slot * 7;
    const int off = isSb ? p.sbOff[slot] : (isPose ? p.poseOff[slot] : p.extOff[slot]);
    double x[9], v[9], y[9], xo[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      x[k] = k < len ? xp[k] : 0.0;
      v[k] = (off >= 0 && k < nd) ? p.vC[off + k] : 0.0;
      y[k] = (off >= 0 && k < nd) ? p.yC[off + k] : 0.0;
    }
    retractBlockValues(isSb, off >= 0, p.ownsCamera != 0, x, v, y, cg, cn, xo, acc);
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (k < len) xc[k] = xo[k];
  } else if (i < nBlk + p.L) {
    const int l = i - nBlk;
    const double4 xx = reinterpret_cast<const double4*>(p.lm)[l];
    const double x[4] = {xx.x, xx.y, xx.z, xx.w};
    double v[3], y[3], xo[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) { v[k] = p.vL[3 * l + k]; y[k] = p.yL[3 * l + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      xo[k] = x[k] + (cg * v[k] - cn * y[k]);
      acc[0] += (x[k] - xo[k]) * (x[k] - xo[k]);
      acc[1] += x[k] * x[k];
    }
    xo[3] = x[3] + 0.0;
    acc[1] += x[3] * x[3];
    reinterpret_cast<double4*>(p.lmC)[l] = double4{xo[0], xo[1], xo[2], xo[3]};
  }
}

// stand-alone dogleg step + retraction (re-used linearisation after a rejected step, multi-GPU mode, wide windows);
// the last block reduces the norms
__device__ __forceinline__ void k_step_retract_body(const DeviceProblem& p, double radius) {
  __shared__ double red[16 * 2];
  __shared__ int lastFlag;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const SolverScalars& sc = *p.scal;
  const DoglegCoeff c = doglegCoefficients(sc.gHatSq, sc.jgSq, sc.gnHatSq, sc.gDotGn, sc.jySq, sc.jvDotJy, sc.jvDotR, sc.jyDotR, radius);
  if (i == 0) { p.scal->doglegStepNorm = c.stepNorm; p.scal->jdSq = c.jdSq; p.scal->jdDotR = c.jdDotR; }
  double acc[2] = {0, 0};  // |x - x_cand|^2, |x|^2
  retractItem(p, i, c.cg, c.cn, acc);
  const double mine = blockSumK<2>(acc, red, -1);
  if (threadIdx.x < 2) cstore(p.partial + (size_t)(threadIdx.x == 0 ? PS_STEP : PS_XNORM) * kMaxPartials + blockIdx.x, mine);
  if (!lastBlockDoneLight(&p.tickets[TK_STEP], &lastFlag)) return;
  for (int k = 0; k < 2; ++k) {
    double s = 0;
    const double* src = p.partial + (size_t)(k == 0 ? PS_STEP : PS_XNORM) * kMaxPartials;
    for (int j = threadIdx.x; j < (int)gridDim.x; j += blockDim.x) s += cload(src + j);
    acc[k] = s;
  }
  const double tot = blockSumK<2>(acc, red, -1);
  if (threadIdx.x == 0) p.scal->stepNormSq = tot;
  if (threadIdx.x == 1) p.scal->xNormSq = tot;
  if (threadIdx.x == 0) p.tickets[TK_STEP] = 0;
}
__global__ __launch_bounds__(256) void k_step_retract(DeviceProblem p, double radius) { k_step_retract_body(p, radius); }
// (batched form: blockIdx.y = the window of the batch, its problem and trust-region scalars from the slot table)
__global__ __launch_bounds__(256) void k_step_retract_batch(const BatchSlot* __restrict__ slots) {
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchReuse)) return;
  k_step_retract_body(sl.p, sl.radius);
}


// partial slots of the post-solve pass
constexpr int kPostK = 9;  // A=|Jv|^2 B=|Jy|^2 C=Jv.Jy D=Jv.r E=Jy.r gHat gnHat gDotGn gradMax
__device__ __constant__ int kPostSlot[kPostK] = {PS_JV_SQ, PS_JY_SQ, PS_JVJY, PS_JV_DOT, PS_JY_DOT, PS_GHAT, PS_GNHAT, PS_GDOTGN, PS_GRADMAX};
// the same table for compile-time indices (a __constant__ lookup is a memory round trip: nine of them, each followed by its
// dependent partial load, used to serialise the tail of the post-solve pass)
__device__ constexpr int postSlotC(int k) {
  constexpr int tbl[kPostK] = {PS_JV_SQ, PS_JY_SQ, PS_JVJY, PS_JV_DOT, PS_JY_DOT, PS_GHAT, PS_GNHAT, PS_GDOTGN, PS_GRADMAX};
  return tbl[k];
}

// One pass over the whole linearisation right after the reduced solve (y_C, and v_C = g/htil from the solver):
//  landmark blocks (16 lanes per landmark): back-substitution y_l = Vinv (b_l - sum_i Jl_i^T Jc_i y_C), v_l = g_l/htil_l,
//      then per observation u_v = J v and u_y = J y and the five sums that price every dogleg step of this
//      linearisation (J*delta = cg*u_v - cn*u_y), plus the landmark part of the scaled-gradient norms;
//  factor blocks (one wave per factor): the same five sums over the rows of the small factors;
//  last block: camera part of the norms and the marginalisation prior (H-space);
//  whichever block finishes last reduces all partials into SolverScalars group B.
template <bool WITH_EXT>
__device__ __forceinline__ void k_post_solve_body(const DeviceProblem& p, int nLmBlocks, int nFacBlocks, double fuseRadius) {
  __shared__ double red[16 * kPostK];
  __shared__ int lastFlag;
  const int t = threadIdx.x, b = blockIdx.x;
  double acc[kPostK];
#pragma unroll
  for (int k = 0; k < kPostK; ++k) acc[k] = 0;
  SVIN_ARGS(SA(p.lmPtr), SA(p.yC), SA(p.vC), SA(p.poseOff), SA(p.extOff), SA(p.scal), SA(p.pose), SA(p.sb), SA(p.sbOff), SA(p.obsIdx),
            SA(p.JpCur), SA(p.JlCur), SA(p.rCur), SA(p.S), SA(p.d), SA(p.L), SA(p.N), SA(p.nPose), SA(p.nExt), SA(p.nSb));
  TRACE(0);
  // clear the accumulators of the next linearisation (nothing reads S / gRed / hC after the solve; gFull is
  // still needed by the last block below and is cleared there)
  for (int i = b * blockDim.x + t; i < p.ldS * p.d; i += gridDim.x * blockDim.x) p.S[i] = 0.0;
  for (int i = b * blockDim.x + t; i < p.d; i += gridDim.x * blockDim.x) { p.gRed[i] = 0.0; p.hC[i] = 0.0; }
  // Staged in LDS by every block at its start: the camera-side solution vectors (tiny, read by every observation: one
  // copy instead of a dependent global load per observation; wide windows keep reading them through L2), the block ->
  // row maps, and -- because any block may turn out to be the last one, whose tail is the serial end of the
  // iteration -- the parameter blocks themselves, so that the fused retraction below starts without a memory round trip
  constexpr int kStageMax = 1024, kStageBlk = 128, kStageItems = 96;   // (1024: config #4's d = 960 -- through L2 the camera vectors were a third dependent round trip per observation)
  __shared__ double sYV[2 * kStageMax];
  __shared__ int sOff[2 * kStageBlk];
  __shared__ double sItemX[kStageItems * 9];
  __shared__ int sItemOff[kStageItems];
  const bool staged = p.d <= kStageMax;
  const bool stagedOff = p.nPose <= kStageBlk && p.nExt <= kStageBlk;
  const int nBlkItems = p.nPose + p.nExt + p.nSb;
  const bool stagedItems = fuseRadius > 0.0 && staged && nBlkItems <= kStageItems;
  const int cholFailIn = p.scal->cholFail;  // set by earlier kernels only
  // Memory round trips of a landmark block: (1) the observation range of the lane group's first landmark, (2) that
  // observation's Jacobians and residual.  The staging loads are issued between the two and land in their shadow.
  struct RawObs { uint32_t idx; double jp[12], je[12], jl[6], rr[2]; };
  const size_t N = (size_t)p.N;
  auto loadRaw = [&](size_t o, RawObs& w) {
    w.idx = p.obsIdx[o];
#pragma unroll
    for (int k = 0; k < 12; ++k) w.jp[k] = p.JpCur[k * N + o];
    if (WITH_EXT) {
#pragma unroll
      for (int k = 0; k < 12; ++k) w.je[k] = p.JeCur[k * N + o];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) w.jl[k] = p.JlCur[k * N + o];
    w.rr[0] = p.rCur[o]; w.rr[1] = p.rCur[N + o];
  };
  int firstStart = 0, firstEnd = 0;
  const bool lmBlock = b < nLmBlocks && b * 16 + (t >> 4) < p.L;
  if (b < nLmBlocks) {
    // the five range ends a wave needs (four landmarks) through the scalar cache: wave-uniform addresses, a read-only
    // table that stays resident from iteration to iteration -- a vector load from L2 / HBM was the first of this
    // block's two dependent round trips
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int base = b * 16 + wv * 4;
    int e[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) e[k] = p.lmPtr[min(base + k, p.L)];
    const int g4 = (t >> 4) & 3;
    firstStart = g4 == 0 ? e[0] : (g4 == 1 ? e[1] : (g4 == 2 ? e[2] : e[3]));
    firstEnd = g4 == 0 ? e[1] : (g4 == 1 ? e[2] : (g4 == 2 ? e[3] : e[4]));
  }
  double syv[2 * (kStageMax / 256)];
  if (staged) {
#pragma unroll
    for (int k = 0; k < kStageMax / 256; ++k) {
      const bool in = t + 256 * k < p.d;
      syv[2 * k] = in ? p.yC[t + 256 * k] : 0.0;
      syv[2 * k + 1] = in ? p.vC[t + 256 * k] : 0.0;
    }
  }
  int so0 = -1, so1 = -1;
  if (stagedOff) {
    if (t < p.nPose) so0 = p.poseOff[t];
    if (t < p.nExt) so1 = p.extOff[t];
  }
  double ix[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ioff = -1;
  if (stagedItems && t < nBlkItems) {
    const bool isSb = t >= p.nPose + p.nExt, isPose = t < p.nPose;
    const int slot = isSb ? t - p.nPose - p.nExt : (isPose ? t : t - p.nPose);
    const double* xp = isSb ? p.sb + (size_t)slot * 9 : (isPose ? p.pose : p.ext) + (size_t)slot * 7;
    ioff = isSb ? p.sbOff[slot] : (isPose ? p.poseOff[slot] : p.extOff[slot]);
#pragma unroll
    for (int k = 0; k < 9; ++k) ix[k] = (k < (isSb ? 9 : 7)) ? xp[k] : 0.0;
  }
  RawObs raw0;
  const bool has0 = lmBlock && (t & 15) < firstEnd - firstStart;
  if (has0) loadRaw((size_t)firstStart + (t & 15), raw0);
  if (staged) {
#pragma unroll
    for (int k = 0; k < kStageMax / 256; ++k)
      if (t + 256 * k < p.d) { sYV[t + 256 * k] = syv[2 * k]; sYV[kStageMax + t + 256 * k] = syv[2 * k + 1]; }
  }
  if (stagedOff) {
    if (t < p.nPose) sOff[t] = so0;
    if (t < p.nExt) sOff[kStageBlk + t] = so1;
  }
  if (stagedItems && t < nBlkItems) {
    sItemOff[t] = ioff;
#pragma unroll
    for (int k = 0; k < 9; ++k) sItemX[t * 9 + k] = ix[k];
  }
  if (staged || stagedOff || stagedItems) __syncthreads();
  if (b < nLmBlocks) {
   // the body is instantiated twice, once on the staged LDS copies and once on the global arrays: ONE pointer variable that
   // may hold either address space makes every access through it a FLAT instruction (15-36 of them per observation here)
   auto landmarkBlock = [&](const double* yCs, const double* vCs, const int* poseOffS, const int* extOffS) __attribute__((always_inline)) {
    const int grp = t >> 4, gl = t & 15;
    // u_y = Jc y_C, u_v = Jc v_C, Jl and r of an observation whose raw data are in registers
    auto finishObs = [&](const RawObs& w, double* jl, double* uy, double* uv, double* rr) {
      const int offP = poseOffS[w.idx & 0xfff];
      uy[0] = uy[1] = uv[0] = uv[1] = 0;
      if (offP >= 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double j0 = w.jp[a], j1 = w.jp[6 + a], y = yCs[offP + a], v = vCs[offP + a];
          uy[0] += j0 * y; uy[1] += j1 * y; uv[0] += j0 * v; uv[1] += j1 * v;
        }
      }
      if (WITH_EXT) {
        const int offE = extOffS[(w.idx >> 12) & 0xfff];
        if (offE >= 0) {
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            const double j0 = w.je[a], j1 = w.je[6 + a], y = yCs[offE + a], v = vCs[offE + a];
            uy[0] += j0 * y; uy[1] += j1 * y; uv[0] += j0 * v; uv[1] += j1 * v;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) jl[k] = w.jl[k];
      rr[0] = w.rr[0]; rr[1] = w.rr[1];
    };
    auto loadObs = [&](size_t o, double* jl, double* uy, double* uv, double* rr) {
      RawObs w;
      loadRaw(o, w);
      finishObs(w, jl, uy, uv, rr);
    };
    for (int l = b * 16 + grp; l < p.L; l += nLmBlocks * 16) {
      const bool first = l == b * 16 + grp;
      const int start = first ? firstStart : p.lmPtr[l], n = (first ? firstEnd : p.lmPtr[l + 1]) - start;
      // the landmark's own quantities do not depend on the observation loop: requested up front
      const double g0 = p.bl[3 * l], g1 = p.bl[3 * l + 1], g2 = p.bl[3 * l + 2];
      const double* vi = p.Vinv + 6 * (size_t)l;
      const double vi0 = vi[0], vi1 = vi[1], vi2 = vi[2], vi3 = vi[3], vi4 = vi[4], vi5 = vi[5];
      const double h0 = p.hL[3 * l], h1 = p.hL[3 * l + 1], h2 = p.hL[3 * l + 2];
      double t0 = 0, t1 = 0, t2 = 0;
      double cjl[6], cuy[2], cuv[2], crr[2];  // first observation of this lane stays in registers
      for (int i = gl; i < n; i += 16) {
        double jl[6], uy[2], uv[2], rr[2];
        if (first && i == gl) finishObs(raw0, jl, uy, uv, rr);
        else loadObs((size_t)start + i, jl, uy, uv, rr);
        t0 += jl[0] * uy[0] + jl[3] * uy[1];
        t1 += jl[1] * uy[0] + jl[4] * uy[1];
        t2 += jl[2] * uy[0] + jl[5] * uy[1];
        if (i == gl) {
#pragma unroll
          for (int k = 0; k < 6; ++k) cjl[k] = jl[k];
          cuy[0] = uy[0]; cuy[1] = uy[1]; cuv[0] = uv[0]; cuv[1] = uv[1]; crr[0] = rr[0]; crr[1] = rr[1];
        }
      }
      t0 = rowSum16(t0); t1 = rowSum16(t1); t2 = rowSum16(t2);
      const double q0 = g0 - t0, q1 = g1 - t1, q2 = g2 - t2;
      const double y0 = vi0 * q0 + vi1 * q1 + vi2 * q2;
      const double y1 = vi1 * q0 + vi3 * q1 + vi4 * q2;
      const double y2 = vi2 * q0 + vi4 * q1 + vi5 * q2;
      const double v0 = g0 / h0, v1 = g1 / h1, v2 = g2 / h2;
      if (gl == 0) {
        cstore(p.yL + 3 * l, y0); cstore(p.yL + 3 * l + 1, y1); cstore(p.yL + 3 * l + 2, y2);   // read by the last block's fused step
        cstore(p.vL + 3 * l, v0); cstore(p.vL + 3 * l + 1, v1); cstore(p.vL + 3 * l + 2, v2);
        acc[5] += g0 * g0 / h0 + g1 * g1 / h1 + g2 * g2 / h2;
        acc[6] += h0 * y0 * y0 + h1 * y1 * y1 + h2 * y2 * y2;
        acc[7] += -(g0 * y0 + g1 * y1 + g2 * y2);
        acc[8] = fmax(acc[8], fmax(fabs(g0), fmax(fabs(g1), fabs(g2))));
      }
      for (int i = gl; i < n; i += 16) {
        double jl[6], uy[2], uv[2], rr[2];
        if (i == gl) {
#pragma unroll
          for (int k = 0; k < 6; ++k) jl[k] = cjl[k];
          uy[0] = cuy[0]; uy[1] = cuy[1]; uv[0] = cuv[0]; uv[1] = cuv[1]; rr[0] = crr[0]; rr[1] = crr[1];
        } else {
          loadObs((size_t)start + i, jl, uy, uv, rr);
        }
        const double jv0 = uv[0] + jl[0] * v0 + jl[1] * v1 + jl[2] * v2, jv1 = uv[1] + jl[3] * v0 + jl[4] * v1 + jl[5] * v2;
        const double jy0 = uy[0] + jl[0] * y0 + jl[1] * y1 + jl[2] * y2, jy1 = uy[1] + jl[3] * y0 + jl[4] * y1 + jl[5] * y2;
        acc[0] += jv0 * jv0 + jv1 * jv1;
        acc[1] += jy0 * jy0 + jy1 * jy1;
        acc[2] += jv0 * jy0 + jv1 * jy1;
        acc[3] += jv0 * rr[0] + jv1 * rr[1];
        acc[4] += jy0 * rr[0] + jy1 * rr[1];
      }
    }
   };
   if (staged && stagedOff) landmarkBlock(sYV, sYV + kStageMax, sOff, sOff + kStageBlk);
   else landmarkBlock(p.yC, p.vC, p.poseOff, p.extOff);
  } else if (b < nLmBlocks + nFacBlocks) {
    {
      // one wave per factor, lane = (row a = lane & 15, column quarter lane >> 4): the row's products with v_C and y_C are
      // split over four lanes (a thread per row walked up to 30 columns of dependent global loads: these blocks were the
      // last to finish), the solution vectors come from the staged copy
      const int wave = t >> 6, lane = t & 63, a = lane & 15, cq = lane >> 4;
     auto factorBlock = [&](const double* yCf, const double* vCf) __attribute__((always_inline)) {
      for (int f = (b - nLmBlocks) * 4 + wave; f < p.F; f += nFacBlocks * 4) {
        const FactorLin& lin = p.linCur[f];
        const int m = lin.m, ncols = lin.ncols;
        const int o0 = lin.off[0], o1 = lin.off[1], o2 = lin.off[2], o3 = lin.off[3];
        const int d0 = lin.dim[0], d1 = lin.dim[1], d2 = lin.dim[2];
        double uv = 0, uy = 0;
        if (a < m) {
          for (int c = cq; c < ncols; c += 4) {
            // column c -> (block, offset within the block)
            const int bb = (c >= d0) + (c >= d0 + d1) + (c >= d0 + d1 + d2);
            const int off = bb == 0 ? o0 : (bb == 1 ? o1 : (bb == 2 ? o2 : o3));
            const int cc = c - (bb == 0 ? 0 : (bb == 1 ? d0 : (bb == 2 ? d0 + d1 : d0 + d1 + d2)));
            if (off >= 0) {
              const double j = lin.J[a * ncols + c];
              uv += j * vCf[off + cc];
              uy += j * yCf[off + cc];
            }
          }
        }
        // the four column quarters of a row sit in the four 16-lane rows of the wave
        const double uvA = uv + __shfl_xor(uv, 16, 64), uyA = uy + __shfl_xor(uy, 16, 64);
        const double uvT = uvA + __shfl_xor(uvA, 32, 64), uyT = uyA + __shfl_xor(uyA, 32, 64);
        if (cq == 0 && a < m) {
          const double r = lin.r[a];
          acc[0] += uvT * uvT; acc[1] += uyT * uyT; acc[2] += uvT * uyT; acc[3] += uvT * r; acc[4] += uyT * r;
        }
      }
     };
     if (staged) factorBlock(sYV, sYV + kStageMax);
     else factorBlock(p.yC, p.vC);
    }
  } else if (p.ownsCamera) {
    // camera part of the scaled-gradient norms
    for (int i = t; i < p.d; i += blockDim.x) {
      const double g = p.gFull[i], ht = p.htilC[i], y = p.yC[i];
      acc[5] += g * g / ht;
      acc[6] += ht * y * y;
      acc[7] += -g * y;
      acc[8] = fmax(acc[8], fabs(g));
    }
    // marginalisation prior: |J_eff v|^2 = (Mv)^T Ht (Mv), (J_eff v).r = (Mv)^T grad  (same with y)
    const int m = p.priorM;
    if (m > 0) {
      for (int i = t; i < m; i += blockDim.x) {
        const int bi = priorFindBlock(p, i);
        const PriorBlock& B = p.priorBlk[bi];
        const int off = blockOff(p, B.kind, B.slot);
        const int li = i - B.ord;
        double v = 0, y = 0;
        if (off >= 0) {
          if (B.kind != B_SB && li >= 3) {
            for (int c = 0; c < 3; ++c) {
              const double w = p.priorM3[9 * bi + (li - 3) * 3 + c];
              v += w * p.vC[off + 3 + c];
              y += w * p.yC[off + 3 + c];
            }
          } else {
            v = p.vC[off + li];
            y = p.yC[off + li];
          }
        }
        p.priorMv[i] = v;
        p.priorMy[i] = y;
      }
      __syncthreads();
      for (int i = t; i < m; i += blockDim.x) {
        double sv = 0, sy = 0;
        const double* row = p.priorH + (size_t)i * m;
        for (int k = 0; k < m; ++k) { sv += row[k] * p.priorMv[k]; sy += row[k] * p.priorMy[k]; }
        const double mv = p.priorMv[i], my = p.priorMy[i], gr = p.priorGrad[i];
        acc[0] += mv * sv; acc[1] += my * sy; acc[2] += mv * sy; acc[3] += mv * gr; acc[4] += my * gr;
      }
    }
  }
  TRACE(1);
  TRACE(b < nLmBlocks ? 8 : (b < nLmBlocks + nFacBlocks ? 9 : 10));
  const double mine = blockSumK<kPostK>(acc, red, 8);
  TRACE(11);
  if (t < kPostK) cstore(p.partial + (size_t)kPostSlot[t] * kMaxPartials + b, mine);
  if (!lastBlockDoneLight(&p.tickets[TK_POST], &lastFlag)) return;   // partials, y_l and v_l are cstore()d
  TRACE(2);
  // The tail from here on is the serial end of the iteration: one round trip for the partials of the nine sums, then the
  // first round of landmarks with their y_l / v_l (eight per thread) in the shadow of the arithmetic.
  constexpr int kPer = 8;
  double lx[kPer][4], lv[kPer][3], ly[kPer][3];
  auto loadLandmarks = [&](int l0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int l = min(l0 + u * (int)blockDim.x + t, p.L - 1);
      const double4 xx = reinterpret_cast<const double4*>(p.lm)[l];
      lx[u][0] = xx.x; lx[u][1] = xx.y; lx[u][2] = xx.z; lx[u][3] = xx.w;
#pragma unroll
      for (int k = 0; k < 3; ++k) { lv[u][k] = cload(p.vL + 3 * l + k); ly[u][k] = cload(p.yL + 3 * l + k); }
    }
  };
  // final reduction over the blocks, fixed order: thread k-strided per slot; the first round's nine loads per thread are
  // all in flight before the first one is consumed (and ahead of the landmark loads: the memory counter retires in order)
  {
    double x0[kPostK];
#pragma unroll
    for (int k = 0; k < kPostK; ++k) x0[k] = (t < (int)gridDim.x) ? cload(p.partial + (size_t)postSlotC(k) * kMaxPartials + t) : 0.0;
#pragma unroll
    for (int k = 0; k < kPostK; ++k) {
      double s = x0[k];
      const double* src = p.partial + (size_t)postSlotC(k) * kMaxPartials;
      for (int i = t + blockDim.x; i < (int)gridDim.x; i += blockDim.x) { const double x = cload(src + i); s = (k == 8) ? fmax(s, x) : s + x; }
      acc[k] = s;
    }
  }
  TRACE(12);
  // the landmark loads (64 per thread: ~2 us of this CU's load pipe) go out once the partials are in -- queued ahead of
  // them they delay the partial loads of the other waves -- and land while the sums, the dogleg coefficients and the block
  // retraction are being computed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const bool lmHere = fuseRadius > 0.0 && p.L > 0 && !p.lmDeferred;
  if (lmHere) loadLandmarks(0);
  const double tot = blockSumK<kPostK>(acc, red, 8);
  TRACE(3);
  __shared__ double grpB[8];
  // slot order A B C D E gHat gnHat gDotGn -> fields 1 4 5 6 7 0 2 3 of group B
  const int field = (t == 0) ? 1 : (t == 1) ? 4 : (t == 2) ? 5 : (t == 3) ? 6 : (t == 4) ? 7 : (t == 5) ? 0 : (t == 6) ? 2 : 3;
  if (t < 8) grpB[field] = tot;
  // (the global stores of the record, the clearing of gFull and the ticket reset wait until the very end: a global store
  // ahead of the barrier below would hold the whole block until it has completed)
  auto publishGroupB = [&]() {
    if (t < kPostK) {
      double* dst = &p.scal->gHatSq;
      if (t < 8) dst[field] = tot;
      else {
        p.scal->gradMax = tot; p.scal->failMax = (double)cholFailIn; p.scal->cholFail = 0;
        if (p.world > 1 || p.rank < 0) {   // sharded: my pair in my slot, zeros elsewhere (the sum all-reduce gathers)
          const int me = p.rank < 0 ? 0 : p.rank;
          for (int k = 0; k < kScalGatherSlots / 2; ++k) { p.scal->gather[2 * k] = (k == me) ? tot : 0.0; p.scal->gather[2 * k + 1] = (k == me) ? (double)cholFailIn : 0.0; }
        }
      }
    }
    for (int i = t; i < p.d; i += blockDim.x) p.gFull[i] = 0.0;
    if (t == 0) p.tickets[TK_POST] = 0;
  };
  if (fuseRadius <= 0.0) publishGroupB();
  if (fuseRadius > 0.0) {
    // single-GPU narrow windows: this block also takes the dogleg step and retracts (k_step_retract's work) --
    // one launch less on the critical path of every accepted iteration
    __syncthreads();
    const DoglegCoeff c = doglegCoefficients(grpB[0], grpB[1], grpB[2], grpB[3], grpB[4], grpB[5], grpB[6], grpB[7], fuseRadius);
    if (t == 0) { p.scal->doglegStepNorm = c.stepNorm; p.scal->jdSq = c.jdSq; p.scal->jdDotR = c.jdDotR; }
    if (t == 1 && p.lmDeferred) { p.scal->spareA0 = c.cg; p.scal->spareA1 = c.cn; }   // for the candidate evaluation's landmark step
    double a2[2] = {0, 0};
    TRACE(4);
    if (stagedItems) {
      // pose / extrinsics blocks on wave 0, speed-and-bias blocks on wave 1: the two code paths run side by side instead of
      // one after the other inside a wave
      const int nPE = p.nPose + p.nExt;
      const bool split = nPE <= 64 && p.nSb <= 64;
      const int item = split ? (t < 64 ? (t < nPE ? t : -1) : (t < 128 && t - 64 < p.nSb ? nPE + t - 64 : -1)) : (t < nBlkItems ? t : -1);
      if (item >= 0) {
        const bool isSb = item >= nPE, isPose = item < p.nPose;
        const int slot = isSb ? item - nPE : (isPose ? item : item - p.nPose);
        const int off = sItemOff[item], nd = isSb ? 9 : 6;
        double x[9], v[9], y[9], xo[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          x[k] = sItemX[item * 9 + k];
          v[k] = (off >= 0 && k < nd) ? sYV[kStageMax + off + k] : 0.0;
          y[k] = (off >= 0 && k < nd) ? sYV[off + k] : 0.0;
        }
        retractBlockValues(isSb, off >= 0, p.ownsCamera != 0, x, v, y, c.cg, c.cn, xo, a2);
        double* xc = isSb ? p.sbC + (size_t)slot * 9 : (isPose ? p.poseC : p.extC) + (size_t)slot * 7;
#pragma unroll
        for (int k = 0; k < 9; ++k)
          if (k < (isSb ? 9 : 7)) xc[k] = xo[k];
      }
    } else {
      for (int i = t; i < nBlkItems; i += blockDim.x) retractItem(p, i, c.cg, c.cn, a2);
    }
    TRACE(5);
    // landmarks: eight per thread per round, all loads of a round issued before its first store (the first round was
    // requested at the top of the tail)
    for (int l0 = 0; lmHere && l0 < p.L; l0 += kPer * (int)blockDim.x) {
      if (l0 > 0) loadLandmarks(l0);
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int l = l0 + u * (int)blockDim.x + t;
        if (l < p.L) {
          double xo[4];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            xo[k] = lx[u][k] + (c.cg * lv[u][k] - c.cn * ly[u][k]);
            a2[0] += (lx[u][k] - xo[k]) * (lx[u][k] - xo[k]);
            a2[1] += lx[u][k] * lx[u][k];
          }
          xo[3] = lx[u][3] + 0.0;
          a2[1] += lx[u][3] * lx[u][3];
          reinterpret_cast<double4*>(p.lmC)[l] = double4{xo[0], xo[1], xo[2], xo[3]};
        }
      }
    }
    TRACE(6);
    const double tt = blockSumK<2>(a2, red, -1);
    if (t == 0) p.scal->stepNormSq = tt;
    if (t == 1) p.scal->xNormSq = tt;
    publishGroupB();
  }
  TRACE(7);
}
template <bool WITH_EXT>
__global__ __launch_bounds__(256) void k_post_solve(DeviceProblem p, int nLmBlocks, int nFacBlocks, double fuseRadius) { k_post_solve_body<WITH_EXT>(p, nLmBlocks, nFacBlocks, fuseRadius); }
// (the same body held to 256 registers -- two workgroups per CU -- for grids of more blocks than the chip has CUs: wide windows)
template <bool WITH_EXT>
__global__ __launch_bounds__(256, 2) void k_post_solve_wide(DeviceProblem p, int nLmBlocks, int nFacBlocks, double fuseRadius) { k_post_solve_body<WITH_EXT>(p, nLmBlocks, nFacBlocks, fuseRadius); }
// (batched form: blockIdx.y = the window of the batch, its problem and trust-region scalars from the slot table)
template <bool WITH_EXT>
__global__ __launch_bounds__(256, SVIN_BATCH_OCC_POST) void k_post_solve_batch(const BatchSlot* __restrict__ slots, int nLmBlocks, int nFacBlocks) {
  const BatchSlot& sl = batchSlot(slots);
  if (!(sl.stages & kBatchFull)) return;
  k_post_solve_body<WITH_EXT>(sl.p, nLmBlocks, nFacBlocks, sl.radius);
}


void launchDoglegPrepare(const DeviceProblem& p, hipStream_t s, double fuseRadius) {
  const int nLm = (p.L > 0 && p.N > 0) ? min((p.L + 15) / 16, 1024) : 0;
  const int nFac = p.F > 0 ? min((p.F + 3) / 4, 1024) : 0;
  if (nLm + nFac + 1 > kEvalSplitBlocks && !optOn(kOptNoEvalSplit)) {
    if (p.anyExtVariable) hipLaunchKernelGGL(k_post_solve_wide<true>, dim3(nLm + nFac + 1), dim3(256), 0, s, p, nLm, nFac, fuseRadius);
    else hipLaunchKernelGGL(k_post_solve_wide<false>, dim3(nLm + nFac + 1), dim3(256), 0, s, p, nLm, nFac, fuseRadius);
    return;
  }
  if (p.anyExtVariable) hipLaunchKernelGGL(k_post_solve<true>, dim3(nLm + nFac + 1), dim3(256), 0, s, p, nLm, nFac, fuseRadius);
  else hipLaunchKernelGGL(k_post_solve<false>, dim3(nLm + nFac + 1), dim3(256), 0, s, p, nLm, nFac, fuseRadius);
}

// final single-block reduction of the cost partials into SolverScalars (used when no later evaluation kernel
// can take it over, see evaluateAll)
__global__ __launch_bounds__(256) void k_reduce_cost(DeviceProblem p, int nA, int nB) {
  __shared__ double red[72];
  reduceCost(p, nA, nB, red);
}

void launchCost(const DeviceProblem& p, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_cost, dim3(1), dim3(256), 0, s, p, p.N > 0 ? evalGrid(p.N) : 0, p.F);
}

void launchDoglegStep(const DeviceProblem& p, double radius, hipStream_t s) {
  const int nB = (p.nPose + p.nExt + p.nSb + p.L + 255) / 256;
  hipLaunchKernelGGL(k_step_retract, dim3(nB), dim3(256), 0, s, p, radius);
}

// ================================================================ batched rounds (kernels.hpp: BatchSlot)
// The windows of a batch have the same launch geometry: every grid below is the one the launcher of the single window computes
// from `geom`, with the window as blockIdx.y -- gridDim.x, which the kernels' last-block logic and partial sums read, is what it is
// for the window on its own, and so is every reduction order: a window ends bit for bit where it ends alone.
bool batchSupported(const DeviceProblem& p) {
  if (!(p.L > 0 && p.N > 0 && p.dC > 0 && p.schurDense && p.d > 0)) return false;
  if (!canFuseEvaluation(p) || optOn(kOptSplitEval) || optOn(kOptNoFuseStep) || optOn(kOptNoDeferLm)) return false;
  if ((p.nPose + p.nExt + p.nSb + p.L) > 16384) return false;            // (the fused dogleg step of k_post_solve)
  if (solverClass(p.d, p.sPadded != 0) != 0 || cholBorderRows(p.d, p.sPadded != 0) != 0) return false;   // LDS-resident solver, no border
  if (priorAccBlocks(p) > 0 && !p.ownsCamera) return false;
  if (p.nLocked > 0) return false;   // (reduced pose manifolds: k_lock_rows has no batched form)
  if (p.nHostFactors > 0) return false;
  return true;
}
void launchBatchRound(const BatchSlot* dSlots, const DeviceProblem& geom, int n, int stagesUnion, bool cand, hipStream_t s) {
  const DeviceProblem& p = geom;
  (void)hipGetLastError();
  if (stagesUnion & kBatchFull) {
    const DenseSchurPlan plan = denseSchurPlan(p);
    const int nFac = p.F, nPri = priorAccBlocks(p);
    const dim3 grid(p.nSlabs + nFac + nPri, n);
#define LAUNCH(MAXT, E, NWV)                                                                                        \
  do {                                                                                                              \
    ensureDynamicLds((const void*)k_schur_dense_batch<MAXT, E, NWV>, plan.ldsBytes);                                \
    hipLaunchKernelGGL((k_schur_dense_batch<MAXT, E, NWV>), grid, dim3(64 * NWV), plan.ldsBytes, s, dSlots, p.nSlabs, nFac); \
  } while (0)
    if (plan.nTr > 12) LAUNCH(17, true, 8);
    else if (plan.nTr > 8) LAUNCH(10, true, 8);
    else if (plan.aMfma) LAUNCH(9, true, 4);
    else LAUNCH(9, false, 4);
#undef LAUNCH
    const int nRed = p.dC * p.dC + 3 * p.dC;
    hipLaunchKernelGGL(k_reduce_slabs_batch, dim3((nRed + 15) / 16, n), dim3(256), 0, s, dSlots);
    const int dpad = ((p.d + 15) / 16) * 16;
    const size_t ldsChol = cholLdsBytes(dpad / 16);
    ensureDynamicLds((const void*)k_chol_solve_lds_batch<0>, ldsChol);
    hipLaunchKernelGGL(k_chol_solve_lds_batch<0>, dim3(1, n), dim3(kCholLdsThreads), ldsChol, s, dSlots, dpad, 1, 0, (const double*)nullptr);
    const int nLm = min((p.L + 15) / 16, 1024);
    const int nFacP = p.F > 0 ? min((p.F + 3) / 4, 1024) : 0;
    if (p.anyExtVariable) hipLaunchKernelGGL(k_post_solve_batch<true>, dim3(nLm + nFacP + 1, n), dim3(256), 0, s, dSlots, nLm, nFacP);
    else hipLaunchKernelGGL(k_post_solve_batch<false>, dim3(nLm + nFacP + 1, n), dim3(256), 0, s, dSlots, nLm, nFacP);
  }
  if (stagesUnion & kBatchReuse) {
    const int nB = (p.nPose + p.nExt + p.nSb + p.L + 255) / 256;
    hipLaunchKernelGGL(k_step_retract_batch, dim3(nB, n), dim3(256), 0, s, dSlots);
  }
  if (stagesUnion & kBatchEval) {
    const int nR = (p.N + 255) / 256, pri = p.priorM > 0 ? 1 : 0;
    const size_t stage = evalSplitStageBytes(p);
    if (p.anyExtVariable) hipLaunchKernelGGL(k_eval_reproj_batch<true>, dim3(nR, n), dim3(256), stage, s, dSlots, cand ? 1 : 0);
    else hipLaunchKernelGGL(k_eval_reproj_batch<false>, dim3(nR, n), dim3(256), stage, s, dSlots, cand ? 1 : 0);
    hipLaunchKernelGGL(k_eval_rest_batch, dim3(p.F + pri, n), dim3(256), 0, s, dSlots, cand ? 1 : 0, nR, pri);
  }
  checkSolverLaunches();
}

// ================================================================ K9: landmark quality (Estimator.cpp:902-923)
// H = sum J_lm^T J_lm over all observations WITHOUT loss correction (Map::getLhs), symmetric 3x3
// eigenvalues by cyclic Jacobi, quality = sqrt(lmin)/sqrt(lmax) (0 if lmin < 1e-12).
// 16 lanes per landmark, one observation per lane and turn (a thread per landmark walked its ~10 observations serially:
// 23 us for 2 000 landmarks)
__global__ __launch_bounds__(256) void k_landmark_quality(DeviceProblem p, double* __restrict__ quality) {
  const int l = blockIdx.x * 16 + (threadIdx.x >> 4), gl = threadIdx.x & 15;
  if (l >= p.L) return;   // whole 16-lane rows leave together
  double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
  const double4 hp = reinterpret_cast<const double4*>(p.lm)[l];
  const double hpw[4] = {hp.x, hp.y, hp.z, hp.w};
  const int oEnd = p.lmPtr[l + 1];
  for (int o = p.lmPtr[l] + gl; o < oEnd; o += 16) {
    const uint32_t idx = p.obsIdx[o];
    double rr[2], jp[12], jl[6], je[12];
    if (((idx >> 24) & 0xf) == kPriorCam) {   // landmark prior: its rows of S (Map::getLhs sums every residual of the block)
      const double* pr = p.lmPrior + 12 * (int)p.obsUv[2 * (size_t)o];
      const bool second = p.obsUv[2 * (size_t)o + 1] != 0.0;
      for (int k = 0; k < 3; ++k) { jl[k] = pr[3 + (second ? 6 : 0) + k]; jl[3 + k] = second ? 0.0 : pr[6 + k]; }
    } else {
      reprojEval(p.cams[(idx >> 24) & 0xf], p.pose + (size_t)(idx & 0xfff) * 7, hpw, p.ext + (size_t)((idx >> 12) & 0xfff) * 7,
                 p.obsUv[2 * (size_t)o], p.obsUv[2 * (size_t)o + 1], fabs(p.obsW[o]), rr, jp, jl, je);   // (Map::getLhs does not ask whether the block is constant)
    }
    a00 += jl[0] * jl[0] + jl[3] * jl[3]; a01 += jl[0] * jl[1] + jl[3] * jl[4]; a02 += jl[0] * jl[2] + jl[3] * jl[5];
    a11 += jl[1] * jl[1] + jl[4] * jl[4]; a12 += jl[1] * jl[2] + jl[4] * jl[5]; a22 += jl[2] * jl[2] + jl[5] * jl[5];
  }
  a00 = rowSum16(a00); a01 = rowSum16(a01); a02 = rowSum16(a02); a11 = rowSum16(a11); a12 = rowSum16(a12); a22 = rowSum16(a22);
  if (gl != 0) return;
  // cyclic Jacobi on the symmetric 3x3 (one lane per landmark: this part is serial).  A sweep ends the iteration when the
  // off-diagonal mass is below 1e-16 of the trace -- the eigenvalues then move by less than off^2 / gap -- instead of
  // waiting for exact zeros (up to 12 sweeps of IEEE divisions and square roots were 2/3 of this kernel's 18 us); the
  // rotation uses reciprocal / reciprocal-square-root with Newton steps.
  auto rotate = [](double& app, double& aqq, double& apq, double& apr, double& aqr) {
    // annihilates apq; (p, q) diagonal entries, apr / aqr the third row's couplings
    const double th = (aqq - app) * 0.5 * rcpNewton(apq);
    const double tt = copysign(1.0, th) * rcpNewton(fabs(th) + sqrt(th * th + 1.0));
    const double c = rsqrtNewton(tt * tt + 1.0), sn = tt * c;
    const double npp = app - tt * apq, nqq = aqq + tt * apq;
    const double npr = c * apr - sn * aqr, nqr = sn * apr + c * aqr;
    app = npp; aqq = nqq; apq = 0.0; apr = npr; aqr = nqr;
  };
  const double tiny = 1.0e-16 * (fabs(a00) + fabs(a11) + fabs(a22));
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    if (off <= tiny) break;
    const double skip = 1.0e-20 * tiny;   // an entry this small is zero for every purpose (and 1 / it would overflow)
    if (fabs(a01) > skip) rotate(a00, a11, a01, a02, a12);
    if (fabs(a02) > skip) rotate(a00, a22, a02, a01, a12);
    if (fabs(a12) > skip) rotate(a11, a22, a12, a01, a02);
  }
  const double smallest = fmin(a00, fmin(a11, a22)), largest = fmax(a00, fmax(a11, a22));
  quality[l] = (smallest < 1.0e-12) ? 0.0 : sqrt(smallest) / sqrt(largest);
}

void launchLandmarkQuality(const DeviceProblem& p, double* quality, hipStream_t s) {
  if (p.L == 0) return;
  hipLaunchKernelGGL(k_landmark_quality, dim3((p.L + 15) / 16), dim3(256), 0, s, p, quality);
}

}  // namespace svin
