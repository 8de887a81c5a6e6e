// svin_amd: device data layout and kernel launch interface (host side sees plain structs).
//
// HBM layout (all FP64 unless noted; SoA so that lane i of a wave touches element i):
//   parameter tables   pose[nPose][7]  ext[nExt][7]  sb[nSb][9]  lm[L][4]   (+ candidate copies)
//   reduced-system map poseOff/extOff/sbOff: first row of the block in the reduced system or -1 (fixed)
//                      order: [poses | variable extrinsics | speed-biases]  -> the leading dC rows are
//                      the only ones reprojection factors touch
//   observations       landmark-major CSR: lmPtr[L+1]; per observation uv(2) w(1) packed index (u32)
//                      and landmark index (i32)
//   linearisation      r[2][N] Jp[12][N] Jl[6][N] Je[12][N]  (component-major: a wave writes/reads
//                      512 contiguous bytes per component) -- two sets (current / candidate)
//   small factors      DevFactor[F] + FactorLin[F] (r[15], J[15x30]) ; DevImu[nImu] pre-integration state
//   prior              H-space form of the marginalisation prior (Ht m x m, bp m, c0) + block table
//   normal equations   S[d][d], gRed[d], gFull[d], hC[d], per-landmark Vinv[6] bl[3] hL[3] scaleL[3]
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "dmath.hpp"
#include "options.hpp"

namespace svin {

constexpr int kMaxCams = 8;
constexpr int kDensePoseCap = 256;   // k_schur_dense stages the pose -> row map of at most this many poses in LDS
constexpr int kPriorCam = 15;   // camera field of a packed index that marks a landmark-prior pseudo-observation (HomogeneousPointError)

// packed per-observation index: pose slot (12 bit) | ext slot (12 bit) | camera (4 bit)
inline __host__ __device__ uint32_t packObs(int pose, int ext, int cam) {
  return (uint32_t)pose | ((uint32_t)ext << 12) | ((uint32_t)cam << 24);
}

struct ImuParams {
  double a_max, g_max, sigma_g_c, sigma_a_c, sigma_bg, sigma_ba, sigma_gw_c, sigma_aw_c, tau, g;
  double a0[3];
};

struct DevImu {
  int sampleStart, sampleCount;  // into the imu sample pool
  uint32_t t0[2], t1[2];
  ImuParams par;
  int redo, redoCounter;
  double Delta_t;
  double Delta_q[4], C_integral[9], C_doubleintegral[9], acc_integral[3], acc_doubleintegral[3];
  double dalpha_db_g[9], dv_db_g[9], dp_db_g[9];
  double P_delta[225], information[225], sqrtInfo[225];
  double sb_ref[9];
};

enum FactorKind : int { F_IMU = 0, F_POSE_PRIOR = 1, F_SB_PRIOR = 2, F_RELPOSE = 3, F_SONAR = 4, F_DEPTH = 5,
                        F_HOST = 6 };   // residual and minimal Jacobians computed by a callback of the host (Window::evaluateHostFactors) before every evaluation launch
enum BlockKind : int { B_POSE = 0, B_EXT = 1, B_SB = 2, B_LM = 3 };

struct DevFactor {
  int kind, nblk, m, imuIndex;
  int blkKind[4], blkSlot[4];
  double meas[9];       // pose prior: T(7); sb prior: 9; sonar: range, heading, mean(3); depth: depth, firstDepth
  double aux[8];        // sonar: T_SSo(7)
  double sqrtInfo[81];  // row-major m x m (upper-triangular L^T)
};
struct FactorLin {
  double r[15];
  double J[15 * 30];  // row-major m x ncols, blocks concatenated in order
  int off[4], dim[4], m, ncols;
};

struct PriorBlock {
  int kind, slot;  // BlockKind / table slot (B_LM unsupported in the solver: priors never keep landmarks)
  int ord, mdim;   // first row in the prior, minimal dimension (0 when the block was fixed)
  double lin[9];   // linearisation point
};

// scalar results of one iteration, read back by the host once per iteration
struct SolverScalars {
  // --- group A (8 summable scalars): produced by the step (retraction) and the candidate evaluation
  double cost;            // cost at the evaluated point (current or candidate)
  double costReproj, costFactors, costPrior;
  double stepNormSq, xNormSq;
  double spareA0, spareA1;
  // --- group B (8 summable scalars): produced once per linearisation by the post-solve pass.  With
  // v = g / htil (steepest descent) and y = Gauss-Newton solution, J*delta is linear in the dogleg
  // coefficients, so the five J sums below price ANY dogleg step of this linearisation.
  double gHatSq;          // |g_hat|^2
  double jgSq;            // |J v|^2                    (Cauchy point)
  double gnHatSq;         // |gn_hat|^2
  double gDotGn;          // g_hat . gn_hat
  double jySq;            // |J y|^2
  double jvDotJy;         // (J v).(J y)
  double jvDotR;          // (J v).r
  double jyDotR;          // (J y).r
  // --- sharded mode: every rank's (gradMax, failMax) pair in its own slot, the other slots zero, so that the SUM all-reduce
  // of [group B | gather] in ONE message gathers them and the host takes the max (a max all-reduce of its own before)
  double gather[16];
  // --- 2 max-reduced scalars (one GPU: written directly; sharded: the host fills them from `gather`)
  double gradMax;         // max |g_full|
  double failMax;         // (double)cholFail
  // --- derived on the device from the (all-reduced) sums and the trust-region radius
  double jdSq, jdDotR;    // |J delta|^2 , (J delta).r   (model cost change)
  double doglegStepNorm;
  int cholFail;           // bits 1 | 2: S or a landmark block is not positive definite (numerical: the mu ladder retries);
                          // bits 4 | 8 (kCholFailSync): a bounded device-side wait of a solver kernel gave up (a fault: solve() throws)
  int pad;
};
// pinned-host mailbox: the kernel that sums the cost publishes all scalars + the sequence number of that evaluation
struct ScalarMailbox {
  SolverScalars scal;
  unsigned long long seq;
};
constexpr int kCholFailSync = 4 | 8;
constexpr int kScalGroupA = 0, kScalGroupB = 8, kScalGather = 16, kScalGatherSlots = 16;  // offsets (doubles) of the all-reduced groups

// wide windows: 16-landmark chunks one workgroup of k_schur_panels works through (the host builds the work list: Window::pack).
// Config #4 on one GPU, 8 / 12 / 16: k_schur_panels 523 / 545 / 551 us, k_reduce_panel_slabs (one 74 KB slab per workgroup) 45 / 25 /
// 19 us -- a wash on one GPU, and a rank of an 8-GPU run has an eighth of the chunks: 8 keeps its ~210 workgroups from becoming ~105.
#ifndef SVIN_PANEL_CHUNKS
#define SVIN_PANEL_CHUNKS 8
#endif
constexpr int kPanelChunksPerBlock = SVIN_PANEL_CHUNKS;
// block-pair form (round 6): entries (landmark x panel pair) per workgroup of k_schur_rows, its waves (the host deals the block
// rows of a panel pair to them), records a batch stages in LDS (x 20 doubles = 160 bytes: two buffers of 36.8 KB, two workgroups
// per CU; the last record of a buffer is never staged: all zero, the B operand of the padding pairs), pair words per wave and batch
constexpr int kBlkMinWordsPerBlock = 1024;    // pair words per workgroup, at least (Window::pack cuts the work list by words)
constexpr int kBlkWaves = 8;
constexpr int kBlkBatchRecs = 230;
constexpr int kBlkBatchWords = 128;
constexpr int kBlkRec = 18;                   // doubles per slot record: E_la (6 x 3), rec[6 k + row]
#ifndef SVIN_SLOTS_PER_WG
#define SVIN_SLOTS_PER_WG 1024
#endif
constexpr int kBlkSlotsPerWorkgroup = SVIN_SLOTS_PER_WG;   // slots (one thread each, four trips) per workgroup of k_blocks_slots
constexpr int kBlkMaxPoseBlocks = 512;      // the per-pose accumulators of k_blocks_slots live in LDS (28 doubles per pose block)

struct DeviceProblem {
  // sizes
  int nPose, nExt, nSb, L, N, F, nImu, d, dC, priorM, priorBlocks, nCam;
  int anyExtVariable;
  int ownsCamera;   // landmark-sharded mode: only one rank takes the marginalisation prior and the camera-side norms (the small
                    // factors are dealt to the ranks frame by frame at pack() time: each rank's F counts its own)
  int rank, world;  // sharded mode (0, 1 otherwise)
  // tables
  double *pose, *ext, *sb, *lm;
  double *poseC, *extC, *sbC, *lmC;
  int *poseOff, *extOff, *sbOff;
  CameraModel* cams;
  // observations
  int* lmPtr;
  int schurDense;                            // narrow window: Schur complement as a Gram matrix on MFMA
  int schurPanels, nPanelBlocks, nPanelPairs;  // wide window: the same per 96-row panel pair (k_schur_panels)
  const int4* panelWork;                     // per workgroup: panel I, panel J, first chunk entry, chunk count
  const int* panelChunks;                    // chunk ids (16 landmarks each) of the work list
  const int* panelPairPtr;                   // per panel pair: first workgroup (nPanelPairs + 1 entries)
  // wide window, block-pair form (round 6: k_blocks_slots / k_schur_rows): a SLOT is a (landmark, distinct variable pose) pair
  int schurBlocks, nSlots;                   // 1: the panel work list is processed by k_schur_rows (0: the tile form k_schur_panels)
  const int* slotPtr;                        // per landmark: first slot (L + 1 entries); a landmark's slots ascend with the pose
  const unsigned short* slotBlk;             // per slot: pose block of the reduced camera system (poseOff / 6)
  const int* slotObsPtr;                     // per slot: its observations (nSlots + 1 entries into slotObs)
  const int* slotObs;                        // observation numbers
  const int* slotLm;                         // per slot: its landmark
  double* slotRec;                           // per slot kBlkRec doubles, written once per build: E = (sum Jp^T Jl) L^-T (kernels.hip)
  // work list of k_schur_rows (host-built, Window::pack): panelWork.z / .w = first batch / batches of the workgroup
  const int4* blkOwn;                        // per workgroup: the two block rows of wave w in bytes 2 w, 2 w + 1 (255: none)
  const int2* blkBatch;                      // per batch: first entry of blkRecSlot, records
  const int4* blkWaveTab;                    // per (batch, wave): first pair word, pair words of its first / second row (multiples of 8, together at most kBlkBatchWords)
  const int* blkRecSlot;                     // per staged record: its slot
  const uint32_t* blkPairs;                  // pair words: 2 x pose block in J | B record's byte offset << 8 | A record << 24; words 2 j and 2 j + 1 share their A record
  double* blkPartial;                        // per workgroup of k_blocks_slots: (dC / 6) x 28 sums of Jp^T Jp (21) and Jp^T r (6)
  double *obsUv, *obsW;
  uint32_t* obsIdx;
  int dCPose, aBlocks;   // rows of the variable poses in the reduced camera system; dense Schur: A accumulated block-wise in LDS
  const int* obsOrder;   // dense Schur with the A part on MFMA: the observations of every 16-landmark chunk sorted by pose (or null)
  int* obsLm;
  // linearisation buffers (cur = accepted point, cand = candidate)
  double *rCur, *JpCur, *JlCur, *JeCur;
  double *rCand, *JpCand, *JlCand, *JeCand;
  // factors
  DevFactor* factors;
  FactorLin *linCur, *linCand;
  DevImu* imus;
  uint32_t* imuT;    // [M][2]
  double* imuMeas;   // [M][6]
  // prior (H-space)
  double *priorH, *priorBp;
  const double* priorC0;                     // e0.e0 of the prior (device scalar)
  PriorBlock* priorBlk;
  double *priorDchi, *priorGrad, *priorM3;      // linearisation of the prior at the accepted point: m, m, 9 per block
  double *priorDchiC, *priorGradC, *priorM3C;   // ... at the candidate (swapped on acceptance)
  double *priorMv, *priorMy;                    // scratch m
  // normal equations
  double *S, *gRed, *gFull, *hC, *htilC, *scaleC;
  int sPadded;   // S has ((d + 15) / 16) * 16 rows of ldS doubles, zero beyond row / column d (the window's own buffer)
  int ldS;   // leading dimension of S.  The window sets it to d rounded up to 16 doubles (every 16-column tile row segment is then ONE
             // 128-byte cache line: the tile loads of the solvers touch half as many lines); 0 = d for a caller that only uses
             // launchSolveReduced on a principal block of its own matrix (pose graph root)
  double *Vinv, *bl, *hL, *scaleL;           // per landmark 6 / 3 / 3 / 3
  double *slabs; int nSlabs;                 // per-workgroup private copies of the leading dC x dC block (+2 dC vectors)
  double *yC, *yL;                           // Gauss-Newton solution (cam d, landmarks 3L)
  double *deltaC, *deltaL;                   // trust-region step
  double *vC, *vL;                           // generic vector for J*v passes
  double *cholL;                             // factor of S
  SolverScalars* scal;
  double* partial;                           // reduction scratch
  unsigned int* tickets;                     // last-block-done counters of the fused reductions
  ScalarMailbox* mailbox;                    // host-visible copy of the scalars (nullptr: disabled)
  unsigned long long mailboxSeq;             // sequence number to publish with this evaluation
  int lmDeferred;                            // fused step: the landmark part of the retraction is taken by the candidate evaluation
  int padDeferred;
  const double* lmPrior;                     // landmark priors: 12 doubles each (measurement xyz, upper-triangular sqrt information row-major)
  double* lmFactor;                          // wide windows: per landmark L^-1 of (V + mu D) (6) and c = L^-1 b (3), written by
                                             // k_panels_landmarks once per build, read by every panel pair of k_schur_panels
  int sbChain;                               // > 0: the rows dC .. d are sbChain variable speed / bias blocks (9 rows each) that
                                             // couple only with their neighbours in this order (IMU factors) and with the kept rows:
                                             // the blocked solver eliminates them as a chain first (k_sb_factor ...)
  // poses / extrinsics on a reduced manifold (Map::Pose3d / Pose4d / Pose2d, PoseManifold.cpp:173-466): the rows of the reduced
  // system that belong to their LOCKED tangent directions.  launchSolveReduced turns each into an identity row with a zero
  // right-hand side first (k_lock_rows) -- the same system as with those Jacobian columns removed, the step stays zero there
  const int* lockedRows;
  int nLocked;
  int nHostFactors;   // factors of kind F_HOST in `factors` (such a window is not batched: the host evaluates between the launches)
  // Set by Window::solve for the duration of a one-GPU solve: launches may fork onto the side stream of the solver's stream
  // (kernels.hip sideLaneOf).  Wide windows: the build also launches the small factors and the speed / bias chain's factorisation
  // and forward substitution (k_factors_only, k_sb_factor, k_sb_forward) there, beside the landmark elimination, whose results
  // they do not read, and launchSolveReduced skips them -- build and solve of an iteration must then see the same mu / initScale,
  // which is what the trust-region loop does anyway.
  int sideLane;
};

// ---- batched solve (svin_ba_solve_prepared_batch: B independent windows of equal launch geometry through ONE launch sequence per
// trust-region round, the window as blockIdx.y).  One slot per window, refilled by the host every round: the window's problem as
// it stands (buffer sets swapped by its accepted steps, mailbox sequence number of this round's evaluation) and the scalars its
// own trust region hands the kernels; `stages` says which launches of the round the window takes part in.
enum : int { kBatchFull = 1,    // build + reduced solve + post-solve pass with the fused dogleg step (a fresh linearisation)
             kBatchReuse = 2,   // k_step_retract only (a rejected step: smaller radius on the same Gauss-Newton / Cauchy pair)
             kBatchEval = 4 };  // the candidate (or initial) evaluation
struct BatchSlot {
  DeviceProblem p;
  double mu, radius;
  int initScale, stages;
};
void releaseSideLane(hipStream_t s);   // frees the side stream / events kernels.hip keeps for solver stream `s` (before `s` is destroyed)
bool batchSupported(const DeviceProblem& p);   // the geometry the batched kernels cover (otherwise the window is solved on its own)
// one round for the `n` windows of dSlots (device copy of the slot table); `geom` = any window of the batch (equal geometry),
// `stagesUnion` = OR of the slots' stages, `cand` as for launchEvalAll
void launchBatchRound(const BatchSlot* dSlots, const DeviceProblem& geom, int n, int stagesUnion, bool cand, hipStream_t s);
int schurDenseABlocks(const DeviceProblem& p);   // DeviceProblem::aBlocks as launchAccumulateNormalEquations chooses it

// ---- launch wrappers (kernels.hip).  `cand` selects candidate tables/buffers.
void launchEvalReproj(const DeviceProblem& p, bool cand, bool robust, hipStream_t s);
void launchEvalFactors(const DeviceProblem& p, bool cand, hipStream_t s, bool sumCost = false);
void launchEvalPrior(const DeviceProblem& p, bool cand, hipStream_t s, bool sumCost = false, int reprojBlocks = -1);
// fused evaluation (small factors + reprojection residuals in one launch, 256 observations per block)
bool canFuseEvaluation(const DeviceProblem& p);
void launchEvalAll(const DeviceProblem& p, bool cand, bool sumCost, hipStream_t s);
int costSummedBy(const DeviceProblem& p);  // 2 = prior evaluation, 1 = factor evaluation, 0 = separate launchCost
void launchBuildNormalEquations(const DeviceProblem& p, double mu, bool initScale, hipStream_t s);
// the same in two halves, so that a landmark-sharded solve can all-reduce [S | gRed | gFull | hC] in between
void launchAccumulateNormalEquations(const DeviceProblem& p, double mu, bool initScale, hipStream_t s, bool zeroFirst = true);
void launchZeroBuild(const DeviceProblem& p, hipStream_t s);
// Staged upload: the host arrays of a window arrive as ONE block (one DMA from pinned memory); the segment table at
// the head of the block tells this kernel where each array belongs.  Offsets and sizes are multiples of 16 bytes.
struct StageSegment { unsigned long long srcOff, bytes; void* dst; };
constexpr unsigned long long kStageClear = ~0ull;   // srcOff of a segment that clears `bytes` (a multiple of 16) at dst instead of copying
void launchScatterStaged(const void* block, int nSeg, hipStream_t s);
// the reverse for the read-back: up to 8 device arrays gathered into one block (offsets / sizes multiples of 16 bytes,
// sizes rounded up: the sources are over-allocated accordingly)
struct GatherArgs { const void* src[8]; unsigned long long off[8], bytes[8]; int n; };
void launchGatherStaged(void* block, const GatherArgs& a, hipStream_t s);
void launchFinalizeNormalEquations(const DeviceProblem& p, double mu, bool initScale, hipStream_t s);
// Cholesky + GN step of the reduced system; fuseFinalize applies k_finalize_diag (metric + damping) while loading S
void launchSolveReduced(const DeviceProblem& p, hipStream_t s, double mu = 0.0, bool initScale = false, bool fuseFinalize = false);
// grants a kernel `bytes` of dynamic LDS (hipFuncSetAttribute) once per (device, kernel) and only ever upwards: the attribute
// belongs to the function for the whole process, so the bookkeeping is process-wide and mutex-protected, not per handle
void ensureDynamicLds(const void* fn, size_t bytes);
// doubles DeviceProblem::cholL must hold for a reduced system of d unknowns: the LDS-resident solver's spill copy, or
// the blocked solver's (d64 + 64) x d64 matrix + 1/L_ii + factorised diagonal blocks + block-ready flags
// (withChain: room for the speed / bias chain elimination next to either solver -- the compact kept system, the chain's records,
// Y and t; a window's buffer is sized with it, the pose graph's root solve has no chain)
inline size_t solveReducedScratchDoubles(int d, bool withChain = false) {
  const size_t dpad = ((size_t)d + 15) / 16 * 16, d64 = ((size_t)d + 63) / 64 * 64, nb = d64 / 64;
  const size_t big = (d64 + 64) * d64 + d64 + d64 * 64 + ((nb + 3) * nb + 1) / 2 + 2;
  const size_t chain = withChain ? 2 * dpad * dpad + (dpad + 8) * (dpad + 32) + 64 * 264 + 64 : 0;
  return (dpad * dpad > big ? dpad * dpad : big) + chain;
}
// post-solve pass (back-substitution, J*v / J*y sums, norms); fuseRadius > 0: its last block also takes the dogleg
// step with that radius and retracts (single GPU, narrow windows) -- launchDoglegStep is then not needed
void launchDoglegPrepare(const DeviceProblem& p, hipStream_t s, double fuseRadius = -1.0);
void launchDoglegStep(const DeviceProblem& p, double radius, hipStream_t s);  // delta, J*delta pass, candidate, norms
// sharded mode: the (all-reduced) scalars + sequence number into the pinned-host mailbox, as one wave-wide store
void launchPublishScalars(const SolverScalars* scal, ScalarMailbox* mailbox, unsigned long long seq, hipStream_t s);
void launchSetStopVote(SolverScalars* scal, double vote, hipStream_t s);
// sharded mode: [lower triangle of S | gRed | gFull | hC] <-> one contiguous message in p.cholL (packedSystemDoubles long)
size_t packedSystemDoubles(const DeviceProblem& p);
void launchPackSystem(const DeviceProblem& p, bool unpack, hipStream_t s);
void launchCost(const DeviceProblem& p, hipStream_t s);                  // sums partial costs into scal->cost
void launchImuPropagation(const DevImu* im /*device*/, const uint32_t* T, const double* M, double* io, double* jac, double* cov,
                          int* used, hipStream_t s);
void launchLandmarkQuality(const DeviceProblem& p, double* quality, hipStream_t s);
// Jacobian-evaluation micro-benchmark entry: B independent copies of the observation set
void launchEvalReprojBatched(const DeviceProblem& p, int copies, double* rOut, double* JpOut, double* JlOut,
                             double* JeOut, hipStream_t s);

}  // namespace svin
