// svin_amd: the device-resident window.
//
// The reference mirrors every residual in four hash containers (okvis_ceres/src/Map.cpp:341-376, :467-492) and Ceres
// re-reads the whole graph at every solve.  Here the landmark-major observation CSR of kernels.hpp LIVES on the device
// from frame to frame: the host sends only what changed since the last solve -- the new observation records, the
// (landmark, sequence number) pairs of the removed ones, new / host-set landmark points and the small state tables --
// and ONE single-workgroup kernel rebuilds the CSR in place of Window::pack()'s host pass and full upload:
//
//   phase 0  tombstones (a removed observation is looked up in its landmark's segment of the old CSR), per-landmark
//            counts, landmark points set by the host
//   phase 1  exclusive scan over the landmark handles: CSR slot of every landmark that still has observations, lmPtr
//   phase 2  surviving observations keep their order inside their landmark; the new ones follow in insertion order
//   phase 3  packed (pose slot | extrinsics slot | camera) indices from the stable block handles through this frame's
//            slot tables, landmark points gathered by slot, per-chunk pose ordering for the dense Schur kernels
//
// Landmarks are addressed by a HANDLE (their creation number, renumbered in id order when the handle space has become
// sparse); the CSR lists the landmarks with at least one observation in handle order, the observations of a landmark in
// insertion order -- the order Window::pack() (the host path, kept for wide windows and as the reference the tests
// compare against) produces.  After a solve the landmark points and qualities stay on the device, keyed by handle;
// the host fetches them when somebody asks (Window::syncLandmarks).
//
// The marginalisation job's observation tables (okvis_ceres/src/Estimator.cpp:671-766 decides which residuals are
// linearised into the prior) are gathered from the same CSR by k_window_marg_gather: the host policy only touches the
// landmarks the leaving frames have seen and never copies an observation record.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace svin {

struct WinAdd {          // a new observation (Estimator::addObservation)
  int lmH;               // landmark handle; -1: withdrawn before it reached the device
  uint32_t seq;          // low 32 bits of the residual id: identity inside its landmark, ordering key of one frame's additions
  uint32_t hnd;          // packObs(pose handle, extrinsics handle, camera)
  uint32_t pad;
  double u, v, w;        // key point, sqrt information 8 / size
};
struct WinRem { int lmH; uint32_t seq; };                 // a removed observation
struct WinLmSet { int h, setQuality; double hp[4]; double quality; };   // new landmark / Estimator::setLandmark

struct ResidentArgs {
  int nAdd, nRem, nSet, H;        // delta sizes; handles in use: [0, H)
  int Nold, Lold, Nnew, Lnew;     // sizes of the old CSR and -- as the host graph counts them -- of the new one
  int wantOrder, nPoseSlots;      // per-chunk pose ordering (DeviceProblem::obsOrder) wanted; pose slots of the window
  int valuesOnly, Hprev;          // no observation was added or removed since the last flush: the CSR stays as it is ("old" and
                                  // "new" name the same set), only landmark points, slot-packed indices and the order are refreshed;
                                  // handles [Hprev, H) are new landmarks without observations
  const WinAdd* adds; const WinRem* rems; const WinLmSet* sets;
  // old CSR (read) / new CSR (written): the two sets take turns
  const int* lmPtrOld; const int* handleOfSlotOld; const int* slotOfHOld;
  const double* uvOld; const double* wOld; const uint32_t* hndOld; const uint32_t* seqOld; const int* obsLmOld;
  unsigned char* live;            // per old observation: 1 = alive (reset to 1 for the new CSR at the end)
  int* lmPtrNew; int* handleOfSlotNew; int* slotOfHNew;
  double* uvNew; double* wNew; uint32_t* hndNew; uint32_t* seqNew;
  // per landmark handle
  int* cnt; int* addsH; int* addCur;
  double* lmHp; double* qualH;
  // what the solver reads (DeviceProblem)
  uint32_t* obsIdx; int* obsLm; double* lm; int* obsOrder;
  const int* poseSlotOfH; const int* extSlotOfH;   // this frame's slot of every pose / extrinsics block handle
  int* status;                    // host-mapped: != 0 when the device's counts disagree with the host graph (a bug, reported)
};
void launchWindowRebuild(const ResidentArgs& a, hipStream_t s);

// after a solve: landmark points and qualities back into the per-handle tables (quality 0 for a landmark outside the CSR,
// Estimator.cpp:902-923 with Map::getLhs = 0); quality == nullptr: the points only (the qualities are computed when somebody
// asks for landmarks, Window::syncLandmarks)
void launchWindowStoreLandmarks(int H, const int* slotOfH, const double* lm, const double* quality, double* lmHp, double* qualH,
                                hipStream_t s);

// The same plus the read-back of the states in ONE launch: up to 8 device arrays (pose / extrinsics / speed-bias tables, IMU
// pre-integration states) are written straight into a host-mapped pinned block, the last workgroup to finish publishes a
// sequence number there -- the host polls it instead of paying for a gather launch, a DMA start and a stream synchronisation
// (35 us per optimize() of a sliding window; ~10 us this way).
struct FinishArgs {
  int H, nLmBlocks;
  const int* slotOfH; const double* lm; double* lmHp;
  GatherArgs ga;                      // sources, byte offsets into the host block, sizes (multiples of 16)
  unsigned char* hostBlock;           // device address of the mapped host block
  unsigned long long* hostSeq;        // device address of the mapped sequence number
  unsigned long long seq;
  unsigned int* ticket;               // device counter (zero between launches)
};
void launchWindowFinish(const FinishArgs& a, hipStream_t s);

// Marginalisation job tables out of the resident CSR.  Every pose handle carries the class bits the policy loop tests
// (Estimator.cpp:671-766): kMargRemove = the frame leaves the window, kMargLin = it is outside the IMU window ("linearised"),
// kMargNew = its id is not older than the current keyframe.
constexpr int kMargRemove = 1, kMargLin = 2, kMargNew = 4;
// what the policy loop does with one reprojection residual of a landmark some leaving frame has seen (Estimator.cpp:713-752),
// given the landmark's three summary values (first pass of the loop, :689-709): 0 = stays, 1 = removed, 2 = linearised into the
// prior.  Shared by the host policy (marg.hip) and the device gather, which therefore pick the same residuals.
inline __host__ __device__ int margObsAction(int cls, bool hasNewObservations, bool marginalize, int obsCount) {
  const bool inRemove = (cls & kMargRemove) != 0, inLin = (cls & kMargLin) != 0;
  if ((inRemove && hasNewObservations) || (!inLin && marginalize)) return 1;
  if (marginalize && inLin) return obsCount < 2 ? 1 : 2;
  return 0;
}
// Several clears / small device-to-device copies as ONE launch (a hipMemsetAsync or hipMemcpyAsync costs ~3 us of host time and a
// kernel boundary each; pack() needs four clears, the marginalisation job four clears and two copies).  Units are 8-byte words;
// a job is a matrix of rowWords-wide rows with its own pitch on either side (src == nullptr: clear).
struct FillJob { void* dst; const void* src; unsigned long long words, rowWords, dstPitch, srcPitch; };
struct FillJobs { FillJob job[8]; int n; };
inline void addFill(FillJobs& f, void* dst, const void* src, size_t words, size_t rowWords = 0, size_t dstPitch = 0, size_t srcPitch = 0) {
  if (words == 0) return;
  FillJob& j = f.job[f.n++];
  j.dst = dst; j.src = src; j.words = words;
  j.rowWords = rowWords ? rowWords : words; j.dstPitch = dstPitch ? dstPitch : j.rowWords; j.srcPitch = srcPitch ? srcPitch : j.rowWords;
}
void launchFillJobs(const FillJobs& f, hipStream_t s);

struct MargGatherArgs {
  int L, H;                          // CSR slots, handles
  int expectN, expectLm;             // what the host policy counted
  const int* lmPtr; const int* handleOfSlot;
  const double* uv; const double* w; const uint32_t* hnd;
  const double* lmHp;
  const unsigned char* poseClass;    // per pose handle
  const int* jobPoseSlot; const int* jobExtSlot;   // per handle: slot in the job's tables (-1: not part of the job)
  // job tables (marg.hip's sub-problem)
  int* jLmPtr; int* jObsLm; uint32_t* jIdx; double* jUv; double* jW; double* jLm;
  int* scratch;                      // 2 ints per CSR slot
  int* status;
};
void launchWindowMargGather(const MargGatherArgs& a, hipStream_t s);

}  // namespace svin
