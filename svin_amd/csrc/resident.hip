// svin_amd: kernels of the device-resident window (see resident.hpp for the design).
//
// Sizes: a sliding window holds ~10^4 observations of ~10^3 landmarks and changes by ~10^3 records per frame, so every
// phase is a handful of strided passes for ONE workgroup of 1 024 threads -- a single launch with workgroup barriers between
// the phases instead of five dependent launches (each ~3 us of host enqueueing and a kernel boundary on the device).  The
// kernel is bound by the latency of a few dependent global accesses per phase, not by bandwidth.
#include "resident.hpp"

namespace svin {

namespace {

constexpr int kRebuildThreads = 1024;

// Values other threads of the workgroup produced with atomics in an earlier phase.  A workgroup shares its CU's vector L1,
// the atomics execute in L2 and the kernel has not read these lines before the barrier that follows them, so a plain load
// after that barrier sees them (and the compiler may issue a thread's loads back to back: an atomic load waits for each).
__device__ __forceinline__ int ldAtomic(const int* p) { return *p; }

// exclusive prefix sums of a pair of ints over the workgroup; wsum: 17 int2 of LDS.  Ends with a barrier.
__device__ int2 blockScanExclusive(int2 v, int2* wsum, int2& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int2 inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int ux = __shfl_up(inc.x, off), uy = __shfl_up(inc.y, off);
    if (lane >= off) { inc.x += ux; inc.y += uy; }
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const int2 t = lane < nw ? wsum[lane] : make_int2(0, 0);
    int2 ti = t;
    for (int off = 1; off < 64; off <<= 1) {
      const int ux = __shfl_up(ti.x, off), uy = __shfl_up(ti.y, off);
      if (lane >= off) { ti.x += ux; ti.y += uy; }
    }
    if (lane < nw) wsum[lane] = make_int2(ti.x - t.x, ti.y - t.y);
    if (lane == nw - 1) wsum[16] = ti;
  }
  __syncthreads();
  const int2 base = wsum[wave];
  total = wsum[16];
  return make_int2(base.x + inc.x - v.x, base.y + inc.y - v.y);
}

__global__ __launch_bounds__(kRebuildThreads) void k_window_rebuild(ResidentArgs a) {
  __shared__ int2 wsum[17];
  __shared__ int err;
  const int t = threadIdx.x, nt = blockDim.x;
  if (t == 0) err = 0;
  __syncthreads();
  // ---- phase 0: the delta
  for (int i = t; i < a.nSet; i += nt) {
    const WinLmSet s = a.sets[i];
    for (int k = 0; k < 4; ++k) a.lmHp[4 * (size_t)s.h + k] = s.hp[k];
    if (s.setQuality) a.qualH[s.h] = s.quality;
  }
  if (a.valuesOnly)
    for (int h = a.Hprev + t; h < a.H; h += nt) a.slotOfHNew[h] = -1;
  for (int i = t; i < a.nRem; i += nt) {
    const WinRem r = a.rems[i];
    const int s = a.slotOfHOld[r.lmH];
    int at = -1;
    if (s >= 0)   // no early exit: the loads of the whole segment go out together (a sequence number occurs once per landmark)
      for (int o = a.lmPtrOld[s], e = a.lmPtrOld[s + 1]; o < e; ++o) at = (a.seqOld[o] == r.seq) ? o : at;
    if (at >= 0 && a.live[at]) { a.live[at] = 0; atomicSub(&a.cnt[r.lmH], 1); }
    else atomicOr(&err, 1);
  }
  for (int i = t; i < a.nAdd; i += nt) {
    const int h = a.adds[i].lmH;
    if (h < 0) continue;
    atomicAdd(&a.cnt[h], 1);
    atomicAdd(&a.addsH[h], 1);
  }
  __syncthreads();
  // ---- phase 1: slots and segment starts of the landmarks that have observations, in handle order
  if (!a.valuesOnly) {
    const int per = (a.H + nt - 1) / nt;
    const int h0 = min(a.H, t * per), h1 = min(a.H, h0 + per);
    int2 mine = make_int2(0, 0);
    for (int h = h0; h < h1; ++h) {
      const int c = ldAtomic(&a.cnt[h]);
      if (c < 0) atomicOr(&err, 2);
      if (c > 0) { mine.x += 1; mine.y += c; }
    }
    int2 total;
    int2 at = blockScanExclusive(mine, wsum, total);
    for (int h = h0; h < h1; ++h) {
      const int c = ldAtomic(&a.cnt[h]);
      if (c > 0) {
        a.slotOfHNew[h] = at.x; a.handleOfSlotNew[at.x] = h; a.lmPtrNew[at.x] = at.y;
        at.x += 1; at.y += c;
      } else {
        a.slotOfHNew[h] = -1;
      }
    }
    if (t == 0) {
      a.lmPtrNew[total.x] = total.y;
      if (total.x != a.Lnew || total.y != a.Nnew) atomicOr(&err, 4);
    }
  }
  __syncthreads();
  if (err) {   // the counts disagree with the host graph: report, and leave an EMPTY BUT VALID CSR behind -- the host has
    // already switched to the new set and the solve is enqueued right behind this launch; it throws when it reads the status
    // (downloadStates), but until then the solve's kernels walk these arrays: every landmark without observations, every
    // observation record pointing at slot 0 with weight 0
    if (t == 0) *a.status = err;
    for (int s = t; s <= a.Lnew; s += nt) a.lmPtrNew[s] = 0;
    for (int o = t; o < a.Nnew; o += nt) {
      a.obsIdx[o] = packObs(0, 0, 0); a.obsLm[o] = 0; a.wNew[o] = 0.0;
      a.uvNew[2 * (size_t)o] = 0.0; a.uvNew[2 * (size_t)o + 1] = 0.0;
    }
    return;
  }
  // ---- phase 2: surviving observations first (order kept), this frame's additions behind them
  if (!a.valuesOnly)
  for (int o = t; o < a.Nold; o += nt) {   // one thread per old observation: its rank among the survivors of its landmark
    if (!a.live[o]) continue;
    const int s = a.obsLmOld[o];
    int rank = 0;
    for (int q = a.lmPtrOld[s]; q < o; ++q) rank += a.live[q];
    const int at = a.lmPtrNew[a.slotOfHNew[a.handleOfSlotOld[s]]] + rank;
    a.uvNew[2 * (size_t)at] = a.uvOld[2 * (size_t)o]; a.uvNew[2 * (size_t)at + 1] = a.uvOld[2 * (size_t)o + 1];
    a.wNew[at] = a.wOld[o]; a.hndNew[at] = a.hndOld[o]; a.seqNew[at] = a.seqOld[o];
  }
  for (int i = t; i < a.nAdd; i += nt) {
    const WinAdd ad = a.adds[i];
    if (ad.lmH < 0) continue;
    const int r = atomicAdd(&a.addCur[ad.lmH], 1);   // any order here; phase 3 sorts a landmark's additions by sequence number
    const int at = a.lmPtrNew[a.slotOfHNew[ad.lmH]] + ldAtomic(&a.cnt[ad.lmH]) - ldAtomic(&a.addsH[ad.lmH]) + r;
    a.uvNew[2 * (size_t)at] = ad.u; a.uvNew[2 * (size_t)at + 1] = ad.v;
    a.wNew[at] = ad.w; a.hndNew[at] = ad.hnd; a.seqNew[at] = ad.seq;
  }
  __syncthreads();
  // ---- phase 3: per landmark: additions into insertion order, packed slot indices, the landmark point by slot
  for (int s = t; s < a.Lnew; s += nt) {
    const int h = a.handleOfSlotNew[s];
    const int beg = a.lmPtrNew[s], end = a.lmPtrNew[s + 1];
    const int nNew = ldAtomic(&a.addsH[h]);
    for (int i = end - nNew + 1; i < end; ++i) {   // insertion sort of the tail (a frame adds one or two observations per landmark)
      const double u = a.uvNew[2 * (size_t)i], v = a.uvNew[2 * (size_t)i + 1], w = a.wNew[i];
      const uint32_t hn = a.hndNew[i], sq = a.seqNew[i];
      int j = i - 1;
      while (j >= end - nNew && (int32_t)(a.seqNew[j] - sq) > 0) {   // wrap-safe "j was inserted after i"
        a.uvNew[2 * (size_t)(j + 1)] = a.uvNew[2 * (size_t)j]; a.uvNew[2 * (size_t)(j + 1) + 1] = a.uvNew[2 * (size_t)j + 1];
        a.wNew[j + 1] = a.wNew[j]; a.hndNew[j + 1] = a.hndNew[j]; a.seqNew[j + 1] = a.seqNew[j];
        --j;
      }
      a.uvNew[2 * (size_t)(j + 1)] = u; a.uvNew[2 * (size_t)(j + 1) + 1] = v;
      a.wNew[j + 1] = w; a.hndNew[j + 1] = hn; a.seqNew[j + 1] = sq;
    }
    for (int o = beg; o < end; ++o) {
      const uint32_t hn = a.hndNew[o];
      const int ps = a.poseSlotOfH[hn & 0xfff], es = a.extSlotOfH[(hn >> 12) & 0xfff];
      if (ps < 0 || es < 0) atomicOr(&err, 16);   // (reported at the end; the record stays addressable: slot 0)
      a.obsIdx[o] = packObs(ps < 0 ? 0 : ps, es < 0 ? 0 : es, (int)(hn >> 24));
      a.obsLm[o] = s;
    }
    for (int k = 0; k < 4; ++k) a.lm[4 * (size_t)s + k] = a.lmHp[4 * (size_t)h + k];
    a.addsH[h] = 0; a.addCur[h] = 0;
  }
  for (int i = t; i < a.Nnew; i += nt) a.live[i] = 1;
  __syncthreads();
  // (phase 4 -- the per-chunk pose order of the dense Schur kernels with the A part on MFMA -- is k_window_order, one workgroup
  // per chunk of 16 landmarks, launched right behind this kernel: as the tail of this ONE workgroup it was 140 of the 180 us
  // a stereo_rig_v2 frame's rebuild took)
  __syncthreads();
  if (t == 0 && err) *a.status = err;
}

// Within every chunk of 16 landmarks the observations pose by pose (stable counting sort): what the dense Schur kernels with the
// A part on MFMA walk (Window::pack: orderObs).  One wave per chunk; the chunk's pose slots are staged in LDS once (every lane
// used to read all of them from global memory, twice).
__global__ __launch_bounds__(64) void k_window_order(ResidentArgs a) {
  constexpr int kStage = 1024;
  __shared__ int sPose[kStage];
  const int c = blockIdx.x, lane = threadIdx.x;
  const int oBeg = a.lmPtrNew[16 * c], oEnd = a.lmPtrNew[min(a.Lnew, 16 * c + 16)], cnt = oEnd - oBeg;
  const bool staged = cnt <= kStage;
  if (staged)
    for (int i = lane; i < cnt; i += 64) sPose[i] = (int)(a.obsIdx[oBeg + i] & 0xfff);
  __syncthreads();
  int carry = oBeg;
  for (int pb = 0; pb < a.nPoseSlots + 1; pb += 64) {
    const int p = pb + lane;
    int n = 0;
    if (staged) for (int i = 0; i < cnt; ++i) n += (sPose[i] == p) ? 1 : 0;
    else for (int o = oBeg; o < oEnd; ++o) n += ((int)(a.obsIdx[o] & 0xfff) == p) ? 1 : 0;
    int inc = n;
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(inc, off);
      if (lane >= off) inc += u;
    }
    int at = carry + inc - n;
    if (n > 0) {
      if (staged) { for (int i = 0; i < cnt; ++i) if (sPose[i] == p) a.obsOrder[at++] = oBeg + i; }
      else { for (int o = oBeg; o < oEnd; ++o) if ((int)(a.obsIdx[o] & 0xfff) == p) a.obsOrder[at++] = o; }
    }
    carry += __shfl(inc, 63);
  }
}

__global__ __launch_bounds__(256) void k_window_store_landmarks(int H, const int* slotOfH, const double* lm, const double* quality,
                                                                double* lmHp, double* qualH) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const int s = slotOfH[h];
  if (s < 0) { if (quality) qualH[h] = 0.0; return; }
  for (int k = 0; k < 4; ++k) lmHp[4 * (size_t)h + k] = lm[4 * (size_t)s + k];
  if (quality) qualH[h] = quality[s];
}

constexpr int kFinishBlocksPerSegment = 4;
__global__ __launch_bounds__(256) void k_window_finish(FinishArgs a) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < a.nLmBlocks) {
    const int h = b * blockDim.x + t;
    if (h < a.H) {
      const int s = a.slotOfH[h];
      if (s >= 0)
        for (int k = 0; k < 4; ++k) a.lmHp[4 * (size_t)h + k] = a.lm[4 * (size_t)s + k];
    }
  } else {
    const int g = b - a.nLmBlocks, seg = g / kFinishBlocksPerSegment, part = g % kFinishBlocksPerSegment;
    if (seg < a.ga.n) {
      const uint4* src = reinterpret_cast<const uint4*>(a.ga.src[seg]);
      uint4* dst = reinterpret_cast<uint4*>(a.hostBlock + a.ga.off[seg]);
      const size_t n16 = a.ga.bytes[seg] / 16;
      for (size_t i = (size_t)part * blockDim.x + t; i < n16; i += (size_t)kFinishBlocksPerSegment * blockDim.x) dst[i] = src[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) {
    const unsigned int done = atomicAdd(a.ticket, 1u);
    if (done == gridDim.x - 1) {   // every workgroup's stores are out (each fenced before it took its ticket)
      *a.ticket = 0u;
      __threadfence_system();
      *reinterpret_cast<volatile unsigned long long*>(a.hostSeq) = a.seq;
    }
  }
}

// The landmark part of the marginalisation policy (Estimator.cpp:671-766) over the resident CSR: which reprojection residuals
// are linearised into the prior, as the job tables marg.hip's M1 kernels read.  Landmarks in CSR order, the residuals of a
// landmark in insertion order -- the order the host policy walks them in.
__global__ __launch_bounds__(kRebuildThreads) void k_window_marg_gather(MargGatherArgs a) {
  __shared__ int2 wsum[17];
  const int t = threadIdx.x, nt = blockDim.x;
  const int per = (a.L + nt - 1) / nt;
  const int s0 = min(a.L, t * per), s1 = min(a.L, s0 + per);
  int2 mine = make_int2(0, 0);
  for (int s = s0; s < s1; ++s) {
    bool skip = true, hasNew = false, marg = true;
    int obsCount = 0;
    const int beg = a.lmPtr[s], end = a.lmPtr[s + 1];
    for (int o = beg; o < end; ++o) {
      const int cls = a.poseClass[a.hnd[o] & 0xfff];
      if (cls & kMargRemove) skip = false;
      if (cls & kMargNew) { marg = false; hasNew = true; }
      if (cls & kMargLin) ++obsCount;
    }
    int nJob = 0;
    if (!skip)
      for (int o = beg; o < end; ++o)
        if (margObsAction(a.poseClass[a.hnd[o] & 0xfff], hasNew, marg, obsCount) == 2) ++nJob;
    a.scratch[2 * s] = nJob;
    a.scratch[2 * s + 1] = (hasNew ? 1 : 0) | (marg ? 2 : 0) | (obsCount << 2);
    if (nJob > 0) { mine.x += 1; mine.y += nJob; }
  }
  int2 total;
  int2 at = blockScanExclusive(mine, wsum, total);
  for (int s = s0; s < s1; ++s) {
    const int nJob = a.scratch[2 * s];
    if (nJob == 0) continue;
    const int code = a.scratch[2 * s + 1];
    const bool hasNew = code & 1, marg = code & 2;
    const int obsCount = code >> 2;
    const int h = a.handleOfSlot[s];
    a.jLmPtr[at.x] = at.y;
    for (int k = 0; k < 4; ++k) a.jLm[4 * (size_t)at.x + k] = a.lmHp[4 * (size_t)h + k];
    for (int o = a.lmPtr[s], e = a.lmPtr[s + 1]; o < e; ++o) {
      const uint32_t hn = a.hnd[o];
      if (margObsAction(a.poseClass[hn & 0xfff], hasNew, marg, obsCount) != 2) continue;
      const int ps = a.jobPoseSlot[hn & 0xfff];
      const int es = a.jobExtSlot[(hn >> 12) & 0xfff];
      if (ps < 0) *a.status = 32;
      a.jUv[2 * (size_t)at.y] = a.uv[2 * (size_t)o]; a.jUv[2 * (size_t)at.y + 1] = a.uv[2 * (size_t)o + 1];
      a.jW[at.y] = a.w[o];
      a.jIdx[at.y] = packObs(max(ps, 0), max(es, 0), (int)(hn >> 24));
      a.jObsLm[at.y] = at.x;
      ++at.y;
    }
    ++at.x;
  }
  if (t == 0) {
    a.jLmPtr[total.x] = total.y;
    if (total.x != a.expectLm || total.y != a.expectN) *a.status = 64;
  }
}

__global__ __launch_bounds__(256) void k_fill_jobs(FillJobs f) {
  const FillJob j = f.job[blockIdx.y];
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(j.dst);
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(j.src);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < j.words; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long r = i / j.rowWords, c = i - r * j.rowWords;
    dst[r * j.dstPitch + c] = src ? src[r * j.srcPitch + c] : 0ull;
  }
}

}  // namespace

void launchFillJobs(const FillJobs& f, hipStream_t s) {
  if (f.n <= 0) return;
  hipLaunchKernelGGL(k_fill_jobs, dim3(64, f.n), dim3(256), 0, s, f);
}
void launchWindowRebuild(const ResidentArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_window_rebuild, dim3(1), dim3(kRebuildThreads), 0, s, a);
  if (a.wantOrder && a.Lnew > 0) hipLaunchKernelGGL(k_window_order, dim3((a.Lnew + 15) / 16), dim3(64), 0, s, a);
}
void launchWindowStoreLandmarks(int H, const int* slotOfH, const double* lm, const double* quality, double* lmHp, double* qualH,
                                hipStream_t s) {
  if (H <= 0) return;
  hipLaunchKernelGGL(k_window_store_landmarks, dim3((H + 255) / 256), dim3(256), 0, s, H, slotOfH, lm, quality, lmHp, qualH);
}
void launchWindowFinish(const FinishArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_window_finish, dim3(a.nLmBlocks + kFinishBlocksPerSegment * a.ga.n), dim3(256), 0, s, a);
}
void launchWindowMargGather(const MargGatherArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_window_marg_gather, dim3(1), dim3(kRebuildThreads), 0, s, a);
}

}  // namespace svin
