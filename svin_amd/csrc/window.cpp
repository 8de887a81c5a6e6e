// svin_amd host core (see window.hpp).  Reference line numbers cite
// /root/reference/okvis_ros/okvis/okvis_ceres/src/Estimator.cpp unless another file is named.
#include "window.hpp"
#include "trust_region.hpp"
#include <array>
#include <atomic>
#include <cstdint>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <limits>
#include <map>
#include <mutex>
#include <stdexcept>

#include <rccl/rccl.h>   // declarations only; librccl is opened with dlopen when the sharded mode is switched on

namespace svin {

std::string& lastError() {
  static thread_local std::string e;
  return e;
}

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// compute units of the current device (asked once)
static int deviceComputeUnits() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}
static double nowSec() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double dtSecHost(TimeStamp a, TimeStamp b) {  // okvis Duration normalisation + toSec (Time.hpp:146)
  long long s = (long long)a.sec - (long long)b.sec;
  long long ns = (long long)a.nsec - (long long)b.nsec;
  while (ns < 0) { ns += 1000000000LL; s -= 1; }
  while (ns >= 1000000000LL) { ns -= 1000000000LL; s += 1; }
  return (double)s + 1e-9 * (double)ns;
}

// ------------------------------------------------------------------------------------------ debug / A-B options (options.hpp)
namespace {
struct OptionTable {
  std::atomic<int> v[kOptCount];
  const char* name[kOptCount];
  OptionTable() {
    static const struct { DebugOption which; const char* env; } kNames[] = {
        {kOptNoMailbox, "SVIN_NO_MAILBOX"}, {kOptHostPack, "SVIN_HOST_PACK"}, {kOptSchurPairwise, "SVIN_SCHUR_PAIRWISE"},
        {kOptForceDistributed, "SVIN_FORCE_DISTRIBUTED"}, {kOptNoEarlyImu, "SVIN_NO_EARLY_IMU"}, {kOptPackTiming, "SVIN_PACK_TIMING"},
        {kOptNoZeroCopyStates, "SVIN_NO_ZERO_COPY_STATES"}, {kOptSplitEval, "SVIN_SPLIT_EVAL"}, {kOptNoFuseStep, "SVIN_NO_FUSE_STEP"},
        {kOptNoDeferLm, "SVIN_NO_DEFER_LM"}, {kOptNoSpeculation, "SVIN_NO_SPECULATION"}, {kOptCholTiming, "SVIN_CHOL_TIMING"},
        {kOptPgTiming, "SVIN_PG_TIMING"}, {kOptMargTiming, "SVIN_MARG_TIMING"}, {kOptMargKeepPre, "SVIN_MARG_KEEP_PRE"},
        {kOptMargSyncEnqueue, "SVIN_MARG_SYNC_ENQUEUE"}, {kOptMargEig, "SVIN_MARG_EIG"}, {kOptSchurAMfma, "SVIN_SCHUR_A_MFMA"},
        {kOptPanelsOld, "SVIN_PANELS_OLD"}, {kOptNoLL, "SVIN_NO_LL"}, {kOptNoSbElim, "SVIN_NO_SB_ELIM"},
        {kOptNoLdsBorder, "SVIN_NO_LDS_BORDER"}, {kOptBlkRounds, "SVIN_BLK_ROUNDS"}, {kOptBatchLanes, "SVIN_BATCH_LANES"},
        {kOptBatchTiming, "SVIN_BATCH_TIMING"}, {kOptNoEvalSplit, "SVIN_NO_EVAL_SPLIT"},
        {kOptSlabChunks, "SVIN_SLAB_CHUNKS"}, {kOptNoSbEarly, "SVIN_NO_SB_EARLY"},
        {kOptNoRowSplit, "SVIN_NO_ROW_SPLIT"}};
    static_assert(sizeof(kNames) / sizeof(kNames[0]) == kOptCount, "every option has its environment variable");
    for (const auto& n : kNames) {
      name[n.which] = n.env;
      const char* e = std::getenv(n.env);   // the library's ONE look at the environment for its switches
      int val = e ? 1 : 0;
      if (e && (n.which == kOptBlkRounds || n.which == kOptBatchLanes || n.which == kOptSlabChunks)) val = std::atoi(e);
      if (e && n.which == kOptMargEig) {
        const std::string w(e);
        val = w == "direct" ? 1 : (w == "jacobi" ? 3 : 2);   // any other value selects the Cholesky-preconditioned Jacobi solve alone
      }
      v[n.which].store(val, std::memory_order_relaxed);
    }
  }
};
OptionTable& optionTable() {
  static OptionTable t;   // (thread-safe initialisation; svin_ba_create touches it before the first handle exists)
  return t;
}
}  // namespace
int debugOption(DebugOption which) { return optionTable().v[which].load(std::memory_order_relaxed); }
int debugOptionByName(const char* name, int* value) {
  if (!name) return 0;
  OptionTable& t = optionTable();
  for (int k = 0; k < kOptCount; ++k)
    if (std::strcmp(t.name[k], name) == 0) { if (value) *value = t.v[k].load(std::memory_order_relaxed); return 1; }
  return 0;
}
int setDebugOption(const char* name, int value) {
  if (!name) return 0;
  OptionTable& t = optionTable();
  for (int k = 0; k < kOptCount; ++k)
    if (std::strcmp(t.name[k], name) == 0) { t.v[k].store(value, std::memory_order_relaxed); return 1; }
  return 0;
}

// sqrtInformationUpper: dmath.hpp (shared with svin_host_pose_information)
static void normalisedPose(const double* T, double* out);

// ------------------------------------------------------------------------------------------ RCCL (resolved at run time)
// The library is looked up with dlopen when the landmark-sharded mode is switched on: a process that already holds an
// RCCL (torch.distributed's) gets that one (same SONAME), a single-GPU user never needs it to be installed.
namespace {
// Types and enums come from the installed header (declarations only: no link-time dependency, the symbols are resolved
// with dlsym below), so every call is made through the library's own prototype.
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
};
RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {   // an exception leaves the flag unset: the next call tries again from scratch
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) throw std::runtime_error(std::string("RCCL not found: ") + dlerror());
    auto sym = [&](const char* n) {
      void* f = dlsym(a.lib, n);
      if (!f) throw std::runtime_error(std::string("RCCL symbol missing: ") + n);
      return f;
    };
    a.getUniqueId = reinterpret_cast<decltype(a.getUniqueId)>(sym("ncclGetUniqueId"));
    a.commInitRank = reinterpret_cast<decltype(a.commInitRank)>(sym("ncclCommInitRank"));
    a.allReduce = reinterpret_cast<decltype(a.allReduce)>(sym("ncclAllReduce"));
    a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(sym("ncclCommDestroy"));
    a.getErrorString = reinterpret_cast<decltype(a.getErrorString)>(sym("ncclGetErrorString"));
    api = a;   // published only when every symbol has resolved
  });
  return api;
}
void rcclCheck(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + rccl().getErrorString(r));
}
}  // namespace

int Window::rcclUniqueId(unsigned char* out128) {
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  rcclCheck(rccl().getUniqueId(&id), "ncclGetUniqueId");
  static_assert(sizeof(id.internal) == 128, "svin_ba.h hands the id over as 128 bytes");
  std::memcpy(out128, id.internal, 128);
  return 1;
}
int Window::setDistributedRccl(int rank, int world, const unsigned char* id128) {
  quiesce();
  HIP_OK(hipSetDevice(device_));
  ncclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  dropRcclComm();
  ncclComm_t comm = nullptr;
  rcclCheck(rccl().commInitRank(&comm, world, id, rank), "ncclCommInitRank");
  rcclComm_ = comm;
  rank_ = rank; world_ = world; allreduce_ = nullptr; allreduceUser_ = nullptr;
  return 1;
}
void Window::dropRcclComm() {
  if (rcclComm_) { (void)rccl().commDestroy(static_cast<ncclComm_t>(rcclComm_)); rcclComm_ = nullptr; }
}
void Window::setDistributed(int rank, int world, AllReduceFn fn, void* user) {
  quiesce();
  dropRcclComm();   // the callback form replaces a native communicator (solve() prefers rcclComm_ when it is set)
  rank_ = rank; world_ = world; allreduce_ = fn; allreduceUser_ = user;
}

Window::Window(int device) : device_(device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw std::runtime_error("svin_ba: no HIP device available (this backend has no CPU fallback)");
  if (device < 0 || device >= count) throw std::runtime_error("svin_ba: invalid device index");
  HIP_OK(hipSetDevice(device));
  HIP_OK(hipStreamCreate(&stream_));
  HIP_OK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
  HIP_OK(hipEventCreateWithFlags(&evUploaded_, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&evImuReady_, hipEventDisableTiming));
  std::memset(&prob_, 0, sizeof(prob_));
  // zero-copy mailbox for the per-iteration scalars: the last evaluation kernel stores SolverScalars and a sequence
  // number straight into pinned host memory, the host polls it -- no copy kernel, no stream synchronisation on the
  // critical path of an iteration (SVIN_NO_MAILBOX=1 falls back to memcpy + synchronize)
  if (!optOn(kOptNoMailbox) &&
      hipHostMalloc(reinterpret_cast<void**>(&mailbox_), sizeof(ScalarMailbox), hipHostMallocMapped) == hipSuccess) {
    std::memset(mailbox_, 0, sizeof(ScalarMailbox));
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&mailboxDev_), mailbox_, 0) != hipSuccess) {
      (void)hipHostFree(mailbox_);
      mailbox_ = nullptr; mailboxDev_ = nullptr;
    }
  } else {
    mailbox_ = nullptr;
  }
}
Window::~Window() {
  try { quiesce(); } catch (...) {}
  {
    std::lock_guard<std::mutex> lock(enqueueMutex_);
    enqueueStop_ = true;
    enqueueCv_.notify_all();
  }
  if (enqueueThread_.joinable()) enqueueThread_.join();
  dropRcclComm();
  if (stageEvt_) (void)hipEventDestroy(stageEvt_);
  if (stageHost_) (void)hipHostFree(stageHost_);
  if (resStatus_) (void)hipHostFree(resStatus_);
  if (statesHost_) (void)hipHostFree(statesHost_);
  if (batchSlotsHost_) (void)hipHostFree(batchSlotsHost_);
  if (lmSyncHost_) (void)hipHostFree(lmSyncHost_);
  if (imuPropHost_) (void)hipHostFree(imuPropHost_);
  if (mailbox_) (void)hipHostFree(mailbox_);
  if (evUploaded_) (void)hipEventDestroy(evUploaded_);
  if (evImuReady_) (void)hipEventDestroy(evImuReady_);
  if (stream2_) (void)hipStreamDestroy(stream2_);
  if (stream_) { releaseSideLane(stream_); (void)hipStreamDestroy(stream_); }
}

// ------------------------------------------------------------------------------------------ sensors
int Window::addCamera(int model, const double* intr, const double* dist, int nDist, int w, int h, const double* sig) {
  quiesce();
  if ((int)cameras_.size() >= 15) return -1;
  CameraModel c;
  std::memset(&c, 0, sizeof(c));
  c.fu = intr[0]; c.fv = intr[1]; c.cu = intr[2]; c.cv = intr[3];
  for (int i = 0; i < 8; ++i) c.k[i] = (dist && i < nDist) ? dist[i] : 0.0;
  c.model = model; c.width = w; c.height = h;
  cameras_.push_back(c);
  ExtrinsicsSigmas e;
  e.abs_t = sig[0]; e.abs_r = sig[1]; e.rel_t = sig[2]; e.rel_r = sig[3];
  extrinsics_.push_back(e);
  return (int)cameras_.size() - 1;
}
int Window::setCameraGeometry(size_t cam, int model, const double* intr, const double* dist, int nDist, int w, int h) {
  quiesce();
  // the reference learns the camera geometry from the multi-frame of each observation (implementation/Estimator.hpp:62-66:
  // multiFramePtr->geometryAs<GEOMETRY_TYPE>(camIdx)); a shim that registers the extrinsics parameters first
  // (Estimator::addCamera) hands the geometry over with this call when the first multi-frame arrives
  if (cam >= cameras_.size()) return 0;
  CameraModel& c = cameras_[cam];
  c.fu = intr[0]; c.fv = intr[1]; c.cu = intr[2]; c.cv = intr[3];
  for (int i = 0; i < 8; ++i) c.k[i] = (dist && i < nDist) ? dist[i] : 0.0;
  c.model = model; c.width = w; c.height = h;
  return 1;
}
int Window::addImu(const ImuParams& p) {  // :83-90
  if (imus_.size() > 1) return -1;
  imus_.push_back(p);
  return (int)imus_.size() - 1;
}

// ------------------------------------------------------------------------------------------ graph helpers
Block* Window::addBlock(uint64_t id, int kind, const double* x) {  // Map::addParameterBlock refuses a known id (Map.cpp:257-260)
  if (idInUse(id)) return nullptr;
  Block b;
  b.id = id; b.kind = kind;
  std::memcpy(b.x, x, sizeof(double) * (kind == B_SB ? 9 : 7));
  if (!freeBlockH_[kind].empty()) { b.handle = freeBlockH_[kind].back(); freeBlockH_[kind].pop_back(); }
  else b.handle = nextBlockH_[kind]++;
  if (b.handle > 4095) throw std::runtime_error("window too wide for the packed index");
  Block* nb = &(blocks_[id] = b);
  if ((int)blockByHandle_[kind].size() <= nb->handle) blockByHandle_[kind].resize(nb->handle + 1, nullptr);
  blockByHandle_[kind][nb->handle] = nb;
  return nb;
}
Block* Window::findBlock(uint64_t id) {
  auto it = blocks_.find(id);
  return it == blocks_.end() ? nullptr : &it->second;
}
const Block* Window::findBlock(uint64_t id) const {
  auto it = blocks_.find(id);
  return it == blocks_.end() ? nullptr : &it->second;
}
// the observations of a frame arrive together: its pose block and the two extrinsics blocks answer from four remembered
// pointers instead of a hash lookup each (unordered_map nodes do not move; removeBlock clears the cache)
Block* Window::cachedBlock(uint64_t id) {
  for (Block* b : blockCache_)
    if (b && b->id == id) return b;
  Block* b = findBlock(id);
  if (b) { blockCache_[blockCacheNext_] = b; blockCacheNext_ = (blockCacheNext_ + 1) & 3; }
  return b;
}
static void eraseOne(std::vector<uint64_t>& v, uint64_t x) {
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == x) { v.erase(v.begin() + i); return; }
}
uint64_t Window::addFactor(Factor&& f) {
  f.id = nextResId_++;
  f.dealKey = factorSeq_++;
  for (int b = 0; b < f.nblk; ++b) blocks_.at(f.blocks[b]).residuals.push_back(f.id);
  const uint64_t id = f.id;
  factors_[id] = std::move(f);
  return id;
}
void Window::removeFactor(uint64_t id) {
  auto it = factors_.find(id);
  if (it == factors_.end()) return;
  for (int b = 0; b < it->second.nblk; ++b) {
    Block* blk = findBlock(it->second.blocks[b]);
    if (blk) eraseOne(blk->residuals, id);
  }
  factors_.erase(it);
}
// everything a removed observation takes with it except its slot in lm.obs (the caller erases it, or compacts the list once)
void Window::detachObsRecord(Landmark& lm, const Observation& o) {
  if (Block* b = blockByHandle_[B_POSE][o.poseH]) b->nObs--;
  if (Block* b = blockByHandle_[B_EXT][o.extH]) b->nObs--;
  obsRes2Lm_.erase(o.resId);
  if (residentValid_) {   // the device copy learns about it with the next flush (resident.hpp)
    if (o.pendEpoch == epoch_) addLog_[o.pendIdx].lmH = -1;   // never got there: withdrawn
    else remLog_.push_back(WinRem{lm.handle, (uint32_t)o.resId});
  }
  --numObs_;
}
void Window::afterObsRemoval(Landmark& lm) {
  if (lm.obs.empty()) { --numLmObserved_; emptyLm_.push_back(lm.handle); }
  lm.minPose = UINT64_MAX;
  for (const Observation& q : lm.obs) lm.minPose = std::min(lm.minPose, q.poseId);
}
void Window::removeObsRecord(Landmark& lm, size_t idx) {
  detachObsRecord(lm, lm.obs[idx]);
  lm.obs.erase(lm.obs.begin() + idx);
  afterObsRemoval(lm);
}
void Window::eraseLandmark(Landmark& lm) {
  while (!lm.obs.empty()) removeObsRecord(lm, lm.obs.size() - 1);
  const int h = lm.handle;
  if (lm.fixed) --numFixedLandmarks_;
  lmByHandle_[h] = nullptr;
  lmIndex_.erase(lm.id);
  landmarks_.erase(lmIterByHandle_[h]);
}
void Window::removeBlock(uint64_t id) {  // Map::removeParameterBlock cascades (Map.cpp:322-333)
  Block* b = findBlock(id);
  if (!b) return;
  const std::vector<uint64_t> res = b->residuals;
  for (uint64_t rid : res)
    if (factors_.count(rid)) removeFactor(rid);
  if (b->nObs > 0) {  // reprojection residuals are only listed per landmark: the (rare) cascade scans for them
    for (auto& kv : landmarks_) {
      Landmark& lm = kv.second;
      for (size_t i = 0; i < lm.obs.size();)
        if (lm.obs[i].poseId == id || (b->kind == B_EXT && lm.obs[i].extH == b->handle)) removeObsRecord(lm, i);
        else ++i;
    }
  }
  for (Block*& c : blockCache_) c = nullptr;
  obsCacheValid_ = false;
  blockByHandle_[b->kind][b->handle] = nullptr;
  freeBlockH_[b->kind].push_back(b->handle);
  blocks_.erase(id);
}

// ------------------------------------------------------------------------------------------ IMU prediction
int Window::imuPropagation(const uint32_t* imuT, const double* imuM, int n, const ImuParams& par, double* T, double* sb,
                           TimeStamp t0, TimeStamp t1, double* cov, double* jac, double* integrals) {
  quiesce();   // (the enqueue thread has handed its launches over; the device may still be running them)
  if (n <= 0) return -1;
  // One pinned block and its device twin, kept across calls (addStates calls this once per frame: five allocations, four small
  // copies in and four out cost ~0.1 ms), on the SIDE stream: nothing here depends on what the main stream still holds -- the
  // marginalisation job of the previous frame above all, which a synchronisation of the main stream would wait for.
  //   in : DevImu | io (T 7, sb 9, integrals 7, pad 1) | T (2 n uint32) | M (6 n doubles)
  //   out: io | jac 225 | cov 225 | used
  constexpr size_t kIo = 24, kOut = kIo + 450 + 2;
  const size_t offIm = 0, offIo = (sizeof(DevImu) + 15) / 16 * 16, offT = offIo + kOut * 8,
               offM = offT + ((size_t)2 * n * sizeof(uint32_t) + 15) / 16 * 16, total = offM + (size_t)6 * n * 8;
  if (total > imuPropCap_) {
    if (imuPropHost_) (void)hipHostFree(imuPropHost_);
    imuPropCap_ = total * 2;
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&imuPropHost_), imuPropCap_, hipHostMallocDefault));
    imuPropDev_.reserve(imuPropCap_);
  }
  DevImu im;
  std::memset(&im, 0, sizeof(im));
  im.sampleStart = 0; im.sampleCount = n;
  im.t0[0] = t0.sec; im.t0[1] = t0.nsec; im.t1[0] = t1.sec; im.t1[1] = t1.nsec;
  im.par = par;
  std::memcpy(imuPropHost_ + offIm, &im, sizeof(im));
  double* io = reinterpret_cast<double*>(imuPropHost_ + offIo);
  std::memset(io, 0, kOut * 8);
  std::memcpy(io, T, 7 * sizeof(double));
  std::memcpy(io + 7, sb, 9 * sizeof(double));
  std::memcpy(imuPropHost_ + offT, imuT, sizeof(uint32_t) * 2 * n);
  std::memcpy(imuPropHost_ + offM, imuM, sizeof(double) * 6 * n);
  unsigned char* dev = imuPropDev_.p;
  double* dIo = reinterpret_cast<double*>(dev + offIo);
  int* dUsed = reinterpret_cast<int*>(dIo + kIo + 450);
  HIP_OK(hipMemcpyAsync(dev, imuPropHost_, total, hipMemcpyHostToDevice, stream2_));
  launchImuPropagation(reinterpret_cast<const DevImu*>(dev + offIm), reinterpret_cast<const uint32_t*>(dev + offT),
                       reinterpret_cast<const double*>(dev + offM), dIo, jac ? dIo + kIo : nullptr, cov ? dIo + kIo + 225 : nullptr, dUsed,
                       stream2_);
  HIP_OK(hipMemcpyAsync(io, dIo, kOut * 8, hipMemcpyDeviceToHost, stream2_));
  HIP_OK(hipStreamSynchronize(stream2_));
  const int used = *reinterpret_cast<const int*>(io + kIo + 450);
  if (jac) std::memcpy(jac, io + kIo, 225 * sizeof(double));
  if (cov) std::memcpy(cov, io + kIo + 225, 225 * sizeof(double));
  if (used >= 0) {
    std::memcpy(T, io, 7 * sizeof(double));
    std::memcpy(sb, io + 7, 9 * sizeof(double));
    if (integrals) std::memcpy(integrals, io + 16, 7 * sizeof(double));
  }
  return used;
}

// initPoseFromImu (:848-873): gravity alignment of the very first pose.  A handful of scalar operations
// on the mean accelerometer reading, done once per session at construction time.
bool Window::initPoseFromImu(const double* imuM, int n, double* T) {
  T[0] = T[1] = T[2] = 0; T[3] = T[4] = T[5] = 0; T[6] = 1;
  if (n == 0) return false;
  double acc[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) { acc[0] += imuM[6 * i + 3]; acc[1] += imuM[6 * i + 4]; acc[2] += imuM[6 * i + 5]; }
  acc[0] /= (double)n; acc[1] /= (double)n; acc[2] /= (double)n;
  const double an = std::sqrt(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]);
  const double e[3] = {acc[0] / an, acc[1] / an, acc[2] / an};
  double c[3] = {0.0 * e[2] - 1.0 * e[1], 1.0 * e[0] - 0.0 * e[2], 0.0};
  const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  if (cn > 0) { c[0] /= cn; c[1] /= cn; c[2] /= cn; }
  const double angle = std::acos(e[2]);
  const double delta[6] = {0, 0, 0, -c[0] * angle, -c[1] * angle, -c[2] * angle};
  double To[7];
  poseOplus(T, delta, To);
  std::memcpy(T, To, sizeof(To));
  return true;
}

// ------------------------------------------------------------------------------------------ addStates (:98-411)
int Window::addStates(uint64_t frameId, TimeStamp stamp, uint64_t numKeypoints, const double* T_SC, int nCam,
                      const uint32_t* imuT, const double* imuM, int nImu, bool asKeyframe, const double* sonar,
                      int nSonar, const double* depth, int nDepth, double firstDepth) {
  obsCacheValid_ = false;
  if (nCam != (int)cameras_.size()) { lastError() = "addStates: T_SC count != number of cameras"; return -1; }
  if (imus_.empty()) { lastError() = "addStates: no IMU added"; return -1; }
  if (imuM == nullptr || imuT == nullptr) nImu = 0;
  double T_WS[7], sb[9], integrals[7];
  const bool first = states_.empty();
  bool insertedIntegrals = false;
  if (first) {
    if (!initPoseFromImu(imuM, nImu, T_WS)) return 0;  // :110-113
    if (!(numKeypoints > 10)) return 0;                 // :116-122
    for (double& v : sb) v = 0;
    for (int k = 0; k < 3; ++k) sb[6 + k] = imus_[0].a0[k];
  } else {
    const State& last = states_.rbegin()->second;
    std::memcpy(T_WS, blocks_.at(last.pose.id).x, sizeof(T_WS));
    std::memcpy(sb, blocks_.at(last.sb.at(0).id).x, sizeof(sb));
    const int used = imuPropagation(imuT, imuM, nImu, imus_[0], T_WS, sb, last.stamp, stamp, nullptr, nullptr, integrals);
    if (used < 1) return 0;  // :159-162
    insertedIntegrals = imuIntegrals_.count(frameId) == 0;
    setImuPreIntegral(frameId, integrals);  // :165
  }
  ++stateCount_;  // :171 (before the id check, like the reference)
  if (idInUse(frameId)) return 0;  // Map::addParameterBlock refuses the pose block (:186-193)
  // internal block ids: drawn up front so that a colliding provider leaves the graph untouched
  const State* prevState = first ? nullptr : &states_.rbegin()->second;
  std::vector<uint64_t> extIds(cameras_.size(), 0), sbIds(imus_.size(), 0);
  {
    std::vector<uint64_t> drawn{frameId};
    // a provider's id that is already taken is the host's error; the built-in counter simply skips what callers chose
    auto draw = [&]() -> uint64_t {
      for (int tries = idProvider_ ? 1 : (1 << 20); tries > 0; --tries) {
        const uint64_t id = newId();
        if (id == 0 || idInUse(id) || std::find(drawn.begin(), drawn.end(), id) != drawn.end()) continue;
        drawn.push_back(id);
        return id;
      }
      return 0;
    };
    bool ok = true;
    for (size_t i = 0; i < cameras_.size(); ++i) {
      if ((extrinsics_[i].rel_t < 1e-12 || extrinsics_[i].rel_r < 1e-12) && !first) extIds[i] = prevState->ext.at(i).id;
      else ok &= (extIds[i] = draw()) != 0;
    }
    for (size_t i = 0; i < imus_.size(); ++i) ok &= (sbIds[i] = draw()) != 0;
    if (!ok) {
      lastError() = "addStates: the id provider returned an id that is already in use (frames, landmarks and the "
                    "estimator's internal blocks share ONE id space, see svin_ba_set_id_provider / svin_ba_reserve_ids)";
      --stateCount_;                                              // the error path leaves the window as it was (svin_ba.h)
      if (insertedIntegrals) imuIntegrals_.erase(frameId);
      return -1;
    }
  }
  State st;
  st.id = frameId; st.stamp = stamp; st.isKeyframe = asKeyframe;
  st.pose.id = frameId; st.pose.exists = true;
  addBlock(frameId, B_POSE, T_WS);
  const State* prev = first ? nullptr : &states_.rbegin()->second;
  for (size_t i = 0; i < cameras_.size(); ++i) {  // :203-229
    StateInfo info;
    info.exists = true;
    info.id = extIds[i];
    if (!((extrinsics_[i].rel_t < 1e-12 || extrinsics_[i].rel_r < 1e-12) && !first)) addBlock(info.id, B_EXT, T_SC + 7 * i);
    st.ext.push_back(info);
  }
  for (size_t i = 0; i < imus_.size(); ++i) {  // :232-246
    StateInfo info;
    info.exists = true;
    info.id = sbIds[i];
    addBlock(info.id, B_SB, sb);
    st.sb.push_back(info);
  }
  if (nDepth > 0) {  // :248-262
    double mean_depth = 0.0;
    for (int i = 0; i < nDepth; ++i) mean_depth += depth[i];
    mean_depth = mean_depth / nDepth;
    Factor f;
    f.kind = F_DEPTH; f.nblk = 1; f.blocks[0] = frameId; f.m = 1;
    f.meas[0] = mean_depth; f.meas[1] = firstDepth;
    f.sqrtInfo[0] = std::sqrt(5.0);
    addFactor(std::move(f));
  }
  if (nSonar > 0) {  // :265-316
    const double range = sonar[2 * (nSonar - 1)], heading = sonar[2 * (nSonar - 1) + 1];
    // sonar point in the world frame: T_WS * T_SSo * [range cos h, range sin h, 0] -- the same composition the
    // device factor uses; evaluated here only to select the visual patch (a scan over cached landmark positions)
    const Quat qws = qnormalized(Quat{T_WS[3], T_WS[4], T_WS[5], T_WS[6]});
    const Mat3 Cws = quatToR(qws);
    const Quat qso = qnormalized(Quat{T_SSo_[3], T_SSo_[4], T_SSo_[5], T_SSo_[6]});
    const Vec3 rso = rotate(Cws, Vec3{T_SSo_[0], T_SSo_[1], T_SSo_[2]});
    const Mat3 Cwso = quatToR(qnormalized(qmul(qws, qso)));
    const Vec3 pp = rotate(Cwso, Vec3{range * std::cos(heading), range * std::sin(heading), 0.0});
    const double sl[3] = {pp.x + rso.x + T_WS[0], pp.y + rso.y + T_WS[1], pp.z + rso.z + T_WS[2]};
    double mean[3] = {0, 0, 0};
    size_t cnt = 0;
    double vl[3] = {0, 0, 0};
    syncLandmarks();
    for (auto rit = landmarks_.rbegin(); rit != landmarks_.rend(); ++rit) {
      const double* pt = rit->second.hp;
      if (std::fabs(pt[3]) > 1.0e-8) { vl[0] = pt[0] / pt[3]; vl[1] = pt[1] / pt[3]; vl[2] = pt[2] / pt[3]; }
      if (std::fabs(sl[0] - vl[0]) < 0.1 && std::fabs(sl[1] - vl[1]) < 0.1 && std::fabs(sl[2] - vl[2]) < 0.1) {
        mean[0] += vl[0]; mean[1] += vl[1]; mean[2] += vl[2];
        ++cnt;
      }
    }
    if (cnt > 0) {
      Factor f;
      f.kind = F_SONAR; f.nblk = 1; f.blocks[0] = frameId; f.m = 1;
      f.meas[0] = range; f.meas[1] = heading;
      f.meas[2] = mean[0] / cnt; f.meas[3] = mean[1] / cnt; f.meas[4] = mean[2] / cnt;
      std::memcpy(f.aux, T_SSo_, sizeof(T_SSo_));
      f.sqrtInfo[0] = std::sqrt(1.0);
      addFactor(std::move(f));
    }
  }
  if (first) {
    {  // pose prior (:319-327)
      double information[36] = {0};
      information[35] = 1.0e8; information[0] = 1.0e8; information[7] = 1.0e8; information[14] = 1.0e8;
      Factor f;
      f.kind = F_POSE_PRIOR; f.nblk = 1; f.blocks[0] = frameId; f.m = 6;
      std::memcpy(f.meas, T_WS, 7 * sizeof(double));
      sqrtInformationUpper(information, 6, f.sqrtInfo);
      addFactor(std::move(f));
    }
    for (size_t i = 0; i < cameras_.size(); ++i) {  // :330-350
      const double tv = extrinsics_[i].abs_t * extrinsics_[i].abs_t, rv = extrinsics_[i].abs_r * extrinsics_[i].abs_r;
      if (tv > 1.0e-16 && rv > 1.0e-16) {
        double information[36] = {0};
        for (int k = 0; k < 3; ++k) { information[k * 7] = 1.0 * 1.0 / tv; information[(k + 3) * 7] = 1.0 * 1.0 / rv; }
        Factor f;
        f.kind = F_POSE_PRIOR; f.nblk = 1; f.blocks[0] = st.ext[i].id; f.m = 6;
        std::memcpy(f.meas, T_SC + 7 * i, 7 * sizeof(double));
        sqrtInformationUpper(information, 6, f.sqrtInfo);
        addFactor(std::move(f));
      } else {
        blocks_.at(st.ext[i].id).fixed = true;
      }
    }
    for (size_t i = 0; i < imus_.size(); ++i) {  // :351-364
      const double sbg = imus_[0].sigma_bg, sba = imus_[0].sigma_ba;
      double information[81] = {0};
      for (int k = 0; k < 3; ++k) {
        information[k * 10] = 1.0 * 1.0 / 1.0;
        information[(k + 3) * 10] = 1.0 * 1.0 / (sbg * sbg);
        information[(k + 6) * 10] = 1.0 * 1.0 / (sba * sba);
      }
      Factor f;
      f.kind = F_SB_PRIOR; f.nblk = 1; f.blocks[0] = st.sb[i].id; f.m = 9;
      std::memcpy(f.meas, sb, sizeof(sb));
      sqrtInformationUpper(information, 9, f.sqrtInfo);
      addFactor(std::move(f));
    }
  } else {
    for (size_t i = 0; i < imus_.size(); ++i) {  // :368-382
      Factor f;
      f.kind = F_IMU; f.nblk = 4; f.m = 15;
      f.blocks[0] = prev->id; f.blocks[1] = prev->sb.at(i).id; f.blocks[2] = st.id; f.blocks[3] = st.sb[i].id;
      f.imuT.assign(imuT, imuT + 2 * (size_t)nImu);
      f.imuMeas.assign(imuM, imuM + 6 * (size_t)nImu);
      std::memset(&f.imu, 0, sizeof(f.imu));
      f.imu.sampleCount = nImu;
      f.imu.t0[0] = prev->stamp.sec; f.imu.t0[1] = prev->stamp.nsec;
      f.imu.t1[0] = stamp.sec; f.imu.t1[1] = stamp.nsec;
      f.imu.par = imus_[i];
      f.imu.redo = 1;
      f.imu.Delta_q[3] = 1.0;
      addFactor(std::move(f));
    }
    for (size_t i = 0; i < cameras_.size(); ++i) {  // :385-404
      if (prev->ext.at(i).id != st.ext[i].id) {
        const double dt = dtSecHost(stamp, prev->stamp);
        const double tv = extrinsics_[i].rel_t * extrinsics_[i].rel_t * dt;
        const double rv = extrinsics_[i].rel_r * extrinsics_[i].rel_r * dt;
        double information[36] = {0};
        for (int k = 0; k < 3; ++k) { information[k * 7] = 1.0 * 1.0 / tv; information[(k + 3) * 7] = 1.0 * 1.0 / rv; }
        Factor f;
        f.kind = F_RELPOSE; f.nblk = 2; f.blocks[0] = prev->ext.at(i).id; f.blocks[1] = st.ext[i].id; f.m = 6;
        sqrtInformationUpper(information, 6, f.sqrtInfo);
        addFactor(std::move(f));
      }
    }
  }
  states_[frameId] = st;
  return 1;
}

int Window::addLandmark(uint64_t id, const double* hp) {  // :414-429
  if (idInUse(id)) return 0;
  Landmark lm;
  lm.id = id;
  std::memcpy(lm.hp, hp, sizeof(lm.hp));
  lm.quality = 0.0;
  double dist = std::numeric_limits<double>::max();
  if (std::fabs(hp[3]) > 1.0e-8) {
    const double e[3] = {hp[0] / hp[3], hp[1] / hp[3], hp[2] / hp[3]};
    dist = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  }
  lm.distance = dist;
  lm.handle = nextLmHandle_++;
  auto ins = landmarks_.emplace(id, lm).first;
  Landmark* node = &ins->second;
  node->obs.reserve(12);   // (a landmark collects about ten observations: no reallocation on the way there)
  lmByHandle_.push_back(node);
  lmIterByHandle_.push_back(ins);
  lmIndex_.set(id, (uint64_t)lm.handle);
  emptyLm_.push_back(lm.handle);
  if (residentValid_) {
    WinLmSet st;
    st.h = lm.handle; st.setQuality = 1; st.quality = 0.0;
    std::memcpy(st.hp, hp, sizeof(st.hp));
    setLog_.push_back(st);
  }
  return 1;
}

uint64_t Window::addObservation(uint64_t lmId, uint64_t poseId, uint64_t cam, uint64_t kp, const double* uv,
                                double size) {  // implementation/Estimator.hpp:47-87
  uint64_t hnd = 0;
  if (!lmIndex_.find(lmId, &hnd)) return 0;
  return addObservationTo(*lmByHandle_[(size_t)hnd], poseId, cam, kp, uv, size);
}
int Window::addObservations(int n, const uint64_t* lmIds, const uint64_t* pose, const uint64_t* cam, const uint64_t* kp,
                            const double* uv, const double* size, uint64_t* outIds) {
  // Two passes: the landmark records of a frame's matches are scattered over the heap (std::map nodes, one vector each);
  // resolving the ids first and prefetching node and observation list a few entries ahead hides most of those misses.
  static thread_local std::vector<Landmark*> nodes;
  nodes.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    uint64_t hnd = 0;
    nodes[i] = lmIndex_.find(lmIds[i], &hnd) ? lmByHandle_[(size_t)hnd] : nullptr;
    if (nodes[i]) __builtin_prefetch(nodes[i]);
  }
  constexpr int kAhead = 8;
  int added = 0;
  for (int i = 0; i < n; ++i) {
    if (i + kAhead < n && nodes[i + kAhead]) {
      const Landmark& nx = *nodes[i + kAhead];
      const Observation* d = nx.obs.data();
      if (d) { __builtin_prefetch(d); __builtin_prefetch(d + nx.obs.size(), 1); }
    }
    const uint64_t id = nodes[i] ? addObservationTo(*nodes[i], pose[i], cam[i], kp[i], uv + 2 * (size_t)i, size[i]) : 0;
    if (outIds) outIds[i] = id;
    added += id != 0;
  }
  return added;
}
uint64_t Window::addObservationTo(Landmark& lm, uint64_t poseId, uint64_t cam, uint64_t kp, const double* uv, double size) {
  if (cam >= cameras_.size()) return 0;
  Block* pb = cachedBlock(poseId);
  if (!pb || pb->kind != B_POSE) return 0;
  // the extrinsics blocks of the frame: the state table is consulted once per frame, then obsCacheExt_ answers
  if (!obsCacheValid_ || obsCachePose_ != poseId) {
    auto sit = states_.find(poseId);
    if (sit == states_.end()) return 0;
    for (size_t c = 0; c < cameras_.size() && c < 16; ++c) obsCacheExt_[c] = sit->second.ext.at(c).id;
    obsCachePose_ = poseId;
    obsCacheValid_ = true;
  }
  if (poseId <= lm.maxPose)   // (the first observation from a new frame cannot repeat an older one)
    for (const Observation& o : lm.obs)
      if (o.poseId == poseId && (uint64_t)o.cam == cam && o.kp == kp) return 0;  // duplicate -> NULL
  Block* eb = cachedBlock(obsCacheExt_[cam]);
  if (!eb) return 0;
  return addObservationRecord(lm, pb, eb, cam, kp, uv, size);
}
uint64_t Window::addObservationRecord(Landmark& lm, Block* pb, Block* eb, uint64_t cam, uint64_t kp, const double* uv, double size) {
  const uint64_t poseId = pb->id;
  Observation o;
  o.resId = nextResId_++;
  o.poseId = poseId;
  o.cam = (uint8_t)cam;
  o.kp = kp;
  o.uv[0] = uv[0]; o.uv[1] = uv[1];
  o.size = size;
  o.poseH = (uint16_t)pb->handle; o.extH = (uint16_t)eb->handle;
  if (residentValid_) {
    o.pendIdx = (uint32_t)addLog_.size(); o.pendEpoch = epoch_;
    WinAdd ad;
    ad.lmH = lm.handle; ad.seq = (uint32_t)o.resId; ad.hnd = packObs(o.poseH, o.extH, o.cam); ad.pad = 0;
    if (size != lastObsSize_) { lastObsSize_ = size; lastObsWeight_ = obsWeight(size); }   // (key point sizes repeat: one division and root per size)
    ad.u = uv[0]; ad.v = uv[1]; ad.w = lastObsWeight_;
    addLog_.push_back(ad);
  }
  if (lm.obs.empty()) ++numLmObserved_;
  ++numObs_;
  lm.obs.push_back(o);
  lm.minPose = std::min(lm.minPose, o.poseId);
  lm.maxPose = std::max(lm.maxPose, o.poseId);
  pb->nObs++;
  eb->nObs++;
  pb->seenLm.push_back(lm.handle);
  obsRes2Lm_.set(o.resId, (uint64_t)reinterpret_cast<uintptr_t>(&lm));
  return o.resId;
}
// HomogeneousPointError(measurement, information) (HomogeneousPointError.cpp:58-75): squareRootInformation_ = L^T with
// information = L L^T (Eigen::LLT)
uint64_t Window::addLandmarkPrior(uint64_t lmId, const double* meas4, const double* info9) {
  auto lit = landmarks_.find(lmId);
  if (lit == landmarks_.end() || !meas4 || !info9) return 0;
  double L[9] = {0};
  for (int j = 0; j < 3; ++j) {
    double dsum = info9[j * 3 + j];
    for (int k = 0; k < j; ++k) dsum -= L[j * 3 + k] * L[j * 3 + k];
    if (!(dsum > 0)) { lastError() = "addLandmarkPrior: information matrix is not positive definite"; return 0; }
    L[j * 3 + j] = std::sqrt(dsum);
    for (int i = j + 1; i < 3; ++i) {
      double v = info9[i * 3 + j];
      for (int k = 0; k < j; ++k) v -= L[i * 3 + k] * L[j * 3 + k];
      L[i * 3 + j] = v / L[j * 3 + j];
    }
  }
  Landmark::Prior pr;
  pr.resId = nextResId_++;
  std::memcpy(pr.meas, meas4, sizeof(pr.meas));
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) pr.sqrtInfo[a * 3 + b] = L[b * 3 + a];
  lit->second.priors.push_back(pr);
  lmPriorRes2Lm_[pr.resId] = lmId;
  ++numLandmarkPriors_;
  return pr.resId;
}
int Window::removeLandmarkPrior(uint64_t resId) {
  auto it = lmPriorRes2Lm_.find(resId);
  if (it == lmPriorRes2Lm_.end()) return 0;
  Landmark& lm = landmarks_.at(it->second);
  for (size_t i = 0; i < lm.priors.size(); ++i)
    if (lm.priors[i].resId == resId) { lm.priors.erase(lm.priors.begin() + i); break; }
  lmPriorRes2Lm_.erase(it);
  --numLandmarkPriors_;
  return 1;
}
// ------------------------------------------------------------------------------------------ Map interface (Map.cpp:255-376)
int Window::mapAddParameterBlock(uint64_t id, int type, const double* values) {
  if (!values) return -1;
  if (type == 3) return addLandmark(id, values);
  if (type != 0 && type != 2) return -1;
  if (idInUse(id)) return 0;   // Map.cpp:257-260
  double x[9];
  if (type == 0) normalisedPose(values, x);
  else std::memcpy(x, values, 9 * sizeof(double));
  return addBlock(id, type == 0 ? B_POSE : B_SB, x) ? 1 : 0;
}
int Window::mapSetParameterBlock(uint64_t id, const double* values) {
  if (!values) return -1;
  if (lmIndex_.count(id)) return setLandmark(id, values);
  Block* b = findBlock(id);
  if (!b) return 0;
  if (b->kind == B_SB) std::memcpy(b->x, values, 9 * sizeof(double));
  else normalisedPose(values, b->x);
  return 1;
}
int Window::mapRemoveParameterBlock(uint64_t id) {   // Map.cpp:322-333: the residuals of the block go with it
  uint64_t hnd = 0;
  if (lmIndex_.find(id, &hnd)) {
    Landmark& lm = *lmByHandle_[(size_t)hnd];
    while (!lm.priors.empty()) removeLandmarkPrior(lm.priors.back().resId);
    eraseLandmark(lm);
    return 1;
  }
  if (!findBlock(id) || states_.count(id)) return 0;   // (the blocks of a frame leave through applyMarginalizationStrategy)
  for (const auto& kv : states_) {
    for (const StateInfo& e : kv.second.ext) if (e.id == id) return 0;
    for (const StateInfo& b : kv.second.sb) if (b.id == id) return 0;
  }
  removeBlock(id);
  return 1;
}
uint64_t Window::mapAddPoseError(uint64_t blockId, const double* meas7, const double* information36) {
  Block* b = findBlock(blockId);
  if (!b || b->kind == B_SB || !meas7 || !information36) return 0;
  Factor f;
  f.kind = F_POSE_PRIOR; f.nblk = 1; f.blocks[0] = blockId; f.m = 6;
  normalisedPose(meas7, f.meas);
  sqrtInformationUpper(information36, 6, f.sqrtInfo);
  return addFactor(std::move(f));
}
uint64_t Window::mapAddSpeedAndBiasError(uint64_t blockId, const double* meas9, const double* information81) {
  Block* b = findBlock(blockId);
  if (!b || b->kind != B_SB || !meas9 || !information81) return 0;
  Factor f;
  f.kind = F_SB_PRIOR; f.nblk = 1; f.blocks[0] = blockId; f.m = 9;
  std::memcpy(f.meas, meas9, 9 * sizeof(double));
  sqrtInformationUpper(information81, 9, f.sqrtInfo);
  return addFactor(std::move(f));
}
uint64_t Window::mapAddRelativePoseError(uint64_t block0, uint64_t block1, const double* information36) {
  Block *b0 = findBlock(block0), *b1 = findBlock(block1);
  if (!b0 || !b1 || b0->kind == B_SB || b1->kind == B_SB || b0->kind != b1->kind || !information36) return 0;
  Factor f;
  f.kind = F_RELPOSE; f.nblk = 2; f.blocks[0] = block0; f.blocks[1] = block1; f.m = 6;
  sqrtInformationUpper(information36, 6, f.sqrtInfo);
  return addFactor(std::move(f));
}
// ImuError(measurements, parameters, t_0, t_1) on (pose_0, speed/bias_0, pose_1, speed/bias_1)  (ImuError.cpp:58-75, Map.cpp:341-376):
// what Estimator::addStates creates between consecutive frames, with the blocks named by the caller
uint64_t Window::mapAddImuError(const uint64_t ids[4], const uint32_t* imuT, const double* imuM, int nImu, const ImuParams& par,
                                TimeStamp t0, TimeStamp t1) {
  if (!ids || !imuT || !imuM || nImu < 2) return 0;
  Block *p0 = findBlock(ids[0]), *s0 = findBlock(ids[1]), *p1 = findBlock(ids[2]), *s1 = findBlock(ids[3]);
  if (!p0 || !s0 || !p1 || !s1 || p0->kind != B_POSE || p1->kind != B_POSE || s0->kind != B_SB || s1->kind != B_SB) return 0;
  Factor f;
  f.kind = F_IMU; f.nblk = 4; f.m = 15;
  for (int k = 0; k < 4; ++k) f.blocks[k] = ids[k];
  f.imuT.assign(imuT, imuT + 2 * (size_t)nImu);
  f.imuMeas.assign(imuM, imuM + 6 * (size_t)nImu);
  std::memset(&f.imu, 0, sizeof(f.imu));
  f.imu.sampleCount = nImu;
  f.imu.t0[0] = t0.sec; f.imu.t0[1] = t0.nsec;
  f.imu.t1[0] = t1.sec; f.imu.t1[1] = t1.nsec;
  f.imu.par = par;
  f.imu.redo = 1;
  f.imu.Delta_q[3] = 1.0;
  return addFactor(std::move(f));
}
// SonarError(range, heading, information, landmark patch) on a pose block  (SonarError.cpp:57-183; T_SSo as set for the handle:
// the reference only ever passes the identity); the residual uses the MEAN of the patch (:124-131)
uint64_t Window::mapAddSonarError(uint64_t poseBlock, double range, double heading, double information, const double* patch, int nPatch) {
  Block* b = findBlock(poseBlock);
  if (!b || b->kind != B_POSE || !patch || nPatch < 1 || !(information > 0.0)) return 0;
  Factor f;
  f.kind = F_SONAR; f.nblk = 1; f.blocks[0] = poseBlock; f.m = 1;
  f.meas[0] = range; f.meas[1] = heading;
  double mean[3] = {0, 0, 0};
  for (int i = 0; i < nPatch; ++i) { mean[0] += patch[3 * i]; mean[1] += patch[3 * i + 1]; mean[2] += patch[3 * i + 2]; }
  f.meas[2] = mean[0] / nPatch; f.meas[3] = mean[1] / nPatch; f.meas[4] = mean[2] / nPatch;
  std::memcpy(f.aux, T_SSo_, sizeof(T_SSo_));
  f.sqrtInfo[0] = std::sqrt(information);
  return addFactor(std::move(f));
}
// DepthError(depth, information, first depth) on a pose block  (DepthError.cpp:50-139)
uint64_t Window::mapAddDepthError(uint64_t poseBlock, double depth, double information, double firstDepth) {
  Block* b = findBlock(poseBlock);
  if (!b || b->kind != B_POSE || !(information > 0.0)) return 0;
  Factor f;
  f.kind = F_DEPTH; f.nblk = 1; f.blocks[0] = poseBlock; f.m = 1;
  f.meas[0] = depth; f.meas[1] = firstDepth;
  f.sqrtInfo[0] = std::sqrt(information);
  return addFactor(std::move(f));
}
// Map::addResidualBlock with a cost function the library has no kernel for (Map.cpp:341-376; the reference hands ANY
// ::ceres::CostFunction to Ceres).  The function is evaluated by the HOST: before every evaluation launch the blocks it names are
// read back, the callback computes the residual and the Jacobians in MINIMAL coordinates (6 columns per pose / extrinsics block,
// 9 per speed / bias block -- what ErrorInterface::EvaluateWithMinimalJacobians returns), and the record the device kernels
// consume (FactorLin) is written for it.  A slow path by construction -- one stream synchronisation per evaluation -- for graphs
// built through okvis::ceres::Map by third parties; no loss function, no landmark blocks (they are eliminated on the device
// from reprojection residuals alone), residual dimension <= 15, at most 4 blocks / 30 minimal columns.
uint64_t Window::mapAddHostResidual(const uint64_t* blockIds, int nBlocks, int residualDim, int (*fn)(void*, const double* const*, double*, double**),
                                    void* user) {
  if (!blockIds || !fn || nBlocks < 1 || nBlocks > 4 || residualDim < 1 || residualDim > 15) return 0;
  int cols = 0;
  for (int b = 0; b < nBlocks; ++b) {
    const Block* blk = findBlock(blockIds[b]);
    if (!blk) return 0;   // (unknown, or a landmark)
    for (int c = 0; c < b; ++c)
      if (blockIds[c] == blockIds[b]) return 0;
    cols += blk->kind == B_SB ? 9 : 6;
  }
  if (cols > 30) return 0;
  Factor f;
  f.kind = F_HOST; f.nblk = nBlocks; f.m = residualDim;
  for (int b = 0; b < nBlocks; ++b) f.blocks[b] = blockIds[b];
  f.hostFn = fn; f.hostUser = user;
  return addFactor(std::move(f));
}
void Window::evaluateHostFactors(bool cand, hipStream_t s) {
  if (hostFactors_.empty()) return;
  const DeviceProblem& p = prob_;
  HIP_OK(hipStreamSynchronize(s));   // (the candidate blocks are the device's: k_post_solve / k_step_retract wrote them)
  std::vector<double> hp((size_t)std::max(p.nPose, 1) * 7), he((size_t)std::max(p.nExt, 1) * 7), hs((size_t)std::max(p.nSb, 1) * 9);
  if (p.nPose > 0) HIP_OK(hipMemcpy(hp.data(), cand ? p.poseC : p.pose, sizeof(double) * 7 * (size_t)p.nPose, hipMemcpyDeviceToHost));
  if (p.nExt > 0) HIP_OK(hipMemcpy(he.data(), cand ? p.extC : p.ext, sizeof(double) * 7 * (size_t)p.nExt, hipMemcpyDeviceToHost));
  if (p.nSb > 0) HIP_OK(hipMemcpy(hs.data(), cand ? p.sbC : p.sb, sizeof(double) * 9 * (size_t)p.nSb, hipMemcpyDeviceToHost));
  FactorLin* dst = cand ? p.linCand : p.linCur;
  for (const auto& hf : hostFactors_) {
    const Factor& f = factors_.at(hf.second);
    const double* params[4] = {nullptr, nullptr, nullptr, nullptr};
    double jac[4][15 * 9];
    double* jp[4] = {jac[0], jac[1], jac[2], jac[3]};
    FactorLin L;
    std::memset(&L, 0, sizeof(L));
    L.m = f.m;
    for (int b = 0; b < f.nblk; ++b) {
      const Block& blk = blocks_.at(f.blocks[b]);
      if (blk.kind == B_POSE) { const int sl = poseSlot_.at(blk.id); params[b] = &hp[(size_t)7 * sl]; L.off[b] = hPoseOffKeep_[(size_t)sl]; L.dim[b] = 6; }
      else if (blk.kind == B_EXT) { const int sl = extSlot_.at(blk.id); params[b] = &he[(size_t)7 * sl]; L.off[b] = hExtOffKeep_[(size_t)sl]; L.dim[b] = 6; }
      else { const int sl = sbSlot_.at(blk.id); params[b] = &hs[(size_t)9 * sl]; L.off[b] = hSbOffKeep_[(size_t)sl]; L.dim[b] = 9; }
      L.ncols += L.dim[b];
      std::memset(jac[b], 0, sizeof(jac[b]));
    }
    for (int b = f.nblk; b < 4; ++b) L.off[b] = -1;
    if (!f.hostFn(f.hostUser, params, L.r, jp)) throw std::runtime_error("svin_ba: a host cost function (residual " + std::to_string(f.id) + ") reported a failure");
    int col0 = 0;
    for (int b = 0; b < f.nblk; ++b) {
      for (int a = 0; a < f.m; ++a)
        for (int c = 0; c < L.dim[b]; ++c) L.J[a * L.ncols + col0 + c] = jac[b][a * L.dim[b] + c];
      col0 += L.dim[b];
    }
    HIP_OK(hipMemcpy(dst + hf.first, &L, sizeof(L), hipMemcpyHostToDevice));
  }
}
// ReprojectionError<GEOMETRY>(geometry of camera `cam`, uv, information) with CauchyLoss(1) on (pose, landmark, extrinsics):
// what Estimator::addObservation creates, with the blocks named by the caller.  The device kernels store ONE weight per
// residual (Estimator only ever passes 64 / size^2 * I), so the information has to be a multiple of the identity.
uint64_t Window::mapAddReprojectionError(uint64_t poseBlock, uint64_t landmark, uint64_t extBlock, uint64_t cam, const double* uv,
                                         const double* information4) {
  uint64_t hnd = 0;
  if (!uv || !information4 || cam >= cameras_.size() || !lmIndex_.find(landmark, &hnd)) return 0;
  if (information4[1] != 0.0 || information4[2] != 0.0 || information4[0] != information4[3] || !(information4[0] > 0.0)) {
    lastError() = "map_add_reprojection_error: the information matrix must be a positive multiple of the identity";
    return 0;
  }
  Block *pb = findBlock(poseBlock), *eb = findBlock(extBlock);
  if (!pb || !eb || pb == eb || pb->kind != B_POSE) return 0;
  if (eb->kind == B_POSE) {
    // a 7-dimensional block added through the Map interface becomes an extrinsics block with its first use as one
    if (eb->nObs != 0 || states_.count(extBlock)) { lastError() = "map_add_reprojection_error: the extrinsics block is in use as a pose"; return 0; }
    for (uint64_t rid : eb->residuals) {
      auto it = factors_.find(rid);
      if (it != factors_.end() && it->second.kind == F_RELPOSE) { lastError() = "map_add_reprojection_error: relative-pose residuals tie the block to poses"; return 0; }
    }
    blockByHandle_[B_POSE][eb->handle] = nullptr;
    freeBlockH_[B_POSE].push_back(eb->handle);
    eb->kind = B_EXT;
    if (!freeBlockH_[B_EXT].empty()) { eb->handle = freeBlockH_[B_EXT].back(); freeBlockH_[B_EXT].pop_back(); }
    else eb->handle = nextBlockH_[B_EXT]++;
    if ((int)blockByHandle_[B_EXT].size() <= eb->handle) blockByHandle_[B_EXT].resize(eb->handle + 1, nullptr);
    blockByHandle_[B_EXT][eb->handle] = eb;
    for (Block*& c : blockCache_) c = nullptr;
  }
  if (eb->kind != B_EXT) return 0;
  Landmark& lm = *lmByHandle_[(size_t)hnd];
  // (no duplicate rule at this level: Map::addResidualBlock accepts any number of residuals on the same blocks; the key point
  // index of the record is the residual id it is about to get)
  return addObservationRecord(lm, pb, eb, cam, nextResId_, uv, -std::sqrt(information4[0]));
}
int Window::mapRemoveResidualBlock(uint64_t resId) {   // Map.cpp:467-492
  if (obsRes2Lm_.count(resId)) return removeObservationById(resId);
  if (lmPriorRes2Lm_.count(resId)) return removeLandmarkPrior(resId);
  if (!factors_.count(resId)) return 0;
  removeFactor(resId);
  return 1;
}

int Window::removeObservation(uint64_t lmId, uint64_t poseId, uint64_t cam, uint64_t kp) {  // :452-474
  auto lit = landmarks_.find(lmId);
  if (lit == landmarks_.end()) return 0;
  for (size_t i = 0; i < lit->second.obs.size(); ++i) {
    const Observation& o = lit->second.obs[i];
    if (o.poseId == poseId && (uint64_t)o.cam == cam && o.kp == kp) { removeObsRecord(lit->second, i); return 1; }
  }
  return 0;
}
int Window::removeObservationById(uint64_t resId) {  // :432-449
  uint64_t node = 0;
  if (!obsRes2Lm_.find(resId, &node)) return 0;
  Landmark& lm = *reinterpret_cast<Landmark*>((uintptr_t)node);
  for (size_t i = 0; i < lm.obs.size(); ++i)
    if (lm.obs[i].resId == resId) { removeObsRecord(lm, i); return 1; }
  return 0;
}

// ------------------------------------------------------------------------------------------ getters / setters
int Window::get_T_WS(uint64_t id, double* T) const {
  auto it = states_.find(id);
  if (it == states_.end() || !it->second.pose.exists) return 0;
  std::memcpy(T, blocks_.at(it->second.pose.id).x, 7 * sizeof(double));
  return 1;
}
int Window::getSpeedAndBias(uint64_t id, size_t imu, double* sb) const {
  auto it = states_.find(id);
  if (it == states_.end() || imu >= it->second.sb.size() || !it->second.sb[imu].exists) return 0;
  std::memcpy(sb, blocks_.at(it->second.sb[imu].id).x, 9 * sizeof(double));
  return 1;
}
int Window::getCameraSensorStates(uint64_t id, size_t cam, double* T) const {
  auto it = states_.find(id);
  if (it == states_.end() || cam >= it->second.ext.size() || !it->second.ext[cam].exists) return 0;
  std::memcpy(T, blocks_.at(it->second.ext[cam].id).x, 7 * sizeof(double));
  return 1;
}
const Landmark* Window::landmark(uint64_t id) const {
  auto it = landmarks_.find(id);
  if (it == landmarks_.end()) return nullptr;
  syncLandmarks();
  return &it->second;
}
static void normalisedPose(const double* T, double* out) {  // PoseParameterBlock::setEstimate keeps a Transformation
  const Quat q = qnormalized(Quat{T[3], T[4], T[5], T[6]});
  out[0] = T[0]; out[1] = T[1]; out[2] = T[2];
  out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}
int Window::set_T_WS(uint64_t id, const double* T) {
  auto it = states_.find(id);
  if (it == states_.end() || !it->second.pose.exists) return 0;
  normalisedPose(T, blocks_.at(it->second.pose.id).x);
  return 1;
}
int Window::setSpeedAndBias(uint64_t id, size_t imu, const double* sb) {
  auto it = states_.find(id);
  if (it == states_.end() || imu >= it->second.sb.size() || !it->second.sb[imu].exists) return 0;
  std::memcpy(blocks_.at(it->second.sb[imu].id).x, sb, 9 * sizeof(double));
  return 1;
}
int Window::setCameraSensorStates(uint64_t id, size_t cam, const double* T) {
  auto it = states_.find(id);
  if (it == states_.end() || cam >= it->second.ext.size() || !it->second.ext[cam].exists) return 0;
  normalisedPose(T, blocks_.at(it->second.ext[cam].id).x);
  return 1;
}
int Window::setLandmark(uint64_t id, const double* hp) {
  auto it = landmarks_.find(id);
  if (it == landmarks_.end()) return 0;
  syncLandmarks();   // (a later fetch must not bring back the value this call replaces)
  std::memcpy(it->second.hp, hp, 4 * sizeof(double));
  if (residentValid_) {
    WinLmSet st;
    st.h = it->second.handle; st.setQuality = 0; st.quality = 0.0;
    std::memcpy(st.hp, hp, sizeof(st.hp));
    setLog_.push_back(st);
  }
  return 1;
}
int Window::setLandmarkInitialized(uint64_t id, bool init) {
  auto it = landmarks_.find(id);
  if (it == landmarks_.end()) return 0;
  it->second.initialized = init;
  return 1;
}
int Window::setKeyframe(uint64_t frameId, bool isKf) {
  auto it = states_.find(frameId);
  if (it == states_.end()) return 0;
  it->second.isKeyframe = isKf;
  return 1;
}
int Window::getImuPreIntegral(uint64_t poseId, double* out7) const {
  auto it = imuIntegrals_.find(poseId);
  if (it == imuIntegrals_.end()) return 0;
  std::memcpy(out7, it->second.data(), 7 * sizeof(double));
  return 1;
}
void Window::setImuPreIntegral(uint64_t poseId, const double* in7) {
  std::array<double, 7> a;
  std::memcpy(a.data(), in7, sizeof(a));
  imuIntegrals_.insert(std::make_pair(poseId, a));
}
int Window::setParameterBlockConstant(uint64_t id, bool constant) {
  Block* b = findBlock(id);
  if (!b) {
    uint64_t hnd = 0;
    if (!lmIndex_.find(id, &hnd)) return 0;
    // a constant landmark (okvis::Estimator never makes one; Map-level callers and the reference's TestMap do): its observations
    // are packed with a negative weight, which the evaluation turns into a zero landmark Jacobian.  Such windows take the host path.
    Landmark& lm = *lmByHandle_[(size_t)hnd];
    if (lm.fixed != constant) { lm.fixed = constant; numFixedLandmarks_ += constant ? 1 : -1; }
    return 1;
  }
  b->fixed = constant;
  return 1;
}
// Map::resetParameterization (Map.cpp:513-543).  The reference removes the block and adds it again with the other manifold; here
// the block keeps its place and only the set of held tangent directions changes.  Map::Parameterization (Map.hpp:97-105):
// 0 HomogeneousPoint, 1 Pose6d, 2 Pose3d (orientation varies, position held: PoseManifold.cpp:173-178), 3 Pose4d (position + yaw,
// :276-282), 4 Pose2d (roll / pitch, :372-376), 5 Trivial.  1 = done, 0 = unknown block (the reference's false), negative = a
// manifold the block's type cannot take (the reference would hand Ceres a manifold of the wrong ambient size and abort).
int Window::resetParameterization(uint64_t id, int parameterization) {
  Block* b = findBlock(id);
  if (!b) {
    uint64_t hnd = 0;
    if (!lmIndex_.find(id, &hnd)) return 0;
    return parameterization == 0 ? 1 : -1 /* SVIN_ERR_INVALID_ARG */;
  }
  if (b->kind == B_SB) return parameterization == 5 ? 1 : -1 /* SVIN_ERR_INVALID_ARG */;
  unsigned char lock;
  switch (parameterization) {
    case 1: lock = 0; break;
    case 2: lock = 0x07; break;          // delta = (0, 0, 0, d0, d1, d2)
    case 3: lock = 0x18; break;          // delta = (d0, d1, d2, 0, 0, d3)
    case 4: lock = 0x27; break;          // delta = (0, 0, 0, d0, d1, 0)
    default: return -1 /* SVIN_ERR_INVALID_ARG */;
  }
  b->lock = lock;
  return 1;
}
int Window::parameterization(uint64_t id) const {
  const Block* b = findBlock(id);
  if (!b) {
    uint64_t hnd = 0;
    return lmIndex_.find(id, &hnd) ? 0 : -2 /* SVIN_ERR_NOT_FOUND */;
  }
  if (b->kind == B_SB) return 5;
  return b->lock == 0 ? 1 : (b->lock == 0x07 ? 2 : (b->lock == 0x18 ? 3 : 4));
}
int Window::isParameterBlockConstant(uint64_t id) const {
  const Block* b = findBlock(id);
  if (!b) {
    uint64_t hnd = 0;
    if (!lmIndex_.find(id, &hnd)) return -2;
    return lmByHandle_[(size_t)hnd]->fixed ? 1 : 0;
  }
  return b->fixed ? 1 : 0;
}
int Window::residualsOf(uint64_t blockId, std::vector<uint64_t>& out) const {
  out.clear();
  auto lit = landmarks_.find(blockId);
  if (lit != landmarks_.end()) {
    for (const Observation& o : lit->second.obs) out.push_back(o.resId);
    for (const Landmark::Prior& pr : lit->second.priors) out.push_back(pr.resId);
    std::sort(out.begin(), out.end());
    return 1;
  }
  const Block* b = findBlock(blockId);
  if (!b) return 0;
  out = b->residuals;   // small factors and the prior, in insertion order
  if (b->nObs > 0)      // reprojection residuals live in their landmark's list only
    for (const auto& kv : landmarks_)
      for (const Observation& o : kv.second.obs)
        if (o.poseId == blockId || (b->kind == B_EXT && o.extH == b->handle)) out.push_back(o.resId);
  std::sort(out.begin(), out.end());   // ids grow with insertion: this IS insertion order across both kinds
  return 1;
}
int Window::residualKind(uint64_t resId) const {
  if (obsRes2Lm_.count(resId)) return 100;
  if (hasPrior_ && resId == priorResId_) return 101;
  if (lmPriorRes2Lm_.count(resId)) return 102;
  auto it = factors_.find(resId);
  return it == factors_.end() ? -1 : it->second.kind;
}
int Window::residualInfo(int n, const uint64_t* resIds, int32_t* kind, int32_t* m, int32_t* nBlocks, int32_t* dims4) const {
  int known = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t rid = resIds[i];
    int k = residualKind(rid), mm = 0, nb = 0, d[4] = {0, 0, 0, 0};
    if (k == 100) { mm = 2; nb = 3; d[0] = 7; d[1] = 4; d[2] = 7; }                       // ReprojectionErrorBase.hpp:50-54
    else if (k == 101) { mm = priorM_; nb = (int)priorBlocks_.size(); }                   // (its block list: svin_ba_parameters_of)
    else if (k == 102) { mm = 3; nb = 1; d[0] = 4; }
    else if (k >= 0) {
      const Factor& f = factors_.at(rid);
      mm = f.m; nb = f.nblk;
      for (int b = 0; b < f.nblk; ++b) {
        const Block* blk = findBlock(f.blocks[b]);
        d[b] = blk ? (blk->kind == B_SB ? 9 : 7) : 0;
      }
    }
    if (k >= 0) ++known;
    if (kind) kind[i] = k;
    if (m) m[i] = mm;
    if (nBlocks) nBlocks[i] = nb;
    if (dims4) for (int b = 0; b < 4; ++b) dims4[4 * i + b] = d[b];
  }
  return known;
}
int Window::parametersOf(uint64_t resId, std::vector<uint64_t>& out) const {
  out.clear();
  uint64_t obsNode = 0;
  if (obsRes2Lm_.find(resId, &obsNode)) {   // ReprojectionError: pose, landmark, extrinsics (ReprojectionErrorBase.hpp:50-54)
    const Landmark& lm = *reinterpret_cast<const Landmark*>((uintptr_t)obsNode);
    for (const Observation& o : lm.obs)
      if (o.resId == resId) { out = {o.poseId, lm.id, extIdOf(o)}; return 1; }
    return 0;
  }
  if (hasPrior_ && resId == priorResId_) {
    for (const PriorBlockHost& pb : priorBlocks_) out.push_back(pb.id);
    return 1;
  }
  auto pt = lmPriorRes2Lm_.find(resId);
  if (pt != lmPriorRes2Lm_.end()) { out = {pt->second}; return 1; }
  auto it = factors_.find(resId);
  if (it == factors_.end()) return 0;
  for (int b = 0; b < it->second.nblk; ++b) out.push_back(it->second.blocks[b]);
  return 1;
}
uint64_t Window::currentKeyframeId() const {
  for (auto rit = states_.rbegin(); rit != states_.rend(); ++rit)
    if (rit->second.isKeyframe) return rit->first;
  return 0;
}
uint64_t Window::frameIdByAge(size_t age) const {
  auto rit = states_.rbegin();
  for (size_t i = 0; i < age; ++i) { ++rit; if (rit == states_.rend()) return 0; }
  return rit == states_.rend() ? 0 : rit->first;
}
bool Window::isInImuWindow(uint64_t id) const {
  auto it = states_.find(id);
  if (it == states_.end() || it->second.sb.empty()) return false;
  return it->second.sb[0].exists;
}
int Window::describeBlock(uint64_t id, uint64_t* frame, int32_t* kind, int32_t* index) const {
  for (const auto& kv : states_) {
    const State& s = kv.second;
    if (s.pose.id == id) { *frame = s.id; *kind = 0; *index = 0; return 1; }
    for (size_t i = 0; i < s.ext.size(); ++i) if (s.ext[i].id == id) { *frame = s.id; *kind = 1; *index = (int)i; return 1; }
    for (size_t i = 0; i < s.sb.size(); ++i) if (s.sb[i].id == id) { *frame = s.id; *kind = 2; *index = (int)i; return 1; }
  }
  return 0;
}

int Window::getParameterBlock(uint64_t id, int32_t* type, double* values, uint32_t* sec, uint32_t* nsec, int32_t* fixed,
                              int32_t* initialized) const {
  auto lit = landmarks_.find(id);
  if (lit != landmarks_.end()) {
    syncLandmarks();
    if (type) *type = 3;
    if (values) std::memcpy(values, lit->second.hp, 4 * sizeof(double));
    if (sec) *sec = 0;
    if (nsec) *nsec = 0;
    if (fixed) *fixed = lit->second.fixed ? 1 : 0;
    if (initialized) *initialized = lit->second.initialized ? 1 : 0;
    return 4;
  }
  const Block* b = findBlock(id);
  if (!b) return -2;
  const int dim = b->kind == B_SB ? 9 : 7;
  if (type) *type = b->kind == B_POSE ? 0 : b->kind == B_EXT ? 1 : 2;
  if (values) std::memcpy(values, b->x, dim * sizeof(double));
  if (fixed) *fixed = b->fixed ? 1 : 0;
  if (initialized) *initialized = 1;
  // every block addStates creates carries the frame's time stamp (Estimator.cpp:126, :208-214, :232); a fixed block
  // whose frame has left the window keeps none
  uint64_t frame = 0;
  int32_t k = 0, idx = 0;
  TimeStamp t;
  t.sec = 0; t.nsec = 0;
  if (describeBlock(id, &frame, &k, &idx) == 1) t = states_.at(frame).stamp;
  if (sec) *sec = t.sec;
  if (nsec) *nsec = t.nsec;
  return dim;
}
void Window::parameterBlockIds(std::vector<uint64_t>& out) const {
  out.clear();
  for (const auto& kv : blocks_) out.push_back(kv.first);
  for (const auto& kv : landmarks_) out.push_back(kv.first);
  std::sort(out.begin(), out.end());
}

// ------------------------------------------------------------------------------------------ enqueue thread
void Window::enqueueLoop() {
  (void)hipSetDevice(device_);
  std::unique_lock<std::mutex> lock(enqueueMutex_);
  while (true) {
    enqueueCv_.wait(lock, [&] { return enqueueStop_ || (enqueueBusy_ && enqueueJob_); });
    if (enqueueStop_) return;
    std::function<void()> job = std::move(enqueueJob_);
    enqueueJob_ = nullptr;
    lock.unlock();
    std::exception_ptr err;
    try { job(); } catch (...) { err = std::current_exception(); }
    job = nullptr;   // (releases the host tables of the job)
    lock.lock();
    enqueueError_ = err;
    enqueueBusy_ = false;
    enqueueCv_.notify_all();
  }
}
void Window::enqueueAsync(std::function<void()> job) {
  quiesce();
  std::lock_guard<std::mutex> lock(enqueueMutex_);
  if (!enqueueThread_.joinable()) enqueueThread_ = std::thread([this] { enqueueLoop(); });
  enqueueJob_ = std::move(job);
  enqueueBusy_ = true;
  enqueueCv_.notify_all();
}
void Window::quiesce() const {
  std::unique_lock<std::mutex> lock(enqueueMutex_);
  enqueueCv_.wait(lock, [&] { return !enqueueBusy_; });
  if (enqueueError_) {
    std::exception_ptr e = enqueueError_;
    enqueueError_ = nullptr;
    std::rethrow_exception(e);
  }
}

// everything the handle has enqueued -- the marginalisation job of the last applyMarginalizationStrategy above all -- has run
void Window::waitIdle() {
  quiesce();
  HIP_OK(hipStreamSynchronize(stream_));
  HIP_OK(hipStreamSynchronize(stream2_));
}

// ------------------------------------------------------------------------------------------ device-resident window
bool Window::useResident() const {   // (called by pack() once the state tables are known)
  const bool forceHost = optOn(kOptHostPack);
  return packMode_ == 0 && !forceHost && world_ <= 1 && rcclComm_ == nullptr && numLandmarkPriors_ == 0 && numFixedLandmarks_ == 0 &&
         poseIds_.size() <= (size_t)kResidentPoseCap;
}
void Window::invalidateResident() {
  syncLandmarks();
  residentValid_ = false;
  addLog_.clear(); remLog_.clear(); setLog_.clear();
  ++epoch_;
}
// Handles are creation numbers and are never re-used, so the per-handle tables grow with the landmarks ever created.  When
// the live ones have become a small part of the range the survivors are renumbered in id order (the order the reference's
// std::map walks them in) and the device copy is rebuilt from the graph.  Host graph authoritative (invalidateResident first).
void Window::renumberLandmarkHandles() {
  std::vector<int> newOf(lmByHandle_.size(), -1);
  std::vector<Landmark*> fresh;
  fresh.reserve(landmarks_.size());
  lmIterByHandle_.clear();
  for (auto it = landmarks_.begin(); it != landmarks_.end(); ++it) {
    newOf[it->second.handle] = (int)fresh.size();
    it->second.handle = (int)fresh.size();
    lmIndex_.set(it->first, (uint64_t)it->second.handle);
    fresh.push_back(&it->second);
    lmIterByHandle_.push_back(it);
  }
  lmByHandle_.swap(fresh);
  nextLmHandle_ = (int)lmByHandle_.size();
  auto remap = [&](std::vector<int>& v) {
    size_t k = 0;
    for (int h : v)
      if (h >= 0 && h < (int)newOf.size() && newOf[h] >= 0) v[k++] = newOf[h];
    v.resize(k);
  };
  remap(emptyLm_);
  for (auto& kv : blocks_) remap(kv.second.seenLm);
}
void Window::checkResidentStatus() {
  if (resStatus_ && *resStatus_ != 0) {
    const int code = *resStatus_;
    *resStatus_ = 0;
    invalidateResident();
    throw std::runtime_error("svin_ba: the device-resident window disagrees with the host graph (status " + std::to_string(code) + ")");
  }
}
void Window::flushPendingQuality() const {
  if (!qualityPending_) return;
  quiesce();
  qualityPending_ = false;
  if (qualityProb_.L > 0) launchLandmarkQuality(qualityProb_, dQuality_.p, stream_);
  launchWindowStoreLandmarks(res_.H, res_.slotOfH[res_.cur].p, qualityProb_.lm, dQuality_.p, res_.lmHp.p, res_.qualH.p, stream_);
}
void Window::syncLandmarks() const {
  if (!lmStale_) return;
  quiesce();
  flushPendingQuality();
  lmStale_ = false;
  const size_t H = (size_t)hFlushed_;
  if (H == 0) return;
  if (5 * H > lmSyncCap_) {
    if (lmSyncHost_) (void)hipHostFree(lmSyncHost_);
    lmSyncCap_ = std::max<size_t>(10 * H, 1 << 14);
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&lmSyncHost_), lmSyncCap_ * sizeof(double), hipHostMallocDefault));
  }
  HIP_OK(hipMemcpyAsync(lmSyncHost_, res_.lmHp.p, sizeof(double) * 4 * H, hipMemcpyDeviceToHost, stream_));
  HIP_OK(hipMemcpyAsync(lmSyncHost_ + 4 * H, res_.qualH.p, sizeof(double) * H, hipMemcpyDeviceToHost, stream_));
  HIP_OK(hipStreamSynchronize(stream_));
  for (size_t h = 0; h < H && h < lmByHandle_.size(); ++h) {
    Landmark* lm = lmByHandle_[h];
    if (!lm) continue;
    std::memcpy(lm->hp, lmSyncHost_ + 4 * h, 4 * sizeof(double));
    lm->quality = lmSyncHost_[4 * H + h];
  }
}
// stages this frame's delta (or, when the device copy is not valid, the whole graph as one delta) and fills the rebuild
// kernel's arguments; pack() launches it behind the scatter of the staged block
void Window::flushResident(hipStream_t s, bool wantOrder, std::vector<StagedCopy>& pending, ResidentArgs& ra) {
  Resident& R = res_;
  if (residentValid_ && (size_t)nextLmHandle_ > std::max<size_t>(8192, 8 * landmarks_.size())) invalidateResident();
  const bool full = !residentValid_;
  if (full) {
    if ((size_t)nextLmHandle_ > std::max<size_t>(4096, 2 * landmarks_.size())) renumberLandmarkHandles();
    addLog_.clear(); remLog_.clear(); setLog_.clear();
    ++epoch_;
    for (Landmark* lm : lmByHandle_) {
      if (!lm) continue;
      WinLmSet st;
      st.h = lm->handle; st.setQuality = 1; st.quality = lm->quality;
      std::memcpy(st.hp, lm->hp, sizeof(st.hp));
      setLog_.push_back(st);
      for (Observation& o : lm->obs) {
        o.pendEpoch = 0;
        WinAdd ad;
        ad.lmH = lm->handle; ad.seq = (uint32_t)o.resId; ad.hnd = packObs(o.poseH, o.extH, o.cam); ad.pad = 0;
        ad.u = o.uv[0]; ad.v = o.uv[1]; ad.w = obsWeight(o.size);
        addLog_.push_back(ad);
      }
    }
    R.N = 0; R.L = 0; R.H = 0;
  }
  const size_t H = (size_t)nextLmHandle_, N = numObs_, L = numLmObserved_;
  // nothing was added or removed (a second optimize() of the same window, the benchmark's loop, Map::solve after setEstimate):
  // the table stays, only what depends on values and slots is refreshed
  const bool valuesOnly = !full && addLog_.empty() && remLog_.empty() && N == (size_t)R.N && L == (size_t)R.L;
  const int cur = R.cur, nxt = valuesOnly ? R.cur : 1 - R.cur;
  // per-handle tables keep their contents; new handles start from zero
  const size_t Hc = std::max<size_t>(H, 1);
  R.cnt.growKeep(Hc, R.H, s); R.addsH.growKeep(Hc, R.H, s); R.addCur.growKeep(Hc, R.H, s);
  R.lmHp.growKeep(4 * Hc, 4 * (size_t)R.H, s); R.qualH.growKeep(Hc, R.H, s);
  R.slotOfH[cur].growKeep(Hc, R.H, s);
  if (full) {
    HIP_OK(hipMemsetAsync(R.cnt.p, 0, sizeof(int) * R.cnt.cap, s));
    HIP_OK(hipMemsetAsync(R.addsH.p, 0, sizeof(int) * R.addsH.cap, s));
    HIP_OK(hipMemsetAsync(R.addCur.p, 0, sizeof(int) * R.addCur.cap, s));
  }
  R.live.growKeep(std::max<size_t>(std::max(N, (size_t)R.N), 1), R.N, s);
  // the set the new CSR is written into
  R.slotOfH[nxt].reserve(Hc);
  R.lmPtr[nxt].reserve(L + 2); R.handleOfSlot[nxt].reserve(L + 1);
  R.uv[nxt].reserve(2 * N + 2); R.w[nxt].reserve(N + 1); R.hnd[nxt].reserve(N + 1); R.seq[nxt].reserve(N + 1); R.obsLm[nxt].reserve(N + 1);
  if (!R.lmPtr[cur].p) { R.lmPtr[cur].reserve(2); R.handleOfSlot[cur].reserve(1); R.uv[cur].reserve(2); R.w[cur].reserve(1); R.hnd[cur].reserve(1); R.seq[cur].reserve(1); R.obsLm[cur].reserve(1); }
  auto stage = [&](auto& buf, const auto& host) {
    using T = typename std::decay_t<decltype(host)>::value_type;
    buf.reserve(std::max<size_t>(host.size() + 16 / sizeof(T) + 1, 1));
    if (!host.empty()) pending.push_back({host.data(), sizeof(T) * host.size(), buf.p});
  };
  stage(R.adds, addLog_); stage(R.rems, remLog_); stage(R.sets, setLog_);
  if (!resStatus_) {
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&resStatus_), 64, hipHostMallocMapped));
    *resStatus_ = 0;
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&resStatusDev_), resStatus_, 0));
  }
  ra.nAdd = (int)addLog_.size(); ra.nRem = (int)remLog_.size(); ra.nSet = (int)setLog_.size(); ra.H = (int)H;
  ra.valuesOnly = valuesOnly ? 1 : 0; ra.Hprev = R.H;
  ra.Nold = R.N; ra.Lold = R.L; ra.Nnew = (int)N; ra.Lnew = (int)L;
  ra.wantOrder = wantOrder ? 1 : 0;
  ra.adds = R.adds.p; ra.rems = R.rems.p; ra.sets = R.sets.p;
  ra.lmPtrOld = R.lmPtr[cur].p; ra.handleOfSlotOld = R.handleOfSlot[cur].p; ra.slotOfHOld = R.slotOfH[cur].p;
  ra.uvOld = R.uv[cur].p; ra.wOld = R.w[cur].p; ra.hndOld = R.hnd[cur].p; ra.seqOld = R.seq[cur].p; ra.obsLmOld = R.obsLm[cur].p;
  ra.obsLm = R.obsLm[nxt].p;
  ra.live = R.live.p;
  ra.lmPtrNew = R.lmPtr[nxt].p; ra.handleOfSlotNew = R.handleOfSlot[nxt].p; ra.slotOfHNew = R.slotOfH[nxt].p;
  ra.uvNew = R.uv[nxt].p; ra.wNew = R.w[nxt].p; ra.hndNew = R.hnd[nxt].p; ra.seqNew = R.seq[nxt].p;
  ra.cnt = R.cnt.p; ra.addsH = R.addsH.p; ra.addCur = R.addCur.p; ra.lmHp = R.lmHp.p; ra.qualH = R.qualH.p;
  ra.status = resStatusDev_;
  // from here on the device copy is what the logs are relative to
  R.cur = nxt; R.N = (int)N; R.L = (int)L; R.H = (int)H;
  hFlushed_ = (int)H;
  residentValid_ = true;
}

// ------------------------------------------------------------------------------------------ pack: host graph -> HBM
template <class T>
static void upload(DevBuf<T>& buf, const std::vector<T>& host, hipStream_t s) {
  buf.reserve(std::max<size_t>(host.size(), 1));
  if (!host.empty()) HIP_OK(hipMemcpyAsync(buf.p, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice, s));
}

// Every host array of a job goes into ONE pinned block behind a segment table: one DMA, one scatter kernel (separate
// pageable copies cost ~4 us of enqueueing and ~4 us of draining EACH).  The sources are copied here, the caller's vectors
// may go away right after the call.
void Window::flushStaged(const std::vector<StagedCopy>& pending, hipStream_t s) {
  if (pending.empty()) return;
  auto r16 = [](size_t b) { return (b + 15) / 16 * 16; };
  const size_t tableBytes = r16(sizeof(StageSegment) * pending.size());
  size_t total = tableBytes;
  for (const StagedCopy& pe : pending) if (pe.src) total += r16(pe.bytes);
  if (stageEvt_) HIP_OK(hipEventSynchronize(stageEvt_));   // the previous block may still be on its way
  else HIP_OK(hipEventCreateWithFlags(&stageEvt_, hipEventDisableTiming));
  if (total > stageHostCap_) {
    if (stageHost_) (void)hipHostFree(stageHost_);
    stageHostCap_ = std::max<size_t>(2 * total, 1 << 20);
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&stageHost_), stageHostCap_, hipHostMallocDefault));
  }
  stageDev_.reserve(std::max<size_t>(total, 16));
  StageSegment* table = reinterpret_cast<StageSegment*>(stageHost_);
  size_t off = tableBytes;
  for (size_t i = 0; i < pending.size(); ++i) {
    const size_t b16 = r16(pending[i].bytes);
    if (!pending[i].src) {   // a clear (the destination is over-allocated to the rounded size like every staged array)
      table[i] = StageSegment{kStageClear, (unsigned long long)b16, pending[i].dst};
      continue;
    }
    table[i] = StageSegment{(unsigned long long)off, (unsigned long long)b16, pending[i].dst};
    std::memcpy(stageHost_ + off, pending[i].src, pending[i].bytes);
    if (b16 > pending[i].bytes) std::memset(stageHost_ + off + pending[i].bytes, 0, b16 - pending[i].bytes);
    off += b16;
  }
  HIP_OK(hipMemcpyAsync(stageDev_.p, stageHost_, total, hipMemcpyHostToDevice, s));
  HIP_OK(hipEventRecord(stageEvt_, s));
  launchScatterStaged(stageDev_.p, (int)pending.size(), s);
}

void Window::pack(bool solveFollows) {
  quiesce();
  // the qualities of the last solve: a new solve replaces them before anybody can look; any other caller (inspection hooks,
  // prepare()) keeps them -- they are computed now, while that solve's tables are still intact
  if (solveFollows) qualityPending_ = false;
  else flushPendingQuality();
  const double tPack0 = nowSec();
  poseIds_.clear(); extIds_.clear(); sbIds_.clear(); lmIds_.clear(); factorIds_.clear();
  poseSlot_.clear(); extSlot_.clear(); sbSlot_.clear();
  for (const auto& kv : states_) {
    const State& s = kv.second;
    if (s.pose.exists) { poseSlot_[s.pose.id] = (int)poseIds_.size(); poseIds_.push_back(s.pose.id); }
    for (const StateInfo& e : s.ext)
      if (e.exists && !extSlot_.count(e.id)) { extSlot_[e.id] = (int)extIds_.size(); extIds_.push_back(e.id); }
    for (const StateInfo& b : s.sb)
      if (b.exists) { sbSlot_[b.id] = (int)sbIds_.size(); sbIds_.push_back(b.id); }
  }
  // Fixed blocks are never marginalised (Estimator.cpp:643-645, "we never eliminate fixed blocks"): when their frame
  // leaves the window they stay in the graph as constants -- e.g. the fixed extrinsics of the first frame, still tied
  // to the next frame's by a RelativePoseError, or listed (with no columns) in the marginalisation prior.  The
  // factors and the prior address them by slot like any other block.
  {
    std::vector<uint64_t> orphans;
    for (const auto& kv : blocks_) {
      const Block& b = kv.second;
      if (b.residuals.empty() && b.nObs == 0) continue;   // (a variable block outside every frame: added through the Map interface)
      const bool known = (b.kind == B_POSE) ? poseSlot_.count(b.id) : (b.kind == B_EXT ? extSlot_.count(b.id) : sbSlot_.count(b.id));
      if (!known) orphans.push_back(b.id);
    }
    std::sort(orphans.begin(), orphans.end());
    for (uint64_t id : orphans) {
      const Block& b = blocks_.at(id);
      if (b.kind == B_POSE) { poseSlot_[id] = (int)poseIds_.size(); poseIds_.push_back(id); }
      else if (b.kind == B_EXT) { extSlot_[id] = (int)extIds_.size(); extIds_.push_back(id); }
      else { sbSlot_[id] = (int)sbIds_.size(); sbIds_.push_back(id); }
    }
  }
  if (poseIds_.size() > 4095 || extIds_.size() > 4095) throw std::runtime_error("window too wide for the packed index");
  std::vector<double> hPose(poseIds_.size() * 7), hExt(std::max<size_t>(extIds_.size(), 1) * 7), hSb(sbIds_.size() * 9);
  std::vector<int> hPoseOff(poseIds_.size()), hExtOff(std::max<size_t>(extIds_.size(), 1), -1), hSbOff(sbIds_.size());
  redBlockIds_.clear(); redBlockOff_.clear();
  int d = 0;
  bool anyExtVar = false;
  for (size_t i = 0; i < poseIds_.size(); ++i) {
    const Block& b = blocks_.at(poseIds_[i]);
    std::memcpy(&hPose[7 * i], b.x, 7 * sizeof(double));
    if (b.fixed) hPoseOff[i] = -1;
    else { hPoseOff[i] = d; redBlockIds_.push_back(b.id); redBlockOff_.push_back(d); d += 6; }
  }
  // A graph whose blocks are all landmarks (or all constant): okvis_ceres/test/TestHomogeneousPointError.cpp builds one.  The
  // landmark elimination lives in the Schur kernels, which walk a landmark's residuals against a camera system -- so such a
  // graph gets ONE phantom pose block behind the real ones: identity pose, a unit PoseError at the identity (residual and
  // gradient zero, decoupled from everything), six rows in the reduced system.  It exists in the packed arrays only.
  bool anyVariable = d > 0;
  for (uint64_t id : extIds_) anyVariable |= !blocks_.at(id).fixed;
  for (uint64_t id : sbIds_) anyVariable |= !blocks_.at(id).fixed;
  const bool phantom = !anyVariable && (numObs_ + numLandmarkPriors_) > 0;
  const int phantomSlot = (int)poseIds_.size();
  if (phantom) {
    const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
    hPose.insert(hPose.end(), ident, ident + 7);
    hPoseOff.push_back(d);
    d += 6;
  }
  const int dCPose = d;
  for (size_t i = 0; i < extIds_.size(); ++i) {
    const Block& b = blocks_.at(extIds_[i]);
    std::memcpy(&hExt[7 * i], b.x, 7 * sizeof(double));
    if (b.fixed) hExtOff[i] = -1;
    else { hExtOff[i] = d; redBlockIds_.push_back(b.id); redBlockOff_.push_back(d); d += 6; anyExtVar = true; }
  }
  const int dC = d;
  // blocks on a reduced manifold: the reduced-system rows of their held tangent directions (kernels.hip k_lock_rows)
  std::vector<int> hLocked;
  for (size_t k = 0; k < redBlockIds_.size(); ++k) {
    const Block& b = blocks_.at(redBlockIds_[k]);
    for (int a = 0; a < 6; ++a)
      if ((b.lock >> a) & 1) hLocked.push_back(redBlockOff_[k] + a);
  }
  for (size_t i = 0; i < sbIds_.size(); ++i) {
    const Block& b = blocks_.at(sbIds_[i]);
    std::memcpy(&hSb[9 * i], b.x, 9 * sizeof(double));
    if (b.fixed) hSbOff[i] = -1;
    else { hSbOff[i] = d; redBlockIds_.push_back(b.id); redBlockOff_.push_back(d); d += 9; }
  }
  // landmarks + observations (landmark-major).  Two ways to the same arrays: the device-resident window takes this frame's
  // delta and rebuilds its CSR on the device (resident.hpp); the host path below walks the whole graph and uploads everything
  // (wide windows with their panel work lists, landmark priors, sharded mode, and the reference the tests hold the resident
  // path against).  Both list the landmarks with observations in HANDLE order, the observations in insertion order.
  const bool resident = useResident();
  residentUsed_ = resident;
  if (!resident) {
    syncLandmarks();         // the host graph becomes the authority again ...
    invalidateResident();    // ... and the device copy is rebuilt from it when the window next qualifies
  }
  size_t nLmObs = 0, nObs = 0;
  if (!resident)
    for (const Landmark* lm : lmByHandle_)
      if (lm && (!lm->obs.empty() || !lm->priors.empty())) { ++nLmObs; nObs += lm->obs.size() + 2 * lm->priors.size(); }
  std::vector<double> hLmPrior;   // 12 doubles per HomogeneousPointError: measurement xyz, sqrt information (row-major)
  std::vector<double> hLm(4 * nLmObs), hUv(2 * nObs), hW(nObs);
  std::vector<int> hLmPtr(resident ? 0 : nLmObs + 1), hObsLm(nObs);
  std::vector<uint32_t> hIdx(nObs);
  lmIds_.resize(nLmObs);
  struct SlotCache {
    const std::unordered_map<uint64_t, int>& map;
    std::vector<uint64_t> ids;
    std::vector<int> slots;
    explicit SlotCache(const std::unordered_map<uint64_t, int>& m) : map(m) {
      if (m.size() <= 64)
        for (const auto& kv : m) { ids.push_back(kv.first); slots.push_back(kv.second); }
    }
    int at(uint64_t id) const {
      for (size_t i = 0; i < ids.size(); ++i)
        if (ids[i] == id) return slots[i];
      return map.at(id);
    }
  };
  const SlotCache poseCache(poseSlot_), extCache(extSlot_);
  // landmark order of the CSR: handle order (creation order; id order when ids grow with time, as the reference's IdProvider
  // makes them); for wide windows sorted by visibility signature (below), so that a chunk of 16 consecutive landmarks
  // touches few 96-row panels of the camera matrix and the same tile rows inside them (k_schur_panels work list)
  std::vector<const Landmark*> lmOrder;
  if (!resident) {
    lmOrder.reserve(nLmObs);
    for (const Landmark* lm : lmByHandle_)
      if (lm && (!lm->obs.empty() || !lm->priors.empty())) lmOrder.push_back(lm);
    if (poseIds_.size() > (size_t)kResidentPoseCap) {
      // Wide windows (k_schur_panels): order by VISIBILITY SIGNATURE -- the set of 16-row tiles of the camera matrix a landmark's
      // observations write to (first tile, last tile, then the bit pattern) -- so that the 16 landmarks of a chunk hit the same
      // tile rows and the kernel's step masks drop whole products.  A product step (tile row, tile column, 4 columns of G) runs
      // when both tile rows hold something in those columns; modelled on the host (tools/panel_order_model.py) for the bench
      // window of configs[3]: executed / algorithmic MFMA flops 9.28 ordered by first pose (measured 9.3), 7.84 by (first, last)
      // pose, 6.30 by signature.  The rest is granularity: a landmark there sees 9 poses scattered over a span of 28 (its 55 rows
      // live in ~8 tiles of 16), which no order of the landmarks changes.
      struct Key { int first, last; uint64_t lo, hi; const Landmark* lm; };
      std::vector<Key> keyed;
      keyed.reserve(lmOrder.size());
      for (const Landmark* lm : lmOrder) {
        Key k{INT32_MAX, -1, 0, 0, lm};
        for (const Observation& ob : lm->obs) {
          const int off = hPoseOff[poseCache.at(ob.poseId)];
          if (off < 0) continue;
          for (int tr : {off >> 4, (off + 5) >> 4}) {
            k.first = std::min(k.first, tr); k.last = std::max(k.last, tr);
            if (tr < 64) k.lo |= 1ull << tr; else if (tr < 128) k.hi |= 1ull << (tr - 64);
          }
        }
        if (k.last < 0) k.first = 0;
        keyed.push_back(k);
      }
      std::stable_sort(keyed.begin(), keyed.end(), [](const Key& a, const Key& b) {
        if (a.first != b.first) return a.first < b.first;
        if (a.last != b.last) return a.last < b.last;
        if (a.hi != b.hi) return a.hi < b.hi;
        return a.lo < b.lo;
      });
      for (size_t i = 0; i < keyed.size(); ++i) lmOrder[i] = keyed[i].lm;
    }
    size_t slot = 0, o = 0;
    hLmPtr[0] = 0;
    for (const Landmark* lmp : lmOrder) {
      const Landmark& lm = *lmp;
      lmIds_[slot] = lm.id;
      std::memcpy(&hLm[4 * slot], lm.hp, 4 * sizeof(double));
      for (const Observation& ob : lm.obs) {
        hUv[2 * o] = ob.uv[0]; hUv[2 * o + 1] = ob.uv[1];
        // information = I * 64/size^2 ; sqrt information = its (scalar) Cholesky factor
        hW[o] = lm.fixed ? -obsWeight(ob.size) : obsWeight(ob.size);
        hIdx[o] = packObs(poseCache.at(ob.poseId), extCache.at(extIdOf(ob)), ob.cam);
        hObsLm[o] = (int)slot;
        ++o;
      }
      // a HomogeneousPointError = two pseudo-observations of its landmark (rows 0-1 and row 2 of S); their pose / extrinsics
      // Jacobians are written as zeros by the evaluation, so the slots they name (0, 0) only receive zeros
      for (const Landmark::Prior& pr : lm.priors) {
        const double k = (double)(hLmPrior.size() / 12);
        hLmPrior.insert(hLmPrior.end(), pr.meas, pr.meas + 3);
        hLmPrior.insert(hLmPrior.end(), pr.sqrtInfo, pr.sqrtInfo + 9);
        for (int part = 0; part < 2; ++part) {
          hUv[2 * o] = k; hUv[2 * o + 1] = (double)part;
          hW[o] = 1.0;
          hIdx[o] = packObs(0, 0, kPriorCam);
          hObsLm[o] = (int)slot;
          ++o;
        }
      }
      hLmPtr[++slot] = (int)o;
    }
  }
  const int L = resident ? (int)numLmObserved_ : (int)lmIds_.size(), N = resident ? (int)numObs_ : (int)hObsLm.size();
  // factors
  std::vector<DevFactor> hFac;
  hostFactors_.clear();
  std::vector<DevImu> hImu;
  std::vector<uint32_t> hImuT;
  std::vector<double> hImuM;
  auto blkKindSlot = [&](uint64_t id, int& kind, int& slot) {
    const Block& b = blocks_.at(id);
    kind = b.kind;
    slot = (b.kind == B_POSE) ? poseSlot_.at(id) : (b.kind == B_EXT ? extSlot_.at(id) : sbSlot_.at(id));
  };
  // sharded mode: the small factors are dealt to the ranks by creation number (factors are created in frame order: IMU factor
  // k -> k+1, the priors of a frame, its relative-extrinsics and sonar / depth terms); every rank still holds all states
  for (auto& kv : factors_) {
    Factor& f = kv.second;
    if (!ownsFactor(f)) continue;
    DevFactor df;
    std::memset(&df, 0, sizeof(df));
    df.kind = f.kind; df.nblk = f.nblk; df.m = f.m; df.imuIndex = -1;
    for (int b = 0; b < f.nblk; ++b) blkKindSlot(f.blocks[b], df.blkKind[b], df.blkSlot[b]);
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
    if (f.kind == F_HOST) {
      if (world_ > 1 || rcclComm_) throw std::runtime_error("svin_ba: host cost functions are not available in sharded mode");
      hostFactors_.push_back(std::make_pair((int)hFac.size(), f.id));
    }
    factorIds_.push_back(f.id);
    hFac.push_back(df);
  }
  hPoseOffKeep_ = hPoseOff; hExtOffKeep_ = hExtOff; hSbOffKeep_ = hSbOff;
  if (phantom) {
    DevFactor df;
    std::memset(&df, 0, sizeof(df));
    df.kind = F_POSE_PRIOR; df.nblk = 1; df.m = 6; df.imuIndex = -1;
    df.blkKind[0] = B_POSE; df.blkSlot[0] = phantomSlot;
    df.meas[6] = 1.0;
    for (int k = 0; k < 6; ++k) df.sqrtInfo[k * 6 + k] = 1.0;
    hFac.push_back(df);   // (not in factorIds_: the inspection hooks report the graph's factors)
  }
  const int F = (int)hFac.size();
  // prior
  std::vector<PriorBlock> hPb;
  int priorM = 0;
  if (hasPrior_) {
    priorM = priorM_;
    for (const PriorBlockHost& pb : priorBlocks_) {
      PriorBlock q;
      std::memset(&q, 0, sizeof(q));
      q.kind = pb.kind;
      q.slot = (pb.kind == B_POSE) ? poseSlot_.at(pb.id) : (pb.kind == B_EXT ? extSlot_.at(pb.id) : sbSlot_.at(pb.id));
      q.ord = pb.ord; q.mdim = pb.mdim;
      std::memcpy(q.lin, pb.lin, sizeof(q.lin));
      hPb.push_back(q);
    }
  }
  // Do the variable speed / bias blocks form a chain behind the kept rows -- every factor (of ANY rank: the all-reduced system
  // holds them all) and the prior tying two of them only ties neighbours in the order of the rows?  Then the wide-window solver
  // eliminates them ahead of its blocked Cholesky (kernels.hip, k_sb_factor ...).
  int sbChain = 0;
  {
    std::vector<int> chainPos(sbIds_.size(), -1);
    int n = 0;
    bool ok = true;
    for (size_t i = 0; i < sbIds_.size(); ++i)
      if (hSbOff[i] >= 0) { ok = ok && hSbOff[i] == dC + 9 * n; chainPos[i] = n++; }
    auto neighbours = [&](const int* pos, int cnt) {
      for (int x = 0; x < cnt; ++x)
        for (int y = x + 1; y < cnt; ++y) ok = ok && std::abs(pos[x] - pos[y]) == 1;
    };
    for (const auto& kv : factors_) {
      int pos[4], cnt = 0;
      for (int b = 0; b < kv.second.nblk; ++b) {
        const Block& blk = blocks_.at(kv.second.blocks[b]);
        if (blk.kind != B_POSE && blk.kind != B_EXT && !blk.fixed) pos[cnt++] = chainPos[sbSlot_.at(blk.id)];
      }
      neighbours(pos, cnt);
    }
    if (hasPrior_) {
      std::vector<int> pos;
      for (const PriorBlockHost& pb : priorBlocks_)
        if (pb.kind != B_POSE && pb.kind != B_EXT && chainPos[sbSlot_.at(pb.id)] >= 0) pos.push_back(chainPos[sbSlot_.at(pb.id)]);
      neighbours(pos.data(), (int)pos.size());
    }
    if (ok && d == dC + 9 * n) sbChain = n;
  }
  // ---- device allocation + upload
  const double tPack1 = nowSec();
  hipStream_t s = stream_;
  // every host array of the window goes into one pinned block behind a segment table: one DMA, one scatter kernel
  // (18 separate pageable copies cost ~70 us of enqueueing and ~70 us of draining per pack())
  std::vector<StagedCopy> pending;
  auto upload = [&](auto& buf, const auto& host, hipStream_t) {
    using T = typename std::remove_reference<decltype(host)>::type::value_type;
    buf.reserve(std::max<size_t>(host.size() + 16 / sizeof(T) + 1, 1));   // room for the 16-byte rounding of the copy
    if (!host.empty()) pending.push_back({host.data(), sizeof(T) * host.size(), buf.p});
  };
  upload(dPose_, hPose, s); upload(dExt_, hExt, s); upload(dSb_, hSb, s);
  dPoseC_.reserve(std::max<size_t>(hPose.size(), 1)); dExtC_.reserve(std::max<size_t>(hExt.size(), 1));
  dSbC_.reserve(std::max<size_t>(hSb.size(), 1)); dLmC_.reserve(std::max<size_t>((size_t)4 * L, 1));
  upload(dPoseOff_, hPoseOff, s); upload(dExtOff_, hExtOff, s); upload(dSbOff_, hSbOff, s);
  if (!hLocked.empty()) upload(dLockedRows_, hLocked, s);
  upload(dCams_, cameras_, s);
  ResidentArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  // this frame's slot of every pose / extrinsics block handle (the resident observation records name blocks by handle)
  std::vector<int> hPoseSlotOfH, hExtSlotOfH;
  if (resident) {
    hPoseSlotOfH.assign(std::max(nextBlockH_[B_POSE], 1), -1); hExtSlotOfH.assign(std::max(nextBlockH_[B_EXT], 1), -1);
    for (size_t i = 0; i < poseIds_.size(); ++i) hPoseSlotOfH[blocks_.at(poseIds_[i]).handle] = (int)i;
    for (size_t i = 0; i < extIds_.size(); ++i) hExtSlotOfH[blocks_.at(extIds_[i]).handle] = (int)i;
    upload(res_.poseSlotOfH, hPoseSlotOfH, s); upload(res_.extSlotOfH, hExtSlotOfH, s);
    dLm_.reserve(std::max<size_t>((size_t)4 * L, 1)); dObsIdx_.reserve(std::max<size_t>(N, 1));
  } else {
    upload(dLm_, hLm, s);
    upload(dLmPtr_, hLmPtr, s); upload(dObsLm_, hObsLm, s); upload(dObsUv_, hUv, s); upload(dObsW_, hW, s);
    upload(dObsIdx_, hIdx, s);
    upload(dLmPrior_, hLmPrior, s);
  }
  for (int k = 0; k < 2; ++k) { dLin_[k].reserve(std::max<size_t>((size_t)32 * N, 1)); dFacLin_[k].reserve(std::max(F, 1)); }
  upload(dFactors_, hFac, s); upload(dImus_, hImu, s); upload(dImuT_, hImuT, s); upload(dImuM_, hImuM, s);
  if (hasPrior_) {
    upload(dPriorBlk_, hPb, s);  // Ht / bp / c0 stay where the marginalisation kernels left them (margBuf_.bOut)
    dPriorScratch_.reserve((size_t)6 * priorM + 18 * hPb.size() + 16);
  }
  const int dpad = ((d + 15) / 16) * 16;
  // S and the camera-side vectors share one allocation: [S | gRed | gFull | hC | ...] is all-reduced as one message
  const int sS = ((std::max(d, 1) + 15) / 16) * 16;   // row stride of S: whole 128-byte lines per 16-column tile segment
  dS_.reserve((size_t)sS * sS + (size_t)12 * std::max(d, 1) + 64);   // sS rows as well: the solver reads whole tiles without clamping
  dLmVec_.reserve((size_t)(6 + 3 * 7 + 9) * std::max(L, 1));
  {
    const size_t dp64 = ((size_t)d + 63) / 64 * 64;  // multi-workgroup solver: (dp64 + 64) x dp64 matrix + 1/L_ii + diagonal factors
    dChol_.reserve(std::max<size_t>(solveReducedScratchDoubles(d, true), 1));
  }
  dPartial_.reserve((size_t)16 * 4096);
  dScal_.reserve(1);
  dQuality_.reserve(std::max(L, 1));
  const size_t slabSize = (size_t)dC * dC + 3 * dC;
  const bool useLds = slabSize * 8 + (size_t)4 * 64 * 34 * 8 <= 150 * 1024;
  int nSlabs = 1;
  // windows whose camera block fits 16 x 16 MFMA tiles (dC <= 254, e.g. 42 poses or 10 poses with per-frame extrinsics):
  // dense Gram-matrix Schur complement on MFMA
  const bool schurDense = dC > 0 && dC + 2 <= 256 && poseIds_.size() <= (size_t)kDensePoseCap && !optOn(kOptSchurPairwise);
  if (schurDense) {
    nSlabs = std::max(1, std::min(256, (L + 15) / 16));
    // SVIN_SLAB_CHUNKS=n: n chunks of 16 landmarks per workgroup and private slab (default 1 up to 256 workgroups)
    if (debugOption(kOptSlabChunks) > 1) nSlabs = std::max(1, std::min(nSlabs, ((L + 15) / 16 + debugOption(kOptSlabChunks) - 1) / debugOption(kOptSlabChunks)));
  }
  else if (useLds) nSlabs = std::max(1, std::min(256, (L + 7) / 8));
  // dense Schur with the A part on MFMA (variable extrinsics, or more than 8 tile rows): within every chunk of 16
  // landmarks the observations are visited pose by pose, so that a batch only touches a few tile rows (counting sort)
  std::vector<int> hObsOrder;
  const bool orderObs = schurDense && (anyExtVar || (dC + 2 + 15) / 16 > 8) && N > 0;
  if (orderObs && resident) dObsOrder_.reserve((size_t)N);   // counting sort per chunk on the device (k_window_rebuild, phase 4)
  if (orderObs && !resident) {
    hObsOrder.resize(N);
    std::vector<int> cnt;
    for (int l0 = 0; l0 < L; l0 += 16) {
      const int oBeg = hLmPtr[l0], oEnd = hLmPtr[std::min(L, l0 + 16)];
      cnt.assign(poseIds_.size() + 2, 0);
      for (int o = oBeg; o < oEnd; ++o) cnt[(hIdx[o] & 0xfff) + 1]++;
      for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
      for (int o = oBeg; o < oEnd; ++o) hObsOrder[oBeg + cnt[hIdx[o] & 0xfff]++] = o;
    }
    upload(dObsOrder_, hObsOrder, s);
  }
  // wide windows with fixed extrinsics: Gram-matrix Schur complement per pair of 96-row panels (k_schur_panels).
  // Work list: every chunk of 16 landmarks goes to all panel pairs (I >= J) inside the row range its observations touch.
  const bool schurPanels = !schurDense && !anyExtVar && dC > 0 && L > 0 && !optOn(kOptSchurPairwise);
  std::vector<int> hPanelWork, hPanelChunks, hPanelPairPtr;
  int nPanelBlocks = 0, nPanelPairs = 0;
  // Round 6: the block-pair form (k_schur_blocks) is what runs unless SVIN_PANELS_OLD keeps the tile form (k_schur_panels).  Its
  // SLOTS -- one per (landmark, distinct variable pose), ascending with the pose inside a landmark, each with the list of its
  // observations (two for a stereo pair) -- are structure, built here once per pack(); k_blocks_slots writes a 24-double record
  // per slot and build.  A pose block index has to fit 16 bits.
  const bool schurBlocks = schurPanels && !optOn(kOptPanelsOld) && dC / 6 <= kBlkMaxPoseBlocks;
  std::vector<uint32_t> hPairWords;
  std::vector<int> hSlotPtr, hSlotObsPtr, hSlotObs, hSlotLm, hEntries, hBatch, hWaveTab, hRecSlot;   // (staged uploads copy from these when the block is flushed: they live to the end of pack())
  std::vector<unsigned short> hSlotBlk;
  if (schurBlocks) {
    hSlotPtr.resize((size_t)L + 1); hSlotObs.reserve(N); hSlotBlk.reserve(N); hSlotObsPtr.reserve((size_t)N + 1);
    std::vector<std::pair<int, int>> seen;   // (pose block, observation) of one landmark
    for (int l = 0; l < L; ++l) {
      hSlotPtr[l] = (int)hSlotBlk.size();
      seen.clear();
      for (int o = hLmPtr[l]; o < hLmPtr[l + 1]; ++o) {
        const int off = hPoseOff[hIdx[o] & 0xfff];
        if (off >= 0) seen.emplace_back(off / 6, o);
      }
      std::stable_sort(seen.begin(), seen.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
      for (size_t k = 0; k < seen.size(); ++k) {
        if (k == 0 || seen[k].first != seen[k - 1].first) { hSlotBlk.push_back((unsigned short)seen[k].first); hSlotObsPtr.push_back((int)hSlotObs.size()); hSlotLm.push_back(l); }
        hSlotObs.push_back(seen[k].second);
      }
    }
    hSlotPtr[L] = (int)hSlotBlk.size();
    hSlotObsPtr.push_back((int)hSlotObs.size());
    upload(dSlotPtr_, hSlotPtr, s); upload(dSlotBlk_, hSlotBlk, s); upload(dSlotObsPtr_, hSlotObsPtr, s); upload(dSlotObs_, hSlotObs, s);
    upload(dSlotLm_, hSlotLm, s);
    dSlotRec_.reserve(std::max<size_t>(hSlotBlk.size() * kBlkRec, 1));
    dBlkPartial_.reserve(std::max<size_t>(((hSlotBlk.size() + kBlkSlotsPerWorkgroup - 1) / kBlkSlotsPerWorkgroup) * (size_t)(dC / 6) * 34, 1));
    // work list: per panel pair (I >= J; a panel is 16 pose blocks = 96 rows) the landmarks with slots in both panels, as ENTRIES
    // (first slot and count in either panel -- the slots of a panel are a run, they ascend with the pose), cut into workgroups of
    // pair words (below); the pairs in the order k_reduce_panel_slabs expects (panelPairPtr)
    const int nPan = (dC + 95) / 96;
    nPanelPairs = nPan * (nPan + 1) / 2;
    std::vector<std::vector<int>> lists(nPanelPairs);   // four ints per entry: first slot in I, in J, counts, landmark
    std::vector<int> runPanel, runFirst, runCount;
    for (int l = 0; l < L; ++l) {
      runPanel.clear(); runFirst.clear(); runCount.clear();
      for (int sl = hSlotPtr[l]; sl < hSlotPtr[l + 1]; ++sl) {
        const int pan = hSlotBlk[sl] / 16;
        if (runPanel.empty() || runPanel.back() != pan) { runPanel.push_back(pan); runFirst.push_back(sl); runCount.push_back(1); }
        else ++runCount.back();
      }
      for (size_t a = 0; a < runPanel.size(); ++a)
        for (size_t b = 0; b <= a; ++b) {
          std::vector<int>& li = lists[runPanel[a] * (runPanel[a] + 1) / 2 + runPanel[b]];
          li.insert(li.end(), {runFirst[a], runFirst[b], runCount[a] | (runCount[b] << 8), l});
        }
    }
    // ... and for k_schur_rows (kernels.hip), per panel pair: the sixteen block rows dealt to the kernel's eight waves (two each, by
    // their pair counts, heaviest first, per workgroup), the entries cut into workgroups by pair words and those into BATCHES (records
    // staged in LDS at a time: at most kBlkBatchRecs, and at most kBlkBatchWords pair words per wave), and per batch and wave the
    // PAIR WORDS (pairWord below) in the order the wave works through them, its first row's, then
    // its second row's, each sorted by A record (entry, slot in I); the run of an A record is padded to an even length and a row's
    // words to whole eights with pairs whose B operand is the zero record (the kernel takes the A record of words 2 j, 2 j + 1 from
    // word 2 j, and words 2 j, 2 j + 1 must name two accumulators).  A diagonal pair takes the blocks on and below the block diagonal.
    // Workgroups: cut by PAIR WORDS (an entry of a pair of different panels has 17 pairs on the bench window of configs[3], one of a
    // diagonal pair 12), so many that SVIN_BLK_ROUNDS (default 2) workgroups per place run one after the other -- two places per
    // CU.  (900 workgroups of 256 entries: 181 us; one round of equal entry counts: 233 us, the heaviest workgroup is the kernel.)
    // (pair word: twice the accumulator's number | byte offset of the B record in its LDS buffer << 8 | A record << 24 -- what the
    // kernel needs with the fewest scalar instructions; a staged record is 160 bytes)
    auto pairWord = [](int recA, int recB, int pb) { return (uint32_t)(2 * pb) | ((uint32_t)(recB * 160) << 8) | ((uint32_t)recA << 24); };
    static_assert(kBlkBatchRecs <= 256 && kBlkBatchRecs * 160 < 65536, "pair word fields");
    auto entryWords = [&](const int* en, bool dg) {
      const int nA = en[2] & 0xff, nB = en[2] >> 8;
      int wds = 0;
      for (int ka = 0; ka < nA; ++ka) wds += ((dg ? ka + 1 : nB) + 1) & ~1;
      return wds;
    };
    size_t wordsPerWg = 0;
    {
      size_t total = 0;
      for (int I = 0; I < nPan; ++I)
        for (int J = 0; J <= I; ++J) {
          const std::vector<int>& li = lists[I * (I + 1) / 2 + J];
          for (size_t e = 0; e < li.size() / 4; ++e) total += entryWords(&li[4 * e], I == J);
        }
      const int rounds = debugOption(kOptBlkRounds) > 0 ? debugOption(kOptBlkRounds) : 2;
      const size_t places = (size_t)std::max(1, rounds * 2 * deviceComputeUnits() - nPanelPairs);
      wordsPerWg = std::max<size_t>(kBlkMinWordsPerBlock, (total + places - 1) / places);
    }
    hPanelPairPtr.push_back(0);
    size_t balWgMax = 0, balWgAll = 0, balAll = 0, balMax = 0;   // pair words of the busiest wave / of all waves: per workgroup, per batch (a barrier pair per batch)
    for (int I = 0; I < nPan; ++I)
      for (int J = 0; J <= I; ++J) {
        const std::vector<int>& li = lists[I * (I + 1) / 2 + J];
        const size_t nEnt = li.size() / 4;
        const bool dg = I == J;
        for (size_t k = 0; k < nEnt;) {
          size_t kEnd = k, wgWords = 0;
          while (kEnd < nEnt && wgWords < wordsPerWg) wgWords += entryWords(&li[4 * kEnd++], dg);
          // The workgroup's block rows dealt to the sixteen accumulator sets of its eight waves (two each).  Rows no landmark of the
          // list touches get none; the sets that are left go to the heaviest rows as a SECOND set (the row's runs are then shared
          // between two waves -- by the lighter wave of the moment, below -- and k_schur_rows adds both sets into the slab image:
          // two terms, so the sum does not depend on their order).  Sets heaviest first, to the wave with the least so far.
          // (round 6, measured on the bench window: one set per row and rows dealt by load left the busiest wave of a batch with
          //  1.59 x the mean number of pair words and the busiest wave of a workgroup with 1.24 x; rows r, r + 8 to wave r: 1.71;
          //  entries re-ordered round robin by the wave they load most: 1.57)
          long rowLoad[16] = {0};
          for (size_t e = k; e < kEnd; ++e) {
            const int fa = li[4 * e], nA = li[4 * e + 2] & 0xff, nB = li[4 * e + 2] >> 8;
            for (int ka = 0; ka < nA; ++ka) rowLoad[hSlotBlk[fa + ka] - 16 * I] += ((dg ? ka + 1 : nB) + 1) & ~1;
          }
          int nOwn[16] = {0}, ownerWave[16][2], ownerSel[16][2], ownRows[kBlkWaves][2];
          long waveLoad[kBlkWaves] = {0};
          for (int wvv = 0; wvv < kBlkWaves; ++wvv) ownRows[wvv][0] = ownRows[wvv][1] = 255;
          {
            int mult[16], sets = 0;
            for (int r = 0; r < 16; ++r) { mult[r] = rowLoad[r] > 0 ? 1 : 0; sets += mult[r]; }
            const bool split = !optOn(kOptNoRowSplit);
            while (split && sets < 2 * kBlkWaves) {
              int best = -1;
              for (int r = 0; r < 16; ++r)
                if (mult[r] == 1 && rowLoad[r] >= 16 && (best < 0 || rowLoad[r] > rowLoad[best])) best = r;
              if (best < 0) break;
              mult[best] = 2; ++sets;
            }
            struct Unit { int row; long load; };
            std::vector<Unit> units;
            for (int r = 0; r < 16; ++r)
              for (int c = 0; c < mult[r]; ++c) units.push_back(Unit{r, rowLoad[r] / mult[r]});
            std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.load > b.load; });
            for (const Unit& u : units) {
              int best = -1;
              for (int wvv = 0; wvv < kBlkWaves; ++wvv) {
                if (ownRows[wvv][1] != 255) continue;
                if (nOwn[u.row] == 1 && ownerWave[u.row][0] == wvv) continue;   // (the two sets of a row: two waves)
                if (best < 0 || waveLoad[wvv] < waveLoad[best]) best = wvv;
              }
              if (best < 0) continue;   // (only the second set of a row can be left over: the row keeps its first)
              const int sel = ownRows[best][0] == 255 ? 0 : 1;
              ownRows[best][sel] = u.row;
              ownerWave[u.row][nOwn[u.row]] = best; ownerSel[u.row][nOwn[u.row]] = sel; ++nOwn[u.row];
              waveLoad[best] += u.load;
            }
          }
          {
            long mx = 0, sum = 0;
            for (int wvv = 0; wvv < kBlkWaves; ++wvv) { mx = std::max(mx, waveLoad[wvv]); sum += waveLoad[wvv]; }
            balWgMax += (size_t)mx; balWgAll += (size_t)sum;
          }
          int ownWords[4] = {0, 0, 0, 0};
          for (int wvv = 0; wvv < kBlkWaves; ++wvv)
            ownWords[wvv >> 1] |= (ownRows[wvv][0] | (ownRows[wvv][1] << 8)) << (16 * (wvv & 1));
          const int firstBatch = (int)(hBatch.size() / 2);
          size_t e = k;
          while (e < kEnd) {
            std::vector<uint32_t> words[kBlkWaves][2];   // per wave and owned row
            const int firstRec = (int)hRecSlot.size();
            int recs = 0;
            for (; e < kEnd; ++e) {
              const int fa = li[4 * e], fb = li[4 * e + 1], nA = li[4 * e + 2] & 0xff, nB = li[4 * e + 2] >> 8;
              const int need = nA + (dg ? 0 : nB);
              if (recs + need > kBlkBatchRecs - 1) break;
              int add[kBlkWaves] = {0};   // (every run of an A record is padded to an even number of words)
              int pick[64];               // which of its row's sets the run of slot ka goes to: the wave with fewer words in this batch
              for (int ka = 0; ka < nA; ++ka) {
                const int row = hSlotBlk[fa + ka] - 16 * I;
                int c = 0;
                if (nOwn[row] == 2) {
                  const int w0 = ownerWave[row][0], w1 = ownerWave[row][1];
                  const size_t l0 = words[w0][0].size() + words[w0][1].size() + (size_t)add[w0], l1 = words[w1][0].size() + words[w1][1].size() + (size_t)add[w1];
                  c = l1 < l0 ? 1 : 0;
                }
                pick[ka] = c;
                add[ownerWave[row][c]] += ((dg ? ka + 1 : nB) + 1) & ~1;
              }
              bool fits = true;
              for (int wvv = 0; wvv < kBlkWaves; ++wvv) fits = fits && (int)(words[wvv][0].size() + words[wvv][1].size()) + add[wvv] <= kBlkBatchWords - 12;
              if (!fits) break;
              const int recA0 = recs, recB0 = dg ? recs : recs + nA;
              for (int ka = 0; ka < nA; ++ka) hRecSlot.push_back(fa + ka);
              if (!dg) for (int kb = 0; kb < nB; ++kb) hRecSlot.push_back(fb + kb);
              recs += need;
              for (int ka = 0; ka < nA; ++ka) {
                const int row = hSlotBlk[fa + ka] - 16 * I;
                std::vector<uint32_t>& wl = words[ownerWave[row][pick[ka]]][ownerSel[row][pick[ka]]];
                const int cnt = dg ? ka + 1 : nB;
                int pb = 0;
                for (int kb = 0; kb < cnt; ++kb) {
                  pb = hSlotBlk[fb + kb] - 16 * J;
                  wl.push_back(pairWord(recA0 + ka, recB0 + kb, pb));
                }
                // (padding word of the run: the zero record as B, an accumulator other than its partner's)
                if (cnt & 1) wl.push_back(pairWord(recA0 + ka, kBlkBatchRecs - 1, (pb + 1) & 15));
              }
            }
            if (recs == 0) throw std::logic_error("k_schur_rows work list: an entry does not fit a batch");
            hBatch.insert(hBatch.end(), {firstRec, recs});
            {
              size_t mx = 0;
              for (int wvv = 0; wvv < kBlkWaves; ++wvv) { const size_t n = words[wvv][0].size() + words[wvv][1].size(); balAll += n; mx = std::max(mx, n); }
              balMax += mx;
            }
            for (int wvv = 0; wvv < kBlkWaves; ++wvv) {
              for (int sel = 0; sel < 2; ++sel)   // (a row's words in eights: padding words in twos -- both operands the zero record, two accumulators)
                while (words[wvv][sel].size() % 8) {
                  words[wvv][sel].push_back(pairWord(kBlkBatchRecs - 1, kBlkBatchRecs - 1, 0));
                  words[wvv][sel].push_back(pairWord(kBlkBatchRecs - 1, kBlkBatchRecs - 1, 1));
                }
              hWaveTab.insert(hWaveTab.end(), {(int)hPairWords.size(), (int)words[wvv][0].size(), (int)words[wvv][1].size(), 0});
              hPairWords.insert(hPairWords.end(), words[wvv][0].begin(), words[wvv][0].end());
              hPairWords.insert(hPairWords.end(), words[wvv][1].begin(), words[wvv][1].end());
            }
          }
          hPanelWork.insert(hPanelWork.end(), {I, J, firstBatch, (int)(hBatch.size() / 2) - firstBatch});
          hEntries.insert(hEntries.end(), ownWords, ownWords + 4);   // (blkOwn: one int4 per workgroup)
          ++nPanelBlocks;
          k = kEnd;
        }
        hPanelPairPtr.push_back(nPanelBlocks);
      }
    if (optOn(kOptPackTiming))
      std::printf("[svin_ba pack] k_schur_rows work list: %d workgroups, %zu batches, %zu pair words; the busiest wave of a batch has %.2f x the mean, of a workgroup %.2f x\n",
                  nPanelBlocks, hBatch.size() / 2, balAll, balAll ? (double)kBlkWaves * (double)balMax / (double)balAll : 0.0,
                  balWgAll ? (double)kBlkWaves * (double)balWgMax / (double)balWgAll : 0.0);
    hPairWords.resize(hPairWords.size() + 128, 0u);   // (a wave requests its words in 64s)
    hBatch.resize(hBatch.size() + 2 * 3, 0); hWaveTab.resize(hWaveTab.size() + (size_t)4 * kBlkWaves * 3, 0);   // (the kernel reads descriptors three batches ahead, unconditionally)
    upload(dBlkPairs_, hPairWords, s); upload(dBlkBatch_, hBatch, s); upload(dBlkWaveTab_, hWaveTab, s); upload(dBlkRecSlot_, hRecSlot, s);
    upload(dPanelWork_, hPanelWork, s); upload(dPanelChunks_, hEntries, s); upload(dPanelPairPtr_, hPanelPairPtr, s);
    dSlabs_.reserve(std::max<size_t>((size_t)nPanelBlocks * (96 * 96 + 3 * 96), 1));
  }
  if (schurPanels && !schurBlocks) {
    constexpr int kRows = 96, kChunk = 16, kPerBlock = kPanelChunksPerBlock;
    const int nPan = (dC + kRows - 1) / kRows;
    nPanelPairs = nPan * (nPan + 1) / 2;
    std::vector<std::vector<int>> lists(nPanelPairs);
    const int nChunks = (L + kChunk - 1) / kChunk;
    for (int c = 0; c < nChunks; ++c) {
      int lo = INT32_MAX, hi = -1;
      const int o0 = hLmPtr[c * kChunk], o1 = hLmPtr[std::min(L, (c + 1) * kChunk)];
      for (int o = o0; o < o1; ++o) {
        const int off = hPoseOff[hIdx[o] & 0xfff];
        if (off < 0) continue;
        lo = std::min(lo, off); hi = std::max(hi, off);
      }
      // a chunk without variable poses still has to produce V^-1, b, htil for its landmarks: give it to pair (0, 0)
      const int pLo = hi < 0 ? 0 : lo / kRows, pHi = hi < 0 ? 0 : hi / kRows;
      for (int I = pLo; I <= pHi; ++I)
        for (int J = pLo; J <= I; ++J) lists[I * (I + 1) / 2 + J].push_back(c);
    }
    hPanelPairPtr.push_back(0);
    for (int I = 0; I < nPan; ++I)
      for (int J = 0; J <= I; ++J) {
        const std::vector<int>& li = lists[I * (I + 1) / 2 + J];
        for (size_t k = 0; k < li.size(); k += kPerBlock) {
          const int cnt = (int)std::min<size_t>(kPerBlock, li.size() - k);
          hPanelWork.insert(hPanelWork.end(), {I, J, (int)hPanelChunks.size(), cnt});
          hPanelChunks.insert(hPanelChunks.end(), li.begin() + k, li.begin() + k + cnt);
          ++nPanelBlocks;
        }
        hPanelPairPtr.push_back(nPanelBlocks);
      }
    upload(dPanelWork_, hPanelWork, s); upload(dPanelChunks_, hPanelChunks, s); upload(dPanelPairPtr_, hPanelPairPtr, s);
    dSlabs_.reserve(std::max<size_t>((size_t)nPanelBlocks * (kRows * kRows + 3 * kRows), 1));
  } else if (!schurPanels) {
    dSlabs_.reserve(std::max<size_t>(slabSize * nSlabs, 1));
  }

  // the accumulators start clear (the trust-region loop never launches k_zero_build: k_post_solve re-clears them); the clears
  // ride in the scatter launch of the staged block
  const int sSq = ((std::max(d, 1) + 15) / 16) * 16;
  pending.push_back({nullptr, sizeof(double) * ((size_t)sSq * sSq + (size_t)12 * std::max(d, 1)), dS_.p});
  pending.push_back({nullptr, sizeof(SolverScalars), dScal_.p});
  pending.push_back({nullptr, sizeof(double) * 16 * 4096, dPartial_.p});
  if (hasPrior_) pending.push_back({nullptr, sizeof(double) * (6 * (size_t)priorM + 18 * hPb.size()), dPriorScratch_.p});
  if (schurPanels && resident) throw std::logic_error("resident window needs a panel work list");
  if (resident) flushResident(s, orderObs, pending, ra);   // stages the delta; the rebuild kernel follows the scatter
  flushStaged(pending, s);
  if (solveFollows && resident) HIP_OK(hipEventRecord(evUploaded_, s));   // (the side stream of the early IMU pre-integration waits for the tables)
  if (resident) {
    addLog_.clear(); remLog_.clear(); setLog_.clear();   // (copied into the staged block by flushStaged)
    ++epoch_;
    ra.nPoseSlots = (int)poseIds_.size();
    ra.obsIdx = dObsIdx_.p; ra.lm = dLm_.p; ra.obsOrder = dObsOrder_.p;
    ra.poseSlotOfH = res_.poseSlotOfH.p; ra.extSlotOfH = res_.extSlotOfH.p;
    launchWindowRebuild(ra, s);
  }

  DeviceProblem& p = prob_;
  std::memset(&p, 0, sizeof(p));
  p.nPose = (int)(hPose.size() / 7); p.nExt = (int)std::max<size_t>(extIds_.size(), 1); p.nSb = (int)sbIds_.size();
  p.L = L; p.N = N; p.F = F; p.nImu = (int)hImu.size(); p.d = d; p.dC = dC; p.nCam = (int)cameras_.size();
  p.priorM = priorM; p.priorBlocks = (int)hPb.size(); p.anyExtVariable = anyExtVar ? 1 : 0;
  p.ownsCamera = (world_ <= 1 || rank_ == 0) ? 1 : 0;
  p.rank = (world_ <= 1 && rcclComm_ && optOn(kOptForceDistributed)) ? -1 : rank_;   // -1: one-rank communicator exercising the sharded path
  p.world = world_;
  if (!p.ownsCamera) p.priorM = 0;   // the prior is evaluated and accumulated on one rank only
  p.pose = dPose_.p; p.ext = dExt_.p; p.sb = dSb_.p; p.lm = dLm_.p;
  p.obsOrder = orderObs ? dObsOrder_.p : nullptr;
  p.dCPose = dCPose;
  p.lockedRows = hLocked.empty() ? nullptr : dLockedRows_.p; p.nLocked = (int)hLocked.size();
  p.nHostFactors = (int)hostFactors_.size();
  p.poseC = dPoseC_.p; p.extC = dExtC_.p; p.sbC = dSbC_.p; p.lmC = dLmC_.p;
  p.poseOff = dPoseOff_.p; p.extOff = dExtOff_.p; p.sbOff = dSbOff_.p;
  p.cams = dCams_.p;
  p.schurDense = schurDense ? 1 : 0;
  p.schurPanels = schurPanels ? 1 : 0; p.nPanelBlocks = nPanelBlocks; p.nPanelPairs = nPanelPairs;
  p.schurBlocks = schurBlocks ? 1 : 0; p.nSlots = (int)hSlotBlk.size();
  p.slotPtr = dSlotPtr_.p; p.slotBlk = dSlotBlk_.p; p.slotObsPtr = dSlotObsPtr_.p; p.slotObs = dSlotObs_.p; p.slotLm = dSlotLm_.p; p.slotRec = dSlotRec_.p;
  p.blkOwn = reinterpret_cast<const int4*>(dPanelChunks_.p); p.blkPartial = dBlkPartial_.p; p.blkPairs = dBlkPairs_.p;
  p.blkBatch = reinterpret_cast<const int2*>(dBlkBatch_.p); p.blkWaveTab = reinterpret_cast<const int4*>(dBlkWaveTab_.p); p.blkRecSlot = dBlkRecSlot_.p;
  p.panelWork = reinterpret_cast<const int4*>(dPanelWork_.p); p.panelChunks = dPanelChunks_.p; p.panelPairPtr = dPanelPairPtr_.p;
  p.lmPtr = dLmPtr_.p; p.obsUv = dObsUv_.p; p.obsW = dObsW_.p; p.obsIdx = dObsIdx_.p; p.obsLm = dObsLm_.p;
  if (resident) { p.lmPtr = res_.lmPtr[res_.cur].p; p.obsUv = res_.uv[res_.cur].p; p.obsW = res_.w[res_.cur].p; p.obsLm = res_.obsLm[res_.cur].p; }
  p.lmPrior = dLmPrior_.p;
  curSet_ = 0;
  auto setLin = [&](int set, double*& r, double*& Jp, double*& Jl, double*& Je) {
    double* base = dLin_[set].p;
    r = base; Jp = base + (size_t)2 * N; Jl = base + (size_t)14 * N; Je = base + (size_t)20 * N;
  };
  setLin(0, p.rCur, p.JpCur, p.JlCur, p.JeCur);
  setLin(1, p.rCand, p.JpCand, p.JlCand, p.JeCand);
  p.factors = dFactors_.p; p.linCur = dFacLin_[0].p; p.linCand = dFacLin_[1].p;
  p.imus = dImus_.p; p.imuT = dImuT_.p; p.imuMeas = dImuM_.p;
  if (hasPrior_) {
    {
      const size_t n2 = (size_t)priorM * priorM;
      const double* out = margBuf_.bOut.p;  // k_marg_final: G | Q | J | Ht | e0 | bp | scal
      p.priorH = const_cast<double*>(out) + 3 * n2; p.priorBp = const_cast<double*>(out) + 4 * n2 + priorM;
      p.priorC0 = out + 4 * n2 + 2 * priorM; p.priorBlk = dPriorBlk_.p;
    }
    double* ps = dPriorScratch_.p;
    p.priorDchi = ps; p.priorGrad = ps + priorM; p.priorDchiC = ps + 2 * priorM; p.priorGradC = ps + 3 * priorM;
    p.priorMv = ps + 4 * priorM; p.priorMy = ps + 5 * priorM;
    p.priorM3 = ps + 6 * priorM; p.priorM3C = p.priorM3 + 9 * hPb.size();
  }
  const int dd = std::max(d, 1);
  // accumulators start clear: the trust-region loop never launches k_zero_build (k_post_solve re-clears them)

  p.S = dS_.p;
  p.ldS = sS;
  p.sPadded = 1;
  double* vecBase = dS_.p + (size_t)sS * sS;
  p.gRed = vecBase; p.gFull = vecBase + dd; p.hC = vecBase + 2 * dd; p.htilC = vecBase + 3 * dd;
  p.scaleC = vecBase + 4 * dd; p.yC = vecBase + 5 * dd; p.deltaC = vecBase + 6 * dd; p.vC = vecBase + 7 * dd;
  const size_t LL = std::max(L, 1);
  p.Vinv = dLmVec_.p; p.bl = dLmVec_.p + 6 * LL; p.hL = dLmVec_.p + 9 * LL; p.scaleL = dLmVec_.p + 12 * LL;
  p.yL = dLmVec_.p + 15 * LL; p.deltaL = dLmVec_.p + 18 * LL; p.vL = dLmVec_.p + 21 * LL;
  p.lmFactor = dLmVec_.p + 27 * LL;
  p.sbChain = sbChain;
  p.slabs = dSlabs_.p; p.nSlabs = nSlabs;
  p.cholL = dChol_.p;
  p.scal = dScal_.p;
  p.partial = dPartial_.p;
  p.tickets = reinterpret_cast<unsigned int*>(dPartial_.p + (size_t)14 * 4096);  // zeroed with the partials
  static_assert(sizeof(SolverScalars) % 16 == 0, "cleared in 16-byte words by the scatter kernel");
  // optimize() on the resident path: a factor that still has to be pre-integrated (the frame's new ImuError, redo_ = true,
  // ImuError.cpp:739) is evaluated once on a side stream while the main stream rebuilds the observation table -- the two
  // do not depend on each other, and the ~40 us integration chain is otherwise the first thing the solve waits for.  The
  // results of this evaluation are discarded (the solve's first evaluation repeats it, now without the integration); only
  // the pre-integration state stays.  Not in prepare(): there the upload is outside the measured region and the solve is not.
  const bool noEarlyImu = optOn(kOptNoEarlyImu);
  if (solveFollows && resident && !noEarlyImu && F > 0) {
    bool anyRedo = false;
    for (const DevImu& im : hImu) anyRedo |= im.redo != 0;
    if (anyRedo) {
      HIP_OK(hipStreamWaitEvent(stream2_, evUploaded_, 0));
      launchEvalFactors(p, false, stream2_, false);
      HIP_OK(hipEventRecord(evImuReady_, stream2_));
      HIP_OK(hipStreamWaitEvent(s, evImuReady_, 0));
    }
  }
  if (optOn(kOptPackTiming)) {
    const double tPack2 = nowSec();
    HIP_OK(hipStreamSynchronize(s));
    std::printf("[pack] host graph -> arrays %.1f us, allocation + enqueue %.1f us, drain %.1f us\n", 1e6 * (tPack1 - tPack0),
                1e6 * (tPack2 - tPack1), 1e6 * (nowSec() - tPack2));
  }
}

void Window::downloadStates() {
  quiesce();
  hipStream_t s = stream_;
  const DeviceProblem& p = prob_;
  const bool resident = residentUsed_;   // landmark points and qualities stay on the device, keyed by handle (syncLandmarks fetches)
  std::vector<double> hPose(poseIds_.size() * 7), hExt(std::max<size_t>(extIds_.size(), 1) * 7), hSb(sbIds_.size() * 9),
      hLm(resident ? 0 : (size_t)p.L * 4), hQ(resident ? 0 : p.L);
  std::vector<DevImu> hImu(p.nImu);
  if (p.L > 0 && !resident) launchLandmarkQuality(p, dQuality_.p, s);
  const bool noZeroCopy = optOn(kOptNoZeroCopyStates);   // A/B switch
  const bool zeroCopy = resident && !noZeroCopy;   // the states arrive through a host-mapped block written by k_window_finish
  if (resident && res_.H > 0) {
    if (!zeroCopy) launchWindowStoreLandmarks(res_.H, res_.slotOfH[res_.cur].p, p.lm, nullptr, res_.lmHp.p, res_.qualH.p, s);
    lmStale_ = true;
    qualityPending_ = true;
    qualityProb_ = p;
  }
  {  // one gather kernel + one DMA into the pinned block instead of six read-backs
    GatherArgs ga;
    std::memset(&ga, 0, sizeof(ga));
    struct Out { void* dst; size_t bytes; };
    Out outs[8];
    size_t off = 0;
    auto add = [&](const void* src, void* dst, size_t bytes) {
      if (bytes == 0) return;
      const size_t b16 = (bytes + 15) / 16 * 16;
      ga.src[ga.n] = src; ga.off[ga.n] = off; ga.bytes[ga.n] = b16;
      outs[ga.n] = Out{dst, bytes};
      ++ga.n;
      off += b16;
    };
    add(p.pose, hPose.data(), sizeof(double) * hPose.size());
    if (!extIds_.empty()) add(p.ext, hExt.data(), sizeof(double) * extIds_.size() * 7);
    add(p.sb, hSb.data(), sizeof(double) * hSb.size());
    if (p.L > 0 && !resident) {
      add(p.lm, hLm.data(), sizeof(double) * hLm.size());
      add(dQuality_.p, hQ.data(), sizeof(double) * p.L);
    }
    if (p.nImu > 0) add(p.imus, hImu.data(), sizeof(DevImu) * p.nImu);
    if (zeroCopy) {
      if (off + 64 > statesHostCap_) {
        HIP_OK(hipStreamSynchronize(s));
        if (statesHost_) (void)hipHostFree(statesHost_);
        statesHostCap_ = std::max<size_t>(2 * (off + 64), 1 << 18);
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&statesHost_), statesHostCap_, hipHostMallocMapped));
        std::memset(statesHost_, 0, 64);
        HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&statesHostDev_), statesHost_, 0));
        if (!finishTicket_.p) { finishTicket_.reserve(4); HIP_OK(hipMemsetAsync(finishTicket_.p, 0, 16, s)); }
      }
      FinishArgs fa;
      std::memset(&fa, 0, sizeof(fa));
      fa.H = res_.H; fa.nLmBlocks = (res_.H + 255) / 256;
      fa.slotOfH = res_.slotOfH[res_.cur].p; fa.lm = p.lm; fa.lmHp = res_.lmHp.p;
      fa.ga = ga;
      fa.hostBlock = statesHostDev_ + 64;   // (the first 64 bytes hold the sequence number)
      fa.hostSeq = reinterpret_cast<unsigned long long*>(statesHostDev_);
      fa.seq = ++statesSeq_;
      fa.ticket = reinterpret_cast<unsigned int*>(finishTicket_.p);
      launchWindowFinish(fa, s);
      volatile unsigned long long* seq = reinterpret_cast<volatile unsigned long long*>(statesHost_);
      const double tSpin = nowSec();
      bool ok = false;
      for (unsigned long long spins = 0;; ++spins) {
        if (*seq == statesSeq_) { ok = true; break; }
        if ((spins & 1023) == 1023 && nowSec() - tSpin > 2.0) break;
      }
      if (!ok) HIP_OK(hipStreamSynchronize(s));
      std::atomic_thread_fence(std::memory_order_acquire);
      for (int i = 0; i < ga.n; ++i) std::memcpy(outs[i].dst, statesHost_ + 64 + ga.off[i], outs[i].bytes);
    } else if (ga.n > 0) {
      if (stageEvt_) HIP_OK(hipEventSynchronize(stageEvt_));
      if (off > stageHostCap_) {
        if (stageHost_) (void)hipHostFree(stageHost_);
        stageHostCap_ = std::max<size_t>(2 * off, 1 << 20);
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&stageHost_), stageHostCap_, hipHostMallocDefault));
      }
      stageDev_.reserve(std::max<size_t>(off, 16));
      launchGatherStaged(stageDev_.p, ga, s);
      HIP_OK(hipMemcpyAsync(stageHost_, stageDev_.p, off, hipMemcpyDeviceToHost, s));
    }
    if (!zeroCopy) {
      HIP_OK(hipStreamSynchronize(s));
      for (int i = 0; i < ga.n; ++i) std::memcpy(outs[i].dst, stageHost_ + ga.off[i], outs[i].bytes);
    }
  }
  for (size_t i = 0; i < poseIds_.size(); ++i) std::memcpy(blocks_.at(poseIds_[i]).x, &hPose[7 * i], 7 * sizeof(double));
  for (size_t i = 0; i < extIds_.size(); ++i) std::memcpy(blocks_.at(extIds_[i]).x, &hExt[7 * i], 7 * sizeof(double));
  for (size_t i = 0; i < sbIds_.size(); ++i) std::memcpy(blocks_.at(sbIds_[i]).x, &hSb[9 * i], 9 * sizeof(double));
  if (!resident) {
    for (size_t i = 0; i < lmIds_.size(); ++i) {
      Landmark& lm = landmarks_.at(lmIds_[i]);
      std::memcpy(lm.hp, &hLm[4 * i], 4 * sizeof(double));
      lm.quality = hQ[i];
    }
    // Estimator::optimize also sets quality of unobserved landmarks: getLhs yields H = 0 -> quality 0 (:910-913)
    for (auto& kv : landmarks_)
      if (kv.second.obs.empty()) kv.second.quality = 0.0;
  } else {
    checkResidentStatus();   // (k_window_store_landmarks does the same per handle on the device)
  }
  int k = 0;
  for (uint64_t fid : factorIds_) {   // the factors this rank packed, in pack() order
    Factor& f = factors_.at(fid);
    if (f.kind == F_IMU) f.imu = hImu[k++];
  }
  // sharded: the IMU factors another rank evaluated were re-integrated (or not) over there; whoever evaluates them here next
  // (rank 0's marginalisation job, a window that goes back to one GPU) starts from a fresh pre-integration
  if (world_ > 1)
    for (auto& kv : factors_)
      if (kv.second.kind == F_IMU && !ownsFactor(kv.second)) kv.second.imu.redo = 1;
}

void Window::evaluateAll(bool cand, hipStream_t s) {
  evaluateHostFactors(cand, s);
  prob_.mailbox = distNative_ ? nullptr : mailboxDev_;   // sharded: published after the all-reduce (launchPublishScalars)
  prob_.mailboxSeq = ++mailboxSeq_;
  const int who = costSummedBy(prob_);
  if (canFuseEvaluation(prob_) && !optOn(kOptSplitEval)) {
    // one launch: the reprojection blocks and the prior run next to the (much longer) IMU factor blocks
    launchEvalAll(prob_, cand, true, s);  // factor, reprojection and prior blocks; the last one sums the cost
    return;
  }
  launchEvalReproj(prob_, cand, true, s);
  launchEvalFactors(prob_, cand, s, who == 1);
  launchEvalPrior(prob_, cand, s, who == 2);
  if (who == 0) launchCost(prob_, s);
}

SolverScalars Window::readScalars() {
  SolverScalars sc;
  if (mailbox_ && (world_ <= 1 || distNative_)) {
    // wait for the sequence number of the last evaluateAll(); bounded spin, then fall back to a real synchronise
    volatile unsigned long long* seq = &mailbox_->seq;
    const double tSpin = nowSec();
    bool ok = false;
    for (unsigned long long spins = 0;; ++spins) {
      if (*seq == mailboxSeq_) { ok = true; break; }
      if ((spins & 1023) == 1023 && nowSec() - tSpin > 2.0) break;
    }
    if (ok) {
      std::atomic_thread_fence(std::memory_order_acquire);
      std::memcpy(&sc, const_cast<SolverScalars*>(&mailbox_->scal), sizeof(sc));
      return sc;
    }
  }
  HIP_OK(hipMemcpyAsync(&sc, prob_.scal, sizeof(sc), hipMemcpyDeviceToHost, stream_));
  HIP_OK(hipStreamSynchronize(stream_));
  return sc;
}

// ------------------------------------------------------------------------------------------ trust-region loop
// Ceres 2.2 TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG) semantics with the options
// Estimator::optimize sets (:878-890): Jacobi scaling, monotonic steps, min_relative_decrease 1e-3,
// initial radius 1e4.  All linear algebra and all residual evaluation run on the device; the host only
// takes the accept/reject decision from one SolverScalars read-back per iteration.
void Window::solve(size_t numIter, bool verbose) {
  const double tStart = nowSec();
  DeviceProblem& p = prob_;
  hipStream_t s = stream_;
  summary_.iterations = 0; summary_.num_successful_steps = 0; summary_.termination = 1;
  if (p.d + 3 * p.L == 0) { summary_.termination = 0; summary_.initial_cost = summary_.final_cost = 0; return; }
  // landmark-sharded mode: partial sums are all-reduced at three points per iteration
  const bool forceDist = optOn(kOptForceDistributed);   // single-rank RCCL: exercises the sharded code path on one GPU
  const bool dist = world_ > 1 || (forceDist && rcclComm_);
  auto AR = [&](void* ptr, size_t n, int op) {
    if (!dist) return;
    if (rcclComm_) {   // native: enqueued on the solver's stream, in place
      rcclCheck(rccl().allReduce(ptr, ptr, n, ncclDouble, op == 0 ? ncclSum : ncclMax, static_cast<ncclComm_t>(rcclComm_), s), "ncclAllReduce");
      return;
    }
    HIP_OK(hipStreamSynchronize(s));
    if (!allreduce_ || allreduce_(ptr, (uint64_t)n, op, allreduceUser_) != 0) throw std::runtime_error("all-reduce callback failed");
  };
  // the scalars reach the host through the mailbox once they are complete: on one GPU the evaluation kernel publishes
  // them itself; in sharded mode they are complete only after the all-reduce, so a one-wave kernel publishes them then
  distNative_ = dist && rcclComm_ != nullptr;
  auto publish = [&]() {
    if (distNative_ && mailbox_) launchPublishScalars(p.scal, mailboxDev_, mailboxSeq_, s);
  };
  double* scalD = reinterpret_cast<double*>(p.scal);  // [0..7] group A, [8..15] group B, [16..31] the ranks' (gradMax, failMax) pairs
  // the post-solve pass can take the dogleg step itself when no all-reduce sits between them and one workgroup
  // retracts the whole window quickly enough
  const bool noFuseStep = optOn(kOptNoFuseStep);   // A/B switch for profiling
  // fused step: the landmark half of the retraction rides in the candidate evaluation (k_eval_all), which reads every
  // landmark anyway -- the serial tail of k_post_solve only moves the ~20 parameter blocks.  With the landmarks deferred the
  // fused step has no size limit (round 6: wide windows took a k_step_retract launch per iteration, 9 us, because of theirs)
  const bool deferPossible = canFuseEvaluation(p) && !optOn(kOptSplitEval) && !optOn(kOptNoDeferLm) && p.L > 0 && p.N > 0;
  const bool fuseStep = !noFuseStep && !dist && ((p.nPose + p.nExt + p.nSb + p.L) <= 16384 || deferPossible);
  const bool deferLm = fuseStep && deferPossible;
  // the stop vote (k_set_stop_vote) travels in the slots k_post_solve uses for the fused dogleg coefficients of the deferred
  // landmark step: the two never meet because a sharded solve takes neither the fused nor the deferred step
  if (dist && (fuseStep || deferLm)) throw std::logic_error("sharded solve with a fused step");
  // one GPU: the builds of this solve also run the speed / bias chain's factorisation, beside the landmark elimination (kernels.hpp
  // DeviceProblem::sideLane; every build below is followed by a solve with the same mu / initScale, or not used at all).  Sharded: the
  // chain's blocks are complete only after the all-reduce.
  struct SbEarlyGuard { DeviceProblem& q; ~SbEarlyGuard() { q.sideLane = 0; } } sbEarlyGuard{p};
  p.sideLane = (!dist && !optOn(kOptNoSbEarly)) ? 1 : 0;
  evaluateAll(false, s);
  TrustRegionHost tr;   // every decision of the loop (trust_region.hpp: HIP-free, replayed on the CPU by the tests)
  // The first build depends on no decision (initial damping, metric fixed here): it is enqueued right behind the initial
  // evaluation instead of after the host has seen that evaluation's cost (one mailbox round trip of an idle device per solve).
  bool firstBuilt = false;
  const double firstMu = tr.mu;
  if (!dist && numIter > 0) {
    launchAccumulateNormalEquations(p, firstMu, true, s, /*zeroFirst=*/false);   // pack() cleared the accumulators
    firstBuilt = true;
  }
  AR(scalD, 4, 0);
  publish();
  SolverScalars sc = readScalars();
  tr.fTol = fTol_; tr.gTol = gTol_; tr.pTol = pTol_;
  tr.maxIterations = (int)numIter;
  tr.start(sc.cost);
  summary_.initial_cost = sc.cost;
  double lastIterTime = 0;
  double stopVotes = 0.0;   // sharded mode: number of ranks whose clock asked to stop (identical on every rank)
  auto swapSets = [&]() { swapStateSets(); };
  auto toTr = [&](const SolverScalars& r) {
    TrScalars t;
    t.cost = r.cost; t.stepNormSq = r.stepNormSq; t.xNormSq = r.xNormSq; t.gradMax = r.gradMax; t.failMax = r.failMax;
    t.jdSq = r.jdSq; t.jdDotR = r.jdDotR; t.doglegStepNorm = r.doglegStepNorm;
    if (dist) {   // every rank's (gradMax, failMax) pair was gathered by the sum all-reduce of [group B | gather]
      t.gradMax = 0.0; t.failMax = 0.0;
      for (int k = 0; k < kScalGatherSlots / 2; ++k) { t.gradMax = std::max(t.gradMax, r.gather[2 * k]); t.failMax = std::max(t.failMax, r.gather[2 * k + 1]); }
    }
    return t;
  };
  // Speculative build (single GPU): most steps are accepted, and the host needs ~7 us from the mailbox to the first
  // launch of the next iteration.  Right behind the candidate evaluation the normal equations of the NEXT iteration
  // are enqueued on the candidate's linearisation (the sets an accepted step swaps in) with the damping an accepted
  // step gets; on acceptance they are simply kept, otherwise the accumulators are re-zeroed before the next build.
  const bool noSpeculation = optOn(kOptNoSpeculation);
  const bool speculate = !noSpeculation && (!dist || distNative_);
  bool accumulatorsClean = true;   // S / gRed / hC zero (pack() or k_post_solve), nothing speculative in them
  bool specValid = false;          // the accumulators hold the build of the candidate with damping specMu
  double specMu = 0;
  while (true) {
    // the time-limit callback (CeresIterationCallback.hpp:80-94).  One GPU: this rank's clock.  Sharded: a rank that left
    // the loop on its own clock would leave the others blocked in the next all-reduce, so the ranks vote (below, with the
    // evaluation's all-reduce) and stop on the common result
    const bool stop = dist ? stopVotes > 0.0
                           : (timeLimit_ >= 0.0 && tr.iteration >= minIterations_ && (nowSec() - tStart) + lastIterTime > timeLimit_);
    if (!tr.beginIteration(stop)) break;
    const double tIter = nowSec();
    while (true) {
      if (!tr.reuse) {
        const bool haveFirst = firstBuilt && tr.initScale && tr.mu == firstMu;   // the build enqueued ahead of the loop
        if (firstBuilt && !haveFirst) accumulatorsClean = false;                 // (a retry with more damping: it has to go)
        firstBuilt = false;
        if (!haveFirst && !(specValid && specMu == tr.mu && !tr.initScale))
          launchAccumulateNormalEquations(p, tr.mu, tr.initScale, s, /*zeroFirst=*/!accumulatorsClean);  // pack() / k_post_solve cleared them
        specValid = false;
        if (dist && p.d > 0) {   // one message: lower triangle of S + gRed + gFull + hC (half of what the full matrix would be)
          launchPackSystem(p, /*unpack=*/false, s);
          AR(p.cholL, packedSystemDoubles(p), 0);
          launchPackSystem(p, /*unpack=*/true, s);
        }
        launchSolveReduced(p, s, tr.mu, tr.initScale, /*fuseFinalize=*/true);
        p.lmDeferred = deferLm ? 1 : 0;
        launchDoglegPrepare(p, s, fuseStep ? tr.radius : -1.0);
        accumulatorsClean = true;
        AR(scalD + kScalGroupB, 8 + kScalGatherSlots, 0);   // group B and every rank's (gradMax, failMax) pair: one message
      }
      if (tr.reuse || !fuseStep) launchDoglegStep(p, tr.radius, s);
      p.lmDeferred = (deferLm && !tr.reuse) ? 1 : 0;
      evaluateAll(true, s);
      p.lmDeferred = 0;
      if (speculate && accumulatorsClean && tr.iteration < (int)numIter) {
        DeviceProblem q = p;   // the problem as it looks after an accepted step
        std::swap(q.pose, q.poseC); std::swap(q.ext, q.extC); std::swap(q.sb, q.sbC); std::swap(q.lm, q.lmC);
        std::swap(q.rCur, q.rCand); std::swap(q.JpCur, q.JpCand); std::swap(q.JlCur, q.JlCand); std::swap(q.JeCur, q.JeCand);
        std::swap(q.linCur, q.linCand);
        std::swap(q.priorDchi, q.priorDchiC); std::swap(q.priorGrad, q.priorGradC); std::swap(q.priorM3, q.priorM3C);
        specMu = tr.muAfterAccept();
        launchAccumulateNormalEquations(q, specMu, false, s, /*zeroFirst=*/false);
        accumulatorsClean = false;
        specValid = true;
      }
      if (dist) {  // this iteration will be complete when the vote is read: `iteration` already counts it, and its own duration
                   // so far stands in for the callback's iteration_time_in_seconds (CeresIterationCallback.hpp:86-89)
        const double tNow = nowSec();
        launchSetStopVote(p.scal, (timeLimit_ >= 0.0 && tr.iteration >= minIterations_ && (tNow - tStart) + (tNow - tIter) > timeLimit_) ? 1.0 : 0.0, s);
      }
      AR(scalD, 8, 0);
      publish();
      sc = readScalars();
      if (dist) stopVotes = sc.spareA0;
      // failMax carries the cholFail bits (max over the ranks): 4 | 8 = a bounded wait inside a solver kernel timed out.  That is
      // a synchronisation fault, not a numerical event -- it must not disappear into the mu ladder as "not positive definite"
      if (toTr(sc).failMax >= 4.0) {
        distNative_ = false;
        throw std::runtime_error("svin_ba: a device-side wait in the reduced-system solver timed out (cholFail " +
                                 std::to_string((int)toTr(sc).failMax) + "): synchronisation fault, not a numerical failure");
      }
      if (tr.retryFactorisation(toTr(sc))) { specValid = false; continue; }   // (the speculative damping assumed an accepted step)
      break;
    }
    const TrustRegionHost::Outcome o = tr.endIteration(toTr(sc));
    if (o == TrustRegionHost::kTerminated) break;
    if (o == TrustRegionHost::kInvalid) { specValid = false; lastIterTime = nowSec() - tIter; continue; }
    if (o == TrustRegionHost::kAccepted) swapSets();
    else specValid = false;   // rejected: the speculative build is discarded (accumulators are re-zeroed before the next one)
    if (verbose)
      std::printf("[svin_ba] it %d cost %.9e rel_dec %.3e radius %.3e step %.3e\n", tr.iteration, tr.x_cost, tr.relative_decrease,
                  tr.radius, tr.last_step_norm);
    lastIterTime = nowSec() - tIter;
  }
  summary_.termination = tr.termination;
  summary_.final_cost = tr.x_cost;
  summary_.iterations = tr.iteration;
  summary_.num_successful_steps = tr.successful;
  summary_.total_time = nowSec() - tStart;
  distNative_ = false;
}

void Window::swapStateSets() {
  DeviceProblem& p = prob_;
  std::swap(p.pose, p.poseC); std::swap(p.ext, p.extC); std::swap(p.sb, p.sbC); std::swap(p.lm, p.lmC);
  std::swap(p.rCur, p.rCand); std::swap(p.JpCur, p.JpCand); std::swap(p.JlCur, p.JlCand); std::swap(p.JeCur, p.JeCand);
  std::swap(p.linCur, p.linCand);
  std::swap(p.priorDchi, p.priorDchiC); std::swap(p.priorGrad, p.priorGradC); std::swap(p.priorM3, p.priorM3C);
}

// ------------------------------------------------------------------------------------------ batched solve
// SURVEY 8(e)'s "independent replicas processing different windows" on ONE GPU: a 10-keyframe window keeps 1-3 % of the chip
// busy (five launches of 10-30 us per iteration, the solver one workgroup), and eight handles on eight streams only reach
// 1.6 x one handle -- launch rate and hardware queues.  Here the windows of a batch share the launches: per trust-region ROUND one
// slot table goes to the device (every window's problem as it stands and the scalars of its own trust region, 816 bytes each)
// and at most six launches follow with the window as blockIdx.y.  The host keeps one TrustRegionHost per window and takes each
// window's decision from its own mailbox record exactly as solve() does; a window whose step was rejected takes the round's
// k_step_retract launch instead of the build / solve / post-solve launches, a window that has terminated takes none.  No
// speculative build: measured (round 6, B = 16 in four lanes) 6.9 x one window with it against 7.4 x without -- the lanes keep the
// chip busy, so the builds of steps that end up rejected are work added, not latency hidden.  For the same reason issuing the initial
// evaluation and the first round together (two slot tables, two mailbox records in flight; one host turnaround of eleven saved)
// gains nothing: 6.8 x against 7.1 x at B = 16, 8.3 / 9.0 x at 32 / 64 either way -- measured, not kept.  The arithmetic of a window is the
// arithmetic of solve(): same kernels bodies, same grids (gridDim.x), same reduction orders.
namespace {
struct BatchKey {
  int v[18];
  bool operator<(const BatchKey& o) const { return std::lexicographical_compare(v, v + 18, o.v, o.v + 18); }
};
BatchKey batchKeyOf(const DeviceProblem& p) {
  // what the grids, the LDS sizes and the uniform kernel arguments of launchBatchRound are computed from (L and N themselves may
  // differ: every kernel reads them from its window's problem)
  return BatchKey{{p.d, p.dC, p.dCPose, (p.L + 15) / 16, (p.N + 255) / 256, p.F, p.nPose, p.nExt, p.nSb, (p.nPose + p.nExt + p.nSb + p.L + 255) / 256,
                   p.nSlabs, p.priorM, p.anyExtVariable, p.ldS, p.sPadded, p.priorBlocks, p.nCam /* (staging area of the evaluation) */, p.schurDense}};
}
}  // namespace

int Window::solvePreparedBatch(Window* const* ws, int n, size_t numIter, bool verbose, int* nBatched) {
  if (nBatched) *nBatched = 0;
  if (n <= 0) return 1;
  std::map<BatchKey, std::vector<Window*>> groups;
  std::vector<Window*> alone;
  for (int i = 0; i < n; ++i) {
    Window* w = ws[i];
    for (int j = 0; j < i; ++j)
      if (ws[j] == w) throw std::invalid_argument("svin_ba_solve_prepared_batch: a handle appears twice");
    w->maxIterationsOption_ = numIter;
    const DeviceProblem& p = w->prob_;
    const bool ok = w->world_ <= 1 && !w->rcclComm_ && w->device_ == ws[0]->device_ && w->mailbox_ && p.d + 3 * p.L > 0 && batchSupported(p);
    if (ok) groups[batchKeyOf(p)].push_back(w);
    else alone.push_back(w);
  }
  for (auto& kv : groups) {
    if (kv.second.size() < 2) { alone.push_back(kv.second[0]); continue; }
    solveBatchGroup(kv.second, numIter, verbose);
    if (nBatched) *nBatched += (int)kv.second.size();
  }
  for (Window* w : alone) w->solvePrepared(numIter, verbose);
  return 1;
}

// The streams of the lanes: created one after the other, once per device, so that they land on DIFFERENT hardware queues (the
// runtime deals streams over its four queues in creation order; the windows' own streams -- every fourth handle on the same
// queue -- would put all the lanes of a batch behind one another).  They live as long as the process.
static hipStream_t laneStream(int device, int k) {
  static std::mutex mu;
  static std::map<int, std::vector<hipStream_t>> pool;
  std::lock_guard<std::mutex> lock(mu);
  std::vector<hipStream_t>& v = pool[device];
  while ((int)v.size() <= k) {
    hipStream_t s = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    v.push_back(s);
  }
  return v[(size_t)k];
}

void Window::solveBatchGroup(const std::vector<Window*>& g, size_t numIter, bool verbose) {
  const double tStart = nowSec();
  const int B = (int)g.size();
  HIP_OK(hipSetDevice(g[0]->device_));
  for (Window* w : g) HIP_OK(hipStreamSynchronize(w->stream_));   // uploads / earlier work of every window: done before a shared stream reads them
  // LANES: the windows are cut into up to kLanes sub-batches, each with the stream and the slot table of its first window.  The
  // reduced solve (one workgroup per window, 30 us) and a re-preintegrating IMU factor (one workgroup, 50 us) are latency, not
  // work: while one lane sits in them the launches of the other lanes fill the chip.  The host serves the lanes round robin --
  // collect a lane's mailbox records, take its windows' decisions, issue its next round, go on to the next lane.
  int nLanes = debugOption(kOptBatchLanes) > 0 ? debugOption(kOptBatchLanes) : 4;
  nLanes = std::max(1, std::min(nLanes, B / 2));
  struct Lane { int first = 0, count = 0; Window* lead = nullptr; hipStream_t s = nullptr; BatchSlot* hs = nullptr; bool cand = false, done = false; };
  std::vector<Lane> lanes((size_t)nLanes);
  for (int k = 0; k < nLanes; ++k) {
    Lane& ln = lanes[(size_t)k];
    ln.first = (int)((long long)B * k / nLanes);
    ln.count = (int)((long long)B * (k + 1) / nLanes) - ln.first;
    ln.lead = g[(size_t)ln.first];
    ln.s = nLanes > 1 ? laneStream(g[0]->device_, k) : ln.lead->stream_;
    ln.lead->batchSlotsDev_.reserve((size_t)ln.count);
    if (ln.lead->batchSlotsHostCap_ < (size_t)ln.count) {   // pinned host copy of the lane's slot table (filled per round)
      if (ln.lead->batchSlotsHost_) (void)hipHostFree(ln.lead->batchSlotsHost_);
      ln.lead->batchSlotsHost_ = nullptr; ln.lead->batchSlotsHostCap_ = 0;
      HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&ln.lead->batchSlotsHost_), sizeof(BatchSlot) * (size_t)ln.count, hipHostMallocDefault));
      ln.lead->batchSlotsHostCap_ = (size_t)ln.count;
    }
    ln.hs = ln.lead->batchSlotsHost_;
  }
  struct State { TrustRegionHost tr; bool active = true; bool deferLm = false; double lastIterTime = 0, tIter = 0; };
  std::vector<State> st((size_t)B);
  for (int i = 0; i < B; ++i) {
    Window* w = g[i];
    const DeviceProblem& p = w->prob_;
    w->summary_.iterations = 0; w->summary_.num_successful_steps = 0; w->summary_.termination = 1;
    w->distNative_ = false;
    st[i].deferLm = p.L > 0 && p.N > 0;   // (batchSupported: fused step, fused evaluation)
    st[i].tr.fTol = w->fTol_; st[i].tr.gTol = w->gTol_; st[i].tr.pTol = w->pTol_;
    st[i].tr.maxIterations = (int)numIter;
  }
  // one round of a lane: the slots of its windows that take part -> device, the launches; false if no window takes part
  auto issue = [&](Lane& ln, bool cand) -> bool {
    int uni = 0;
    for (int k = 0; k < ln.count; ++k) {
      const int i = ln.first + k;
      Window* w = g[i];
      BatchSlot& sl = ln.hs[k];
      sl.stages = 0;
      if (!st[i].active) continue;
      const TrustRegionHost& tr = st[i].tr;
      DeviceProblem& p = w->prob_;
      p.mailbox = w->mailboxDev_;
      p.mailboxSeq = ++w->mailboxSeq_;
      p.lmDeferred = (cand && st[i].deferLm && !tr.reuse) ? 1 : 0;
      p.aBlocks = schurDenseABlocks(p);
      sl.p = p;
      p.lmDeferred = 0;
      sl.mu = tr.mu; sl.radius = tr.radius; sl.initScale = tr.initScale ? 1 : 0;
      sl.stages = !cand ? kBatchEval : (tr.reuse ? (kBatchReuse | kBatchEval) : (kBatchFull | kBatchEval));
      uni |= sl.stages;
    }
    ln.cand = cand;
    if (!uni) return false;
    HIP_OK(hipMemcpyAsync(ln.lead->batchSlotsDev_.p, ln.hs, sizeof(BatchSlot) * (size_t)ln.count, hipMemcpyHostToDevice, ln.s));
    launchBatchRound(ln.lead->batchSlotsDev_.p, ln.lead->prob_, ln.count, uni, cand, ln.s);
    return true;
  };
  auto toTr = [](const SolverScalars& r) {
    TrScalars t;
    t.cost = r.cost; t.stepNormSq = r.stepNormSq; t.xNormSq = r.xNormSq; t.gradMax = r.gradMax; t.failMax = r.failMax;
    t.jdSq = r.jdSq; t.jdDotR = r.jdDotR; t.doglegStepNorm = r.doglegStepNorm;
    return t;
  };
  auto finishWindow = [&](int i) {
    Window* w = g[i];
    const TrustRegionHost& tr = st[i].tr;
    st[i].active = false;
    w->summary_.termination = tr.termination; w->summary_.final_cost = tr.x_cost; w->summary_.iterations = tr.iteration;
    w->summary_.num_successful_steps = tr.successful; w->summary_.total_time = nowSec() - tStart;
  };
  // top of an iteration for window i (solve(): the time-limit callback, then TrustRegionHost::beginIteration)
  auto begin = [&](int i) {
    Window* w = g[i];
    TrustRegionHost& tr = st[i].tr;
    const bool stop = w->timeLimit_ >= 0.0 && tr.iteration >= w->minIterations_ && (nowSec() - tStart) + st[i].lastIterTime > w->timeLimit_;
    if (!tr.beginIteration(stop)) { finishWindow(i); return; }
    st[i].tIter = nowSec();
  };
  // the records of the lane's round in flight, and what each of its windows does with them
  auto collect = [&](Lane& ln) {
    for (int k = 0; k < ln.count; ++k) {
      const int i = ln.first + k;
      if (!ln.hs[k].stages) continue;
      const SolverScalars sc = g[i]->readScalars();
      TrustRegionHost& tr = st[i].tr;
      if (!ln.cand) {   // the initial evaluation
        tr.start(sc.cost);
        g[i]->summary_.initial_cost = sc.cost;
        begin(i);
        continue;
      }
      const TrScalars t = toTr(sc);
      if (t.failMax >= 4.0)
        throw std::runtime_error("svin_ba: a device-side wait in the reduced-system solver timed out (cholFail " + std::to_string((int)t.failMax) +
                                 "): synchronisation fault, not a numerical failure");
      if (tr.retryFactorisation(t)) continue;   // the same iteration again, with more damping (next round: a fresh build)
      const TrustRegionHost::Outcome o = tr.endIteration(t);
      if (o == TrustRegionHost::kTerminated) { finishWindow(i); continue; }
      if (o == TrustRegionHost::kAccepted) g[i]->swapStateSets();
      if (verbose && o != TrustRegionHost::kInvalid)
        std::printf("[svin_ba batch %d] it %d cost %.9e rel_dec %.3e radius %.3e step %.3e\n", i, tr.iteration, tr.x_cost, tr.relative_decrease,
                    tr.radius, tr.last_step_norm);
      st[i].lastIterTime = nowSec() - st[i].tIter;
      begin(i);
    }
  };
  const bool timing = optOn(kOptBatchTiming);
  double tIssue = 0, tCollect = 0, tWaitSum = 0;
  int nRounds = 0;
  try {
    for (Lane& ln : lanes) ln.done = !issue(ln, false);
    // a lane is served as soon as the records of its round are there (the rounds of the lanes drift: rejected steps, windows
    // that have terminated, a re-preintegration); a lane that stays silent for two seconds is collected anyway -- readScalars()
    // then falls back to a synchronise and reports what it finds
    auto ready = [&](const Lane& ln) {
      for (int k = 0; k < ln.count; ++k)
        if (ln.hs[k].stages && !g[ln.first + k]->scalarsReady()) return false;
      return true;
    };
    for (;;) {
      int open = 0;
      Lane* pick = nullptr;
      for (Lane& ln : lanes) open += ln.done ? 0 : 1;
      if (!open) break;
      const double tWait = nowSec();
      for (unsigned long long spins = 0; !pick; ++spins) {
        for (Lane& ln : lanes)
          if (!ln.done && ready(ln)) { pick = &ln; break; }
        if (!pick && (spins & 255) == 255 && nowSec() - tWait > 2.0)
          for (Lane& ln : lanes)
            if (!ln.done) { pick = &ln; break; }
      }
      const double t0 = timing ? nowSec() : 0.0;
      collect(*pick);
      const double t1 = timing ? nowSec() : 0.0;
      pick->done = !issue(*pick, true);
      if (timing) { tCollect += t1 - t0; tWaitSum += t0 - tWait; tIssue += nowSec() - t1; ++nRounds; }
    }
  } catch (...) {
    for (Lane& ln : lanes) (void)hipStreamSynchronize(ln.s);   // nothing of this call is left in flight behind the error
    throw;
  }
  for (Lane& ln : lanes) HIP_OK(hipStreamSynchronize(ln.s));
  if (timing)
    std::printf("[svin_ba batch] %d windows in %d lanes: %.1f us, %d lane rounds: wait %.1f us, collect %.1f us, issue %.1f us each\n", B, nLanes,
                1e6 * (nowSec() - tStart), nRounds, 1e6 * tWaitSum / std::max(1, nRounds), 1e6 * tCollect / std::max(1, nRounds),
                1e6 * tIssue / std::max(1, nRounds));
  for (Window* w : g) w->summary_.solve_time = nowSec() - tStart;
}

int Window::prepare() {
  const double t0 = nowSec();
  pack();
  ++pathCounters_[residentUsed_ ? 0 : 1];   // counted where an optimisation packs, not in pack() itself: linearize(), the
                                            // inspection hooks and debugReducedSolve() pack too
  HIP_OK(hipStreamSynchronize(stream_));
  summary_.upload_time = nowSec() - t0;
  return 1;
}
int Window::solvePrepared(size_t numIter, bool verbose) {
  maxIterationsOption_ = numIter;
  const double t1 = nowSec();
  solve(numIter, verbose);
  HIP_OK(hipStreamSynchronize(stream_));
  summary_.solve_time = nowSec() - t1;
  return 1;
}
int Window::finish() {
  const double t2 = nowSec();
  downloadStates();
  summary_.download_time = nowSec() - t2;
  return 1;
}
int Window::optimize(size_t numIter, bool verbose) {
  // (prepare / solvePrepared / finish with their synchronisation points are the measurement form; here the solve is enqueued
  // right behind the upload and the rebuild of the resident window)
  const double t0 = nowSec();
  pack(/*solveFollows=*/true);
  ++pathCounters_[residentUsed_ ? 0 : 1];   // (svin_ba_get_path_counters: nothing falls back silently)
  const double t1 = nowSec();
  summary_.upload_time = t1 - t0;   // host time of pack(): the device part overlaps the first launches of the solve
  maxIterationsOption_ = numIter;
  solve(numIter, verbose);
  const double t2 = nowSec();
  summary_.solve_time = t2 - t1;    // the trust-region loop ends on a host decision: the device has caught up
  downloadStates();
  summary_.download_time = nowSec() - t2;
  return 1;
}

int Window::setOptimizationTimeLimit(double timeLimit, int minIter) {  // :932-951
  if (hasCallback_) {
    if (timeLimit < 0.0) { minIterations_ = (int)maxIterationsOption_; return 1; }
    timeLimit_ = timeLimit; minIterations_ = minIter;
    return 1;
  } else if (timeLimit >= 0.0) {
    hasCallback_ = true;
    timeLimit_ = timeLimit; minIterations_ = minIter;
    return 1;
  }
  return 1;
}

// ------------------------------------------------------------------------------------------ inspection hooks
int Window::observationIds(uint64_t* rid, uint64_t* lm, uint64_t* pose, int32_t* cam, int cap) {
  pack();
  HIP_OK(hipStreamSynchronize(stream_));
  // same order as pack(): landmarks in CSR order (lmIds_), observations in insertion order
  int n = 0;
  std::vector<uint64_t> order(lmIds_);
  if (residentUsed_) {
    order.clear();
    for (const Landmark* lp : lmByHandle_) if (lp && !lp->obs.empty()) order.push_back(lp->id);
  }
  for (uint64_t id : order) {
    const Landmark& l = landmarks_.at(id);
    for (const Observation& o : l.obs) {
      if (n < cap) {
        if (rid) rid[n] = o.resId;
        if (lm) lm[n] = l.id;
        if (pose) pose[n] = o.poseId;
        if (cam) cam[n] = o.cam;
      }
      ++n;
    }
    for (const Landmark::Prior& pr : l.priors)   // the two pseudo-observations of a HomogeneousPointError: pose 0, camera 15
      for (int part = 0; part < 2; ++part) {
        if (n < cap) {
          if (rid) rid[n] = pr.resId;
          if (lm) lm[n] = l.id;
          if (pose) pose[n] = 0;
          if (cam) cam[n] = kPriorCam;
        }
        ++n;
      }
  }
  return n;
}
int Window::debugCsr(int32_t* L, int32_t* N, int32_t* lmPtr, int32_t* obsLm, uint32_t* obsIdx, double* uv, double* w, double* lm,
                     int32_t* obsOrder, int32_t* residentOut) {
  pack();
  HIP_OK(hipStreamSynchronize(stream_));
  if (residentUsed_) checkResidentStatus();
  const DeviceProblem& p = prob_;
  if (L) *L = p.L;
  if (N) *N = p.N;
  if (residentOut) *residentOut = residentUsed_ ? 1 : 0;
  if (lmPtr && p.L > 0) HIP_OK(hipMemcpy(lmPtr, p.lmPtr, sizeof(int) * (p.L + 1), hipMemcpyDeviceToHost));
  if (lm && p.L > 0) HIP_OK(hipMemcpy(lm, p.lm, sizeof(double) * 4 * p.L, hipMemcpyDeviceToHost));
  if (p.N > 0) {
    if (obsLm) HIP_OK(hipMemcpy(obsLm, p.obsLm, sizeof(int) * p.N, hipMemcpyDeviceToHost));
    if (obsIdx) HIP_OK(hipMemcpy(obsIdx, p.obsIdx, sizeof(uint32_t) * p.N, hipMemcpyDeviceToHost));
    if (uv) HIP_OK(hipMemcpy(uv, p.obsUv, sizeof(double) * 2 * p.N, hipMemcpyDeviceToHost));
    if (w) HIP_OK(hipMemcpy(w, p.obsW, sizeof(double) * p.N, hipMemcpyDeviceToHost));
    if (obsOrder) {
      if (p.obsOrder) HIP_OK(hipMemcpy(obsOrder, p.obsOrder, sizeof(int) * p.N, hipMemcpyDeviceToHost));
      else for (int i = 0; i < p.N; ++i) obsOrder[i] = -1;
    }
  }
  return 1;
}
int Window::evalReprojection(bool robust, double* r, double* Jp, double* Jl, double* Je, int cap) {
  pack();
  const DeviceProblem& p = prob_;
  const int N = p.N;
  if (N == 0) return 0;
  // always materialise the extrinsics Jacobian for inspection
  DeviceProblem q = p;
  q.anyExtVariable = 1;
  launchEvalReproj(q, false, robust, stream_);
  std::vector<double> h((size_t)32 * N);
  HIP_OK(hipMemcpyAsync(h.data(), dLin_[0].p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream_));
  HIP_OK(hipStreamSynchronize(stream_));
  const int n = std::min(N, cap);
  for (int i = 0; i < n; ++i) {
    if (r) { r[2 * i] = h[i]; r[2 * i + 1] = h[(size_t)N + i]; }
    if (Jp) for (int k = 0; k < 12; ++k) Jp[12 * i + k] = h[(size_t)(2 + k) * N + i];
    if (Jl) for (int k = 0; k < 6; ++k) Jl[6 * i + k] = h[(size_t)(14 + k) * N + i];
    if (Je) for (int k = 0; k < 12; ++k) Je[12 * i + k] = h[(size_t)(20 + k) * N + i];
  }
  return N;
}
int Window::evalFactors(int32_t* kind, int32_t* m, int32_t* ncols, double* r, double* J, uint64_t* blocks,
                        uint64_t* rids, int cap) {
  pack();
  const DeviceProblem& p = prob_;
  if (p.F == 0) return 0;
  evaluateHostFactors(false, stream_);
  launchEvalFactors(p, false, stream_);
  std::vector<FactorLin> h(p.F);
  std::vector<DevImu> hImu(p.nImu);
  HIP_OK(hipMemcpyAsync(h.data(), p.linCur, sizeof(FactorLin) * p.F, hipMemcpyDeviceToHost, stream_));
  if (p.nImu > 0) HIP_OK(hipMemcpyAsync(hImu.data(), p.imus, sizeof(DevImu) * p.nImu, hipMemcpyDeviceToHost, stream_));
  HIP_OK(hipStreamSynchronize(stream_));
  // evaluation may re-preintegrate: keep the state, exactly like ImuError's mutable members
  int k = 0;
  for (uint64_t fid : factorIds_) {
    Factor& f = factors_.at(fid);
    if (f.kind == F_IMU) f.imu = hImu[k++];
  }
  int i = 0;
  for (uint64_t fid : factorIds_) {
    if (i >= cap) break;
    const Factor& f = factors_.at(fid);
    if (kind) kind[i] = f.kind;
    if (rids) rids[i] = f.id;
    if (m) m[i] = h[i].m;
    if (ncols) ncols[i] = h[i].ncols;
    if (r) std::memcpy(r + 15 * i, h[i].r, 15 * sizeof(double));
    if (J) std::memcpy(J + 450 * i, h[i].J, 450 * sizeof(double));
    if (blocks) for (int b = 0; b < 4; ++b) blocks[4 * i + b] = b < f.nblk ? f.blocks[b] : 0;
    ++i;
  }
  return p.F;
}
int Window::linearize(double mu, double* S, double* g, uint64_t* blockIds, int32_t* blockOff, int32_t* nBlocks,
                      int capD, double* cost) {
  distNative_ = false;
  pack();
  DeviceProblem& p = prob_;
  if (p.d > capD) return -p.d;
  evaluateAll(false, stream_);
  launchBuildNormalEquations(p, mu, true, stream_);
  SolverScalars sc = readScalars();
  if (cost) *cost = sc.cost;
  if (S) HIP_OK(hipMemcpy2D(S, sizeof(double) * p.d, p.S, sizeof(double) * p.ldS, sizeof(double) * p.d, p.d, hipMemcpyDeviceToHost));
  if (g) HIP_OK(hipMemcpy(g, p.gRed, sizeof(double) * p.d, hipMemcpyDeviceToHost));
  if (nBlocks) *nBlocks = (int)redBlockIds_.size();
  for (size_t i = 0; i < redBlockIds_.size(); ++i) {
    if (blockIds) blockIds[i] = redBlockIds_[i];
    if (blockOff) blockOff[i] = redBlockOff_[i];
  }
  // evaluation may have re-preintegrated IMU factors
  std::vector<DevImu> hImu(p.nImu);
  if (p.nImu > 0) {
    HIP_OK(hipMemcpy(hImu.data(), p.imus, sizeof(DevImu) * p.nImu, hipMemcpyDeviceToHost));
    int k = 0;
    for (uint64_t fid : factorIds_) {
      Factor& f = factors_.at(fid);
      if (f.kind == F_IMU) f.imu = hImu[k++];
    }
  }
  return p.d;
}
// inspection hook: the Gauss-Newton step of the reduced system as the solver kernels compute it (whichever of the four paths the
// size selects), for the tests to hold against a host solve of linearize()'s system
int Window::debugReducedSolve(double mu, double* y, int capD, bool fuseFinalize) {
  distNative_ = false;
  pack();
  DeviceProblem& p = prob_;
  if (p.d > capD) return -p.d;
  evaluateAll(false, stream_);
  if (fuseFinalize) {   // what solve() enqueues: metric and damping applied inside the solver's load phase
    launchAccumulateNormalEquations(p, mu, true, stream_, /*zeroFirst=*/true);
    launchSolveReduced(p, stream_, mu, true, /*fuseFinalize=*/true);
  } else {
    launchBuildNormalEquations(p, mu, true, stream_);
    launchSolveReduced(p, stream_);
  }
  HIP_OK(hipMemcpyAsync(y, p.yC, sizeof(double) * p.d, hipMemcpyDeviceToHost, stream_));
  HIP_OK(hipStreamSynchronize(stream_));
  return p.d;
}
// inspection hook: doubles [off, off + count) of the reduced-system solver's scratch buffer (the factor, the eliminated chain's
// records and Y: layout in kernels.hip, launchSolveReduced) after the last solve
int Window::debugPeekSolverScratch(uint64_t off, uint64_t count, double* out) {
  quiesce();
  if (!prob_.cholL || off + count > solveReducedScratchDoubles(prob_.d, true)) return 0;
  HIP_OK(hipStreamSynchronize(stream_));
  HIP_OK(hipMemcpy(out, prob_.cholL + off, sizeof(double) * count, hipMemcpyDeviceToHost));
  return 1;
}
int Window::getPrior(double* H, double* b0, double* J, double* e0, uint64_t* ids, int32_t* ord, int32_t* mdim,
                     int32_t* nBlocks, int capM) {
  quiesce();
  if (!hasPrior_) return 0;
  const int m = priorM_;
  if (m > capM) return -m;
  if (!priorHostValid_) {  // the prior lives on the device; fetch the inspection copies on demand
    const size_t n2 = (size_t)m * m;
    priorH_.assign(n2, 0.0); priorB0_.assign(m, 0.0); priorJ_.assign(n2, 0.0); priorE0_.assign(m, 0.0);
    HIP_OK(hipMemcpyAsync(priorH_.data(), margBuf_.bHk.p, sizeof(double) * n2, hipMemcpyDeviceToHost, stream_));
    HIP_OK(hipMemcpyAsync(priorB0_.data(), margBuf_.bHk.p + n2, sizeof(double) * m, hipMemcpyDeviceToHost, stream_));
    HIP_OK(hipMemcpyAsync(priorJ_.data(), margBuf_.bOut.p + 2 * n2, sizeof(double) * n2, hipMemcpyDeviceToHost, stream_));
    HIP_OK(hipMemcpyAsync(priorE0_.data(), margBuf_.bOut.p + 4 * n2, sizeof(double) * m, hipMemcpyDeviceToHost, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
    priorHostValid_ = true;
  }
  if (H) std::memcpy(H, priorH_.data(), sizeof(double) * m * m);
  if (b0) std::memcpy(b0, priorB0_.data(), sizeof(double) * m);
  if (J) std::memcpy(J, priorJ_.data(), sizeof(double) * m * m);
  if (e0) std::memcpy(e0, priorE0_.data(), sizeof(double) * m);
  if (nBlocks) *nBlocks = (int)priorBlocks_.size();
  for (size_t i = 0; i < priorBlocks_.size(); ++i) {
    if (ids) ids[i] = priorBlocks_[i].id;
    if (ord) ord[i] = priorBlocks_[i].ord;
    if (mdim) mdim[i] = priorBlocks_[i].mdim;
  }
  return m;
}

// ------------------------------------------------------------------------------------------ measurement hooks
int Window::benchJacobianEval(int copies, int iters, double* meanMs, double* bytes, double* backToBackMs) {
  pack();
  const DeviceProblem& p = prob_;
  const size_t N = (size_t)p.N, NB = N * copies;
  if (N == 0) return 0;
  // replicate the observation arrays; every replica reads its own copy and writes its own output lines
  std::vector<double> hUv(2 * N), hW(N);
  std::vector<uint32_t> hIdx(N);
  std::vector<int> hLm(N);
  HIP_OK(hipMemcpy(hUv.data(), p.obsUv, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hW.data(), p.obsW, sizeof(double) * N, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hIdx.data(), p.obsIdx, sizeof(uint32_t) * N, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hLm.data(), p.obsLm, sizeof(int) * N, hipMemcpyDeviceToHost));
  std::vector<double> hLmTab((size_t)4 * p.L);
  HIP_OK(hipMemcpy(hLmTab.data(), p.lm, sizeof(double) * 4 * p.L, hipMemcpyDeviceToHost));
  DevBuf<double> bUv, bW, bLm, bOut;
  DevBuf<uint32_t> bIdx;
  DevBuf<int> bObsLm;
  bUv.reserve(2 * NB); bW.reserve(NB); bIdx.reserve(NB); bObsLm.reserve(NB); bLm.reserve((size_t)4 * p.L * copies);
  const int rows = p.anyExtVariable ? 32 : 20;
  bOut.reserve((size_t)rows * NB);
  for (int c = 0; c < copies; ++c) {
    std::vector<int> lmShift(N);
    for (size_t i = 0; i < N; ++i) lmShift[i] = hLm[i] + c * p.L;
    HIP_OK(hipMemcpy(bUv.p + 2 * N * c, hUv.data(), sizeof(double) * 2 * N, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bW.p + N * c, hW.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bIdx.p + N * c, hIdx.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bObsLm.p + N * c, lmShift.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(bLm.p + (size_t)4 * p.L * c, hLmTab.data(), sizeof(double) * 4 * p.L, hipMemcpyHostToDevice));
  }
  DeviceProblem q = p;
  q.obsUv = bUv.p; q.obsW = bW.p; q.obsIdx = bIdx.p; q.obsLm = bObsLm.p; q.lm = bLm.p;
  double* r = bOut.p;
  double* Jp = r + 2 * NB;
  double* Jl = Jp + 12 * NB;
  double* Je = p.anyExtVariable ? Jl + 6 * NB : nullptr;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) launchEvalReprojBatched(q, copies, r, Jp, Jl, Je, stream_);
  HIP_OK(hipStreamSynchronize(stream_));
  double total = 0;
  for (int it = 0; it < iters; ++it) {
    HIP_OK(hipEventRecord(e0, stream_));
    launchEvalReprojBatched(q, copies, r, Jp, Jl, Je, stream_);
    HIP_OK(hipEventRecord(e1, stream_));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    total += ms;
  }
  if (meanMs) *meanMs = total / iters;
  // the same launches back to back under ONE event pair: the drain of launch i (its last stores leaving the Infinity Cache)
  // overlaps launch i + 1 instead of being cut off by the stop event -- the figure a per-launch bracket cannot flatter
  if (backToBackMs) {
    HIP_OK(hipEventRecord(e0, stream_));
    for (int it = 0; it < iters; ++it) launchEvalReprojBatched(q, copies, r, Jp, Jl, Je, stream_);
    HIP_OK(hipEventRecord(e1, stream_));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    *backToBackMs = (double)ms / iters;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  // algorithmic bytes per residual: read uv 16 + w 8 + packed index 4 + landmark index 4
  // + landmark 32/obs-per-landmark; write r 16 + Jp 96 + Jl 48 (+ Je 96 when extrinsics are variable)
  const double perRes = 16 + 8 + 4 + 4 + 32.0 * p.L / (double)N + 16 + 96 + 48 + (p.anyExtVariable ? 96 : 0);
  if (bytes) *bytes = perRes * (double)NB;
  return 1;
}

#ifdef SVIN_IMU_TIMING
void debugImuTiming(double* out, bool reset);
#endif
#ifdef SVIN_CHOL_TIMING
void debugCholTiming(double* out, bool reset);
#endif
// the collective of the sharded solve, stand-alone: `iters` in-place sum all-reduces of nDoubles FP64 values on the solver's
// stream between two HIP events (the communicator svin_ba_set_distributed_rccl created; collective: every rank calls it)
int Window::benchAllReduce(size_t nDoubles, int iters, double* meanUs) {
  quiesce();
  if (!rcclComm_) { lastError() = "benchAllReduce: no RCCL communicator (svin_ba_set_distributed_rccl first)"; return -1; }
  if (nDoubles == 0 || iters <= 0) return -1;
  HIP_OK(hipSetDevice(device_));
  DevBuf<double> buf;
  buf.reserve(nDoubles);
  HIP_OK(hipMemsetAsync(buf.p, 0, sizeof(double) * nDoubles, stream_));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    rcclCheck(rccl().allReduce(buf.p, buf.p, nDoubles, ncclDouble, ncclSum, static_cast<ncclComm_t>(rcclComm_), stream_), "ncclAllReduce");
  HIP_OK(hipEventRecord(e0, stream_));
  for (int i = 0; i < iters; ++i)
    rcclCheck(rccl().allReduce(buf.p, buf.p, nDoubles, ncclDouble, ncclSum, static_cast<ncclComm_t>(rcclComm_), stream_), "ncclAllReduce");
  HIP_OK(hipEventRecord(e1, stream_));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (meanUs) *meanUs = 1e3 * (double)ms / iters;
  return 1;
}

int Window::benchKernelTimes(int iters, double* evalMs, double* buildMs, double* solveMs) {
  pack();
#ifdef SVIN_IMU_TIMING
  {
    debugImuTiming(nullptr, true);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) {
      invalidatePreintegration();
      pack();
      launchEvalFactors(prob_, false, stream_);
      HIP_OK(hipStreamSynchronize(stream_));
    }
    double dbg[16];
    debugImuTiming(dbg, false);
    const double n = 9.0 * reps;
    {  // the evaluation without re-integration (most iterations)
      debugImuTiming(nullptr, true);
      for (int i = 0; i < reps; ++i) { launchEvalFactors(prob_, false, stream_); HIP_OK(hipStreamSynchronize(stream_)); }
      double d2[16];
      debugImuTiming(d2, false);
      std::printf("[imu eval cycles, no redo] factors counted %.0f: staging %.0f, serial F/e block %.0f (shared quantities %.0f, position/velocity blocks %.0f; part 1 %.0f, part 3 %.0f), whole block %.0f\n", d2[11],
                  d2[8] / d2[11], d2[9] / d2[11], d2[12] / d2[11], d2[13] / d2[11], d2[14] / d2[11], d2[15] / d2[11], d2[10] / d2[11]);
      hipEvent_t a, b;
      HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
      for (int variant = 0; variant < 3; ++variant) {
        float tot = 0;
        for (int i = 0; i < 20; ++i) {
          HIP_OK(hipEventRecord(a, stream_));
          if (variant == 0) evaluateAll(false, stream_);
          else if (variant == 1) launchEvalFactors(prob_, false, stream_);
          else launchEvalAll(prob_, false, false, stream_);
          HIP_OK(hipEventRecord(b, stream_));
          HIP_OK(hipEventSynchronize(b));
          float ms = 0;
          HIP_OK(hipEventElapsedTime(&ms, a, b));
          tot += ms;
        }
        std::printf("[eval timing, no redo] %s: %.2f us\n", variant == 0 ? "evaluateAll (fused, cost summed)" : variant == 1 ? "factors only" : "fused, no cost sum", 1e3 * tot / 20);
      }
    }
    std::printf("[imu redo cycles] P0 %.0f P1dq %.0f P1cross %.0f P2 %.0f P3 %.0f cov %.0f (segment loop of wave 0 %.0f) integrate %.0f post %.0f\n", dbg[0] / n,
                dbg[1] / n, dbg[6] / n, dbg[7] / n, dbg[5] / n, dbg[2] / n, dbg[14] / n, dbg[3] / n, dbg[4] / n);
  }
#endif
  DeviceProblem& p = prob_;
  hipStream_t s = stream_;
  evaluateAll(false, s);
  launchBuildNormalEquations(p, 1e-8, true, s);
  launchSolveReduced(p, s);
  HIP_OK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  auto timeIt = [&](auto fn) {
    double tot = 0;
    for (int i = 0; i < iters; ++i) {
      HIP_OK(hipEventRecord(e0, s));
      fn();
      HIP_OK(hipEventRecord(e1, s));
      HIP_OK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms;
    }
    return tot / iters;
  };
  if (evalMs) *evalMs = timeIt([&]() { launchEvalReproj(p, false, true, s); });
#ifdef SVIN_SCHUR_TIMING
  HIP_OK(hipMemset(p.partial + (size_t)15 * 4096 + 16, 0, 64));
#endif
  if (buildMs) *buildMs = timeIt([&]() { launchBuildNormalEquations(p, 1e-8, false, s); });
#ifdef SVIN_SCHUR_TIMING
  {
    double dbg[5];
    HIP_OK(hipMemcpy(dbg, p.partial + (size_t)15 * 4096 + 16, sizeof(dbg), hipMemcpyDeviceToHost));
    std::printf("[schur cycles per launch, block 0 wave 0] zero %.0f pass1 %.0f vinv %.0f loop-total %.0f slab-store %.0f\n", dbg[0] / iters,
                dbg[1] / iters, dbg[2] / iters, dbg[3] / iters, dbg[4] / iters);
  }
#endif
  if (optOn(kOptCholTiming)) HIP_OK(hipMemset(p.partial + (size_t)15 * 4096, 0, 64 * 8));
#ifdef SVIN_CHOL_TIMING
  debugCholTiming(nullptr, true);
#endif
  if (solveMs) *solveMs = timeIt([&]() { launchSolveReduced(p, s); });
  if (optOn(kOptCholTiming)) {
    double dbg[6];
    HIP_OK(hipMemcpy(dbg, p.partial + (size_t)15 * 4096, sizeof(dbg), hipMemcpyDeviceToHost));
    std::printf("[chol cycles per launch] load + first pivot tile %.0f  factorisation done (from kernel start) %.0f  backward substitution %.0f (wave 0 asked %.1f times more for a late block)\n",
                dbg[1] / iters, dbg[3] / iters, dbg[4] / iters, dbg[5] / iters);
#ifdef SVIN_CHOL_TIMING
    {
      double w[16];
      HIP_OK(hipMemcpy(w, p.partial + (size_t)15 * 4096 + 32, sizeof(w), hipMemcpyDeviceToHost));
      std::printf("[per wave: cycles until its part of the factorisation was done | failed flag polls (~150-200 cycles each)]");
      for (int k = 0; k < 8; ++k) std::printf("  w%d %.0f | %.0f", k, w[k] / iters, w[8 + k] / iters);
      std::printf("\n");
    }
    double dd[4];
    debugCholTiming(dd, false);
    std::printf("[pivot tiles, cycles per launch] %.0f\n", dd[0] / iters);
    {
      double st[20 * 12];
      HIP_OK(hipMemcpy(st, p.partial + (size_t)15 * 4096 + 64, sizeof(st), hipMemcpyDeviceToHost));
      const char* names[20] = {"w0 pivot start", "w0 pivot end", "w0 pivotDone set", "w0 look-ahead wait (polls)", "w0 past the wait", "w0 panel solved",
                              "w0 xReady set", "owner: look-ahead row ready", "w1 load: issued|arrived|stored|barrier", "w5 (row 6) step start", "w5 pivot seen", "w5 panel tile out",
                              "w5 operands there", "w5 row updated", "back: w0 step start", "back: loop entry", "back: loop exit", "back: requests out",
                              "back: first product done", "back: far part there"};
      std::printf("[stamps of the last launch, cycles from kernel start, per block column]\n");
      for (int w = 0; w < 20; ++w) {
        std::printf("  %-28s", names[w]);
        for (int kb = 0; kb < 11; ++kb) std::printf(" %7.0f", st[w * 12 + kb]);
        std::printf("\n");
      }
    }
#endif
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 1;
}

}  // namespace svin
