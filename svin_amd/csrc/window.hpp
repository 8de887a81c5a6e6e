// svin_amd host core: the okvis::Estimator / okvis::ceres::Map mirror behind the C ABI.
//
// The host keeps the *graph* (states, landmark-major observation lists, small factors, the
// marginalisation prior's bookkeeping) and the policy code; every floating-point operation of
// the hot path (residuals, Jacobians, normal equations, Schur complement, solves, retraction,
// landmark quality, IMU prediction, marginalisation algebra) runs in HIP kernels on the device
// buffers described in kernels.hpp.  There is no CPU arithmetic fallback.
//
// Reference (relative to /root/reference/okvis_ros/okvis/okvis_ceres/): src/Estimator.cpp,
// include/okvis/implementation/Estimator.hpp, src/Map.cpp, src/MarginalizationError.cpp.
#pragma once
#include <array>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>
#include "flat_map.hpp"
#include "kernels.hpp"
#include "resident.hpp"

namespace svin {

constexpr int kResidentPoseCap = 42;   // windows up to this many pose blocks keep their observation CSR on the device; wider ones
                                       // (the panel Schur kernels with their host-built work lists) are re-packed by the host

struct TimeStamp {
  uint32_t sec = 0, nsec = 0;
};

struct ExtrinsicsSigmas {
  double abs_t = 0, abs_r = 0, rel_t = 0, rel_r = 0;
};

// a parameter block that is not a landmark
struct Block {
  uint64_t id = 0;
  int kind = B_POSE;  // B_POSE / B_EXT / B_SB
  bool fixed = false;
  unsigned char lock = 0;           // pose / extrinsics blocks: bit k set = tangent direction k is held (Map::Pose3d / Pose4d / Pose2d)
  double x[9] = {0};
  std::vector<uint64_t> residuals;  // ids of the factors / the prior touching it (insertion order)
  int nObs = 0;                     // reprojection residuals touching it (they live in Landmark::obs only)
  int handle = -1;                  // stable small number among the blocks of its kind (recycled): how the device-resident
                                    // observation records name their pose / extrinsics block (resident.hpp)
  std::vector<int> seenLm;          // pose blocks: handles of the landmarks observed from this frame, one entry per observation
                                    // (candidates of the marginalisation policy when the frame leaves; may name dead landmarks)
};

// one cache line per record: addObservation's duplicate check and the marginalisation policy walk these lists
struct Observation {
  uint64_t resId = 0, poseId = 0;
  uint64_t kp = 0;
  double uv[2] = {0, 0};
  double size = 1.0;
  uint32_t pendIdx = 0, pendEpoch = 0;   // position in the add log while the record has not reached the device (resident.hpp)
  uint16_t poseH = 0, extH = 0;           // Block::handle of the pose and the extrinsics block (the extrinsics id: Window::extIdOf)
  uint8_t cam = 0;
};
static_assert(sizeof(Observation) == 64, "Observation is meant to fill one cache line");
// square-root information of a reprojection residual: Estimator::addObservation passes 64 / size^2 * I (implementation/
// Estimator.hpp:69-71); a residual added through the Map interface with information s * I stores -sqrt(s) in `size`
inline double obsWeight(double size) { return size > 0 ? std::sqrt(64.0 / (size * size)) : -size; }

struct Landmark {
  // what addObservation and the marginalisation policy touch comes first (one or two cache lines of the map node)
  std::vector<Observation> obs;  // insertion order
  uint64_t id = 0;
  uint64_t maxPose = 0;            // upper bound of the frame ids among obs: an observation from a newer frame cannot be a duplicate
  uint64_t minPose = UINT64_MAX;   // smallest frame id among obs (frame ids grow with time): lets the marginalisation policy skip a
                                   // landmark nobody in the leaving frames has seen without walking its observation list
  int handle = -1;               // creation number (resident.hpp): CSR order of the device-resident window
  uint32_t visit = 0;            // scratch stamp of the marginalisation policy (a landmark is handled once per call)
  bool initialized = true;       // HomogeneousPointParameterBlock::initialized_ (constructor default, HomogeneousPointParameterBlock.hpp:68)
  bool fixed = false;            // Map::setParameterBlockConstant: its observations are packed with a negative weight (kernels.hip K1)
  double hp[4] = {0, 0, 0, 1};
  double quality = 0, distance = 0;
  // HomogeneousPointError residuals on this landmark (HomogeneousPointError.cpp:48-117): measurement and the
  // upper-triangular square-root information (row-major); not added by okvis::Estimator, available to callers
  struct Prior { uint64_t resId = 0; double meas[4] = {0, 0, 0, 1}; double sqrtInfo[9] = {0}; };
  std::vector<Prior> priors;
};

struct Factor {
  uint64_t id = 0;
  int kind = F_IMU;
  int nblk = 0;
  uint64_t blocks[4] = {0, 0, 0, 0};
  int m = 0;
  double meas[9] = {0};
  double aux[8] = {0};
  double sqrtInfo[81] = {0};
  // IMU
  std::vector<uint32_t> imuT;   // 2 per sample
  std::vector<double> imuMeas;  // 6 per sample
  DevImu imu;                   // persistent pre-integration state (synced back after each solve)
  uint64_t dealKey = 0;         // creation number among the factors of this window (which rank owns it in sharded mode)
  // F_HOST: the caller's cost function (svin_ba.h svin_cost_function) and its context
  int (*hostFn)(void*, const double* const*, double*, double**) = nullptr;
  void* hostUser = nullptr;
};

struct StateInfo { uint64_t id = 0; bool exists = false; };
struct State {
  uint64_t id = 0;
  TimeStamp stamp;
  bool isKeyframe = false;
  StateInfo pose;
  std::vector<StateInfo> ext;  // per camera
  std::vector<StateInfo> sb;   // per imu
};

struct PriorBlockHost {
  uint64_t id = 0;
  int kind = B_POSE;
  int ord = 0, mdim = 0, dim = 0;
  double lin[9] = {0};
};

struct Summary {
  double initial_cost = 0, final_cost = 0;
  int iterations = 0, num_successful_steps = 0, termination = 0;
  double total_time = 0, upload_time = 0, solve_time = 0, download_time = 0;
};

// growable device buffer
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipFree(p);
    size_t c = cap ? cap : 64;
    while (c < n) c *= 2;
    if (hipMalloc(&p, c * sizeof(T)) != hipSuccess) { p = nullptr; cap = 0; throw std::runtime_error("hipMalloc failed"); }
    cap = c;
  }
  // the same, keeping the first `keep` elements (device-to-device copy on `s`) and clearing the rest of the new allocation
  void growKeep(size_t n, size_t keep, hipStream_t s) {
    if (n <= cap) return;
    size_t c = cap ? cap : 64;
    while (c < n) c *= 2;
    T* q = nullptr;
    if (hipMalloc(&q, c * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
    if (keep > cap) keep = cap;
    if (p && keep) (void)hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
    (void)hipMemsetAsync(q + keep, 0, (c - keep) * sizeof(T), s);
    if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); }
    p = q; cap = c;
  }
};

// device buffers of the marginalisation job (marg.hip), kept across calls
struct MargBuffers {
  DevBuf<double> bPose, bExt, bSb, bLm, bUv, bW, bLin, bU, bW2, bV, bVec, bScratch, bImuM, bPartial, bHk, bOut;
  DevBuf<int> bOP, bOE, bOS, bLmPtr, bObsLm, bIdxList, bFlag;
  DevBuf<uint32_t> bIdx, bImuT;
  DevBuf<DevFactor> bFac;
  DevBuf<FactorLin> bFacLin;
  DevBuf<DevImu> bImu;
  DevBuf<SolverScalars> bScal;
};

// inspection copy of the marginalisation system after M1 (SVIN_MARG_KEEP_PRE=1): U m x m, ba, W m x 3Lm, V 9 Lm, bb 3 Lm,
// and per dense row whether it is about to be marginalised
struct MargPre {
  int m = 0, Lm = 0;
  std::vector<double> U, ba, W, V, bb;
  std::vector<int> margRows;
  std::vector<uint64_t> denseIds, lmIds;   // the dense blocks (first row denseOrd, denseMdim rows each) and the landmarks, in order
  std::vector<int> denseOrd, denseMdim;
};

class Window {
 public:
  explicit Window(int device);
  ~Window();

  // Ids.  Upstream every id -- frames, landmarks AND the estimator's internal extrinsics / speed-bias blocks
  // (Estimator.cpp:217,234) -- comes from ONE process-wide okvis::IdProvider.  The core therefore never invents ids
  // on its own authority: it asks the provider the host installed (the shim forwards to IdProvider::instance().newId()),
  // and without one it counts from the largest id it has been shown so far (reserveIds).  A collision is an error.
  typedef uint64_t (*IdProviderFn)(void* user);
  void setIdProvider(IdProviderFn fn, void* user) { idProvider_ = fn; idProviderUser_ = user; }
  void reserveIds(uint64_t maxSeen) { if (maxSeen > idCounter_) idCounter_ = maxSeen; }
  uint64_t newId() { return idProvider_ ? idProvider_(idProviderUser_) : ++idCounter_; }
  bool idInUse(uint64_t id) const { return blocks_.count(id) || lmIndex_.count(id) || states_.count(id); }
  int addCamera(int model, const double* intr, const double* dist, int nDist, int w, int h, const double* sig);
  int setCameraGeometry(size_t cam, int model, const double* intr, const double* dist, int nDist, int w, int h);
  int addImu(const ImuParams& p);
  void clearCameras() { quiesce(); cameras_.clear(); extrinsics_.clear(); }   // Estimator.cpp:91
  void clearImus() { imus_.clear(); }                                // Estimator.cpp:94
  size_t numCameras() const { return cameras_.size(); }
  size_t numImus() const { return imus_.size(); }
  void setSonarExtrinsics(const double* T) { std::memcpy(T_SSo_, T, sizeof(T_SSo_)); }

  int addStates(uint64_t frameId, TimeStamp stamp, uint64_t numKeypoints, const double* T_SC, int nCam,
                const uint32_t* imuT, const double* imuM, int nImu, bool asKeyframe, const double* sonar, int nSonar,
                const double* depth, int nDepth, double firstDepth);
  int addLandmark(uint64_t id, const double* hp);
  uint64_t addObservation(uint64_t lm, uint64_t pose, uint64_t cam, uint64_t kp, const double* uv, double size);
  // the same for a batch (what a frontend holds after matching): identical results, the landmark records are prefetched
  int addObservations(int n, const uint64_t* lm, const uint64_t* pose, const uint64_t* cam, const uint64_t* kp, const double* uv,
                      const double* size, uint64_t* outIds);
  int removeObservation(uint64_t lm, uint64_t pose, uint64_t cam, uint64_t kp);
  int removeObservationById(uint64_t resId);
  // ---- okvis::ceres::Map::addParameterBlock / addResidualBlock / remove* (Map.cpp:255-376, :322-333, :467-492) for callers
  // that build a graph block by block instead of through addStates (the reference's own tests do).  A block added here belongs
  // to no frame; type: 0 = 7-dimensional pose (T_WS or, once a reprojection residual names it as such, extrinsics T_SC),
  // 2 = speed and bias (9), 3 = homogeneous point (4).
  int mapAddParameterBlock(uint64_t id, int type, const double* values);
  int mapSetParameterBlock(uint64_t id, const double* values);
  int mapRemoveParameterBlock(uint64_t id);
  uint64_t mapAddPoseError(uint64_t blockId, const double* meas7, const double* information36);                 // PoseError.cpp:49-132
  uint64_t mapAddSpeedAndBiasError(uint64_t blockId, const double* meas9, const double* information81);         // SpeedAndBiasError.cpp:47-113
  uint64_t mapAddRelativePoseError(uint64_t block0, uint64_t block1, const double* information36);              // RelativePoseError.cpp:48-147
  uint64_t mapAddImuError(const uint64_t ids[4], const uint32_t* imuT, const double* imuM, int nImu, const ImuParams& par, TimeStamp t0,
                          TimeStamp t1);                                                                         // ImuError.cpp:58-75
  uint64_t mapAddSonarError(uint64_t poseBlock, double range, double heading, double information, const double* patch, int nPatch);   // SonarError.cpp:57-183
  uint64_t mapAddDepthError(uint64_t poseBlock, double depth, double information, double firstDepth);             // DepthError.cpp:50-139
  // a residual block whose cost function the HOST evaluates (Map::addResidualBlock with an arbitrary ::ceres::CostFunction, Map.cpp:341-376)
  uint64_t mapAddHostResidual(const uint64_t* blockIds, int nBlocks, int residualDim, int (*fn)(void*, const double* const*, double*, double**), void* user);
  uint64_t mapAddReprojectionError(uint64_t poseBlock, uint64_t landmark, uint64_t extBlock, uint64_t cam, const double* uv,
                                   const double* information4);                                                 // ReprojectionError + CauchyLoss(1)
  int mapRemoveResidualBlock(uint64_t resId);
  // HomogeneousPointError on a landmark (information = 3x3 symmetric positive definite, row-major); 0 on failure
  uint64_t addLandmarkPrior(uint64_t lm, const double* meas4, const double* information9);
  int removeLandmarkPrior(uint64_t resId);
  int optimize(size_t numIter, bool verbose);
  int prepare();
  int solvePrepared(size_t numIter, bool verbose);
  // B prepared windows through ONE launch sequence per trust-region round (the window as a grid dimension; kernels.hpp BatchSlot).
  // Windows whose geometry the batched kernels do not cover, and groups of one, run solve() one after the other; *nBatched = the
  // number of windows that ran in a batch.  Every window ends where solvePrepared() alone would leave it.
  static int solvePreparedBatch(Window* const* ws, int n, size_t numIter, bool verbose, int* nBatched);
  int finish();
  void invalidatePreintegration() { for (auto& kv : factors_) if (kv.second.kind == F_IMU) kv.second.imu.redo = 1; }
  int setOptimizationTimeLimit(double timeLimit, int minIter);
  int applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, std::vector<uint64_t>& removed);
  // integrals (optional, 7 doubles): acc_doubleintegral(3), acc_integral(3), Delta_t -- the second overload of
  // ImuError::propagation (ImuError.cpp:479-697)
  int imuPropagation(const uint32_t* imuT, const double* imuM, int n, const ImuParams& par, double* T, double* sb,
                     TimeStamp t0, TimeStamp t1, double* cov, double* jac, double* integrals = nullptr);
  static bool initPoseFromImu(const double* imuM, int n, double* T);   // Estimator.cpp:848-873

  // getters / setters
  int get_T_WS(uint64_t id, double* T) const;
  int getSpeedAndBias(uint64_t id, size_t imu, double* sb) const;
  int getCameraSensorStates(uint64_t id, size_t cam, double* T) const;
  const Landmark* landmark(uint64_t id) const;   // (fetches the landmark points from the device first when a solve has moved them)
  int set_T_WS(uint64_t id, const double* T);
  int setSpeedAndBias(uint64_t id, size_t imu, const double* sb);
  int setCameraSensorStates(uint64_t id, size_t cam, const double* T);
  int setLandmark(uint64_t id, const double* hp);
  int setLandmarkInitialized(uint64_t id, bool init);                  // Estimator.cpp:1126-1129
  int setKeyframe(uint64_t frameId, bool isKf);                        // Estimator.hpp:444
  int getImuPreIntegral(uint64_t poseId, double* out7) const;          // Estimator.cpp:1001-1014
  void setImuPreIntegral(uint64_t poseId, const double* in7);          // Estimator.cpp:1081-1087 (std::map::insert: first one wins)
  int stateCount() const { return stateCount_; }                       // Estimator.hpp:450
  // okvis::ceres::Map graph queries answered from the core's own graph (Map.cpp:495-620).  Residual ids are the ids
  // addObservation / the factors / the prior were given; block ids are frame ids (pose), internal ids (extrinsics,
  // speed/bias) and landmark ids.
  bool parameterBlockExists(uint64_t id) const { return blocks_.count(id) || lmIndex_.count(id); }   // Map.cpp:77-80
  int setParameterBlockConstant(uint64_t id, bool constant);     // Map.cpp:495-510
  int resetParameterization(uint64_t id, int parameterization);  // Map.cpp:513-543 (values of Map::Parameterization, Map.hpp:97-105)
  int parameterization(uint64_t id) const;
  int isParameterBlockConstant(uint64_t id) const;               // ParameterBlock::fixed()
  int residualsOf(uint64_t blockId, std::vector<uint64_t>& out) const;    // Map::residuals        Map.cpp:576-587
  int parametersOf(uint64_t resId, std::vector<uint64_t>& out) const;     // Map::parameters       Map.cpp:602-620
  int residualKind(uint64_t resId) const;
  // kind, residual dimension and the ambient dimensions of the blocks of a list of residuals in one call (what the shim's
  // Map::residuals / errorInterfacePtr need per residual: sizes and type, ErrorInterface::residualDim / parameterBlockDim)
  int residualInfo(int n, const uint64_t* resIds, int32_t* kind, int32_t* m, int32_t* nBlocks, int32_t* dims4) const;   // -1 unknown, 100 reprojection, 101 marginalisation prior, 102 landmark prior, else FactorKind
  const std::map<uint64_t, State>& states() const { return states_; }
  const std::map<uint64_t, Landmark>& landmarks() const { syncLandmarks(); return landmarks_; }
  size_t numLandmarks() const { return landmarks_.size(); }
  const Landmark* landmarkGraph(uint64_t id) const {   // structure only (observation list, flags): no fetch from the device
    auto it = landmarks_.find(id);
    return it == landmarks_.end() ? nullptr : &it->second;
  }
  const std::map<uint64_t, Landmark>& landmarksGraph() const { return landmarks_; }
  bool landmarkExists(uint64_t id) const { return lmIndex_.count(id); }
  // ---- device-resident window (resident.hpp): 0 = resident whenever the window qualifies (narrow, one GPU, no landmark
  // priors), 1 = always the host path (Window::pack's graph -> array pass and full upload: the reference the tests compare with)
  void setPackMode(int mode) { packMode_ = mode; }
  // inspection: pack() and copy the observation CSR the solver would read (sizes first: call with null pointers)
  int debugCsr(int32_t* L, int32_t* N, int32_t* lmPtr, int32_t* obsLm, uint32_t* obsIdx, double* uv, double* w, double* lm,
               int32_t* obsOrder, int32_t* resident);
  void syncLandmarks() const;   // landmark points / qualities of the last solve: device tables -> host graph (no-op when current)
  uint64_t currentKeyframeId() const;
  uint64_t frameIdByAge(size_t age) const;
  bool isInImuWindow(uint64_t id) const;
  const Summary& summary() const { return summary_; }
  void setTolerances(double f, double g, double p) { fTol_ = f; gTol_ = g; pTol_ = p; }
  // landmark-sharded multi-GPU mode: every rank holds all states and the factors between them, plus its own
  // range of landmarks; `fn` all-reduces `count` doubles at device address `ptr` in place (op 0 = sum, 1 = max).
  typedef int (*AllReduceFn)(void* ptr, uint64_t count, int op, void* user);
  void setDistributed(int rank, int world, AllReduceFn fn, void* user);
  void dropRcclComm();
  // sharded mode: a small factor belongs to the rank its creation number names.  The number is given once (addFactor) and
  // never shifts when other factors leave the window, so an IMU factor's pre-integration state stays with one rank.
  bool ownsFactor(const Factor& f) const { return world_ <= 1 || (int)(f.dealKey % (uint64_t)world_) == rank_; }
  // the same with RCCL called natively on the handle's stream (no host synchronisation, no callback): `id` is the
  // 128-byte ncclUniqueId rank 0 obtained from rcclUniqueId() and the host distributed to every rank
  static int rcclUniqueId(unsigned char* out128);
  int setDistributedRccl(int rank, int world, const unsigned char* id128);

  // inspection hooks
  int evalReprojection(bool robust, double* r, double* Jp, double* Jl, double* Je, int cap);
  int observationIds(uint64_t* rid, uint64_t* lm, uint64_t* pose, int32_t* cam, int cap);
  int evalFactors(int32_t* kind, int32_t* m, int32_t* ncols, double* r, double* J, uint64_t* blocks, uint64_t* rids,
                  int cap);
  int linearize(double mu, double* S, double* g, uint64_t* blockIds, int32_t* blockOff, int32_t* nBlocks, int capD,
                double* cost);
  void waitIdle();
  const long long* pathCounters() const { return pathCounters_; }
  int debugReducedSolve(double mu, double* y, int capD, bool fuseFinalize = false);
  int debugPeekSolverScratch(uint64_t off, uint64_t count, double* out);
  int getPrior(double* H, double* b0, double* J, double* e0, uint64_t* ids, int32_t* ord, int32_t* mdim,
               int32_t* nBlocks, int capM);
  int describeBlock(uint64_t id, uint64_t* frame, int32_t* kind, int32_t* index) const;
  const MargPre& margPre() const { return margPre_; }
  // Map::parameterBlockPtr / id2parameterBlockMap as values (Map.hpp:166-170, :188): type 0 pose, 1 extrinsics, 2 speed/bias,
  // 3 landmark; returns the ambient dimension (7 / 9 / 4) or SVIN_ERR_NOT_FOUND
  int getParameterBlock(uint64_t id, int32_t* type, double* values, uint32_t* sec, uint32_t* nsec, int32_t* fixed, int32_t* initialized) const;
  void parameterBlockIds(std::vector<uint64_t>& out) const;
  int benchJacobianEval(int copies, int iters, double* meanMs, double* bytes, double* backToBackMs = nullptr);
  int benchAllReduce(size_t nDoubles, int iters, double* meanUs);   // native RCCL all-reduce on the solver stream, HIP events
  int benchKernelTimes(int iters, double* evalMs, double* buildMs, double* solveMs);

 private:
  // graph helpers
  Block* addBlock(uint64_t id, int kind, const double* x);   // nullptr when the id is already in use
  void removeBlock(uint64_t id);
  uint64_t addFactor(Factor&& f);
  void removeFactor(uint64_t id);
  void removeObsRecord(Landmark& lm, size_t idx);
  void detachObsRecord(Landmark& lm, const Observation& o);
  void afterObsRemoval(Landmark& lm);
  void eraseLandmark(Landmark& lm);   // (its observations are gone already)
  uint64_t addObservationTo(Landmark& lm, uint64_t pose, uint64_t cam, uint64_t kp, const double* uv, double size);
  uint64_t addObservationRecord(Landmark& lm, Block* pb, Block* eb, uint64_t cam, uint64_t kp, const double* uv, double size);
  uint64_t extIdOf(const Observation& o) const { return blockByHandle_[B_EXT][o.extH]->id; }
  Block* findBlock(uint64_t id);
  const Block* findBlock(uint64_t id) const;

  // device side
  void pack(bool solveFollows = false);   // host graph -> device arrays (sets prob_); solveFollows: called by optimize()
  void downloadStates();       // device tables -> host blocks / landmarks / imu states
  void evaluateAll(bool cand, hipStream_t s);
  void evaluateHostFactors(bool cand, hipStream_t s);   // F_HOST factors: callbacks at the current / candidate blocks, r and J into their FactorLin records
  void solve(size_t numIter, bool verbose);
  static void solveBatchGroup(const std::vector<Window*>& g, size_t numIter, bool verbose);
  void swapStateSets();   // an accepted step: the candidate sets become the current ones
  SolverScalars readScalars();
  // the record of the last evaluation has reached the mailbox (no mailbox: true -- readScalars() synchronises)
  bool scalarsReady() const { return !mailbox_ || *reinterpret_cast<const volatile unsigned long long*>(&mailbox_->seq) == mailboxSeq_; }

  // marginalisation (device algebra in marg.hip)
  friend class Marginalizer;

  int device_ = 0;
  int rank_ = 0, world_ = 1;
  AllReduceFn allreduce_ = nullptr;
  void* allreduceUser_ = nullptr;
  bool distNative_ = false;    // the current solve all-reduces through rcclComm_ (scalars published after the reduction)
  void* rcclComm_ = nullptr;   // ncclComm_t of the native path (RCCL resolved at run time, see window.cpp)
  hipStream_t stream_ = nullptr;
  hipStream_t stream2_ = nullptr;                       // side stream: the early IMU pre-integration of optimize() (pack())
  hipEvent_t evUploaded_ = nullptr, evImuReady_ = nullptr;
  // staged upload of pack(): pinned host block + its device twin (segment table first), see launchScatterStaged
  struct StagedCopy { const void* src; size_t bytes; void* dst; };
  void flushStaged(const std::vector<StagedCopy>& pending, hipStream_t s);
  unsigned char* stageHost_ = nullptr;
  size_t stageHostCap_ = 0;
  DevBuf<unsigned char> stageDev_;
  hipEvent_t stageEvt_ = nullptr;
  uint64_t idCounter_ = 0;
  IdProviderFn idProvider_ = nullptr;
  void* idProviderUser_ = nullptr;
  int stateCount_ = 0;
  std::map<uint64_t, std::array<double, 7>> imuIntegrals_;   // Estimator::imuIntegralsMap_ (Estimator.hpp:404-408)
  std::vector<CameraModel> cameras_;
  std::vector<ExtrinsicsSigmas> extrinsics_;
  std::vector<ImuParams> imus_;
  double T_SSo_[7] = {0, 0, 0, 0, 0, 0, 1};

  std::map<uint64_t, State> states_;
  std::map<uint64_t, Landmark> landmarks_;
  std::unordered_map<uint64_t, Block> blocks_;
  std::map<uint64_t, Factor> factors_;
  FlatMap64 obsRes2Lm_;        // reprojection residual id -> its Landmark node (address; std::map nodes do not move)
  FlatMap64 lmIndex_;          // landmark id -> handle
  std::vector<Landmark*> lmByHandle_;   // handle -> node of landmarks_ (nullptr: gone); std::map nodes do not move
  std::vector<std::map<uint64_t, Landmark>::iterator> lmIterByHandle_;   // the same as iterators (erase without a search)
  int nextLmHandle_ = 0;
  size_t numObs_ = 0, numLmObserved_ = 0;   // reprojection residuals in the graph / landmarks with at least one
  std::vector<int> emptyLm_;   // handles of landmarks that had no observation at some point (the marginalisation policy erases them)
  std::vector<int> freeBlockH_[3];      // recycled Block::handle per kind
  int nextBlockH_[3] = {0, 0, 0};
  std::vector<Block*> blockByHandle_[3];
  uint32_t visitStamp_ = 0;
  // delta of the graph since the last flush to the device (only kept while the device copy is valid)
  std::vector<WinAdd> addLog_;
  std::vector<WinRem> remLog_;
  std::vector<WinLmSet> setLog_;
  uint32_t epoch_ = 1;         // identifies the add log an Observation::pendIdx refers to
  int packMode_ = 0;
  bool residentValid_ = false; // the device holds the window as of the last flush; the logs describe what changed since
  bool residentUsed_ = false;  // the last pack() took the resident path
  long long pathCounters_[4] = {0, 0, 0, 0};   // pack() resident / host; marginalisation tables gathered on the device / built on the host
  mutable bool lmStale_ = false;   // the device's per-handle landmark tables are newer than Landmark::hp / quality
  int hFlushed_ = 0;           // landmark handles below this have device-side values
  // Landmark qualities (Estimator.cpp:902-923) are only ever read through getLandmark(s): the resident path computes them
  // when somebody asks (or before the tables of that solve are overwritten), not at the end of every optimize()
  mutable bool qualityPending_ = false;
  mutable DeviceProblem qualityProb_;
  void flushPendingQuality() const;
  struct Resident {
    DevBuf<double> uv[2], w[2], lmHp, qualH;
    DevBuf<uint32_t> hnd[2], seq[2];
    DevBuf<int> lmPtr[2], handleOfSlot[2], slotOfH[2], obsLm[2], cnt, addsH, addCur, poseSlotOfH, extSlotOfH, margScratch;
    DevBuf<unsigned char> live, poseClass;
    DevBuf<WinAdd> adds;
    DevBuf<WinRem> rems;
    DevBuf<WinLmSet> sets;
    int cur = 0, N = 0, L = 0, H = 0;
  };
  mutable Resident res_;
  int* resStatus_ = nullptr;      // pinned, device-visible: consistency flag of the rebuild / gather kernels
  int* resStatusDev_ = nullptr;
  unsigned char* statesHost_ = nullptr;      // pinned, device-visible: [sequence number | states of the last solve] (k_window_finish)
  unsigned char* statesHostDev_ = nullptr;
  size_t statesHostCap_ = 0;
  unsigned long long statesSeq_ = 0;
  DevBuf<int> finishTicket_;
  mutable double* lmSyncHost_ = nullptr;   // pinned read-back area of syncLandmarks
  unsigned char* imuPropHost_ = nullptr;   // imuPropagation: pinned in / out block and its device twin
  DevBuf<unsigned char> imuPropDev_;
  size_t imuPropCap_ = 0;
  mutable size_t lmSyncCap_ = 0;
  // The launches of an asynchronous device job (the marginalisation's M1-M3: a DMA, the scatter and 6-13 kernels) are issued by
  // a thread of the handle's own while the call that assembled the job returns; quiesce() -- at the top of everything that
  // touches the stream, the job buffers, the staging block or the prior -- waits for it (and rethrows what it threw).
  void enqueueAsync(std::function<void()> job);
  void quiesce() const;
  void enqueueLoop();
  mutable std::thread enqueueThread_;
  mutable std::mutex enqueueMutex_;
  mutable std::condition_variable enqueueCv_;
  mutable std::function<void()> enqueueJob_;
  mutable bool enqueueBusy_ = false, enqueueStop_ = false;
  mutable std::exception_ptr enqueueError_;
  bool useResident() const;
  void invalidateResident();      // the next pack() uploads the whole graph again
  void renumberLandmarkHandles();
  void flushResident(hipStream_t s, bool wantOrder, std::vector<StagedCopy>& pending, ResidentArgs& ra);
  void checkResidentStatus();
  Block* cachedBlock(uint64_t id);
  Block* blockCache_[4] = {nullptr, nullptr, nullptr, nullptr};
  int blockCacheNext_ = 0;
  double lastObsSize_ = 0.0, lastObsWeight_ = 0.0;
  uint64_t obsCachePose_ = 0;     // frame whose extrinsics block ids obsCacheExt_ holds ...
  bool obsCacheValid_ = false;    // ... if any (frame id 0 is a legal id, so the id itself cannot say)
  uint64_t obsCacheExt_[16] = {0};
  std::unordered_map<uint64_t, uint64_t> lmPriorRes2Lm_;  // HomogeneousPointError residual id -> landmark id
  size_t numLandmarkPriors_ = 0;
  size_t numFixedLandmarks_ = 0;
  DevBuf<double> dLmPrior_;
  uint64_t nextResId_ = 1;
  uint64_t factorSeq_ = 0;      // Factor::dealKey of the next factor

  // marginalisation prior (host bookkeeping + device matrices mirrored on the host for the C API)
  bool hasPrior_ = false;
  uint64_t priorResId_ = 0;
  std::vector<PriorBlockHost> priorBlocks_;
  std::vector<double> priorH_, priorB0_, priorJ_, priorE0_;  // H/b0 as marginalised; J,e0 from M3
  bool priorHostValid_ = false;                              // host copies above fetched from the device (getPrior)
  int priorM_ = 0;

  // solver options
  double fTol_ = 1e-6, gTol_ = 1e-10, pTol_ = 1e-8;
  double timeLimit_ = -1.0;
  int minIterations_ = 0;
  bool hasCallback_ = false;
  size_t maxIterationsOption_ = 50;
  Summary summary_;

  // packing maps (valid after pack())
  std::vector<uint64_t> poseIds_, extIds_, sbIds_, lmIds_, factorIds_;
  std::unordered_map<uint64_t, int> poseSlot_, extSlot_, sbSlot_;
  std::vector<uint64_t> redBlockIds_;
  std::vector<int32_t> redBlockOff_;
  DeviceProblem prob_;

  // device buffers
  DevBuf<double> dPose_, dExt_, dSb_, dLm_, dPoseC_, dExtC_, dSbC_, dLmC_;
  DevBuf<int> dLockedRows_;
  std::vector<std::pair<int, uint64_t>> hostFactors_;   // (index in the packed factor table, factor id) of the F_HOST factors
  std::vector<int> hPoseOffKeep_, hExtOffKeep_, hSbOffKeep_;   // pack()'s block -> reduced-row tables (for the records of host factors)
  DevBuf<int> dPoseOff_, dExtOff_, dSbOff_, dLmPtr_, dObsLm_, dPanelWork_, dPanelChunks_, dPanelPairPtr_, dObsOrder_;
  DevBuf<int> dSlotPtr_, dSlotObsPtr_, dSlotObs_, dSlotLm_, dBlkBatch_, dBlkWaveTab_, dBlkRecSlot_;   // wide windows, block-pair Schur form: (landmark, pose) slots (kernels.hpp)
  DevBuf<unsigned short> dSlotBlk_;
  DevBuf<uint32_t> dBlkPairs_;
  DevBuf<double> dSlotRec_, dBlkPartial_;
  DevBuf<CameraModel> dCams_;
  DevBuf<double> dObsUv_, dObsW_;
  DevBuf<uint32_t> dObsIdx_;
  DevBuf<double> dLin_[2];  // r(2N) Jp(12N) Jl(6N) Je(12N) each
  DevBuf<DevFactor> dFactors_;
  DevBuf<FactorLin> dFacLin_[2];
  DevBuf<DevImu> dImus_;
  DevBuf<uint32_t> dImuT_;
  DevBuf<double> dImuM_;
  DevBuf<double> dPriorH_, dPriorBp_, dPriorScratch_;
  DevBuf<PriorBlock> dPriorBlk_;
  DevBuf<double> dS_, dVec_, dLmVec_, dSlabs_, dChol_, dPartial_, dQuality_;
  MargPre margPre_;
  DevBuf<SolverScalars> dScal_;
  MargBuffers margBuf_;
  ScalarMailbox* mailbox_ = nullptr;      // pinned host memory, written by the device
  ScalarMailbox* mailboxDev_ = nullptr;   // its device-side address
  unsigned long long mailboxSeq_ = 0;
  int curSet_ = 0;
  // batched solve (solvePreparedBatch): the slot table of the batch this window leads -- device copy and pinned host copy
  DevBuf<BatchSlot> batchSlotsDev_;
  BatchSlot* batchSlotsHost_ = nullptr;
  size_t batchSlotsHostCap_ = 0;
};

std::string& lastError();

// inspection hook of the prior's eigen-solver (symeig.hpp; marg.hip): eigenvalues (ascending) and eigenvectors (X[i * n + j] =
// component i of vector j) of a symmetric n x n matrix, n <= 128, on the current device.  1, 0 (n out of range), -1 (not finite).
int debugSymEig(int n, const double* A, double* lam, double* X, double* deviceMs);
}  // namespace svin
