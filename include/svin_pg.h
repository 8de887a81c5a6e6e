/* svin_pg.h -- C ABI of the MI355X-native global pose-graph optimisation.
 *
 * Replaces the numerical core of pose_graph's optimisation thread in AutonomousFieldRoboticsLab/SVIn (paths relative
 * to /root/reference/pose_graph/):
 *   PoseGraph::optimize4DoFPoseGraph   src/pose_graph/PoseGraph.cpp:226-385   (yaw + translation per keyframe)
 *   PoseGraph::optimize6DoFPoseGraph   src/pose_graph/PoseGraph.cpp:387-543   (quaternion + translation)
 * i.e. building the Ceres problem from the keyframe list (sequential edges to the 2 / 4 predecessors of the same
 * sequence, loop edges with HuberLoss(0.1), first keyframe constant), ceres::Solve with SPARSE_NORMAL_CHOLESKY and
 * Levenberg-Marquardt (10 / 5 iterations), and writing the poses back.  Loop detection, BRIEF / DBoW2, the keyframe
 * container and the ROS plumbing stay in pose_graph; they hand keyframes over through svin_pg_add_keyframe.
 *
 * Conventions: opaque handle, externally synchronised (the reference holds kflistMutex_ while it builds the problem);
 * positions 3 doubles, quaternions 4 doubles [x y z w] (Eigen coeffs order); angles in degrees where the reference
 * uses degrees (PoseGraph.h:85-127).  Return 1 = ok, 0 = benign false, <0 = error; nothing throws.  There is no CPU
 * fallback: without a HIP device svin_pg_create() returns NULL.
 */
#ifndef SVIN_PG_H_
#define SVIN_PG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svin_pg svin_pg;

/* six_dof = 0: optimize4DoFPoseGraph, 1: optimize6DoFPoseGraph; max_iterations <= 0: the reference's 10 / 5 */
svin_pg* svin_pg_create(int device, int six_dof, int max_iterations);
void svin_pg_destroy(svin_pg* h);
const char* svin_pg_last_error(void);

/* keyframelist.push_back (PoseGraph.cpp addKeyframe): index, sequence, SVIn pose (Keyframe::getSVInPose);
 * loop_index < 0: no loop; otherwise Keyframe::getLoopRelativeT / getLoopRelativeQ / getLoopRelativeYaw
 * (src/pose_graph/Keyframe.cpp:576-582) */
int svin_pg_add_keyframe(svin_pg* h, int index, int sequence, const double* t, const double* q, int loop_index,
                         const double* loop_rel_t, const double* loop_rel_q, double loop_rel_yaw_deg);
int svin_pg_num_keyframes(const svin_pg* h);

/* one pass of the optimisation thread's loop body for cur_index (PoseGraph.cpp:236-351 / :397-516): keyframes with
 * index >= earliest_loop_index up to cur_index are optimised, their poses updated in place */
int svin_pg_optimize(svin_pg* h, int earliest_loop_index, int cur_index);

/* Keyframe::getPose (the drift-corrected / optimised pose; the SVIn pose handed in stays the optimisation's input) */
int svin_pg_get_pose(const svin_pg* h, int k, double* t, double* q);
/* the same for the first n keyframes of the list: t = n x 3, q = n x 4 (what updatePath() walks, PoseGraph.cpp:545-) */
int svin_pg_get_poses(const svin_pg* h, int n, double* t, double* q);

/* yaw_drift (degrees), r_drift (3x3 row-major), t_drift after the last optimisation (PoseGraph.cpp:356-363 /
 * :521-526); svin_pg_optimize applies them to the keyframes after cur_index, svin_pg_add_keyframe to new keyframes
 * (PoseGraph.cpp:127-132) */
int svin_pg_get_drift(const svin_pg* h, double* yaw_drift_deg, double* r_drift, double* t_drift);

/* solver layout (no reference counterpart; Ceres picks its own ordering): keyframes per piece (0 = keep, default
 * 32 for 4-DoF / 64 for 6-DoF, at most 256 / tangent size) and the free-keyframe count up to which the whole graph is solved as one dense
 * system (default 128).  svin_pg_get_partition: free keyframes, separator keyframes, pieces, largest piece (rows),
 * Schur tiles, seconds of the host-side symbolic step, separator unknowns, number of dense separator solves and their
 * total seconds (HIP events on the solver's stream) of the last svin_pg_optimize */
int svin_pg_set_partition(svin_pg* h, int piece_keyframes, int dense_keyframes);
/* 1 = eliminate the pieces only; 2 (default) = also eliminate the cut keyframes in level-2 pieces (of
 * level2_piece_keyframes cut keyframes, 0 = keep, default 16 / 32) so that the dense root only holds the loop cover */
int svin_pg_set_levels(svin_pg* h, int levels, int level2_piece_keyframes);
/* out10: ... as above, [6] = unknowns of the dense root solve, [9] = level-2 pieces */
int svin_pg_get_partition(const svin_pg* h, double* out10);

/* ceres::Solver::Summary of the last solve: initial_cost, final_cost, iterations, termination (0 convergence,
 * 1 no convergence, 3 failure), successful steps, solve seconds (device work, inputs resident) */
int svin_pg_summary(const svin_pg* h, double* out6);

#ifdef __cplusplus
}
#endif
#endif /* SVIN_PG_H_ */
