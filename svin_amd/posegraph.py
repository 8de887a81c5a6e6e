"""Host-side mirror of pose_graph's optimisation entry points over the C ABI in include/svin_pg.h.

``PoseGraph`` plays the role of the keyframe list + optimisation thread of
/root/reference/pose_graph/src/pose_graph/PoseGraph.cpp (addKeyframe :70-224, optimize4DoFPoseGraph :226-385,
optimize6DoFPoseGraph :387-543); the numerics run on the GPU inside libsvin_ba.so -- there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import estimator

PG_EXPORTS = [
    "svin_pg_create", "svin_pg_destroy", "svin_pg_last_error", "svin_pg_add_keyframe", "svin_pg_num_keyframes",
    "svin_pg_optimize", "svin_pg_get_pose", "svin_pg_get_poses", "svin_pg_get_drift", "svin_pg_summary",
    "svin_pg_set_partition", "svin_pg_get_partition", "svin_pg_set_levels",
]

_BOUND = False


def _lib():
    global _BOUND
    L = estimator.load_library()
    if not _BOUND:
        vp, i32, f64, pd = C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double)

        def sig(name, res, *args):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = list(args)
        sig("svin_pg_create", vp, i32, i32, i32)
        sig("svin_pg_destroy", None, vp)
        sig("svin_pg_last_error", C.c_char_p)
        sig("svin_pg_add_keyframe", i32, vp, i32, i32, pd, pd, i32, pd, pd, f64)
        sig("svin_pg_num_keyframes", i32, vp)
        sig("svin_pg_optimize", i32, vp, i32, i32)
        sig("svin_pg_get_pose", i32, vp, i32, pd, pd)
        sig("svin_pg_get_poses", i32, vp, i32, pd, pd)
        sig("svin_pg_get_drift", i32, vp, pd, pd, pd)
        sig("svin_pg_summary", i32, vp, pd)
        sig("svin_pg_set_partition", i32, vp, i32, i32)
        sig("svin_pg_get_partition", i32, vp, pd)
        sig("svin_pg_set_levels", i32, vp, i32, i32)
        _BOUND = True
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PoseGraph:
    def __init__(self, device=0, six_dof=False, max_iterations=0):
        self.L = _lib()
        self.six = bool(six_dof)
        self.h = self.L.svin_pg_create(device, 1 if six_dof else 0, max_iterations)
        if not self.h:
            raise RuntimeError("svin_pg_create failed: " + self.L.svin_pg_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.svin_pg_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.svin_pg_last_error().decode()))
        return rc

    def add_keyframe(self, index, sequence, t, q, loop=None):
        """loop = (loop_index, rel_t[3], rel_q[4] xyzw, rel_yaw_deg) or None"""
        t = np.ascontiguousarray(t, np.float64)
        q = np.ascontiguousarray(q, np.float64)
        if loop is None:
            self._check(self.L.svin_pg_add_keyframe(self.h, index, sequence, _p(t), _p(q), -1, None, None, 0.0), "add_keyframe")
        else:
            li, rt, rq, ry = loop
            rt = np.ascontiguousarray(rt, np.float64)
            rq = np.ascontiguousarray(rq, np.float64)
            self._check(self.L.svin_pg_add_keyframe(self.h, index, sequence, _p(t), _p(q), int(li), _p(rt), _p(rq), float(ry)),
                        "add_keyframe")

    def optimize(self, earliest_loop_index, cur_index):
        self._check(self.L.svin_pg_optimize(self.h, earliest_loop_index, cur_index), "optimize")
        return self.summary()

    def summary(self):
        s = np.zeros(6)
        self._check(self.L.svin_pg_summary(self.h, _p(s)), "summary")
        return dict(initial_cost=s[0], final_cost=s[1], iterations=int(s[2]), termination=int(s[3]), successful=int(s[4]),
                    solve_seconds=s[5])

    def poses(self):
        n = self.L.svin_pg_num_keyframes(self.h)
        T, Q = np.zeros((n, 3)), np.zeros((n, 4))
        self._check(self.L.svin_pg_get_poses(self.h, n, _p(T), _p(Q)), "get_poses")
        return T, Q

    def drift(self):
        y, r, t = np.zeros(1), np.zeros((3, 3)), np.zeros(3)
        self._check(self.L.svin_pg_get_drift(self.h, _p(y), _p(r), _p(t)), "get_drift")
        return float(y[0]), r, t

    def set_partition(self, piece_keyframes=0, dense_keyframes=128):
        self._check(self.L.svin_pg_set_partition(self.h, piece_keyframes, dense_keyframes), "set_partition")

    def partition(self):
        s = np.zeros(10)
        self._check(self.L.svin_pg_get_partition(self.h, _p(s)), "get_partition")
        return dict(free=int(s[0]), separators=int(s[1]), pieces=int(s[2]), max_rows=int(s[3]), tiles=int(s[4]),
                    symbolic_seconds=float(s[5]), separator_unknowns=int(s[6]), dense_solves=int(s[7]),
                    dense_solve_seconds=float(s[8]), level2_pieces=int(s[9]))

    def set_levels(self, levels=2, level2_piece_keyframes=0):
        self._check(self.L.svin_pg_set_levels(self.h, levels, level2_piece_keyframes), "set_levels")
