#!/usr/bin/env python3
"""Round-2 golden fixtures: SonarError and DepthError, from an INDEPENDENT mpmath restatement of their definitions
(60 digits).  Nothing here calls the oracle or the product.

Definitions restated (reference files, for the reader):
  SonarError   okvis_ceres/src/SonarError.cpp:118-171   e = sqrt(info) * (range - |r_WS - mean(patch)|)
               The Jacobian the reference hands to Ceres is NOT the derivative of that residual: it is
               sqrt(info) * (r_WS - p_sonar) / range on the translation (zero on the rotation), with
               p_sonar = T_WS * T_SSo * [range cos(heading), range sin(heading), 0].  Both are recorded; the fixture
               also stores the true derivative so that a reader can see the difference the reference lives with.
  DepthError   okvis_ceres/src/DepthError.cpp:75-139    e = sqrt(info) * (z_WS - (first_depth - depth)), J = sqrt(info) e_z
  patch selection   okvis_ceres/src/Estimator.cpp:265-316   landmarks inside the +-0.1 m box around p_sonar at the pose
               the frame is ADDED with (T_WS_add), in reverse id order; information 1.0 (sonar), 5.0 (depth)

Run:  python tests/golden/make_golden_r2.py      (writes sonar_depth.npz)
"""
import os

import mpmath as mp
import numpy as np

from make_golden import H, numdiff, pose_plus, qmul, qrot, rand_pose, to_mp  # the same mp helpers (independent of oracle/product)

mp.mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))
# T_SSo of config/config_stereorig_v2.yaml:70-74 as [r | q xyzw]: rotation about z by -90 degrees
T_SSO = [0.015995, 0.125, 0.128, 0.0, 0.0, -np.sqrt(0.5), np.sqrt(0.5)]


def sonar_point(T_WS, T_SSo, rge, hdg):
    q = qmul(T_WS[3:7], T_SSo[3:7])
    n = mp.sqrt(sum(c * c for c in q))
    q = [c / n for c in q]
    r_wso = [a + b for a, b in zip(T_WS[:3], qrot(T_WS[3:7], T_SSo[:3]))]
    p = qrot(q, [rge * mp.cos(hdg), rge * mp.sin(hdg), mp.mpf(0)])
    return [a + b for a, b in zip(r_wso, p)]


def main():
    rng = np.random.default_rng(20250930)
    rows = []
    for c in range(8):
        T_add = np.r_[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]          # initPoseFromImu of a level, resting sensor
        T_eval = rand_pose(rng, 0.3, 0.2)
        rge, hdg = rng.uniform(0.6, 3.0), rng.uniform(-np.pi, np.pi)
        k = int(rng.integers(5, 31))
        p_add = np.array([float(x) for x in sonar_point(to_mp(T_add), to_mp(T_SSO), mp.mpf(float(rge)), mp.mpf(float(hdg)))])
        patch = p_add + rng.uniform(-0.09, 0.09, size=(k, 3))
        outside = p_add + np.array([[0.11, 0.0, 0.0], [0.0, -0.12, 0.05], [0.05, 0.05, 0.101]])  # must NOT be selected
        mT, mS = to_mp(T_eval), to_mp(T_SSO)
        mr, mh = mp.mpf(float(rge)), mp.mpf(float(hdg))
        mean = [sum(mp.mpf(float(v)) for v in patch[:, a]) / k for a in range(3)]

        def err(Tx):
            d = [Tx[a] - mean[a] for a in range(3)]
            return [mr - mp.sqrt(sum(x * x for x in d))]          # sqrt(information = 1.0)
        sp = sonar_point(mT, mS, mr, mh)
        J_ref = np.array([float((mT[a] - sp[a]) / mr) for a in range(3)] + [0.0, 0.0, 0.0])
        J_true = numdiff(lambda d: err(pose_plus(mT, d)), 6)[0]
        depth, first_depth = rng.uniform(0.5, 5.0), rng.uniform(-0.5, 0.5)
        s5 = mp.sqrt(mp.mpf(5))
        rows.append(dict(T_add=T_add, T_eval=T_eval, range=rge, heading=hdg, npatch=k,
                         patch=np.vstack([patch, np.full((30 - k, 3), np.nan)]), outside=outside,
                         r=float(err(mT)[0]), J_ref=J_ref, J_true=J_true, p_sonar=np.array([float(x) for x in sp]),
                         depth=depth, first_depth=first_depth,
                         depth_r=float(s5 * (mT[2] - (mp.mpf(float(first_depth)) - mp.mpf(float(depth))))),
                         depth_J=np.array([0.0, 0.0, float(s5), 0.0, 0.0, 0.0])))
    out = {"T_SSo": np.array(T_SSO)}
    for key in rows[0]:
        out[key] = np.array([row[key] for row in rows])
    np.savez(os.path.join(HERE, "sonar_depth.npz"), **out)
    print("wrote sonar_depth.npz:", len(rows), "cases; |J_ref - J_true| up to",
          max(np.max(np.abs(r["J_ref"] - r["J_true"])) for r in rows))


if __name__ == "__main__":
    main()
