import sys; sys.path.insert(0,'.')
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
from oracle import orc
spec = syn.make_window(P=8, L=250, n_obs=2500, seed=44, rig="euroc", keyframe_every=2, frame_dt=0.3)
def run(est):
    out=[]
    def on_frame(k, fid):
        est.optimize(25)
        s=est.summary()
        ids=est.frame_ids()
        out.append((s["iterations"], s.get("termination"), s["final_cost"], [est.get_T_WS(i).copy() for i in ids]))
        est.apply_marginalization(2,3)
    syn.feed(est, spec, on_frame=on_frame)
    return out
g=Estimator(0); c=orc.OracleEstimator()
for e in (g,c): e.set_solver_options(1e-12,1e-12,1e-12)
a=run(g); b=run(c)
for k,(x,y) in enumerate(zip(a,b)):
    d=max(np.max(np.abs(p-q)) for p,q in zip(x[3],y[3]))
    print(k, "it", x[0], y[0], "term", x[1], y[1], "cost", x[2], y[2], "dpose", d)
