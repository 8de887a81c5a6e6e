"""Host model of the MFMA products k_schur_panels executes (kernels.hip: per 16-landmark chunk and panel pair, a 16 x 16 x 4
product step runs when both tile rows hold something in those four columns of G) against the algorithmic count sum_l 3 (6 n_l)^2,
for different orders of the landmarks of the configs[3] bench window.  Reproduces the counter measurement (9.3,
profiles/r04_config4_mfma.json) and is what chose the order Window::pack() uses for wide windows.  python tools/panel_order_model.py"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from svin_amd import synthetic as syn
def executed(order, lm_ptr_sorted, pose_of_obs_by_lm, kRows=96):
    """order: landmark indices in CSR order. returns executed MFMA count, histogram of live steps"""
    tot=0; L=len(order); hist=np.zeros(13,int)
    for c0 in range(0,L,16):
        lms=order[c0:c0+16]
        kmask={}   # (panel, tile) -> bits
        lo=10**9; hi=-1
        for g,l in enumerate(lms):
            kb=(1<<((3*g)>>2))|(1<<((3*g+2)>>2))
            for pz in pose_of_obs_by_lm[l]:
                off=6*pz
                lo=min(lo,off); hi=max(hi,off)
                I=off//kRows; rl=off-I*kRows
                for tr in {rl>>4,(rl+5)>>4}:
                    kmask[(I,tr)]=kmask.get((I,tr),0)|kb
        if hi<0: continue
        pLo,pHi=lo//kRows,hi//kRows
        for I in range(pLo,pHi+1):
            for J in range(pLo,I+1):
                for ti in range(6):
                    a=kmask.get((I,ti),0)
                    if not a: continue
                    for tj in range(6):
                        if I==J and ti<tj: continue
                        b=kmask.get((J,tj),0)
                        if not b: continue
                        n=bin(a&b).count("1"); tot+=n; hist[n]+=1
    return tot,hist
if __name__=="__main__":
    sp=syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25)
    L=sp.L
    byl=[[] for _ in range(L)]
    for l,f in zip(sp.obs_lm,sp.obs_frame): byl[l].append(int(f))
    n_l=np.array([len(b) for b in byl],float)
    alg=float(np.sum(3.0*(6.0*n_l)**2))
    # distinct poses per landmark
    first=np.array([min(b) if b else 0 for b in byl]); last=np.array([max(b) if b else 0 for b in byl])
    npose=np.array([len(set(b)) for b in byl])
    print("obs per lm mean %.1f, distinct poses mean %.1f, span mean %.1f"%(n_l.mean(),npose.mean(),(last-first+1).mean()))
    seen=np.nonzero(n_l>0)[0]
    def report(name,order):
        ex,h=executed(order,None,byl)
        print("%-40s executed/algorithmic %.2f  (MFMA %d) live-steps hist %s"%(name,ex*2048/alg,ex,h.tolist()))
    o_first=seen[np.argsort(first[seen],kind='stable')]
    report("first pose (current)",o_first)
    o_fl=seen[np.lexsort((last[seen],first[seen]))]
    report("first, then last",o_fl)
    o_lf=seen[np.lexsort((first[seen],last[seen]))]
    report("last, then first",o_lf)
    # tile-row signature: bitmask of 16-row tiles touched
    def tilesig(b):
        m=0
        for pz in set(b):
            off=6*pz
            m|=1<<(off>>4); m|=1<<((off+5)>>4)
        return m
    sig=np.array([tilesig(b) for b in byl],dtype=object)
    firstt=np.array([(int(s)&-int(s)).bit_length() if s else 0 for s in sig]); lastt=np.array([int(s).bit_length() for s in sig])
    o_t=seen[np.lexsort((lastt[seen],firstt[seen]))]
    report("first tile row, last tile row",o_t)
    o_sig=np.array(sorted(seen,key=lambda l:(firstt[l],lastt[l],int(sig[l]))))
    report("first tile, last tile, signature",o_sig)
    mid=(first+last)/2.0
    o_mid=seen[np.lexsort(((last-first)[seen],np.round(mid[seen]*1).astype(int)))]
    report("centre, then span",o_mid)
    o_sp=seen[np.lexsort((first[seen],(last-first)[seen]))]
    report("span, then first",o_sp)
