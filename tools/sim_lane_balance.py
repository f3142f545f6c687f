#!/usr/bin/env python3
"""Cost model of an in-wave balancing scheme for phase 2 of ya::grid_force_bits on real hit counts (uniform points at
rho = 9.8, cube-sorted, 64 consecutive cells per wavefront, hits per stencil plane): every lane evaluates its own first
C0 = mean + delta hits as today; the excess hits of all lanes are listed, dealt to the lanes in rounds of 64, evaluated
(EVAL instructions each: owner context from LDS, term parked in LDS) and added by their owners in order (ADD each).
Prints today's cost, the free-balance bound (tools/ab/bits_probe.inc measures that one on the GPU) and the scheme's.
CPU only.  DESIGN.md section 6, round 6 item 5."""
import numpy as np
from scipy.spatial import cKDTree
rng=np.random.default_rng(2)
L=28; n=int(9.8*L**3)
X=rng.random((n,3))*L
cube=np.floor(X).astype(int)
cid=cube[:,0]+L*cube[:,1]+L*L*cube[:,2]
order=np.lexsort((np.arange(n),cid)); X=X[order]; cube=cube[order]
tree=cKDTree(X); pairs=tree.query_pairs(1.0,output_type='ndarray')
h=np.zeros((n,3),int)
cz=cube[:,2]
for a,b in ((pairs[:,0],pairs[:,1]),(pairs[:,1],pairs[:,0])):
    dz=cz[b]-cz[a]; idx=np.where(dz==0,0,np.where(dz==-1,1,2)); np.add.at(h,(a,idx),1)
h[:,0]+=1
inner=np.all((cube>=2)&(cube<L-2),axis=1)
T=n//64
H=h[:T*64].reshape(T,64,3); I=inner[:T*64].reshape(T,64).all(1); H=H[I]
PAIR=65.0
cur=(np.ceil(H.max(1)/2)*2*PAIR).sum(1)
ideal=(np.ceil(H.mean(1)/2)*2*PAIR).sum(1)
print("tiles",len(H),"current %.0f  free-balance %.0f (%.3f)"%(cur.mean(),ideal.mean(),ideal.mean()/cur.mean()))
def scheme(H,delta,LIST=10,EVAL=85,ADD=10,FIX=60):
    tot=np.zeros(len(H))
    for p in range(3):
        hp=H[:,:,p]
        C0=np.ceil((hp.mean(1)+delta)/2)*2
        C0=np.minimum(C0,np.ceil(hp.max(1)/2)*2)
        over=np.maximum(hp-C0[:,None],0)
        cost=C0*PAIR
        # rounds of 64 items; per-round imbalance: approx max over owners of items in the round
        tot_over=over.sum(1)
        rounds=np.ceil(tot_over/64)
        # items of an owner are contiguous: max per-owner items in a round <= max overflow
        mx=over.max(1)
        per_round_max=np.where(rounds>0, np.minimum(mx, 64), 0)
        # if several rounds, an owner's items may split; approximate per-round max by mx for first round and mx/2 for others
        cost+= np.where(rounds>0, FIX*rounds + EVAL*rounds + (LIST+ADD)*mx*np.minimum(rounds,1) + (LIST+ADD)*np.maximum(rounds-1,0)*mx*0.5, 0)
        tot+=cost
    return tot
for d in (-1,0,1,2,3,4):
    t=scheme(H,d)
    print("delta %+d: scheme %.0f  ratio %.3f"%(d,t.mean(),t.mean()/cur.mean()))

# regrouping: blocks of 256 consecutive cells, sorted by total hits, dealt to 4 waves; one running pass over all planes
T4=(len(H)//4)*4
B=H[:T4].reshape(-1,4,64,3)
tot=B.sum(3).reshape(len(B),256)
tot_sorted=np.sort(tot,axis=1)[:,::-1].reshape(len(B),4,64)
regroup=(np.ceil(tot_sorted.max(2)/2)*2*PAIR).sum(1)            # per block of 4 waves
cur4=(np.ceil(B.max(2)/2)*2*PAIR).sum(2).sum(1)
merged=(np.ceil(B.sum(3).max(2)/2)*2*PAIR).sum(1)               # merged planes only, no regrouping
print("per 256 cells: current %.0f  merged planes %.0f (%.3f)  regrouped+merged %.0f (%.3f)  ideal %.0f"%(cur4.mean(), merged.mean(), merged.mean()/cur4.mean(), regroup.mean(), regroup.mean()/cur4.mean(), (np.ceil(tot.mean(1)/2)*2*PAIR*4).mean()))
