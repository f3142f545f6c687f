#!/usr/bin/env python3
"""Would trimming a lane's stencil-row segments by the cubes its cell cannot reach (corner / edge cubes further than the
cut-off from the cell) shorten phase 1 of ya::grid_force_bits?  Candidates per lane fall by 24 % on average, the
wavefront's cost (its slowest lane, per plane) by 1.4 %: not built.  CPU only; DESIGN.md section 6, round 6 item 5."""
import numpy as np
rng=np.random.default_rng(1)
# uniform density 9.8 in a box 24^3, cubes size 1
L=24; n=int(9.8*L**3)
X=rng.random((n,3))*L
cube=np.floor(X).astype(int)
cid=cube[:,0]+L*cube[:,1]+L*L*cube[:,2]
order=np.lexsort((np.arange(n),cid)); X=X[order]; cube=cube[order]; cid=cid[order]
counts=np.bincount(cid,minlength=L**3).reshape(L,L,L)  # [z][y][x]
def cnt(x,y,z):
    ok=(x>=0)&(x<L)&(y>=0)&(y<L)&(z>=0)&(z<L)
    return np.where(ok,counts[np.clip(z,0,L-1),np.clip(y,0,L-1),np.clip(x,0,L-1)],0)
f=X-cube  # fractional
glo=f; ghi=1-f
m=1e-4
def gap(axis,d):
    g=np.where(d==0,0.0,np.where(d<0,glo[:,axis],ghi[:,axis]))
    return np.maximum(g-m,0)**2*(d!=0)
full=np.zeros((n,3)); trim=np.zeros((n,3))
for zi,dz in enumerate((0,-1,1)):
    for dy in (0,-1,1):
        base=gap(1,np.full(n,dy))+gap(2,np.full(n,dz))
        row_skip=base>1.0
        for dx in (-1,0,1):
            c=cnt(cube[:,0]+dx,cube[:,1]+dy,cube[:,2]+dz)
            full[:,zi]+=c
            skip=row_skip|((base+gap(0,np.full(n,dx)))>1.0)
            trim[:,zi]+=np.where(skip,0,c)
# interior tiles only
inner=np.all((cube>=2)&(cube<L-2),axis=1)
T=n//64
F=full[:T*64].reshape(T,64,3); R=trim[:T*64].reshape(T,64,3); I=inner[:T*64].reshape(T,64).all(1)
r4=lambda a: np.ceil(a/4)*4
print("mean cand per cell full %.1f trimmed %.1f"%(full[inner].sum(1).mean(), trim[inner].sum(1).mean()))
print("wave cost (sum over planes of max lane) full %.1f trimmed %.1f  ratio %.3f"%(r4(F[I]).max(1).sum(1).mean(), r4(R[I]).max(1).sum(1).mean(), r4(R[I]).max(1).sum(1).mean()/r4(F[I]).max(1).sum(1).mean()))
