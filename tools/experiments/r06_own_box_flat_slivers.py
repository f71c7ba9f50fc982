#!/usr/bin/env python3
"""Round 6: what decided HOW the own-box rule of DESIGN.md 3.5 treats a candidate that lies before its box's entry.  Triangles lying flat in an axis
plane (zero-thickness boxes: entry = exit = the plane's slab distance): fp32 Moeller-Trumbore's t differs from that distance by rounding x the
triangle's condition number.  R = the first form of the rule (reject unless entry <= t x (1 + 2^-17)): 7 % / 44 % of the hits on flat slivers of
condition ~200 / ~2000 rejected -- holes in walls.  R' = the adopted form (the ray must meet the box; t is RAISED to the entry): none rejected.
numpy float32 in the operation order of 3.5.  Output: profiles/r06_own_box_flat_slivers.txt"""
import numpy as np
f=np.float32
rng=np.random.default_rng(2)
N=1_000_000
def run(p0,p1,p2,label):
    p0=np.array(p0,dtype=f);p1=np.array(p1,dtype=f);p2=np.array(p2,dtype=f)
    o=(rng.uniform(0,10,(N,3))*[0.45,1,1]).astype(f)
    b=rng.uniform(0,1,(N,2)); b[b.sum(1)>1]=1-b[b.sum(1)>1]
    tgt=(p0+(p1-p0)*b[:,:1].astype(f)+(p2-p0)*b[:,1:].astype(f)).astype(f)
    d=(tgt-o).astype(f); d=(d/np.sqrt((d[:,0]*d[:,0]+d[:,1]*d[:,1])+d[:,2]*d[:,2])[:,None]).astype(f)
    def dot(a,b): return ((a[...,0]*b[...,0]+a[...,1]*b[...,1])+a[...,2]*b[...,2]).astype(f)
    def cross(a,b): return np.stack([(a[...,1]*b[...,2])-(a[...,2]*b[...,1]),(a[...,2]*b[...,0])-(a[...,0]*b[...,2]),(a[...,0]*b[...,1])-(a[...,1]*b[...,0])],-1).astype(f)
    e1=(p1-p0).astype(f);e2=(p2-p0).astype(f)
    E1=np.broadcast_to(e1,d.shape);E2=np.broadcast_to(e2,d.shape)
    pv=cross(d,E2);det=dot(E1,pv);idet=(f(1)/det).astype(f);tv=(o-p0).astype(f);u=(dot(tv,pv)*idet).astype(f);qv=cross(tv,E1);v=(dot(d,qv)*idet).astype(f);th=(dot(E2,qv)*idet).astype(f)
    valid=(np.abs(det)>=1e-8)&(u>=0)&(v>=0)&(u+v<=1)&(th>1e-4)
    inv=(f(1)/d).astype(f)
    V=np.stack([p0,p1,p2])
    with np.errstate(all='ignore'):
        s=((V[None,:,:]-o[:,None,:])*inv[:,None,:]).astype(f)
    near=np.fmin(np.fmin(s[:,0],s[:,1]),s[:,2]);far=np.fmax(np.fmax(s[:,0],s[:,1]),s[:,2])
    tn=np.fmax(np.fmax(near[:,0],near[:,1]),np.fmax(near[:,2],f(1e-4)))
    tfb=np.fmin(np.fmin(far[:,0],far[:,1]),far[:,2])
    for name,pad in (("R  pad 1+2^-17",f(1+2.0**-17)),):
        tf=np.fmin(tfb,th); ok=tn<=(tf*pad).astype(f)
        print(label,name,"valid",valid.sum(),"rejected",(valid&~ok).sum(),"rate %.2e"%((valid&~ok).sum()/max(1,valid.sum())))
    for name,pad in (("R' pad 1+2^-21",f(1+2.0**-21)),("R' pad 1",f(1))):
        ok=tn<=(tfb*pad).astype(f)
        print(label,name,"valid",valid.sum(),"rejected",(valid&~ok).sum(),"rate %.2e"%((valid&~ok).sum()/max(1,valid.sum())), " snapped", (valid&ok&(tn>th)).sum())
run((5,0,0),(5,10,0),(5,10,10),"flat wall            ")
run((5,0,0),(5,10,1),(5,10,1.01),"flat sliver kappa~200")
run((5,0,0),(5,10,1),(5,10,1.001),"flat sliver kappa~2000")
run((5,0,0),(5.3,10,1),(4.9,10,1.01),"fat-box sliver       ")
run((5,0,0),(5.3,10,0),(4.9,10,10),"fat triangle         ")
