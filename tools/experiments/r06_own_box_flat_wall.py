#!/usr/bin/env python3
"""Round 6: how far fp32 Moeller-Trumbore's t lies from the slab distance of the SAME plane for triangles that lie flat in an axis plane (every wall of a
Cornell box), in units of 2^-24 -- what decided the pads of DESIGN.md 3.4 / 3.5: the own-box rule with pbrt-v3's 1 + 2 gamma_3 (6 u) would sit within 0.2 u
of rejecting legitimate hits on such walls (deviation up to 5.8 u in 2 M rays); with kOwnPad = 1 + 2^-17 (128 u) nothing is rejected.  numpy float32 in
the operation order of DESIGN.md 3.5.  Output: profiles/r06_own_box_flat_wall.txt"""
import numpy as np
f=np.float32
rng=np.random.default_rng(1)
N=2_000_000
# wall x=5 (axis aligned), triangle p0=(5,0,0) p1=(5,10,0) p2=(5,10,10)  (second: p0,(5,10,10),(5,0,10))
def run(p0,p1,p2,label,scale=1.0,shift=0.0):
    p0=(np.array(p0,dtype=f)*f(scale)+f(shift)).astype(f);p1=(np.array(p1,dtype=f)*f(scale)+f(shift)).astype(f);p2=(np.array(p2,dtype=f)*f(scale)+f(shift)).astype(f)
    o=(rng.uniform(0,10,(N,3))*[0.45,1,1]).astype(f)*f(scale)+f(shift); o=o.astype(f)
    tgt=(np.stack([np.full(N,5.0),rng.uniform(0,10,N),rng.uniform(0,10,N)],1)).astype(f)*f(scale)+f(shift)
    d=(tgt.astype(f)-o).astype(f); d=(d/np.sqrt((d[:,0]*d[:,0]+d[:,1]*d[:,1])+d[:,2]*d[:,2])[:,None]).astype(f)
    def dot(a,b): return ((a[...,0]*b[...,0]+a[...,1]*b[...,1])+a[...,2]*b[...,2]).astype(f)
    def cross(a,b): return np.stack([(a[...,1]*b[...,2])-(a[...,2]*b[...,1]),(a[...,2]*b[...,0])-(a[...,0]*b[...,2]),(a[...,0]*b[...,1])-(a[...,1]*b[...,0])],-1).astype(f)
    e1=(p1-p0).astype(f);e2=(p2-p0).astype(f)
    pv=cross(d,np.broadcast_to(e2,d.shape));det=dot(np.broadcast_to(e1,d.shape),pv)
    idet=(f(1)/det).astype(f);tv=(o-p0).astype(f);u=(dot(tv,pv)*idet).astype(f);qv=cross(tv,np.broadcast_to(e1,d.shape));v=(dot(d,qv)*idet).astype(f);th=(dot(np.broadcast_to(e2,d.shape),qv)*idet).astype(f)
    valid=(np.abs(det)>=1e-8)&(u>=0)&(v>=0)&(u+v<=1)&(th>1e-4)
    inv=(f(1)/d).astype(f)
    lo=np.minimum(np.minimum(p0,p1),p2);hi=np.maximum(np.maximum(p0,p1),p2)
    with np.errstate(all='ignore'):
        a0=((lo-o)*inv).astype(f);a1=((hi-o)*inv).astype(f)
    near=np.fmin(a0,a1);far=np.fmax(a0,a1)
    btn=np.fmax(np.fmax(near[:,0],near[:,1]),np.fmax(near[:,2],f(1e-4)))
    for padname,pad in (("1+2g3",f(1.0000003576)),("1+2^-17",f(1+2.0**-17)),("1+2^-20",f(1+2.0**-20))):
        btf=np.fmin(np.fmin(far[:,0],far[:,1]),np.fmin(far[:,2],th))
        ok=btn<=(btf*pad).astype(f)
        print(label,padname,"valid",valid.sum(),"rejected by R",(valid&~ok).sum(), "rate %.2e"%((valid&~ok).sum()/max(1,valid.sum())))
    rel=((th-btn)/th)[valid]
    print("  rel dev th vs btn (units of 2^-24): min %.1f max %.1f"%(rel.min()*2**24, rel.max()*2**24))
run((5,0,0),(5,10,0),(5,10,10),"flat wall")
run((5,0,0),(5,10,0),(5,10,10),"flat wall @+1000",1.0,1000.0)
run((5,0,0),(5,10,0),(5,3.7,9.1),"flat wall general e2")
run((5,0,0),(5.000001,10,0),(5,10,10),"1-ulp-ish tilt")
run((5,0,0),(5.3,10,0),(4.9,10,10),"fat")
