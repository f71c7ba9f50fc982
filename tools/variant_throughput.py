#!/usr/bin/env python3
"""Throughput of the kernel's instantiations outside the default path on BASELINE C3's scene at 64 spp: another box filter
radius (fixed-point film), samplers 1, 2 and 3 (3.10, 3.12, 3.13), integrator 2 (MIS, 3.14: render_kernel_x), beside the
default.  usage (GPU box): variant_throughput.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

sd = scenes.random_mesh_scene(1_000_000, 2048, 2048)
with pbrt_amd.Scene(sd, builder="gpu") as sc:
    for name, kw in (("default filter, stratified", {}), ("box filter radius 1.5 (fixed-point film)", dict(filter_width=(1.5, 1.5))),
                     ("box filter radius 1.3 (footprints change from sample to sample)", dict(filter_width=(1.3, 1.3))),
                     ("box filter radius 4.0", dict(filter_width=(4.0, 4.0))),
                     ("sampler 1 (padded 0,2)", dict(sampler="sobol")), ("sampler 2 (Sobol' proper, 128 dimensions)", dict(sampler="sobol_nd")),
                     ("sampler 3 (Halton)", dict(sampler="halton")), ("integrator 2 (MIS), stratified", dict(integrator=2)),
                     ("integrator 2 (MIS), Halton", dict(integrator=2, sampler="halton")), ("default again", {})):
        film, st = sc.render(max_depth=8, spp=(8, 8), seed=0, **kw)
        print(f"C3 scene, 2048x2048, 64 spp, {name}: kernel {st['kernel_ms']:.1f} ms, {st['samples'] / st['kernel_ms'] / 1e3:.1f} Msamples/s")
