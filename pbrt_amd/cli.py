"""Command line of the render path, mirroring the reference binary (src/bin/pbrt.rs:24-44):

    python -m pbrt_amd.cli [-n N] [--quick] [-q] [-v] [-o FILE] scene.pbrt ...

parses each scene (the C++ parser behind pbrt_hip_load_file), renders it on the GPU through
pbrt_hip_render (what `world_end` would do) and writes the image named by Film "string filename"
(or -o), PNG or PFM, via pbrt_hip_write_image."""
import argparse
import os
import sys
import time

from . import api, loader


def main(argv=None):
    ap = argparse.ArgumentParser(prog="pbrt", description="Parses a scene file and renders it on an MI355X.")
    ap.add_argument("-n", "--nthreads", type=int, default=1, help="accepted for compatibility; the GPU path ignores it (the reference never reads it either, lib.rs:61)")
    ap.add_argument("--quick", action="store_true", help="quick render: a quarter of the samples per pixel")
    ap.add_argument("-q", "--quiet", action="store_true", help="squelch all non-error output")
    ap.add_argument("-v", "--verbose", action="store_true", help="enable extra logging output")
    ap.add_argument("-o", "--outfile", default="", help="path to store the rendered output")
    ap.add_argument("scene_files", nargs="*")
    args = ap.parse_args(argv)

    def log(level, msg):  # quiet=1 / default=2 / verbose=3 (bin/pbrt.rs:48-62)
        if (args.verbose and level <= 3) or (not args.quiet and level <= 2) or level <= 1:
            print(msg, file=sys.stderr)

    if not args.scene_files:
        log(1, "no scene files given")
        return 1
    for path in args.scene_files:
        try:
            ls = loader.load_file(path)
        except api._lib.PbrtHipError as e:
            log(1, f"{path}: {e}")
            return 1
        for w in ls.warnings:
            log(1, f"warning: {w}")  # WARN passes every level (pbrt.rs:51-53: quiet is "only WARN and higher")
        kw = ls.render_kwargs()
        if args.quick:
            kw["spp"] = (max(1, kw["spp"][0] // 2), max(1, kw["spp"][1] // 2))
        sd = ls.scene
        log(2, f"{path}: {sd.idx.shape[0]} triangles, {len(sd.spheres)} spheres, {len(sd.lights)} lights, "
               f"{sd.xres}x{sd.yres}, {kw['spp'][0] * kw['spp'][1]} spp, integrator {ls.names['integrator']}")
        t0 = time.time()
        with api.Scene(sd) as sc:
            film, st = sc.render(**kw)
        rgb = api.film_to_rgb(film, scale=ls.film_scale)  # Film::write_image multiplies by Film "scale" (film.rs:368-371)
        out = args.outfile or ls.filename
        if not os.path.splitext(out)[1]:
            out += ".png"
        api.write_image(out, rgb)
        log(2, f"wrote {out}: kernel {st['kernel_ms']:.1f} ms ({st['samples'] / st['kernel_ms'] / 1e3:.1f} Msamples/s), "
               f"total {time.time() - t0:.2f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
