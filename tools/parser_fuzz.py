#!/usr/bin/env python3
"""Mutated scene files through the parser (pbrt_hip_parse_*: C++ behind the C ABI, no device): insertions of tokens the grammar knows and of
garbage, deletions, truncations, duplicated stretches, numbers replaced by 0 / -1 / 2^32 / 2^24 + 1 / 1e30 -- every file must either load or
be refused with PbrtHipError (an error code and a message through the ABI); a crash or an exception of another kind is a finding.
python3 tools/parser_fuzz.py SEED N      (profiles/r06u_parser_fuzz.txt: 1.2 M files; tests/test_parser.py runs 4 000)
python3 tools/parser_fuzz.py SEED N --hip   on a GPU box: every file that loads goes on to pbrt_hip_scene_create (the device builder) and, when
its film is at most 262 144 pixels, one sample per pixel at depth 3: created and rendered, or refused with PbrtHipError -- the VALUES are
scene_create's to check (a resolution of 0, a camera matrix of NaNs from `LookAt 1e30 ...`, vertices at 1e30 ...)."""
import os
import random
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbrt_amd import loader  # noqa: E402
from pbrt_amd._lib import PbrtHipError  # noqa: E402

base = open(os.path.join(ROOT, "scenes", "c0_check_sphere.pbrt")).read()
# a second seed text with more directives
extra = '''
LookAt 0 0 5  0 0 0  0 1 0
Camera "perspective" "float fov" [45]
Sampler "stratified" "integer xsamples" 2 "integer ysamples" 2 "bool jitter" "true"
Integrator "path" "integer maxdepth" 5 "bool mis" "false"
PixelFilter "box" "float xwidth" 1.5 "float ywidth" 1.0
Film "image" "integer xresolution" [32] "integer yresolution" [24] "string filename" "x.png" "float cropwindow" [0.1 0.9 0.2 0.8] "float maxsampleluminance" 2
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [4 4 4]
  Translate 0 2 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1 1 0 -1 1 0 1 -1 0 1] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
LightSource "point" "point from" [1 2 3] "rgb I" [5 5 5]
LightSource "distant" "point from" [0 0 0] "point to" [0 0 -1] "rgb L" [1 1 1]
LightSource "infinite" "rgb L" [.2 .2 .3]
Texture "c" "spectrum" "checkerboard" "float uscale" [4] "float vscale" [4] "rgb tex1" [.1 .1 .1] "rgb tex2" [.9 .9 .9]
ObjectBegin "o"
  Material "matte" "texture Kd" "c"
  Shape "sphere" "float radius" 0.5
ObjectEnd
TransformBegin
  Scale 2 2 2  Rotate 30 0 1 0  ConcatTransform [1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1]
  ObjectInstance "o"
TransformEnd
Material "mirror" "rgb Kr" [.9 .9 .9]
Shape "sphere" "float radius" 1
WorldEnd
'''
texts = [base, extra]
toks = ['"', '[', ']', '#', '\n', ' ', '1e39', '-', 'nan', 'inf', '"integer indices"', '"point P"', '99999999999', '-1', '0', 'WorldBegin', 'WorldEnd', 'AttributeBegin', 'AttributeEnd', 'ObjectBegin "o"', 'ObjectEnd', 'ObjectInstance "o"', 'Shape "trianglemesh"', 'Include "x"', 'Texture', '"bool x" "true"', '"string filename"', '\x00', '\xff', '"float fov" [0]', '"float fov" [180]', 'Scale 0 0 0', 'Transform [0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0]']

def mutate(t, rnd):
    for _ in range(rnd.randint(1, 4)):
        m = rnd.randint(0, 5)
        i = rnd.randrange(len(t) + 1)
        if m == 0:
            t = t[:i] + rnd.choice(toks) + t[i:]
        elif m == 1:
            t = t[:i] + t[min(len(t), i + rnd.randint(1, 40)):]
        elif m == 2:
            t = t[:i]
        elif m == 3:
            w = t.split(' ')
            if len(w) > 2:
                w[rnd.randrange(len(w))] = rnd.choice(toks)
                t = ' '.join(w)
        elif m == 4:
            j = min(len(t), i + rnd.randint(1, 60))
            t = t[:i] + t[i:j] * rnd.randint(2, 5) + t[j:]
        else:
            t = re.sub(r'\d+', lambda mm: rnd.choice([mm.group(0), '0', '-1', '4294967296', '1e30', '16777217']), t, count=rnd.randint(1, 3))
    return t


def run(seed, n):
    """-> (files that loaded, files refused with PbrtHipError); anything else propagates"""
    rnd = random.Random(seed)
    ok = err = 0
    for _ in range(n):
        try:
            loader.load_string(mutate(rnd.choice(texts), rnd))
            ok += 1
        except PbrtHipError:
            err += 1
    return ok, err


def run_hip(seed, n):
    import pbrt_amd
    rnd = random.Random(seed)
    loaded = created = rendered = refused = 0
    reasons = {}
    for _ in range(n):
        try:
            sd = loader.load_string(mutate(rnd.choice(texts), rnd)).scene
        except PbrtHipError:
            continue
        loaded += 1
        try:
            with pbrt_amd.Scene(sd) as sc:
                created += 1
                if sd.xres * sd.yres <= 262144:
                    sc.render(spp=(1, 1), max_depth=3, seed=1)
                    rendered += 1
        except PbrtHipError as e:
            refused += 1
            k = str(e).split(": ", 1)[-1][:70]
            reasons[k] = reasons.get(k, 0) + 1
    return loaded, created, rendered, refused, reasons


if __name__ == "__main__":
    t0 = time.time()
    if "--hip" in sys.argv:
        loaded, created, rendered, refused, reasons = run_hip(int(sys.argv[1]), int(sys.argv[2]))
        print(f"{sys.argv[2]} mutated scene files (seed {sys.argv[1]}): {loaded} loaded; of those {created} created on the device ({rendered} rendered), {refused} refused by "
              f"scene_create / render with PbrtHipError {reasons}; no crash, no hang, no other exception, {time.time() - t0:.0f} s")
        sys.exit(0)
    ok, err = run(int(sys.argv[1]), int(sys.argv[2]))
    print(f"{sys.argv[2]} mutated scene files (seed {sys.argv[1]}): {ok} loaded, {err} refused with PbrtHipError, no crash, no other exception, {time.time() - t0:.0f} s")
