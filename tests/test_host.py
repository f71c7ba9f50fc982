"""CPU-side tests of the product's host logic and of the C-ABI library itself (no GPU, no
compute calls): symbols, error behaviour, the BVH builder, the sharding geometry, the inputs."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

import pbrt_amd
from pbrt_amd import _lib, scenes
from util import SMALL_SCENES, assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Both headers: pbrt_hip.h (what a host binds) and pbrt_hip_debug.h (the hooks of tests and tools); no symbol in both, the
    boundary's calls of SURVEY.md 8(b) in the first, and the ctypes binding covers exactly their union."""
    header = open(os.path.join(ROOT, "include", "pbrt_hip.h")).read()
    debug = open(os.path.join(ROOT, "include", "pbrt_hip_debug.h")).read()
    stable = set(re.findall(r"\b(pbrt_hip_[a-z_0-9]+)\s*\(", header))
    hooks = set(re.findall(r"\b(pbrt_hip_[a-z_0-9]+)\s*\(", debug))
    assert not (stable & hooks), stable & hooks
    assert {"pbrt_hip_device_count", "pbrt_hip_scene_create", "pbrt_hip_render", "pbrt_hip_scene_destroy", "pbrt_hip_last_error",
            "pbrt_hip_render_multi", "pbrt_hip_multi_create", "pbrt_hip_load_file", "pbrt_hip_write_image"} <= stable
    assert {"pbrt_hip_scene_export_quads", "pbrt_hip_scene_export_bvh", "pbrt_hip_scene_walk_info", "pbrt_hip_render_stack_plan", "pbrt_hip_bvh_build_host",
            "pbrt_hip_quad_build_host", "pbrt_hip_quad_build_host_ex", "pbrt_hip_tokenize", "pbrt_hip_loaded_state", "pbrt_hip_sobol_matrices"} <= hooks
    declared = stable | hooks
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name)
    assert b"gfx950" in l.pbrt_hip_version()


def test_build_id_is_the_hash_of_the_sources():
    """the library carries the identity of what it was built from (sources under csrc/, the public
    header, kernel flags); profiles/pmc_<workload>.json records the id of the library its counters were taken on and
    bench.py withholds roofline.frac when they differ (tests/test_gpu_parity.py::test_bench_withholds_a_stale_profile)."""
    from pbrt_amd import build
    bid = pbrt_amd.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", bid), bid
    assert bid == build.source_id(os.environ.get("PBRT_HIP_EXTRA_FLAGS", "").split())
    assert build.source_id(["-DX"]) != build.source_id([])  # the flags are part of the identity


def test_bench_gpus_n_picks_a_launch_path_however_it_is_started():
    """`python bench.py --gpus N` must produce a line however it is launched: under torch.distributed.run one
    rank per GPU; bare, with no WORLD_SIZE, all N GPUs from the one process through pbrt_hip_multi_* (RCCL inside the library).  The
    choice is a pure function of argv and the environment, taken before torch or HIP are imported -- nothing is re-executed."""
    import subprocess
    import sys
    import bench
    assert bench.launch_mode(1, False, {}) == ("single", 1)
    assert bench.launch_mode(1, False, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}) == ("ranks", 1)  # torchrun with one rank: RCCL group of one
    for n in (2, 4, 8):
        assert bench.launch_mode(n, False, {}) == ("in-process", 1)                       # bare: the in-library path
        assert bench.launch_mode(n, True, {}) == ("in-process", 1)                        # --single-process
        assert bench.launch_mode(n, False, {"WORLD_SIZE": str(n), "RANK": "1"}) == ("ranks", n)  # the driver's launch
        with pytest.raises(SystemExit):  # --single-process inside a multi-rank launch: every rank would drive all the GPUs
            bench.launch_mode(n, True, {"WORLD_SIZE": str(n), "RANK": "1"})
    for env in ({"WORLD_SIZE": "4", "RANK": "0"}, {"WORLD_SIZE": "1", "RANK": "0"}):  # a launcher that disagrees with --gpus
        with pytest.raises(SystemExit):
            bench.launch_mode(8, False, env)
    if pbrt_amd.device_count() == 0:  # here: the bare N > 1 command gets as far as "no device", not "use another launcher"
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "needs a HIP device" in r.stderr and "torch.distributed.run" not in r.stderr


def test_kernel_isa_ids():
    """pbrt_amd/isa_id.py: a profile is keyed by the machine code of the kernel it measured.  Every production
    instantiation of render_kernel has an id, different instantiations have different ones, the profiler's and the demangler's
    spelling of a name find the same kernel, and a kernel the library does not have has none."""
    from pbrt_amd import isa_id
    ids = {isa_id.normalise(k): v for k, v in isa_id.kernel_ids(_lib.LIB_PATH).items()}
    prod = {k: v for k, v in ids.items() if re.fullmatch(r"render_kernel<(true|false),false,false,\d+,\d+,(true|false),(true|false)>", k)}
    assert len(prod) >= 8 and len(set(prod.values())) == len(prod), prod
    assert all(re.fullmatch(r"[0-9a-f]{16}", v) for v in prod.values())
    rocprof_name = "void pbrt_hip::(anonymous namespace)::render_kernel<false, false, false, 30, 3, false, false>(pbrt_hip::DevScene, pbrt_hip::RenderParams)"
    assert isa_id.kernel_id(_lib.LIB_PATH, rocprof_name) == prod["render_kernel<false,false,false,30,3,false,false>"]
    assert isa_id.kernel_id(_lib.LIB_PATH, "render_kernel<false, false, false, 31, 3, false, false>(x)") is None
    assert any(k.startswith("intersect_kernel<") for k in ids) and "merge_kernel" in " ".join(ids)


def test_committed_profiles_price_the_built_kernel():
    """profiles/pmc_<workload>.json carry the id of the kernel their counters were taken on; bench.py withholds roofline.frac when the
    loaded library's kernel of that name has another one.  The tree as committed must not be in that state: the production instantiation
    built from these sources IS the one the committed counters belong to (a kernel change without a new measurement set fails here
    before it ships a bench line without its roofline)."""
    import json
    from pbrt_amd import isa_id
    for wl in ("c3", "big"):
        pmc = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{wl}.json")))
        assert pmc.get("kernel_isa_id"), wl
        assert isa_id.kernel_id(_lib.LIB_PATH, pmc["kernel"]) == pmc["kernel_isa_id"], (wl, isa_id.normalise(pmc["kernel"]))
        # round 6: the lookup bench.py does -- by the MANGLED symbol, file parsing alone (no demangler child from a GPU-initialised process) --
        # finds the same kernel; the profile says which TREE its per-ray counters belong to (withheld when the live walk differs), what the
        # compiler allocated (code-object notes, not rocprofv3's granule fields) and how FETCH_SIZE was calibrated for this access shape
        assert isa_id.kernel_id(_lib.LIB_PATH, pmc["kernel"], pmc["kernel_symbol"]) == pmc["kernel_isa_id"]
        res = isa_id.kernel_resources(_lib.LIB_PATH, pmc["kernel_symbol"])
        assert res["vgpr_count"] == pmc["vgpr"] <= 96 and res["vgpr_spill_count"] == pmc["vgpr_spill"] == 0 and pmc["scratch_bytes"] == 0
        assert pmc["lds_bytes"] == pmc["lds_dynamic_bytes"] == 30 * 256 and pmc["waves_per_cu"] == 20  # the overflow variant's 30 rows
        t = pmc["tree"]
        assert t["kernel_fetches_per_ray"] > 30 and t["kernel_tris_per_ray"] > 4 and t["quad_nodes"] > 400_000 and t["spp"] in (512, 64)
        cal = pmc["fetch_size_calibration"]
        assert 0.9 < cal["fetch_size_bytes_per_known_byte"] < 1.1 and os.path.exists(os.path.join(ROOT, cal["source"]))  # x1, not the x2 of a wide stream
        assert abs(pmc["traffic_bytes_calibrated"] - (pmc["FETCH_SIZE_KB_per_launch"] * cal["factor"] + pmc["WRITE_SIZE_KB_per_launch"]) * 1024) < 1e-6 * pmc["traffic_bytes_calibrated"]


def test_library_holds_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"render_kernel" in blob


def test_struct_layouts_match_header(tmp_path):
    """include/pbrt_hip.h compiles as plain C, and the ctypes mirrors have the C sizes."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pbrt_hip.h"\n#include "pbrt_hip_debug.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(pbrt_hip_material),sizeof(pbrt_hip_light),sizeof(pbrt_hip_sphere),sizeof(pbrt_hip_scene_desc),"
                   "sizeof(pbrt_hip_render_desc),sizeof(pbrt_hip_stats),sizeof(pbrt_hip_texture),offsetof(pbrt_hip_scene_desc,tri_uv),"
                   "offsetof(pbrt_hip_material,kd_tex));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    mirrors = [_lib.Material, _lib.Light, _lib.Sphere, _lib.SceneDesc, _lib.RenderDesc, _lib.Stats, _lib.Texture]
    assert sizes[:7] == [C.sizeof(m) for m in mirrors]
    assert sizes[:3] == [32, 32, 32] and sizes[6] == 64
    assert sizes[7] == _lib.SceneDesc.tri_uv.offset and sizes[8] == _lib.Material.kd_tex.offset == 28
    from oracle import binding as ob  # the oracle's independent mirrors have the same layout (a test hands both the same bytes)
    assert [C.sizeof(m) for m in (ob.Material, ob.Light, ob.Sphere, ob.SceneDesc, ob.RenderDesc, ob.Texture)] == sizes[:5] + [64]


@pytest.mark.skipif(pbrt_amd.device_count() > 0, reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    with pytest.raises(_lib.PbrtHipError) as e:
        pbrt_amd.Scene(scenes.sphere_scene(8, 8))
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


@pytest.mark.skipif(pbrt_amd.device_count() > 0, reason="checks the no-device behaviour")
def test_multi_gpu_entry_points_fail_loudly_without_a_device():
    """The in-library multi-GPU path (pbrt_hip_multi_* / pbrt_hip_render_multi) has no CPU fallback either."""
    sd = scenes.cornell_scene(8, 8)
    with pytest.raises(_lib.PbrtHipError) as e:
        pbrt_amd.MultiScene(sd, 2)
    assert e.value.code == -2
    with pytest.raises(_lib.PbrtHipError) as e:
        pbrt_amd.render_multi(sd, 0, spp=(1, 1))
    assert e.value.code == -2


def test_shards_partition_the_film_for_any_gpu_count():
    """Host-side partition / gather index math of the multi-GPU path (pbrt_hip_slab_floats, pbrt_hip_slab_pixel_index): for
    1..9 ranks and ragged or cropped films every pixel belongs to exactly one rank's slab, rank 0's slab is the
    largest (the gather's common count), and a slab is whole 64x64 super-tiles."""
    for (xres, yres, crop) in [(200, 136, (0, 1, 0, 1)), (64, 64, (0, 1, 0, 1)), (1000, 700, (0.1, 0.77, 0.2, 0.9)), (130, 513, (0, 1, 0, 1))]:
        b = pbrt_amd.film_cropped_bounds(xres, yres, crop)
        n_px = (b[2] - b[0]) * (b[3] - b[1])
        for world in range(1, 10):
            seen = np.zeros(n_px, int)
            sizes = []
            for r in range(world):
                idx = pbrt_amd.slab_pixel_index(xres, yres, crop, r, world)
                assert len(idx) % 4096 == 0
                assert len(idx) * 4 == _lib.lib().pbrt_hip_slab_floats(xres, yres, (C.c_float * 4)(*crop), r, world)
                sizes.append(len(idx))
                seen[idx[idx >= 0]] += 1
            assert (seen == 1).all() and sizes[0] == max(sizes)


def test_bad_arguments_are_rejected_before_any_device_work():
    sd = scenes.cornell_scene(8, 8)
    sd.mat_id = sd.mat_id.copy()
    sd.mat_id[0] = 99
    with pytest.raises(_lib.PbrtHipError) as e:
        pbrt_amd.Scene(sd)
    assert e.value.code == -1 and "material id" in str(e.value)
    sd = scenes.cornell_scene(8, 8)
    sd.idx = sd.idx.copy()
    sd.idx[3, 1] = 10_000
    with pytest.raises(_lib.PbrtHipError) as e:
        pbrt_amd.Scene(sd)
    assert e.value.code == -1 and "vertex index" in str(e.value)
    sd = scenes.cornell_scene(8, 8)
    sd.xres = 0
    with pytest.raises(_lib.PbrtHipError):
        pbrt_amd.Scene(sd)
    for bad in (np.nan, np.inf, -np.inf):  # a non-finite vertex would send the SAH bucket index out of range
        sd = scenes.cornell_scene(8, 8)
        sd.P = sd.P.copy()
        sd.P[5, 1] = bad
        with pytest.raises(_lib.PbrtHipError) as e:
            pbrt_amd.Scene(sd)
        assert e.value.code == -1 and "not finite" in str(e.value)
        with pytest.raises(_lib.PbrtHipError):
            pbrt_amd.bvh_build_host(sd.P, sd.idx)
        with pytest.raises(_lib.PbrtHipError):
            pbrt_amd.quad_build_host(sd.P, sd.idx)
    sd = scenes.check_sphere_scene(8, 8)
    sd.spheres = sd.spheres.copy()
    sd.spheres[0, 3] = 0.0
    with pytest.raises(_lib.PbrtHipError):
        pbrt_amd.Scene(sd)
    sd = scenes.cornell_scene(8, 8)
    sd.fov = 180.0
    with pytest.raises(_lib.PbrtHipError):
        pbrt_amd.Scene(sd)
    with pytest.raises(ValueError):
        pbrt_amd.slab_pixel_index(64, 64, (0, 1, 0, 1), 2, 2)
    for crop in ((0.0, np.nan, 0.0, 1.0), (0.0, 1.0, -0.1, 1.0), (0.0, 1e30, 0.0, 1.0)):
        sd = scenes.cornell_scene(8, 8)
        sd.crop = crop
        with pytest.raises(_lib.PbrtHipError) as e:
            pbrt_amd.Scene(sd)
        assert e.value.code == -1 and "crop window" in str(e.value)
    # lights and colours that are not numbers (they would become ray directions / throughputs that nothing prunes)
    for field, row, col, what in (("lights", 0, 2, "light 0"), ("lights", 0, 5, "light 0"), ("materials", 1, 2, "material 1"), ("materials", 0, 5, "material 0")):
        for bad in (np.nan, np.inf):
            sd = scenes.check_sphere_scene(8, 8)
            arr = getattr(sd, field).copy()
            arr[row, col] = bad
            setattr(sd, field, arr)
            with pytest.raises(_lib.PbrtHipError) as e:
                pbrt_amd.Scene(sd)
            assert e.value.code == -1 and what in str(e.value) and "not finite" in str(e.value), (field, row, col, str(e.value))
    # textured materials: a texture number beyond the table, a textured triangle without corner (u, v), non-finite (u, v)
    # or mapping, an unknown texture type
    from util import checker_plane_scene
    for breakit, what in ((lambda sd: setattr(sd, "mat_tex", np.array([2], np.uint32)), "texture number"),
                          (lambda sd: setattr(sd, "tri_uv", np.zeros((0, 6), np.float32)), "tri_uv is NULL"),
                          (lambda sd: sd.tri_uv.__setitem__((1, 3), np.inf), "tri_uv is not finite"),
                          (lambda sd: sd.textures.__setitem__((0, 7), np.nan), "mapping is not finite"),
                          (lambda sd: sd.textures.__setitem__((0, 2), np.inf), "texture colour is not finite"),
                          (lambda sd: sd.textures.__setitem__((0, 0), 5), "unknown texture type")):
        sd, _ = checker_plane_scene(8)
        breakit(sd)
        with pytest.raises(_lib.PbrtHipError) as e:
            pbrt_amd.Scene(sd)
        assert e.value.code == -1 and what in str(e.value), (what, str(e.value))


@pytest.mark.parametrize("name", ["mesh1k", "mesh20k", "cornell", "check_sphere", "sphere", "spheres2k"])
def test_builder_equals_oracle_builder(oracle, name):
    """Two independently written builders of the same spec agree word for word -- spheres included: the oracle bounds a
    sphere by [c - r, c + r], the library's builders by the proxy triangle that spans that box (util.with_sphere_proxies)."""
    from util import with_sphere_proxies
    sd = SMALL_SCENES[name]()
    nodes, order, depth = pbrt_amd.bvh_build_host(*with_sphere_proxies(sd))
    on, oo, od = oracle.OracleScene(sd).bvh()
    assert depth == od
    assert_bit_equal(nodes, on, "nodes")
    assert_bit_equal(order, oo, "order")


def _check_tree(nodes, order, P, idx):
    nt = idx.shape[0]
    assert sorted(order.tolist()) == list(range(nt))  # a permutation: every triangle in exactly one leaf
    f = nodes.view(np.float32)
    seen = np.zeros(len(nodes), bool)
    leaf_slots = 0
    stack = [(0, 1)]
    max_level = 0
    while stack:
        i, level = stack.pop()
        assert not seen[i]
        seen[i] = True
        max_level = max(max_level, level)
        lo, hi = f[i, 0:3], f[i, 3:6]
        cnt, axis = int(nodes[i, 7] & 0xFFFF), int(nodes[i, 7] >> 16)
        if cnt:
            assert axis == 0
            tri = order[nodes[i, 6]:nodes[i, 6] + cnt]
            v = P[idx[tri]].reshape(-1, 3)
            assert np.array_equal(v.min(0), lo) and np.array_equal(v.max(0), hi)  # tight
            leaf_slots += cnt
        else:
            assert axis < 3
            a, b = i + 1, int(nodes[i, 6])
            assert b > a
            for c in (a, b):
                assert (f[c, 0:3] >= lo).all() and (f[c, 3:6] <= hi).all()  # children inside the parent
            assert np.array_equal(np.minimum(f[a, 0:3], f[b, 0:3]), lo) and np.array_equal(np.maximum(f[a, 3:6], f[b, 3:6]), hi)
            stack += [(a, level + 1), (b, level + 1)]
    assert seen.all() and leaf_slots == nt
    return max_level


def test_builder_structure():
    sd = SMALL_SCENES["mesh20k"]()
    nodes, order, depth = pbrt_amd.bvh_build_host(sd.P, sd.idx)
    assert _check_tree(nodes, order, sd.P, sd.idx) == depth
    assert (nodes[:, 7] & 0xFFFF).max() <= 4


def test_builder_edge_cases():
    nodes, order, depth = pbrt_amd.bvh_build_host(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    assert len(nodes) == 0 and depth == 0
    P = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    nodes, order, depth = pbrt_amd.bvh_build_host(P, np.array([[0, 1, 2]], np.uint32))
    assert len(nodes) == 1 and depth == 1 and nodes[0, 7] == 1 and order.tolist() == [0]
    # 600 identical triangles: all centroids coincide -> halved by position into leaves of <= 255
    idx = np.tile(np.array([[0, 1, 2]], np.uint32), (600, 1))
    nodes, order, depth = pbrt_amd.bvh_build_host(P, idx)
    cnt = nodes[:, 7] & 0xFFFF
    assert cnt.max() <= 255 and cnt.sum() == 600
    _check_tree(nodes, order, P, idx)
    # a long chain of nested triangles stays within the 64-entry traversal stack
    k = 3000
    s = (0.5 ** (np.arange(k) / 40.0)).astype(np.float32)
    Pn = np.concatenate([np.stack([s * 0, s * 0, s * 0], 1), np.stack([s, s * 0, s * 0], 1), np.stack([s * 0, s, s * 0], 1)])
    In = np.stack([np.arange(k), k + np.arange(k), 2 * k + np.arange(k)], 1).astype(np.uint32)
    nodes, order, depth = pbrt_amd.bvh_build_host(Pn, In)
    assert depth <= 64
    _check_tree(nodes, order, Pn, In)


def test_slab_pixel_index_partitions_the_film():
    xres, yres, crop = 200, 136, (0.0, 1.0, 0.0, 1.0)
    for world in (1, 2, 3, 8, 13):
        seen = np.zeros(xres * yres, int)
        for r in range(world):
            idx = pbrt_amd.slab_pixel_index(xres, yres, crop, r, world)
            assert len(idx) % 4096 == 0
            ok = idx[idx >= 0]
            seen[ok] += 1
            # super-tile j of rank r is super-tile r + j*world of the film, row-major 64x64
            for j in range(len(idx) // 4096):
                t = r + j * world
                x0, y0 = (t % 4) * 64, (t // 4) * 64
                blk = idx[j * 4096:(j + 1) * 4096].reshape(64, 64)
                assert blk[0, 0] == y0 * xres + x0
                assert (blk[:, min(63, xres - 1 - x0) + 1:] == -1).all()
        assert (seen == 1).all()
    # cropped film: indices are relative to the cropped bounds
    idx = pbrt_amd.slab_pixel_index(256, 256, (0.25, 0.5, 0.5, 1.0), 0, 1)
    assert len(idx) == 2 * 4096 and idx.max() == 64 * 128 - 1
    assert len(pbrt_amd.slab_pixel_index(64, 64, (0, 0, 0, 0), 0, 1)) == 0  # empty film


def test_scene_generators_are_deterministic_and_shaped():
    a = scenes.random_mesh_scene(5000, 32, 32)
    b = scenes.random_mesh_scene(5000, 32, 32)
    assert_bit_equal(a.P, b.P, "P")
    assert a.idx.shape == (5000 + 14, 3) and a.materials.shape == (252, 7)
    tri = a.P[a.idx[:5000]]
    centre_span = (tri.max(1) - tri.min(1)).max()
    assert centre_span <= 2 * 5000 ** (-1 / 3) + 1e-6  # vertices within +-s of the centre
    assert (np.abs(tri) <= 1 + 5000 ** (-1 / 3) + 1e-6).all()
    assert (a.materials[np.arange(0, 250, 5), 0] == 1).all() and (a.materials[1:5, 0] == 0).all()
    le = a.materials[a.mat_id[-2:], 4:7]
    assert (le == 20).all()
    e1 = a.P[a.idx[-1, 1]] - a.P[a.idx[-1, 0]]
    e2 = a.P[a.idx[-1, 2]] - a.P[a.idx[-1, 0]]
    assert np.cross(e1, e2)[2] < 0  # the ceiling light faces down
    c = scenes.random_mesh_scene(5000, 32, 32, seed=99)
    assert not np.array_equal(a.P, c.P)


def test_native_cli_usage():
    import subprocess
    from pbrt_amd.build import CLI_PATH
    r = subprocess.run([CLI_PATH, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--outfile" in r.stderr and "--nthreads" in r.stderr and "--quick" in r.stderr
    assert subprocess.run([CLI_PATH], capture_output=True).returncode == 1  # no scene files
    assert subprocess.run([CLI_PATH, "--bogus"], capture_output=True).returncode == 2


def test_cli_log_levels_are_the_reference_binarys(tmp_path):
    """bin/pbrt.rs:48-62: stderrlog verbosity 1 (quiet: "only WARN and higher"), 2 (default: + INFO), 3 (verbose: + DEBUG) -- a parser
    warning passes every level, the info line only the default and the verbose one.  (No GPU here: the run ends at the render call.)"""
    import subprocess
    from pbrt_amd.build import CLI_PATH
    scene = tmp_path / "s.pbrt"
    scene.write_text('Camera "perspective" "float lensradius" 0.5\nWorldBegin Shape "sphere" WorldEnd\n')
    for flags, info in (([], True), (["-q"], False), (["-v"], True)):
        for cmd in ([CLI_PATH], [sys.executable, "-m", "pbrt_amd.cli"]):
            r = subprocess.run(cmd + flags + [str(scene)], capture_output=True, text=True, cwd=ROOT)
            assert "warning: Camera: \"lensradius\" ignored" in r.stderr, (cmd, flags, r.stderr)
            assert ("1 spheres" in r.stderr) == info, (cmd, flags, r.stderr)


def _check_quads(quads, need, P, idx, order):
    """Decode the quantised 4-wide tree and check its invariants: every decoded child box encloses
    everything below it (exact arithmetic), every leaf slot is reachable exactly once, and no walk can
    hold more stack entries than `need`."""
    f = quads.view(np.float32)
    tri_lo = P[idx[order]].min(1).astype(np.float64)
    tri_hi = P[idx[order]].max(1).astype(np.float64)
    seen = np.zeros(len(order), int)
    worst = 0

    def visit(q, held):
        nonlocal worst
        origin = f[q, 0:3].astype(np.float64)
        cell = np.array([f[q, 3], f[q, 10], f[q, 11]], np.float64)  # powers of two, as f32
        assert all(c > 0 and np.frexp(c)[0] == 0.5 for c in cell)
        qlo = [quads[q, 4 + a] for a in range(3)]
        qhi = [quads[q, 7], quads[q, 8], quads[q, 9]]
        refs = quads[q, 12:16]
        kids = [k for k in range(4) if refs[k] != 0x80000000]  # kEmptyLeafRef: unused slot
        held += len(kids) - 1
        worst = max(worst, held)
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        for k in range(4):
            blo = np.array([origin[a] + ((int(qlo[a]) >> (8 * k)) & 0xFF) * cell[a] for a in range(3)])
            bhi = np.array([origin[a] + ((int(qhi[a]) >> (8 * k)) & 0xFF) * cell[a] for a in range(3)])
            if k not in kids:
                assert k >= 1 and (blo >= bhi).all()  # unused slots come last; inverted box wherever the node has extent
                continue
            r = int(refs[k])
            if r & 0x80000000:
                cnt, first = (r >> 24) & 0x7F, r & 0xFFFFFF
                assert cnt >= 1
                seen[first:first + cnt] += 1
                tlo, thi = tri_lo[first:first + cnt].min(0), tri_hi[first:first + cnt].max(0)
            else:
                assert r % 64 == 0  # an interior child's ref is its byte offset
                tlo, thi = visit(r // 64, held)
            assert (blo <= tlo).all() and (bhi >= thi).all(), (q, k, blo, tlo, bhi, thi)
            lo_all, hi_all = np.minimum(lo_all, tlo), np.maximum(hi_all, thi)
        return lo_all, hi_all

    visit(0, 0)
    assert (seen == 1).all(), f"{(seen != 1).sum()} leaf slots not reached exactly once"
    assert worst <= need, (worst, need)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("name", ["mesh1k", "mesh20k", "cornell", "ties", "deep"])
def test_quantised_quad_tree_invariants(name, split):
    import sys
    sys.setrecursionlimit(10000)
    sd = SMALL_SCENES[name]()
    nodes, order, depth = pbrt_amd.bvh_build_host(sd.P, sd.idx)
    quads, need = pbrt_amd.quad_build_host(sd.P, sd.idx, split_leaves=split)
    assert len(quads) > 0 and need >= 1
    _check_quads(quads, need, sd.P, sd.idx, order)


def test_dp_collapse_option_builds_a_valid_tree():
    """PBRT_HIP_COLLAPSE=dp (SAH-optimal collapse by dynamic programming; read once per process, hence the child
    process): fewer nodes than the greedy rule, same structural invariants."""
    import subprocess
    import sys
    code = (
        "import sys; sys.setrecursionlimit(10000); sys.path.insert(0, 'tests')\n"
        "import pbrt_amd\n"
        "from util import SMALL_SCENES\n"
        "from test_host import _check_quads\n"
        "sd = SMALL_SCENES['mesh20k']()\n"
        "nodes, order, depth = pbrt_amd.bvh_build_host(sd.P, sd.idx)\n"
        "quads, need = pbrt_amd.quad_build_host(sd.P, sd.idx)\n"
        "_check_quads(quads, need, sd.P, sd.idx, order)\n"
        "print(len(quads))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = {}
    for mode in ("dp", "greedy"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root,
                           env=dict(os.environ, PBRT_HIP_COLLAPSE=mode))
        assert r.returncode == 0, r.stderr[-2000:]
        n[mode] = int(r.stdout.split()[-1])
    assert n["dp"] < n["greedy"]


def _kernel_notes():
    """(name, vgpr_count, vgpr_spill_count, private_segment_fixed_size) of every kernel in the library's gfx950 code objects."""
    import glob
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        so = shutil.copy(_lib.LIB_PATH, td)
        # one code object per translation unit lands beside the input as <name>.<n>.hipv4-amdgcn-amd-amdhsa--gfx950
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", os.path.basename(so)], cwd=td, check=True, capture_output=True)
        cos = sorted(glob.glob(os.path.join(td, "*gfx950")))
        assert cos, "no gfx950 code object in the library"
        notes = "".join(subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
                        for co in cos)
    out = []
    for blk in notes.split(".agpr_count:")[1:]:
        f = {k: re.search(r"\." + k + r":\s+(\S+)", blk) for k in ("name", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
        if all(f.values()):
            out.append((f["name"].group(1), int(f["vgpr_count"].group(1)), int(f["vgpr_spill_count"].group(1)),
                        int(f["private_segment_fixed_size"].group(1))))
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="needs the ROCm llvm tools")
def test_production_kernels_do_not_spill():
    """the production instantiations of render_kernel (no counters, no exact walk) and
    intersect_kernel must stay under the register budget of their launch bounds: no VGPR spills, no scratch."""
    import subprocess
    notes = _kernel_notes()
    assert notes, "no kernel notes found in the code object"
    demangled = subprocess.run(["c++filt"], input="\n".join(n[0] for n in notes), capture_output=True, text=True, check=True).stdout.split("\n")
    prod = [(d, n) for d, n in zip(demangled, notes)
            if ("render_kernel<" in d and re.search(r"render_kernel<(true|false), false, false,", d)) or "intersect_kernel<" in d and ", false," in d]
    assert len(prod) >= 4, demangled
    for d, (_, vgpr, spill, scratch) in prod:
        assert spill == 0 and scratch == 0, f"{d}: {vgpr} VGPRs, {spill} spilled, {scratch} B scratch"
        # the triangle-scene render kernels are launched at 5 waves per SIMD (capi.cpp: 20 one-wave workgroups per CU)
        # (also the instantiations for a box filter radius other than 0.5 and for the Sobol' sampler: the last two arguments)
        if re.search(r"render_kernel<false, false, false, \d+, \d+, (true|false), (true|false)>", d):
            assert vgpr <= 96, f"{d}: {vgpr} VGPRs do not fit 5 waves per SIMD"


def test_render_stack_plan():
    """pbrt_hip_render_stack_plan (device_types.h render_stack_plan): LDS comes in granules of 1280 bytes, so <= 28 entries
    (30 rows) run at 20 waves per CU (5 per SIMD) with the whole stack in LDS, deeper trees run the overflow variant with 30 rows,
    also at 20 waves (round 3: very deep trees too -- a walk's stack stays far below its worst-case bound, device_types.h).
    Granules x waves never exceed a CU's 160 KB.  (The kernel-side use is covered by the GPU parity tests, also on a 12-row build.)"""
    def plan(need):
        r, w, x = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        assert _lib.lib().pbrt_hip_render_stack_plan(need, C.byref(r), C.byref(w), C.byref(x)) == 0
        return r.value, w.value, x.value
    assert plan(5) == (8, 20, 0)
    assert plan(28) == (30, 20, 0)
    assert plan(29) == (30, 20, 1)
    assert plan(33) == (30, 20, 5)   # BASELINE config C2
    assert plan(34) == (30, 20, 6)
    assert plan(38) == (30, 20, 10)  # C3
    assert plan(41) == (30, 20, 13)
    assert plan(42) == (30, 20, 14)  # very deep trees (the 12 M-triangle workload: 48): the same 30 rows
    assert plan(60) == (30, 20, 32)
    for need in range(0, 100):
        rows, waves, extra = plan(need)
        assert -(-rows * 256 // 1280) * 1280 * waves <= 160 * 1024 and waves == 20
        assert rows + extra >= need + 2


def test_image_readers_refuse_hostile_headers(tmp_path):
    """a header that announces more pixels than the file holds must come back as an error code,
    not as std::bad_alloc through the C ABI."""
    w, h = C.c_int32(0), C.c_int32(0)
    bomb = tmp_path / "bomb.pfm"
    bomb.write_bytes(b"PF\n60000 60000\n-1\n")
    assert _lib.lib().pbrt_hip_read_image(str(bomb).encode(), None, C.byref(w), C.byref(h)) < 0
    short = tmp_path / "short.pfm"
    short.write_bytes(b"PF\n4 4\n-1\n" + b"\0" * 100)  # 192 bytes announced, 100 present
    assert _lib.lib().pbrt_hip_read_image(str(short).encode(), None, C.byref(w), C.byref(h)) < 0
    # a PNG whose IDAT inflates to far more than its IHDR needs (decompression bomb): zeros behind a 1x1 header
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 2, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\0" * (1 << 24), 9)) + chunk(b"IEND", b"")
    p = tmp_path / "bomb.png"
    p.write_bytes(png)
    assert _lib.lib().pbrt_hip_read_image(str(p).encode(), None, C.byref(w), C.byref(h)) < 0
    # and a legitimate file still reads
    rgb = np.linspace(0, 1, 4 * 3 * 3, dtype=np.float32).reshape(3, 4, 3)
    ok = tmp_path / "ok.pfm"
    pbrt_amd.write_image(str(ok), rgb)
    assert np.array_equal(pbrt_amd.read_image(str(ok)), rgb)


# ---- the production walk's trees checked on the CPU with the kernel's step restated (oracle/quad_walk.cpp) ----
@pytest.mark.parametrize("tree", ["sah", "reinsert"])
@pytest.mark.parametrize("name", ["mesh1k", "mesh20k", "cornell", "ties", "deep"])
def test_production_tree_finds_the_oracles_hits(oracle, name, tree):
    """A hit does not depend on the tree (tie rule): the 4-wide quantised tree collapsed from the canonical
    binned-SAH tree and the one optimised by the DEVICE builder's parallel re-insertion pass run on the host (the same functions,
    reinsert_core.hpp, that bvh_gpu.hip's kernels call) must give the oracle's hit records and occlusion flags ray for ray -- any
    difference is a box that does not enclose what lies below it, or a triangle a move lost.  Also: no walk exceeds the builder's
    stack bound."""
    from pbrt_amd.api import quad_build_host_ex
    from util import SMALL_SCENES, random_rays
    sd = SMALL_SCENES[name]().normalized()
    q = quad_build_host_ex(sd.P, sd.idx, tree=tree)
    o, d, tmax = random_rays(60_000, 33)
    ref = oracle.OracleScene(sd)
    rt, rp, rb1, rb2, _ = ref.intersect(o, d, tmax)
    spheres = rp >= sd.idx.shape[0]  # (the walk covers the triangles; spheres are tested after it)
    got = oracle.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], o, d, tmax)
    keep = ~spheres
    assert np.array_equal(got["prim"][keep], rp[keep]) and np.array_equal(got["t"][keep].view(np.uint32), rt[keep].view(np.uint32))
    assert np.array_equal(got["b1"][keep].view(np.uint32), rb1[keep].view(np.uint32))
    if sd.spheres.shape[0] == 0:
        occ = oracle.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], o, d, tmax, any_hit=True)
        assert np.array_equal(occ["occluded"], ref.occluded(o, d, tmax))
        assert max(got["max_stack"], occ["max_stack"]) <= q["stack_need"]
    assert q["n_refs"] == sd.idx.shape[0]  # every triangle is one reference
    _check_quads(q["quads"], q["stack_need"], sd.P, sd.idx, q["order"])


def test_walk_prunes_rays_parallel_to_an_axis(oracle):
    """The kernel's step restated (oracle/quad_walk.cpp) on rays with direction components that are exactly 0: the slab test of such an axis
    must still prune (the walk multiplies by a huge finite power of two where 1 / d is infinite: a quantised plane's t = q x inf - inf was
    NaN and the axis dropped out) -- about the node steps of a ray in general position instead of a tenth of the tree --, with the
    oracle's hits; also in scenes of size 1e12 and 1e-2 (the stand-in follows the root box's extent)."""
    from pbrt_amd.api import quad_build_host_ex
    from util import SMALL_SCENES
    rng = np.random.default_rng(5)
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [0, -1, 1], [-1, 0, 1], [0, -0.0, 1]], np.float32)
    d = np.repeat(dirs, 300, 0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for scale in (1.0, 1e12, 1e-2):
        sd = SMALL_SCENES["mesh20k"]().normalized()
        sd.P = (sd.P.astype(np.float64) * scale).astype(np.float32)
        o = (rng.uniform(-1.6, 1.6, d.shape) * scale).astype(np.float32)
        o[::3] = (np.round(o[::3].astype(np.float64) / scale * 4) / 4 * scale).astype(np.float32)  # on round planes of the nodes' grids
        tmax = np.full(len(d), np.inf, np.float32)
        q = quad_build_host_ex(sd.P, sd.idx, tree="sah")
        got = oracle.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], o, d, tmax)
        g = rng.normal(size=d.shape)
        dg = (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)
        general = oracle.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], o, dg, tmax)
        rt, rp = oracle.OracleScene(sd).intersect(o, d, tmax)[:2]
        assert np.array_equal(got["prim"], rp) and np.array_equal(got["t"].view(np.uint32), rt.view(np.uint32)), scale
        assert (rp != 0xffffffff).mean() > 0.3
        assert got["steps"].mean() < 1.5 * general["steps"].mean(), (scale, got["steps"].mean(), general["steps"].mean())


def test_reinsertion_cuts_the_walks_work(oracle):
    """PBRT_HIP_SCENE_OPTIMIZED_TREE / PBRT_HIP_TREE_REINSERT (the device builder's parallel re-insertion pass, run here on the host):
    on a mesh of BASELINE's kind the optimised tree costs a walk less work than the binned-SAH tree for the same hits -- here (20 k
    triangles, uniformly random rays) at least 4 % fewer node steps, and less in all with a triangle test priced at 0.7 node steps
    (their instruction counts); on the path-traced ray mix of the 1 M-triangle scene both fall (tools/walk_sim.py: 35.2 -> 33.9
    steps, 4.33 -> 4.18 tests; on the GPU 40.2 -> 38.4 node fetches per ray)."""
    from pbrt_amd.api import quad_build_host_ex
    from util import SMALL_SCENES, random_rays
    sd = SMALL_SCENES["mesh20k"]().normalized()
    o, d, tmax = random_rays(40_000, 7)
    work = {}
    for tree in ("sah", "reinsert"):
        q = quad_build_host_ex(sd.P, sd.idx, tree=tree)
        got = oracle.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], o, d, tmax)
        work[tree] = (got["steps"].sum(), got["tris"].sum(), got["prim"])
    assert np.array_equal(work["sah"][2], work["reinsert"][2])
    cost = {k: float(v[0]) + 0.7 * float(v[1]) for k, v in work.items()}
    assert work["reinsert"][0] < 0.96 * work["sah"][0] and cost["reinsert"] < 0.98 * cost["sah"], (work["sah"][:2], work["reinsert"][:2])


def test_sobol_nd_generator_matrices(oracle):
    """Sampler 2 takes its dimensions from generator matrices the library builds on the host from the
    Joe-Kuo direction numbers: they must equal the oracle's own construction (which tests/test_reference_vectors.py pins to
    the reference's SOBOL_MATRICES32, sobolmatrices.rs:81) column for column, and -- where /root/reference is mounted --
    the reference's table directly."""
    from pbrt_amd.api import sobol_matrices
    m = sobol_matrices()
    ref = oracle.sobol_matrices()
    assert m.shape == (128, 32) and ref.shape[0] >= 128  # 64 requests x 2: all a path of maxdepth 16 makes
    assert np.array_equal(m, ref[:128, :32])
    path = "/root/reference/src/core/sobolmatrices.rs"
    if os.path.exists(path):
        text = open(path).read()
        body = text[text.index("SOBOL_MATRICES32"):]
        body = body[body.index("= [") + 3:]
        vals = []
        for tok in re.finditer(r"0x[0-9a-fA-F]+|\d+", body):
            vals.append(int(tok.group(0), 0))
            if len(vals) >= 128 * 52:
                break
        table = np.array(vals, np.uint64).reshape(128, 52)[:, :32].astype(np.uint32)
        assert np.array_equal(m, table)


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """The PyTorch-ROCm wheel carries its own HIP / HSA runtime; libpbrt_hip.so names /opt/rocm's by the same sonames.  Loaded in the order
    library -> torch, the process used to hold two HSA runtimes and torch found "No HIP GPUs" (seen on the MI355X box with a test subset
    that imported torch late).  pbrt_amd._lib loads torch's copy first when torch is installed: one libamdhip64 and one libhsa-runtime64
    in the process, both before and after `import torch`, in either order."""
    import subprocess
    import sys
    code = r"""
import sys, os
order = sys.argv[1]
def runtimes():
    libs = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime64' in l}
    return sorted(os.path.realpath(p) for p in libs)
if order == 'torch_first':
    import torch
import pbrt_amd
pbrt_amd.device_count()
before = runtimes()
import torch
torch.cuda.is_available()
after = runtimes()
assert before == after and len(after) == 2, (before, after)
assert all(os.sep + 'torch' + os.sep in p for p in after), after
# ... and the RCCL of the in-library multi-GPU path is the one beside that runtime (torch's), not another ROCm release's
import ctypes
from pbrt_amd import _lib
buf = ctypes.create_string_buffer(4096)
assert _lib.lib().pbrt_hip_rccl_library(buf, 4096) == 0, _lib.lib().pbrt_hip_last_error()
assert os.path.dirname(os.path.realpath(buf.value.decode())) == os.path.dirname(after[0]), (buf.value, after)
print('ok')
"""
    for order in ("library_first", "torch_first"):
        r = subprocess.run([sys.executable, "-c", code, order], capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (order, r.stdout, r.stderr[-2000:])
