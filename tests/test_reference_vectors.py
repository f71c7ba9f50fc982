"""Pins the oracle (and the product's host pieces) against every known-answer vector the
reference crate holds for code adjacent to the path.  Each test cites the
reference file:line the literal values are taken from."""
import os

import numpy as np
import pytest

import pbrt_amd

EPS = np.finfo(np.float32).eps


# ---- PCG32: src/core/rng.rs ----
def test_rng_default_u32(oracle):  # rng.rs:131-137
    want = [355248013, 41705475, 3406281715, 4186697710, 483882979, 2766312848, 1713261421, 154902030, 3085534493,
            3877580365]
    assert oracle.rng_default_u32(10).tolist() == want


def test_rng_threshold(oracle):  # rng.rs:145-148
    assert oracle.rng_default_threshold(4095, 10).tolist() == [2668, 1995, 3385, 2470, 1399, 1118, 3511, 465, 1133, 295]


def test_rng_new_zero(oracle):  # rng.rs:152-155
    assert int(oracle.rng_seq_u32(0, 1)[0]) == 1774745655


def test_rng_threshold_terminates(oracle):  # rng.rs:158-163
    oracle.rng_default_threshold(0xFFFFFFFF // 2, 1)


def test_rng_uniform_float(oracle):  # rng.rs:166-176 (assert_approx_eq default 1e-6)
    want = [0.0827126, 0.00971031, 0.793087, 0.974792, 0.112663, 0.644082, 0.3989, 0.0360659, 0.718407, 0.90282]
    assert np.allclose(oracle.rng_default_float(10), want, atol=1e-6, rtol=0)


def test_scene_generator_pcg_matches_oracle(oracle):
    """pbrt_amd.scenes' vectorised PCG32 (host input generator) is the same generator."""
    from pbrt_amd.scenes import PcgStreams
    seqs = np.array([0, 1, 7, 0x5EED0001, 2**40 + 3], np.uint64)
    r = PcgStreams(seqs)
    got_u = np.stack([r.u32() for _ in range(6)], axis=1)
    r = PcgStreams(seqs)
    got_f = np.stack([r.uniform() for _ in range(6)], axis=1)
    for i, s in enumerate(seqs):
        assert got_u[i].tolist() == oracle.rng_seq_u32(int(s), 6).tolist()
        assert np.array_equal(got_f[i], oracle.rng_seq_float(int(s), 6))


# ---- Film: src/core/film.rs ----
CROP_Q = (0.25, 0.75, 0.25, 0.75)
FULL = (0.0, 1.0, 0.0, 1.0)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_film_sample_bounds(oracle, impl):  # film.rs:151-164
    f = oracle if impl == "oracle" else pbrt_amd
    assert f.film_sample_bounds(1920, 1080, CROP_Q, (8.0, 8.0)) == (472, 262, 1448, 818)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_film_tile_bounds(oracle, impl):  # film.rs:251-262
    f = oracle if impl == "oracle" else pbrt_amd
    assert f.film_tile_bounds(1920, 1080, CROP_Q, (8.0, 8.0), (0, 0, 1920, 1080)) == (480, 270, 1440, 810)
    assert f.film_tile_bounds(1920, 1080, CROP_Q, (8.0, 8.0), (500, 500, 600, 600)) == (492, 492, 608, 608)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_film_cropped_bounds(oracle, impl):  # film.rs:92-101
    f = oracle if impl == "oracle" else pbrt_amd
    assert f.film_cropped_bounds(1920, 1080, CROP_Q) == (480, 270, 1440, 810)
    assert f.film_cropped_bounds(200, 10, FULL) == (0, 0, 200, 10)
    assert f.film_cropped_bounds(10, 10, (0.0, 0.0, 0.0, 0.0)) == (0, 0, 0, 0)  # empty film
    assert f.film_cropped_bounds(101, 7, (0.3, 0.61, 0.1, 0.9)) == (31, 1, 62, 7)  # ceil on both ends


def test_film_physical_extent(oracle):  # film.rs:186-216
    for _ in range(2):
        assert np.allclose(oracle.film_physical_extent(800, 600, 100.0), [-0.04, -0.03, 0.04, 0.03], atol=EPS, rtol=0)


def test_film_merge_colour(oracle):  # film.rs:524-534: pixel xyz == Spectrum::to_xyz of the tile colour
    green = oracle.rgb_to_xyz([0, 1, 0])
    red = oracle.rgb_to_xyz([1, 0, 0])
    assert np.array_equal(green, np.array([0.357580, 0.715160, 0.119193], np.float32))  # spectrum.rs:139-145
    assert np.array_equal(red, np.array([0.412453, 0.212671, 0.019334], np.float32))


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_film_write_rgb(oracle, impl):  # film.rs:346-372
    film = np.array([[[0.357580, 0.715160, 0.119193, 1.0], [0.8, 1.4, 0.2, 2.0], [-1.0, 0.1, 0.0, 4.0],
                      [0.3, 0.3, 0.3, 0.0]]], np.float32)
    got = (oracle.film_write_rgb if impl == "oracle" else pbrt_amd.film_to_rgb)(film, 2.0)
    want = np.zeros((1, 4, 3), np.float32)
    for i in range(4):
        c = oracle.xyz_to_rgb(film[0, i, :3])
        w = film[0, i, 3]
        if w != 0:
            c = np.maximum(c * (np.float32(1) / w), 0)
        want[0, i] = c * np.float32(2)
    assert np.array_equal(got, want)
    assert np.allclose(got[0, 0], [0, 2, 0], atol=1e-5)


def test_product_film_to_rgb_equals_oracle(oracle):
    rng = np.random.default_rng(3)
    film = rng.uniform(-0.2, 3.0, (17, 9, 4)).astype(np.float32)
    film[::3, ::2, 3] = 0
    assert np.array_equal(pbrt_amd.film_to_rgb(film, 1.5).view(np.uint32), oracle.film_write_rgb(film, 1.5).view(np.uint32))


# ---- lib.rs ----
def test_quadratic(oracle):  # lib.rs:171-180
    assert oracle.quadratic(1, 1, 1) is None
    assert oracle.quadratic(1, -6, -16) == (-2.0, 8.0)
    assert oracle.quadratic(1, 6, 5) == (-5.0, -1.0)
    assert oracle.quadratic(1, 0, -16) == (-4.0, 4.0)
    assert oracle.quadratic(1, 6, 0) == (-6.0, 0.0)
    t0, t1 = oracle.quadratic(1, 2, -2)
    s3 = np.sqrt(np.float32(3))
    assert abs(t0 - (-1 - s3)) < EPS and abs(t1 - (-1 + s3)) < EPS


def test_gamma_and_to_byte(oracle):  # lib.rs:93-99, imageio.rs:66-68
    assert oracle.gamma_correct(0.0) == 0.0
    assert oracle.gamma_correct(0.002) == np.float32(12.92) * np.float32(0.002)
    assert abs(oracle.gamma_correct(1.0) - 1.0) < 1e-6
    assert oracle.to_byte(0.0) == 0 and oracle.to_byte(1.0) == 255 and oracle.to_byte(7.0) == 255 and oracle.to_byte(-1.0) == 0
    assert oracle.to_byte(0.5) == int(255 * (1.055 * 0.5 ** (1 / 2.4) - 0.055) + 0.5)


# ---- transform.rs ----
def test_matrix_inverse_doctests(oracle):  # transform.rs:142-155
    eye = np.eye(4, dtype=np.float32)
    assert np.array_equal(oracle.matrix_inverse(eye), eye)
    m = np.diag([2, 3, 4, 1]).astype(np.float32)
    assert np.allclose(oracle.matrix_mul(oracle.matrix_inverse(m), m), eye, atol=EPS, rtol=0)
    assert np.allclose(oracle.matrix_mul(m, oracle.matrix_inverse(m)), eye, atol=EPS, rtol=0)


def test_matrix_transpose_and_product_doctests(oracle):  # transform.rs:113-128 (transpose), :270-282 (Mul) with :142-155's products
    m = np.array([[2, 0, 0, 0], [3, 1, 0, 0], [4, 0, 1, 0], [5, 6, 7, 1]], np.float32)
    m_t = np.array([[2, 3, 4, 5], [0, 1, 0, 6], [0, 0, 1, 7], [0, 0, 0, 1]], np.float32)
    assert np.array_equal(oracle.matrix_transpose(m), m_t)
    i4 = np.eye(4, dtype=np.float32)
    d = np.diag([2, 3, 4, 1]).astype(np.float32)
    assert np.array_equal(oracle.matrix_mul(oracle.matrix_inverse(d), d), i4) and np.array_equal(oracle.matrix_mul(d, oracle.matrix_inverse(d)), i4)
    assert np.array_equal(oracle.matrix_mul(oracle.matrix_inverse(i4), i4), i4)


def test_look_at(oracle):  # transform.rs:485-520, camera of scenes/check-sphere.pbrt:1-3
    m, mi = oracle.look_at((3, 4, 1.5), (0.5, 0.5, 0), (0, 0, 1))
    assert np.array_equal(mi[:3, 3], np.array([3, 4, 1.5], np.float32))
    d = np.array([0.5 - 3, 0.5 - 4, -1.5], np.float64)
    assert np.allclose(mi[:3, 2], d / np.linalg.norm(d), atol=1e-6)
    assert np.allclose(oracle.matrix_mul(m, mi), np.eye(4), atol=1e-5)
    assert abs(np.dot(mi[:3, 0], mi[:3, 1])) < 1e-6 and abs(np.dot(mi[:3, 0], mi[:3, 2])) < 1e-6
    # left-handed: right x up = -dir ... new_up = cross(dir, right)
    assert np.allclose(np.cross(mi[:3, 2], mi[:3, 0]), mi[:3, 1], atol=1e-6)


def test_product_look_at_equals_oracle(oracle):
    for args in [((3, 4, 1.5), (0.5, 0.5, 0), (0, 0, 1)), ((0, -1.95, 0), (0, 0, 0), (0, 0, 1)),
                 ((1.5, -2.25, 9), (0.1, 0.2, 0.3), (0.2, 1, 0.1))]:
        a = pbrt_amd.look_at(*args)
        b = oracle.look_at(*args)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_clamp_lerp_and_the_2x2_solver(oracle):
    """lib.rs:104-137 (clamp, lerp) and transform.rs:36-58 (solve_linear_system_2x2): the doc-tests' literals.  (The solver is pbrt-v3's
    helper for a triangle's dp/du; nothing on this path calls it, SURVEY 8(c) lists its vectors with the others.)"""
    assert [oracle.clamp(-1.0, 0.0, 1.0), oracle.clamp(0.5, 0.0, 1.0), oracle.clamp(2.0, 0.0, 1.0)] == [0.0, 0.5, 1.0]
    assert [oracle.clamp(-1, 0, 2), oracle.clamp(1, 0, 2), oracle.clamp(3, 0, 2)] == [0, 1, 2]
    assert [oracle.lerp(0.0, 0.0, 1.0), oracle.lerp(0.5, 0.0, 1.0), oracle.lerp(1.0, 0.0, 1.0), oracle.lerp(0.75, 0.0, 2.0)] == [0.0, 0.5, 1.0, 1.5]
    assert oracle.solve_linear_system_2x2([[3.0, -5.0], [1.0, -4.0]], [4.0, -1.0]) == [3.0, 1.0]
    assert oracle.solve_linear_system_2x2([[2.0, -3.0], [0.0, 4.0]], [-8.0, 8.0]) == [-1.0, 2.0]
    assert oracle.solve_linear_system_2x2([[5.0, -1.0], [3.0, 2.0]], [3.0, 20.0]) == [2.0, 7.0]
    assert oracle.solve_linear_system_2x2([[2.0, -1.0], [-4.0, 2.0]], [7.0, 6.0]) is None
    assert oracle.solve_linear_system_2x2([[2.0, -1.0], [-2.0, 1.0]], [7.0, 3.0]) is None


# ---- imageio.rs ----
def test_png_pfm_roundtrip(tmp_path, oracle):  # imageio.rs:325-390
    rng = np.random.default_rng(5)
    rgb = rng.uniform(0, 1, (6, 5, 3)).astype(np.float32)
    png = tmp_path / "t.png"
    pbrt_amd.write_image(png, rgb)
    from PIL import Image
    got = np.asarray(Image.open(png))
    want = np.vectorize(oracle.to_byte, otypes=[np.uint8])(rgb)
    assert got.shape == (6, 5, 3) and np.array_equal(got, want)  # decoded == to_byte(p), imageio.rs:345-356
    pfm = tmp_path / "t.pfm"
    pbrt_amd.write_image(pfm, rgb)
    raw = pfm.read_bytes()
    assert raw.startswith(b"PF\n5 6\n-1\n")  # imageio.rs:186-196, little-endian host
    data = np.frombuffer(raw[len(b"PF\n5 6\n-1\n"):], "<f4").reshape(6, 5, 3)[::-1]  # rows bottom to top
    assert np.array_equal(data, rgb)  # exact round trip, imageio.rs:363-389
    with pytest.raises(pbrt_amd._lib.PbrtHipError) as e:
        pbrt_amd.write_image(tmp_path / "t.exr", rgb)  # imageio.rs:272: unimplemented!("writing .exr files is not implemented")
    assert "writing .exr files is not implemented" in str(e.value)
    with pytest.raises(pbrt_amd._lib.PbrtHipError) as e:
        pbrt_amd.write_image(tmp_path / "t.jpeg", rgb)  # imageio.rs:281
    assert "unknown file extension jpeg" in str(e.value)
    with pytest.raises(pbrt_amd._lib.PbrtHipError) as e:
        pbrt_amd.write_image(tmp_path / "no" / "such" / "dir.png", rgb)  # imageio.rs:248
    assert "Failed to create file" in str(e.value)


def _reference_test_image():  # imageio.rs:298-309: 64x64, (x/64, y/64, 1)
    y, x = np.mgrid[0:64, 0:64]
    return np.stack([x / 64.0, y / 64.0, np.ones_like(x, float)], axis=-1).astype(np.float32)


def test_roundtrip_png(tmp_path, oracle):  # imageio.rs:325-361
    img = _reference_test_image()
    name = tmp_path / "imageio-roundtrip.png"
    pbrt_amd.write_image(name, img)
    got = pbrt_amd.read_image(name)
    want = np.vectorize(oracle.to_byte, otypes=[np.uint8])(img).astype(np.float32) / np.float32(255)
    assert got.shape == (64, 64, 3) and np.array_equal(got, want)


def test_roundtrip_pfm(tmp_path):  # imageio.rs:363-390
    img = _reference_test_image()
    name = tmp_path / "imageio-roundtrip.pfm"
    pbrt_amd.write_image(name, img)
    got = pbrt_amd.read_image(name)
    assert np.array_equal(got, img)


def _png_idat(path):
    import struct
    b, at, out, kinds = open(path, "rb").read(), 8, b"", []
    while at < len(b):
        n, = struct.unpack(">I", b[at:at + 4])
        kinds.append(b[at + 4:at + 8])
        if kinds[-1] == b"IDAT":
            out += b[at + 8:at + 8 + n]
        at += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND"
    return out


def test_png_writer_compresses_and_other_decoders_agree(tmp_path, oracle):
    """The writer's deflate stream (row filters, LZ77, stored / fixed / dynamic blocks -- imageio.cpp zlib_deflate) read by two decoders
    that share no code with it: the zlib of the Python standard library on the IDAT payload, and PIL on the file.  Pixels are
    to_byte(p) (imageio.rs:345-356) whatever the coding; sizes stay within a few per cent of zlib level 6 on the same filtered rows."""
    import zlib
    from PIL import Image
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:200, 0:300]
    cases = {
        "noise": rng.random((97, 131, 3), dtype=np.float32),                      # incompressible: stored blocks
        "flat": np.full((64, 64, 3), 0.25, np.float32),                           # one long run
        "grad": np.stack([xx / 300, yy / 200, xx / 300], -1).astype(np.float32),  # Sub / Up / Paeth rows of constants
        "one": np.zeros((1, 1, 3), np.float32),                                   # a handful of symbols: the fixed code
        "wide": rng.random((2, 3, 3), dtype=np.float32),
        "render": (0.5 + 0.02 * rng.standard_normal((300, 260, 3))).astype(np.float32),  # small residuals: dynamic codes, > 64 K tokens
        "tiles": np.kron(rng.random((12, 12, 3)), np.ones((25, 25, 1))).astype(np.float32),  # matches at distances up to the window
        "dark": (rng.random((150, 150, 3)) ** 8).astype(np.float32),
    }
    block_kinds = set()
    for name, img in cases.items():
        img = np.ascontiguousarray(img)
        path = tmp_path / f"{name}.png"
        pbrt_amd.write_image(path, img)
        h, w, _ = img.shape
        want = np.vectorize(oracle.to_byte, otypes=[np.uint8])(img)
        z = _png_idat(path)
        block_kinds.add((z[2] >> 1) & 3)  # BTYPE of the first block
        rows = np.frombuffer(zlib.decompress(z), np.uint8).reshape(h, 1 + 3 * w)
        assert set(rows[:, 0].tolist()) <= {0, 1, 2, 4}, name
        assert np.array_equal(np.asarray(Image.open(path)), want), name
        assert np.array_equal(pbrt_amd.read_image(path), want.astype(np.float32) / np.float32(255)), name
        assert len(z) <= len(zlib.compress(rows.tobytes(), 6)) * 1.05 + 64, (name, len(z))
        assert len(z) <= rows.size + 5 * (rows.size // 65535 + 1) + 6, name  # never worse than stored blocks
    assert block_kinds == {0, 1, 2}, block_kinds
    assert (tmp_path / "grad.png").stat().st_size < 2000  # 180 KB of scanlines


def test_png_reader_decodes_compressed_filtered_files(tmp_path):
    """Files from another encoder (PIL: dynamic Huffman blocks, all five row filters), RGB / RGBA / grey."""
    from PIL import Image
    rng = np.random.default_rng(9)
    smooth = (np.add.outer(np.arange(97), np.arange(131)) % 256).astype(np.uint8)
    for mode, arr in [("RGB", rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)),
                      ("RGB", np.stack([smooth, smooth[::-1], smooth // 2], -1)),
                      ("RGBA", rng.integers(0, 256, (17, 23, 4), dtype=np.uint8)), ("L", smooth)]:
        name = tmp_path / f"t_{mode}_{arr.shape[0]}.png"
        Image.fromarray(arr, mode).save(name, optimize=True)
        got = pbrt_amd.read_image(name)
        ref = np.asarray(Image.open(name).convert("RGB")).astype(np.float32) / np.float32(255)
        assert np.array_equal(got, ref), mode
    with pytest.raises(pbrt_amd._lib.PbrtHipError):
        pbrt_amd.read_image(tmp_path / "missing.png")
    with pytest.raises(pbrt_amd._lib.PbrtHipError):
        pbrt_amd.read_image(tmp_path / "x.tga")  # imageio.rs:180


def test_sobol_generator_matrices(oracle):
    """The Sobol' generator matrices the oracle builds from the Joe-Kuo direction numbers (oracle.cpp sobol_matrix;
    DESIGN.md 3.10): structural properties always, and equality with the reference's SOBOL_MATRICES32
    (src/core/sobolmatrices.rs:81, 52 columns per dimension; its size test is at :60808-60814) where /root/reference is
    mounted.  (The reference never uses the table: no sampler exists in the crate.)"""
    m = oracle.sobol_matrices()
    assert m.shape[1] == 52 and m.shape[0] >= 2
    assert [int(v) for v in m[0, :32]] == [1 << (31 - i) for i in range(32)] and not m[0, 32:].any()  # van der Corput
    for d in range(m.shape[0]):  # column i has its leading one at bit 31 - i: an upper-triangular, invertible matrix
        for i in range(32):
            assert int(m[d, i]) >> (31 - i) == 1 or (int(m[d, i]) >> (31 - i)) & 1 == 1
            assert int(m[d, i]) & ((1 << (31 - i)) - 1) == 0
    # the first 2^k points of dimensions (1, 2) are a (0, 2)-net: one point in every elementary interval of area 2^-k
    pts = oracle.sobol_points(64)
    for a in range(7):
        nx, ny = 1 << a, 1 << (6 - a)
        assert len({(int(x * nx), int(y * ny)) for x, y in pts}) == 64
    path = "/root/reference/src/core/sobolmatrices.rs"
    if not os.path.exists(path):
        pytest.skip("reference not mounted")
    import re
    text = open(path).read()
    start = text.index("const SOBOL_MATRICES32")
    body = text[text.index("[", text.index("=", start)):text.index("];", start)]
    ref = np.array([int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", body)[:m.size]], np.uint32).reshape(m.shape)
    assert np.array_equal(m, ref), np.argwhere(m != ref)[:5]
