"""A SECOND implementation of the path of DESIGN.md section 3, in float64 numpy, that shares the RANDOM NUMBERS with the oracle and the
library and nothing else: test infrastructure, like oracle/, written from the spec's text (3.1 sampler and chunks, 3.2 camera, 3.5
Moeller-Trumbore -- the textbook's, every ray against every triangle, no BVH, no own-box rule --, 3.7 surface and BSDFs, 3.8 one-light direct
estimate, 3.9 path loop, roulette and film), with numpy's own sin / cos / sqrt where the spec fixes polynomials.

What it is for: north_star's "PSNR >= 50 dB vs the reference image".  The oracle and the HIP path agree bit for bit, which proves that two
restatements of ONE spec by one author agree; tests/independent_mc.py agrees with them statistically.  This file sits between the two: because
sample s of pixel (x, y) draws the same numbers here, its path is the same path up to rounding, and the IMAGES can be compared directly --
a wrong term (a pdf, a cosine, the n_lights factor, the order of the draws, the depth rule, the roulette weight) would show in every pixel.
Every sampler, integrator and film path of the spec has its second implementation here.  What differs legitimately: float64 against fp32 arithmetic (1e-6 relative), and a handful of paths per image whose discrete decisions (which
triangle at an edge, roulette at the threshold, the picked light) fall the other way.

Supported: triangles and spheres (matte / mirror; emissive triangles = area lights), point / distant / constant-infinite lights, integrators 0
(path), 1 (direct) and 2 (path with the one-sample MIS of 3.14), the stratified sampler, the padded (0,2)-sequence of 3.10 ("sobol") and the
Halton sampler of 3.13 ("halton"), the Sobol' sampler of 3.12 ("sobol_nd": the generator matrices are a table, handed in), checkerboard textures
(3.15; numpy's arctan2 / arccos for a sphere's (u, v)), box filters of any radius (3.11: the fixed-point film) and "maxsampleluminance"."""
import numpy as np

_M = np.uint64(0x5851F42D4C957F2D)
_EPS1 = np.float32(1.0) - np.float32(2.0 ** -23)  # 1 - f32::EPSILON (core/rng.rs:19)


class _Pcg:
    """n PCG32 streams (core/rng.rs:46-93: set_sequence, uniform_u32, uniform_float), advanced only where `mask` says so"""

    def __init__(self, seq):
        seq = np.asarray(seq, np.uint64)
        self.inc = (seq << np.uint64(1)) | np.uint64(1)
        self.state = np.zeros_like(seq)
        self._u32(np.ones(len(seq), bool))
        with np.errstate(over="ignore"):
            self.state = self.state + np.uint64(0x853C49E6748FEA9B)
        self._u32(np.ones(len(seq), bool))

    def _u32(self, mask):
        old = self.state
        with np.errstate(over="ignore"):
            self.state = np.where(mask, old * _M + self.inc, old)
        xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xs >> rot) | (xs << ((~rot + np.uint32(1)) & np.uint32(31)))

    def uniform(self, mask):
        """one uniform_float per stream of `mask` (fp32, as the spec's arithmetic has it); the others keep their state"""
        u = self._u32(mask).astype(np.float32) * np.float32(2.3283064365386963e-10)
        return np.minimum(u, _EPS1)


    def get1(self, mask):
        return self.uniform(mask)

    def get2(self, mask):
        return self.uniform(mask), self.uniform(mask)

    def start_sample(self, s):
        pass


def _mix32(v):
    v = np.asarray(v, np.uint32).copy()
    with np.errstate(over="ignore"):
        v ^= v >> np.uint32(16); v *= np.uint32(0x7FEB352D); v ^= v >> np.uint32(15); v *= np.uint32(0x846CA68B); v ^= v >> np.uint32(16)
    return v


def _bitrev32(v):
    v = np.asarray(v, np.uint32)
    v = ((v >> np.uint32(1)) & np.uint32(0x55555555)) | ((v & np.uint32(0x55555555)) << np.uint32(1))
    v = ((v >> np.uint32(2)) & np.uint32(0x33333333)) | ((v & np.uint32(0x33333333)) << np.uint32(2))
    v = ((v >> np.uint32(4)) & np.uint32(0x0F0F0F0F)) | ((v & np.uint32(0x0F0F0F0F)) << np.uint32(4))
    v = ((v >> np.uint32(8)) & np.uint32(0x00FF00FF)) | ((v & np.uint32(0x00FF00FF)) << np.uint32(8))
    return (v >> np.uint32(16)) | (v << np.uint32(16))


def _primes(n):
    out, k = [], 2
    while len(out) < n:
        if all(k % q for q in out if q * q <= k):
            out.append(k)
        k += 1
    return out


class _Lds:
    """The low-discrepancy samplers as DESIGN.md writes them: "02" = the padded (0,2)-sequence of 3.10, "halton" = 3.13 -- integer arithmetic
    keyed by (pixel, sample number, request number); nothing but the request counter is state."""
    G = np.uint32(0x9E3779B9)

    def __init__(self, kind, q, spp, matrices=None):
        self.mat = None if matrices is None else np.asarray(matrices, np.uint32).reshape(128, 32)
        q = np.asarray(q, np.uint64)
        with np.errstate(over="ignore"):
            self.key = _mix32((q & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ _mix32((q >> np.uint64(32)).astype(np.uint32) + self.G))
        self.kind, self.n = kind, len(q)
        self.mask = np.uint32((1 << int(np.ceil(np.log2(max(spp, 1))))) - 1) if spp > 1 else np.uint32(0)
        self.primes = _primes(128)
        self.j = np.zeros(self.n, np.uint32)
        self.s = 0

    def start_sample(self, s):
        self.s, self.j = s, np.zeros(self.n, np.uint32)

    @staticmethod
    def _u(bits):
        return np.minimum(bits.astype(np.float32) * np.float32(2.3283064365386963e-10), _EPS1)

    def _padded(self, j):
        with np.errstate(over="ignore"):
            a = _mix32(self.key + j * self.G)
        i = np.uint32(self.s) ^ (a & self.mask)
        x = _bitrev32(i)
        y, k, v = np.zeros(self.n, np.uint32), i.copy(), np.full(self.n, 0x80000000, np.uint32)
        while k.any():
            y ^= np.where((k & np.uint32(1)) != 0, v, np.uint32(0))
            k >>= np.uint32(1)
            v ^= v >> np.uint32(1)
        return self._u(x ^ _mix32(a ^ np.uint32(0x68E31DA4))), self._u(y ^ _mix32(a ^ np.uint32(0xB5297A4D)))

    def _halton_dim(self, d):
        """dimension d (an array) of sample self.s under the pixel keys"""
        out = np.zeros(self.n, np.float32)
        i = np.uint32(self.s)
        for dd in np.unique(d):
            sel = d == dd
            b = self.primes[int(dd)]
            with np.errstate(over="ignore"):
                h = _mix32(self.key[sel] + np.uint32(int(dd) + 1) * self.G)
            if b == 2:
                out[sel] = self._u(_bitrev32(np.full(sel.sum(), i, np.uint32)) ^ h)
                continue
            v, nrem, pw = np.zeros(sel.sum(), np.uint32), int(i), 1
            while True:
                pw *= b
                a_k = nrem % b
                nrem //= b
                with np.errstate(over="ignore"):
                    h = h * np.uint32(0x9E3779B1) + np.uint32(0x7F4A7C15)
                w = np.uint32(a_k) * (np.uint32(1) + (((h >> np.uint32(16)) * np.uint32(b - 1)) >> np.uint32(16))) + (((h & np.uint32(0xFFFF)) * np.uint32(b)) >> np.uint32(16))
                v = v * np.uint32(b) + w % np.uint32(b)
                if pw > int(self.mask):
                    break
            with np.errstate(over="ignore"):
                h = h * np.uint32(0x9E3779B1) + np.uint32(0x7F4A7C15)
            u = (v.astype(np.float32) + h.astype(np.float32) * np.float32(2.3283064365386963e-10)) * (np.float32(1.0) / np.float32(pw))
            out[sel] = np.minimum(u, _EPS1)
        return out

    def _sobol_dim(self, d):
        """3.12: dimension d (an array) of the Sobol' sequence at index self.s -- XOR of the generator matrix's columns at the set bits --,
        scrambled per pixel and dimension"""
        x = np.zeros(self.n, np.uint32)
        for b in range(32):
            if (self.s >> b) & 1:
                x ^= self.mat[d, b]
        with np.errstate(over="ignore"):
            return self._u(x ^ _mix32(self.key + (d.astype(np.uint32) + np.uint32(1)) * self.G))

    def get2(self, mask):
        j = self.j.copy()
        assert int(j[mask].max(initial=0)) < 64 or self.kind in ("02", "sobol")  # (requests beyond the 64th fall back to the padded ones: not walked here)
        self.j = np.where(mask, self.j + np.uint32(1), self.j)
        if self.kind in ("02", "sobol"):
            return self._padded(j)
        if self.kind == "sobol_nd":
            return self._sobol_dim(2 * j.astype(np.int64)), self._sobol_dim(2 * j.astype(np.int64) + 1)
        return self._halton_dim(2 * j.astype(np.int64)), self._halton_dim(2 * j.astype(np.int64) + 1)

    def get1(self, mask):  # a 1-D request takes the first coordinate of its pair
        j = self.j.copy()
        self.j = np.where(mask, self.j + np.uint32(1), self.j)
        if self.kind in ("02", "sobol"):
            return self._padded(j)[0]
        if self.kind == "sobol_nd":
            return self._sobol_dim(2 * j.astype(np.int64))
        return self._halton_dim(2 * j.astype(np.int64))


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _closest(o, d, tmax, p0, e1, e2, any_hit=False):
    """every ray against every triangle, float64: (t, triangle, u, v) of the closest hit in (1e-4, tmax), triangle -1 for a miss;
    ties to the lower triangle number"""
    n = len(o)
    t_best, tri, ub, vb = np.full(n, np.inf), np.full(n, -1, np.int64), np.zeros(n), np.zeros(n)
    step = max(64, 400_000 // max(1, len(p0)))
    for a in range(0, n, step):
        sl = slice(a, min(n, a + step))
        with np.errstate(all="ignore"):
            oo, dd = o[sl, None, :], d[sl, None, :]
            pv = np.cross(dd, e2[None])
            det = (e1[None] * pv).sum(-1)
            inv = 1.0 / det
            tv = oo - p0[None]
            u = (tv * pv).sum(-1) * inv
            qv = np.cross(tv, e1[None])
            v = (dd * qv).sum(-1) * inv
            t = (e2[None] * qv).sum(-1) * inv
            ok = (np.abs(det) >= 1e-8) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-4) & (t < tmax[sl, None])
        tt = np.where(ok, t, np.inf)
        k = tt.argmin(1)  # (argmin returns the first minimum: the lower triangle number at equal t)
        r = np.arange(tt.shape[0])
        hit = np.isfinite(tt[r, k])
        t_best[sl], tri[sl] = tt[r, k], np.where(hit, k, -1)
        ub[sl], vb[sl] = u[r, k], v[r, k]
    return t_best, tri, ub, vb


def _closest_prims(o, d, tmax, p0, e1, e2, sph):
    """triangles (numbers 0 .. T - 1) and spheres (T + s; `sph` rows: centre, radius): the closest hit, ties to the lower primitive number"""
    T = len(p0)
    if T:
        t, prim, ub, vb = _closest(o, d, tmax, p0, e1, e2)
    else:
        t, prim, ub, vb = np.full(len(o), np.inf), np.full(len(o), -1, np.int64), np.zeros(len(o)), np.zeros(len(o))
    for k, row in enumerate(sph):
        c, r = row[:3], row[3]
        oc = o - c
        a = (d * d).sum(1)
        b = 2.0 * (d * oc).sum(1)
        cc = (oc * oc).sum(1) - r * r
        with np.errstate(all="ignore"):
            disc = b * b - 4.0 * a * cc
            rd = np.sqrt(np.maximum(disc, 0.0))
            q = np.where(b < 0, -0.5 * (b - rd), -0.5 * (b + rd))
            r0, r1 = q / a, cc / q
        t0, t1 = np.minimum(r0, r1), np.maximum(r0, r1)
        ok0 = (disc >= 0) & (t0 > 1e-4) & (t0 < tmax)
        ok1 = (disc >= 0) & (t1 > 1e-4) & (t1 < tmax)
        ts = np.where(ok0, t0, np.where(ok1, t1, np.inf))
        closer = ts < t  # (a sphere's number is above every triangle's: at equal t the triangle keeps the hit)
        t, prim = np.where(closer, ts, t), np.where(closer, T + k, prim)
    return t, prim, ub, vb


def _cosine_about(n, u1, u2):
    """pbrt-v3 CosineSampleHemisphere through ConcentricSampleDisk, in the frame CoordinateSystem(n) gives; -> (wi, z)"""
    ox, oy = 2.0 * u1 - 1.0, 2.0 * u2 - 1.0
    with np.errstate(all="ignore"):
        wide = np.abs(ox) > np.abs(oy)
        r = np.where(wide, ox, oy)
        phi = np.where(wide, (np.pi / 4) * (oy / ox), np.pi / 2 - (np.pi / 4) * (ox / oy))
    zero = (ox == 0) & (oy == 0)
    dx, dy = np.where(zero, 0.0, r * np.cos(phi)), np.where(zero, 0.0, r * np.sin(phi))
    z = np.sqrt(np.maximum(0.0, 1.0 - dx * dx - dy * dy))
    big_x = np.abs(n[:, 0]) > np.abs(n[:, 1])
    with np.errstate(all="ignore"):
        v2a = np.stack([-n[:, 2], np.zeros(len(n)), n[:, 0]], 1) / np.sqrt(n[:, 0] ** 2 + n[:, 2] ** 2)[:, None]
        v2b = np.stack([np.zeros(len(n)), n[:, 2], -n[:, 1]], 1) / np.sqrt(n[:, 1] ** 2 + n[:, 2] ** 2)[:, None]
    v2 = np.where(big_x[:, None], v2a, v2b)
    v3 = np.cross(n, v2)
    return v2 * dx[:, None] + v3 * dy[:, None] + n * z[:, None], z


def render(sd, integrator=0, max_depth=5, spp=(1, 1), seed=0, sampler="stratified", filter_width=None, max_sample_luminance=0.0, sobol_matrices=None,
           window=None, trace=None):
    """-> film [h, w, 4] float64 {X, Y, Z, weight} of SceneData `sd` (whole image; `window` = (x0, y0, w, h): those pixels of it only -- the
    default filter: a pixel's samples land in it alone --, which is how a scene of a million triangles is affordable by brute force)"""
    sd = sd.normalized()
    sampler = {0: "stratified", 1: "sobol", 2: "sobol_nd", 3: "halton"}.get(sampler, sampler)  # (the C ABI's numbers: what a loaded scene file carries)
    assert integrator in (0, 1, 2) and sampler in ("stratified", "sobol", "sobol_nd", "halton")
    if tuple(sd.crop) != (0.0, 1.0, 0.0, 1.0):  # a crop window (Film::new, film.rs:82-137: ceil(resolution x crop) in float): those pixels of the whole image, streams numbered in it
        assert window is None and not filter_width
        f32 = np.float32
        bx0, bx1 = (int(np.ceil(f32(sd.xres) * f32(c))) for c in sd.crop[:2])
        by0, by1 = (int(np.ceil(f32(sd.yres) * f32(c))) for c in sd.crop[2:])
        import dataclasses
        return render(dataclasses.replace(sd, crop=(0.0, 1.0, 0.0, 1.0)), integrator, max_depth, spp, seed, sampler, None, max_sample_luminance,
                      sobol_matrices, window=(bx0, by0, bx1 - bx0, by1 - by0))
    assert sampler != "sobol_nd" or sobol_matrices is not None  # (3.12's generator matrices are a table: the caller hands them over)
    rx, ry = (float(v) if float(v) != 0.0 else 0.5 for v in (filter_width or (0.5, 0.5)))
    wide = (rx, ry) != (0.5, 0.5)
    pad_x, pad_y = (int(np.ceil(rx - 0.5)), int(np.ceil(ry - 0.5))) if wide else (0, 0)
    mis = integrator == 2
    sph = sd.spheres.astype(np.float64)
    T = sd.idx.shape[0]
    W, H = int(sd.xres), int(sd.yres)
    nx, ny = spp
    n_spp = nx * ny
    K = 1
    while 2 * K <= 16 and 2 * K * 32 <= n_spp:
        K *= 2
    P = sd.P.astype(np.float64)[sd.idx.astype(np.int64)] if T else np.zeros((0, 3, 3))
    p0, p1, p2 = P[:, 0], P[:, 1], P[:, 2]
    e1, e2 = p1 - p0, p2 - p0
    mats = sd.materials.astype(np.float64)
    mat_of = np.concatenate([sd.mat_id.astype(np.int64), sph[:, 4].astype(np.int64)])  # by primitive number: triangles, then spheres
    if len(mat_of) == 0:
        mat_of = np.zeros(1, np.int64)  # (a scene of lights alone: nothing is ever hit)
    tri_area_all = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1) if T else np.zeros(0)
    textures = sd.textures.astype(np.float64)
    n_tex = len(textures)
    mat_tex = sd.mat_tex.astype(np.int64)
    have_uv = sd.tri_uv.shape[0] == T and T > 0
    tri_uv = sd.tri_uv.astype(np.float64) if have_uv else np.zeros((max(T, 1), 6))
    is_mirror = mats[:, 0] == 1
    kcol, le = mats[:, 1:4], mats[:, 4:7]
    # the light list (3.8): explicit lights in order, then every emissive triangle in index order
    L = []
    for row in sd.lights.astype(np.float64):
        L.append(dict(type=int(row[0]), p=row[1:4], c=row[4:7]))
    le_inf = sum((l["c"] for l in L if l["type"] == 2), np.zeros(3))
    has_inf = any(l["type"] == 2 for l in L)
    for t in range(len(P)):
        if (le[mat_of[t]] > 0).any():
            cr = np.cross(e1[t], e2[t])
            L.append(dict(type=3, p=p0[t], p1=p1[t], p2=p2[t], c=le[mat_of[t]], n=cr / np.linalg.norm(cr), area=0.5 * np.linalg.norm(cr)))
    nL = len(L)
    nLf = np.float32(nL)
    ltype = np.array([l["type"] for l in L], np.int64) if nL else np.zeros(0, np.int64)
    lp = np.array([l["p"] for l in L]).reshape(-1, 3)
    lc = np.array([l["c"] for l in L]).reshape(-1, 3)
    lp1 = np.array([l.get("p1", np.zeros(3)) for l in L]).reshape(-1, 3)
    lp2 = np.array([l.get("p2", np.zeros(3)) for l in L]).reshape(-1, 3)
    ln = np.array([l.get("n", np.zeros(3)) for l in L]).reshape(-1, 3)
    larea = np.array([l.get("area", 0.0) for l in L])
    # the camera (3.2)
    c2w = sd.cam_to_world.astype(np.float64)
    aspect = W / H
    x0, x1, y0, y1 = (-aspect, aspect, -1.0, 1.0) if aspect >= 1 else (-1.0, 1.0, -1.0 / aspect, 1.0 / aspect)
    th = float(np.float32(np.tan(float(sd.fov) * np.pi / 360.0)))
    ax, bx, ay, by = (x1 - x0) / W * th, x0 * th, -(y1 - y0) / H * th, y1 * th
    # one stream per (pixel, chunk); the samples of a chunk run in order on their stream
    # the SAMPLED pixels (3.11): the image plus a halo of pad pixels for a box filter wider than 0.5; their streams are numbered in the
    # haloed image W' x H'
    py, px = np.mgrid[-pad_y:H + pad_y, -pad_x:W + pad_x]
    if window is not None:
        assert not wide
        py, px = np.mgrid[window[1]:window[1] + window[3], window[0]:window[0] + window[2]]
    px, py = px.ravel(), py.ravel()
    n_px = len(px)
    Wn, Hn = W + 2 * pad_x, H + 2 * pad_y
    film = np.zeros((n_px, 3))
    acc = np.zeros((H, W, 4), np.int64)  # the fixed-point film of a wide filter
    for c in range(K):
        s_lo, s_hi = (c * n_spp) // K, ((c + 1) * n_spp) // K
        with np.errstate(over="ignore"):
            qpix = np.uint64(seed) * np.uint64(Wn) * np.uint64(Hn) + (py + pad_y).astype(np.uint64) * np.uint64(Wn) + (px + pad_x).astype(np.uint64)
            rng = _Pcg(qpix * np.uint64(K) + np.uint64(c)) if sampler == "stratified" else _Lds(sampler, qpix, n_spp, sobol_matrices)
        part = np.zeros((n_px, 3))
        everyone = np.ones(n_px, bool)
        for s in range(s_lo, s_hi):
            sx, sy = s % nx, s // nx
            rng.start_sample(s)
            u1, u2 = rng.get2(everyone)
            if sampler == "stratified":
                jx = np.minimum((np.float32(sx) + u1) * (np.float32(1) / np.float32(nx)), _EPS1)
                jy = np.minimum((np.float32(sy) + u2) * (np.float32(1) / np.float32(ny)), _EPS1)
            else:  # (3.10: the net's point is the film offset itself)
                jx, jy = u1, u2
            fx = (px.astype(np.float32) + jx).astype(np.float64)
            fy = (py.astype(np.float32) + jy).astype(np.float64)
            dc = _unit(np.stack([fx * ax + bx, fy * ay + by, np.ones(n_px)], 1))
            d = dc @ c2w[:3, :3].T
            o = np.tile(c2w[:3, 3], (n_px, 1))
            Lsum, beta = np.zeros((n_px, 3)), np.ones((n_px, 3))
            alive = np.ones(n_px, bool)
            specular = np.zeros(n_px, bool)
            pb_prev = np.zeros(n_px)  # MIS: the density the ray in flight was drawn with (cos / pi of its cosine sample)
            bounces = 0
            while alive.any():
                a = np.flatnonzero(alive)
                t, tri, ub, vb = _closest_prims(o[a], d[a], np.full(len(a), np.inf), p0, e1, e2, sph)
                hit = tri >= 0
                on_tri = hit & (tri < T)
                wo = -d[a]
                m = mat_of[np.maximum(tri, 0)]
                ti = np.where(on_tri, tri, 0)
                ng = _unit(np.cross(e1[ti], e2[ti])) if T else np.zeros((len(a), 3))
                if len(sph):
                    si = np.clip(tri - T, 0, len(sph) - 1)
                    ph = (o[a] - sph[si, :3]) + d[a] * np.where(hit, t, 0.0)[:, None]
                    ng = np.where(on_tri[:, None], ng, ph / sph[si, 3:4])
                collect = (bounces == 0) | specular[a]
                front = (ng * wo).sum(1) > 0
                emits = hit & front & (le[m] > 0).any(1)
                Lsum[a] += np.where((emits & collect)[:, None], beta[a] * le[m], 0.0)
                if has_inf:
                    Lsum[a] += np.where((~hit & collect)[:, None], beta[a] * le_inf, 0.0)
                if mis:  # 3.14: the BSDF-sampled half of the previous vertex's estimate, where the bounce ray lands on a light
                    with np.errstate(all="ignore"):
                        cl = (ng * wo).sum(1)
                        pl = ((t * t) / (cl * tri_area_all[ti])) / nL if T else np.zeros(len(a))
                        wb = pb_prev[a] ** 2 / (pb_prev[a] ** 2 + pl ** 2)
                    Lsum[a] += np.where((emits & on_tri & ~collect)[:, None], beta[a] * le[m] * wb[:, None], 0.0)
                    if has_inf:
                        Lsum[a] += np.where((~hit & ~collect)[:, None], beta[a] * le_inf * (nL * nL / (nL * nL + 1.0)), 0.0)
                alive[a[~hit]] = False
                if bounces >= max_depth:  # (such a ray was traced for its emission alone)
                    alive[a] = False
                    break
                a, tri, ub, vb, m, ng, wo, on_tri, t = a[hit], tri[hit], ub[hit], vb[hit], m[hit], ng[hit], wo[hit], on_tri[hit], t[hit]
                if len(a) == 0:
                    break
                w = (1.0 - ub) - vb
                ti = np.where(on_tri, tri, 0)
                p = p0[ti] * w[:, None] + p1[ti] * ub[:, None] + p2[ti] * vb[:, None] if T else np.zeros((len(a), 3))
                if len(sph):
                    si = np.clip(tri - T, 0, len(sph) - 1)
                    p = np.where(on_tri[:, None], p, sph[si, :3] + ((o[a] - sph[si, :3]) + d[a] * t[:, None]))
                nf = np.where(((ng * wo).sum(1) < 0)[:, None], -ng, ng)
                po = p + nf * 1e-4
                matte = ~is_mirror[m]
                k = kcol[m]
                if n_tex and (mat_tex[m] > 0).any():
                    # 3.15: a matte Kd from a checkerboard over the primitive's (u, v), point-sampled
                    if T and have_uv:
                        uvc = tri_uv[ti]
                        tu = (uvc[:, 0] * w + uvc[:, 2] * ub) + uvc[:, 4] * vb
                        tv_ = (uvc[:, 1] * w + uvc[:, 3] * ub) + uvc[:, 5] * vb
                    else:
                        tu, tv_ = np.zeros(len(a)), np.zeros(len(a))
                    if len(sph):
                        phi = np.arctan2(ng[:, 1], ng[:, 0])
                        phi = np.where(phi < 0, phi + 2 * np.pi, phi)
                        theta = np.arccos(np.clip(ng[:, 2], -1.0, 1.0))
                        tu = np.where(on_tri, tu, phi / (2 * np.pi))
                        tv_ = np.where(on_tri, tv_, (theta - np.pi) / (0.0 - np.pi))
                    tx = textures[np.maximum(mat_tex[m], 1) - 1]
                    cell = np.floor(tx[:, 7] * tu + tx[:, 9]) + np.floor(tx[:, 8] * tv_ + tx[:, 10])
                    kt = np.where((np.mod(cell, 2) == 0)[:, None], tx[:, 1:4], tx[:, 4:7])
                    k = np.where(((mat_tex[m] > 0) & matte)[:, None], kt, k)
                full = np.zeros(n_px, bool)
                # --- the direct-light estimate at a matte vertex: pick, then the pair, whatever the light's kind (3.1) ---
                lpend = np.zeros((len(a), 3))
                need_shadow = np.zeros(len(a), bool)
                sh_d, sh_t = np.zeros((len(a), 3)), np.full(len(a), np.inf)
                if nL > 0:
                    full[:] = False
                    full[a[matte]] = True
                    xi = rng.get1(full)[a]
                    l1, l2 = (v[a] for v in rng.get2(full))
                    li = np.minimum((xi * nLf).astype(np.int64), nL - 1)  # (fp32 product, truncated: the spec's pick)
                    f = k / np.pi
                    ty = ltype[li]
                    # point
                    dv = lp[li] - po
                    d2 = (dv * dv).sum(1)
                    with np.errstate(all="ignore"):
                        dist = np.sqrt(d2)
                        wi_p = dv / dist[:, None]
                        cs_p = (wi_p * nf).sum(1)
                        ld_p = f * lc[li] * ((cs_p / d2) * nL)[:, None]
                    ok_p = (ty == 0) & (d2 > 0) & (cs_p > 0)
                    # distant
                    cs_d = (lp[li] * nf).sum(1)
                    ok_d = (ty == 1) & (cs_d > 0)
                    ld_d = f * lc[li] * (cs_d * nL)[:, None]
                    # constant infinite: a cosine-sampled direction
                    wi_i, z_i = _cosine_about(nf, l1.astype(np.float64), l2.astype(np.float64))
                    ok_i = (ty == 2) & (z_i != 0)
                    ld_i = k * lc[li] * nL
                    # an emissive triangle: a uniform point (the sqrt form), one-sided
                    su0 = np.sqrt(l1.astype(np.float64))
                    b0 = 1.0 - su0
                    b1 = l2.astype(np.float64) * su0
                    b2 = (1.0 - b0) - b1
                    pl = lp[li] * b0[:, None] + lp1[li] * b1[:, None] + lp2[li] * b2[:, None]
                    dv = pl - po
                    d2t = (dv * dv).sum(1)
                    with np.errstate(all="ignore"):
                        dist_t = np.sqrt(d2t)
                        wi_t = dv / dist_t[:, None]
                        cs_t = (wi_t * nf).sum(1)
                        cl_t = -(wi_t * ln[li]).sum(1)
                        ld_t = f * lc[li] * ((((cs_t * cl_t) * larea[li]) / d2t) * nL)[:, None]
                        if mis:  # power heuristic: this strategy's density against the BSDF's for the same direction
                            pl_t, pb_t = (d2t / (cl_t * larea[li])) / nL, cs_t / np.pi
                            ld_t = ld_t * (pl_t ** 2 / (pl_t ** 2 + pb_t ** 2))[:, None]
                    if mis:
                        ld_i = ld_i * (1.0 / (1.0 + nL * nL))
                    ok_t = (ty == 3) & (d2t > 0) & (cs_t > 0) & (cl_t > 0)
                    need_shadow = matte & (ok_p | ok_d | ok_i | ok_t)
                    with np.errstate(all="ignore"):
                        ld = np.where(ok_p[:, None], ld_p, np.where(ok_d[:, None], ld_d, np.where(ok_i[:, None], ld_i, ld_t)))
                        sh_d = np.where(ok_p[:, None], wi_p, np.where(ok_d[:, None], lp[li], np.where(ok_i[:, None], wi_i, wi_t)))
                        sh_t = np.where(ok_p, dist * 0.9999, np.where(ok_t, dist_t * 0.9999, np.inf))
                    lpend = np.where(need_shadow[:, None], beta[a] * ld, 0.0)
                # --- the BSDF sample: matte = cosine hemisphere (two draws, path integrator only), mirror = reflection (none) ---
                go = np.ones(len(a), bool)
                wi_next = -wo + nf * (2.0 * (wo * nf).sum(1))[:, None]
                if integrator == 1:
                    go &= ~matte  # direct lighting ends at the first matte vertex
                else:
                    full[:] = False
                    full[a[matte]] = True
                    c1, c2 = (v[a] for v in rng.get2(full))
                    wi_c, z_c = _cosine_about(nf, c1.astype(np.float64), c2.astype(np.float64))
                    wi_next = np.where(matte[:, None], wi_c, wi_next)
                    go &= ~(matte & (z_c == 0))
                    pb_prev[a] = z_c / np.pi
                new_beta = np.where(go[:, None], beta[a] * k, beta[a])
                spec_next = ~matte
                go &= ~(new_beta == 0).all(1)
                if bounces > 3:  # Russian roulette (3.9)
                    full[:] = False
                    full[a[go]] = True
                    xr = rng.get1(full)[a]
                    q = np.maximum(0.05, 1.0 - new_beta.max(1))
                    die = go & (xr < q)
                    with np.errstate(all="ignore"):
                        new_beta = np.where((go & ~die)[:, None], new_beta / (1.0 - q)[:, None], new_beta)
                    go &= ~die
                # --- the shadow ray ---
                if need_shadow.any():
                    sidx = np.flatnonzero(need_shadow)
                    _, blocker, _, _ = _closest_prims(po[sidx], sh_d[sidx], sh_t[sidx], p0, e1, e2, sph)
                    lit = np.zeros(len(a), bool)
                    lit[sidx] = blocker < 0
                    Lsum[a] += np.where(lit[:, None], lpend, 0.0)
                beta[a] = new_beta
                specular[a] = spec_next
                # a ray at the depth limit is traced only after a specular bounce, for its emission
                if not mis:  # (with MIS the ray at the limit IS traced: the BSDF half of the last vertex's estimate)
                    go &= ~((bounces + 1 >= max_depth) & ~spec_next)
                alive[a] = go
                o[a], d[a] = po, wi_next
                bounces += 1
            y = 0.212671 * Lsum[:, 0] + 0.715160 * Lsum[:, 1] + 0.072169 * Lsum[:, 2]
            bad = np.isnan(Lsum).any(1) | (y < -1e-5) | np.isinf(y)
            Ls = np.where(bad[:, None], 0.0, Lsum)
            if max_sample_luminance > 0:  # Film "maxsampleluminance" (3.9)
                with np.errstate(all="ignore"):
                    Ls = np.where((y > max_sample_luminance)[:, None] & ~bad[:, None], Ls * (max_sample_luminance / y)[:, None], Ls)
            if trace is not None:  # (debugging: every sample's film point and radiance)
                trace.append((s, px.copy(), py.copy(), fx.copy(), fy.copy(), Ls.copy()))
            if not wide:
                part += Ls
            else:  # 3.11: every pixel within the radius of the film point gets the sample, as 2^-24 fixed point
                q = (np.minimum(np.maximum(Ls, 0.0), 32768.0).astype(np.float32).astype(np.float64) * 16777216.0).astype(np.int64)
                # (d - r and d + r are float32 sums, as 3.11 has them since the random-scene soak: a film point one ulp below 32 under r = 2.5
                # rounds d + r up to 34 and the footprint reaches pixel 34 -- 2e-6 beyond the radius; the weights are compared EXACTLY)
                f32 = np.float32
                dx, dy = fx.astype(f32) - f32(0.5), fy.astype(f32) - f32(0.5)
                x0, x1 = np.maximum(np.ceil(dx - f32(rx)).astype(np.int64), 0), np.minimum(np.floor(dx + f32(rx)).astype(np.int64), W - 1)
                y0, y1 = np.maximum(np.ceil(dy - f32(ry)).astype(np.int64), 0), np.minimum(np.floor(dy + f32(ry)).astype(np.int64), H - 1)
                for oy in range(2 * pad_y + 2):
                    for ox in range(2 * pad_x + 2):
                        xx, yy = x0 + ox, y0 + oy
                        ok = (xx <= x1) & (yy <= y1)
                        np.add.at(acc, (yy[ok], xx[ok]), np.concatenate([q[ok], np.ones((int(ok.sum()), 1), np.int64)], 1))
        film += part
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    if wide:
        return np.concatenate([(acc[..., :3].astype(np.float64) / 16777216.0) @ M.T, acc[..., 3:4].astype(np.float64)], -1)
    out = np.concatenate([film @ M.T, np.full((n_px, 1), float(n_spp))], 1)
    return out.reshape((H, W, 4) if window is None else (window[3], window[2], 4))


def psnr_db(film_a, film_b):
    """PSNR of two {X, Y, Z, weight} films as north_star states it: linear RGB (XYZ / weight through the inverse matrix) clamped to [0, 1], peak 1"""
    Minv = np.linalg.inv(np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]]))

    def rgb(f):
        f = np.asarray(f, np.float64)
        return np.clip((f[..., :3] / np.maximum(f[..., 3:4], 1e-30)) @ Minv.T, 0.0, 1.0)
    mse = float(((rgb(film_a) - rgb(film_b)) ** 2).mean())
    return np.inf if mse == 0 else 10.0 * np.log10(1.0 / mse)
