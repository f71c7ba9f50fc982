"""An independent Monte Carlo estimate of a small scene's image, in float64 numpy: test infrastructure, like oracle/, but sharing no code,
no random numbers and no sampling strategy with either the oracle or the library.

The scene (furnished_box_scene): the closed box [-1, 1]^3 seen from inside -- matte floor / ceiling / back / front walls, a red matte left
wall, a MIRROR right wall --, one square emitter just under the ceiling facing down.  The estimator here has no light sampling at all: a
path scatters (cosine-weighted about the wall's inward normal, by its own formula; mirror: reflection) and collects the emitter's radiance
only when it runs into it, for at most max_depth + 1 segments -- which is the same integral pbrt-v3's PathIntegrator estimates with next
event estimation, Russian roulette and the "emission after a specular bounce" rule.  If the product double-counted or lost
a term (emission after a diffuse bounce; the light's pdf conversion r^2 / (A cos); the x n_lights of uniform light selection; the roulette
reweighting; the depth at which a path ends), the two images would differ by far more than the sampling error compared here."""
import numpy as np

LIGHT_HALF = 0.4
LIGHT_Z = 0.99
LE = np.array([12.0, 11.0, 9.0])
KD = {"floor": (0.7, 0.7, 0.7), "ceiling": (0.6, 0.6, 0.6), "back": (0.5, 0.6, 0.7), "front": (0.4, 0.4, 0.4), "left": (0.65, 0.1, 0.1)}
KR = (0.8, 0.8, 0.8)  # the right wall (x = +1) is a mirror
EYE, LOOK, UP, FOV = (0.0, -0.9, -0.2), (0.1, 0.0, -0.1), (0.0, 0.0, 1.0), 75.0


def furnished_box_scene(xres, yres):
    """The same scene as arrays for pbrt_hip_scene_create / the oracle."""
    from pbrt_amd import MATTE, MIRROR, SceneData, look_at
    mats = [[MATTE, *KD["floor"], 0, 0, 0], [MATTE, *KD["ceiling"], 0, 0, 0], [MATTE, *KD["back"], 0, 0, 0], [MATTE, *KD["front"], 0, 0, 0],
            [MATTE, *KD["left"], 0, 0, 0], [MIRROR, *KR, 0, 0, 0], [MATTE, 0, 0, 0, *LE]]
    quads = [(((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), 0), (((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)), 1),
             (((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)), 2), (((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)), 3),
             (((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)), 4), (((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)), 5),
             (((-LIGHT_HALF, -LIGHT_HALF, LIGHT_Z), (-LIGHT_HALF, LIGHT_HALF, LIGHT_Z), (LIGHT_HALF, LIGHT_HALF, LIGHT_Z), (LIGHT_HALF, -LIGHT_HALF, LIGHT_Z)), 6)]  # normal -z
    V, I, M = [], [], []
    for q, m in quads:
        b = len(V)
        V.extend(q)
        I.extend([[b, b + 1, b + 2], [b, b + 2, b + 3]])
        M.extend([m, m])
    return SceneData(P=np.array(V, np.float32), idx=np.array(I, np.uint32), mat_id=np.array(M, np.uint16), materials=np.array(mats, np.float32),
                     cam_to_world=look_at(EYE, LOOK, UP)[1], fov=FOV, xres=xres, yres=yres).normalized()


def _camera_rays(rng, n, xres, yres):
    eye, look, up = np.array(EYE), np.array(LOOK), np.array(UP)
    fwd = (look - eye) / np.linalg.norm(look - eye)
    right = np.cross(up / np.linalg.norm(up), fwd)
    right /= np.linalg.norm(right)
    new_up = np.cross(fwd, right)
    aspect = xres / yres
    wx, wy = (aspect, 1.0) if aspect >= 1 else (1.0, 1.0 / aspect)
    t = np.tan(np.radians(FOV) / 2)
    fx, fy = rng.random(n) * xres, rng.random(n) * yres  # continuous raster position: uniform over the film
    d = ((2 * fx / xres - 1) * wx * t)[:, None] * right + ((1 - 2 * fy / yres) * wy * t)[:, None] * new_up + fwd
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.tile(eye, (n, 1)), d, fx.astype(np.int64), fy.astype(np.int64)


def _scatter_cosine(rng, normal):
    """A cosine-weighted direction about `normal` [n, 3]: a uniform point in the unit disc by rejection-free polar coordinates, lifted."""
    n = normal.shape[0]
    r, phi = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    a = np.where(np.abs(normal[:, [0]]) > 0.5, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t1 = np.cross(normal, a)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(normal, t1)
    return (r * np.cos(phi))[:, None] * t1 + (r * np.sin(phi))[:, None] * t2 + np.sqrt(np.maximum(0, 1 - r * r))[:, None] * normal


_cache = {}


def block_means(xres, yres, block, max_depth, n_paths, seed=12345, batch=1 << 20):
    """-> (mean[by, bx, 3], standard error[by, bx, 3]) of the radiance over blocks of block x block pixels."""
    key = (xres, yres, block, max_depth, n_paths, seed)
    if key not in _cache:
        _cache[key] = _block_means(xres, yres, block, max_depth, n_paths, seed, batch)
    return _cache[key]


def _block_means(xres, yres, block, max_depth, n_paths, seed, batch):
    rng = np.random.default_rng(seed)
    bx_n, by_n = xres // block, yres // block
    s1 = np.zeros((by_n * bx_n, 3))
    s2 = np.zeros((by_n * bx_n, 3))
    cnt = np.zeros(by_n * bx_n)
    wall_kd = np.array([KD["left"], (0, 0, 0), KD["front"], KD["back"], KD["floor"], KD["ceiling"]])  # index 2 * axis + (exit on the + side)
    done = 0
    while done < n_paths:
        n = min(batch, n_paths - done)
        done += n
        o, d, px, py = _camera_rays(rng, n, xres, yres)
        beta = np.ones((n, 3))
        L = np.zeros((n, 3))
        alive = np.ones(n, bool)
        for _segment in range(max_depth + 1):
            idx = np.nonzero(alive)[0]
            if idx.size == 0:
                break
            oo, dd = o[idx], d[idx]
            with np.errstate(divide="ignore", invalid="ignore"):
                t_axis = (np.where(dd > 0, 1.0, -1.0) - oo) / dd  # distance to the box's plane ahead, per axis
                t_axis = np.where(dd == 0, np.inf, t_axis)
                t_light = np.where(dd[:, 2] != 0, (LIGHT_Z - oo[:, 2]) / dd[:, 2], np.inf)
            axis = np.argmin(t_axis, axis=1)
            t_wall = t_axis[np.arange(idx.size), axis]
            p_l = oo + t_light[:, None] * dd
            on_light = (t_light > 1e-9) & (t_light < t_wall) & (np.abs(p_l[:, 0]) <= LIGHT_HALF) & (np.abs(p_l[:, 1]) <= LIGHT_HALF)
            from_below = on_light & (dd[:, 2] > 0)  # the emitter faces down; its back (seen from the gap above it) is black
            L[idx[from_below]] += beta[idx[from_below]] * LE
            alive[idx[on_light]] = False  # the emitter's Kd is 0: the path ends there
            go = ~on_light
            idx, oo, dd, axis, t_wall = idx[go], oo[go], dd[go], axis[go], t_wall[go]
            p = oo + t_wall[:, None] * dd
            plus = dd[np.arange(idx.size), axis] > 0
            normal = np.zeros((idx.size, 3))
            normal[np.arange(idx.size), axis] = np.where(plus, -1.0, 1.0)  # inward
            wall = 2 * axis + plus
            mirror = wall == 1  # x = +1
            nd = np.empty_like(dd)
            nd[mirror] = dd[mirror] - 2 * (dd[mirror] * normal[mirror]).sum(1, keepdims=True) * normal[mirror]
            beta[idx[mirror]] *= KR
            dif = ~mirror
            nd[dif] = _scatter_cosine(rng, normal[dif])
            beta[idx[dif]] *= wall_kd[wall[dif]]  # f cos / pdf = Kd for a cosine-weighted direction
            o[idx] = p
            d[idx] = nd
        b = (py // block) * bx_n + (px // block)
        for c in range(3):
            s1[:, c] += np.bincount(b, L[:, c], by_n * bx_n)
            s2[:, c] += np.bincount(b, L[:, c] ** 2, by_n * bx_n)
        cnt += np.bincount(b, minlength=by_n * bx_n)
    mean = s1 / cnt[:, None]
    var = np.maximum(s2 / cnt[:, None] - mean ** 2, 0)
    return mean.reshape(by_n, bx_n, 3), np.sqrt(var / cnt[:, None]).reshape(by_n, bx_n, 3)


def compare_with_blocks(rgb, mean, se, block):
    """rgb[y, x, 3]: a render of furnished_box_scene.  -> (largest |difference| in units of the estimate's standard error + 0.4 % of the
    value, the relative difference of the whole image's sum)."""
    by_n, bx_n, _ = mean.shape
    got = rgb.astype(np.float64).reshape(by_n, block, bx_n, block, 3).mean((1, 3))
    z = np.abs(got - mean) / (se + 0.004 * np.abs(mean) + 1e-12)
    return float(z.max()), float(got.sum() / mean.sum() - 1)
