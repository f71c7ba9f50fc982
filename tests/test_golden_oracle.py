"""The oracle reproduces the committed golden films (regression pin of the spec, CPU only)."""
import os

import numpy as np

from util import SMALL_SCENES, assert_bit_equal


def test_oracle_reproduces_golden(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_golden.npz"))
    assert len(g.files) == 6
    for key in g.files:
        name, integ, depth, sx, sy, seed, sampler = key.split("-")
        film, _ = oracle.OracleScene(SMALL_SCENES[name]()).render(integrator=int(integ), max_depth=int(depth),
                                                                 spp=(int(sx), int(sy)), seed=int(seed), sampler=int(sampler))
        assert_bit_equal(film, g[key], key)
        assert film[..., :3].max() > 0  # not a black image


def test_oracle_reproduces_round5_golden(oracle):
    """What round 5 added beside the default path -- sampler 2's dimensions 33 .. 128, the Halton sampler, integrator 2 (MIS), a
    checkerboard Kd -- pinned the same way (tests/golden/make_golden_r05.py); test_oracle_reproduces_golden above still passing on the
    round-2 file is the evidence that the default path's arithmetic did not move."""
    from util import checker_plane_scene, checker_sphere_scene
    scenes = dict(SMALL_SCENES, checker=lambda: checker_plane_scene(40)[0], checkersphere=lambda: checker_sphere_scene(48, 40))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_golden_r05.npz"))
    assert len(g.files) == 11
    for key in g.files:
        name, integ, depth, sx, sy, seed, sampler = key.split("-")
        film, _ = oracle.OracleScene(scenes[name]()).render(integrator=int(integ), max_depth=int(depth),
                                                            spp=(int(sx), int(sy)), seed=int(seed), sampler=int(sampler))
        assert_bit_equal(film, g[key], key)
        assert film[..., :3].max() > 0
