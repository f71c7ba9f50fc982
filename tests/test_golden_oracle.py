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
