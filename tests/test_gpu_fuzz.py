"""Differential parity on random scenes (tools/parity_fuzz.py): triangles and spheres at inexact coordinates, flat axis-aligned emitters,
all BSDF kinds, point lights, every aperture kind, 1..30 materials and up to dozens of emitters (both sides of the LDS-table limit) --
the HIP path against the CPU oracle, pixel for pixel and engine state for engine state.  Seeds 5, 6, 15 and 18 are scenes on which the
shadow-walk pruning that was not exact showed."""
import importlib.util
import os

import numpy as np
import pytest

from tests.util import assert_bits_equal

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("parity_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "parity_fuzz.py"))
parity_fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(parity_fuzz)


@pytest.mark.parametrize("first", [0, 8, 16, 24])
def test_random_scenes_vs_oracle(oracle_lib, first):
    from cpupathtrace_amd import binding, scenes
    import oracle
    w, h, spp = 20, 16, 6
    ys, xs = np.mgrid[0:h, 0:w]
    xs, ys = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32)
    for seed in range(first, first + 8):
        desc, cam = parity_fuzz.random_scene(seed)
        opt = scenes.options(w, h, spp, spp, float(np.random.default_rng(seed + 7).choice([1e-3, 1e-4, 1e-2])))
        states = np.array([binding.seed_to_state(binding.pixel_seed(1000 + seed, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
        scene = binding.Scene(desc)
        try:
            img, after = scene.process_item(cam, opt, binding.pixel_streams(xs, ys, states))
        finally:
            scene.close()
        want, want_after = oracle_lib.scene_create(desc).render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=8)
        assert_bits_equal(img, want, "random scene %d (%d objects)" % (seed, len(desc["obj_kind"])))
        assert_bits_equal(after, want_after, "engine states after random scene %d" % seed)
