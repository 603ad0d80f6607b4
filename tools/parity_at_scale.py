"""One-off parity evidence at BASELINE.json configs[4]'s full size: 16 copies of the 7.2 M-triangle stand-in (115 M triangles), 1024 x 1024,
a few samples per pixel on the GPU, N random pixels rendered by the CPU oracle with the same per-pixel engines, compared bit for bit.
(The oracle needs ~25 GB and minutes for a tree of this size: too heavy for the test suite, so it is run by hand and its output kept
under profiles/.)    python tools/parity_at_scale.py [mesh_n = 1900] [spp = 2] [pixels = 512]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_pixels = int(sys.argv[3]) if len(sys.argv) > 3 else 512
t0 = time.time()
desc, cam = scenes.dragon_grid_scene(*scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM), grid=4)
print("scene: %d triangles, generated in %.1f s" % (len(desc["tri_pos"]), time.time() - t0), flush=True)
opt = scenes.options(1024, 1024, spp, spp)
t0 = time.time()
gpu = binding.Scene(desc)
print("device scene built in %.1f s, BVH depth %d" % (time.time() - t0, gpu.info()["depth"]), flush=True)
frame, st = gpu.process_job(cam, opt, base_seed=77, want_stats=True)
print("GPU frame: %.1f Msamples/s (%.0f ms), %.1f slab tests per ray" % (1024 * 1024 * spp / st["kernel_ms"] / 1e3, st["kernel_ms"],
                                                                         (2.0 * st["node_visits"] + st["rays_traced"]) / st["rays_traced"]), flush=True)
gpu.close()
t0 = time.time()
handle = oracle.Checker("oracle").scene_create(desc)
print("oracle scene built in %.1f s" % (time.time() - t0), flush=True)
rng = np.random.default_rng(5)
xs, ys = rng.integers(0, 1024, n_pixels).astype(np.int32), rng.integers(0, 1024, n_pixels).astype(np.int32)
states = np.array([binding.seed_to_state(binding.pixel_seed(77, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
t0 = time.time()
want, _ = handle.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=16)
print("oracle rendered %d pixels in %.1f s" % (n_pixels, time.time() - t0), flush=True)
same = (frame[ys, xs].view(np.uint32) == want[ys, xs].view(np.uint32)).all(axis=1)
print("bit-identical pixels: %d of %d; hit something: %d" % (int(same.sum()), n_pixels, int((want[ys, xs][:, 3] == 1).sum())))
sys.exit(0 if same.all() else 1)
