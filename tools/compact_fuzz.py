"""Random scenes (tools/parity_fuzz.py's generator) rendered at a frame size whose wavefronts hold several rows of slots, with adaptive
sampling, once with shading passes row by row (PT_COMPACT=0) and once over compacted lists of the ready slots (PT_COMPACT=1): the two
frames must agree bit for bit (which lane shades which slot must not matter), in both variants of the kernel (PT_LDS_SMALL=0 keeps small
trees in HBM); 400 random pixels of every frame are also rendered by the CPU oracle with the same engines and compared bit for bit.
        python tools/compact_fuzz.py [first_seed] [n_scenes] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from cpupathtrace_amd import binding, scenes
from tools.parity_fuzz import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
bad = 0
ol = oracle.Checker("oracle")
for seed in range(first, first + n):
    desc, cam = random_scene(seed, tri_scale=1 + seed % 8)
    rng = np.random.default_rng(seed)
    mn = int(rng.integers(2, 9))
    mx = mn + int(rng.integers(4, 40))
    opt = scenes.options(size, size - 32 * int(rng.integers(0, 5)), mn, mx, float(rng.choice([1e-3, 1e-2, 1e-4])))
    frames, drawn = [], []
    for compact in (0, 1):
        os.environ["PT_COMPACT"] = str(compact)
        os.environ["PT_LDS_SMALL"] = str(seed % 2)
        s = binding.Scene(desc)
        img, st = s.process_job(cam, opt, base_seed=1000 + seed, want_stats=True)
        frames.append(img.copy())
        drawn.append(st["samples"])
        s.close()
    same = bool((frames[0].view(np.uint32) == frames[1].view(np.uint32)).all()) and drawn[0] == drawn[1]
    # the oracle on 400 pixels of the frame
    w, h = opt["image_width"], opt["image_height"]
    xs, ys = rng.integers(0, w, 400).astype(np.int32), rng.integers(0, h, 400).astype(np.int32)
    states = np.array([binding.seed_to_state(binding.pixel_seed(1000 + seed, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
    ho = ol.scene_create(desc)
    want, _ = ho.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=16)
    ho.close()
    g, wv = frames[1][ys, xs], want[ys, xs]
    both_nan = np.isnan(g) & np.isnan(wv)
    same = same and not ((g.view(np.uint32) != wv.view(np.uint32)) & ~both_nan).any()
    bad += 0 if same else 1
    print("seed %d: %d objects, %d..%d spp, %.1f %% of the samples drawn, tree %s: %s" % (
        seed, len(desc["obj_kind"]), mn, mx, 100.0 * drawn[1] / (opt["image_width"] * opt["image_height"] * mx), "in LDS" if seed % 2 else "in HBM",
        "identical, oracle pixels too" if same else "DIFFERENT"), flush=True)
print("%d of %d scenes differ" % (bad, n))
sys.exit(1 if bad else 0)
