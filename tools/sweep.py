"""Knob sweep on one scene (the environment is read when a Scene is created).

    python tools/sweep.py <mesh_n | 0 = cornell | -1 = box> <spp> "<list of dicts of env settings, or a dict of lists (grid)>" [size] [min_spp] [copies_per_side]

With min_spp < spp the per-pixel estimator may stop early (adaptive sampling); the rate is then on the max-spp basis and the share of
the samples really drawn is printed.
"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = eval(sys.argv[3]) if len(sys.argv) > 3 else [{}]
size = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
min_spp = int(sys.argv[5]) if len(sys.argv) > 5 else spp
copies = int(sys.argv[6]) if len(sys.argv) > 6 else 1  # 4 = BASELINE.json configs[4]: 16 copies of the mesh in an enlarged box
if isinstance(spec, dict):
    keys = list(spec)
    spec = [dict(zip(keys, combo)) for combo in itertools.product(*[spec[k] for k in keys])]
if mesh_n > 0:
    pos, nrm = scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM)
    sc, cam = scenes.dragon_box_scene(pos, nrm) if copies <= 1 else scenes.dragon_grid_scene(pos, nrm, grid=copies)
else:
    sc, cam = scenes.cornell_scene(size, size) if mesh_n == 0 else scenes.box_scene()
opt = scenes.options(size, size, min_spp, spp)
touched = set()
for env in spec:
    for k in touched:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
        touched.add(k)
    s = binding.Scene(sc)
    s.process_job(cam, scenes.options(size, size, 4, 4))
    best = None
    for _ in range(2):
        img, st = s.process_job(cam, opt, want_stats=True)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    s.close()
    print(env, "%.1f Msamples/s  (%.1f %% of the samples drawn)  kernel %.1f ms  %.1f walks per wave step  %d shading passes  rays/sample %.2f  nodes/ray %.1f" % (
        size * size * spp / best["kernel_ms"] / 1e3, 100.0 * best["samples"] / (size * size * spp), best["kernel_ms"], (best["node_visits"] + best["leaf_tests"]) / max(best["wave_steps"], 1),
        best["shading_passes"], best["rays_traced"] / max(best["samples"], 1), best["node_visits"] / max(best["rays_traced"], 1)), flush=True)
