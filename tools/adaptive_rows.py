"""Adaptive sampling with fewer resident streams: with PT_ROWS < 4 a 1024 x 1024 job no longer fits the slots in one round, so the streams
beyond the first round are pulled from the counter as slots free up -- dynamic balance for a job whose streams stop at different samples.

    python tools/adaptive_rows.py [mesh_n]   (mesh_n 0 = Cornell)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
size = 1024
if mesh_n == 0:
    sc, cam = scenes.cornell_scene(size, size)
else:
    sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM))
for rows in (4, 3, 2, 1):
    os.environ["PT_ROWS"] = str(rows)
    s = binding.Scene(sc)
    s.process_job(cam, scenes.options(size, size, 4, 4))
    for mn, mx in ((64, 64), (16, 64), (64, 256)):
        best = None
        for _ in range(2):
            img, st = s.process_job(cam, scenes.options(size, size, mn, mx), want_stats=True)
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        st = best
        print("rows %d  %3d..%3d spp: kernel %7.1f ms; %7.1f Msamples/s on the max-spp basis; %5.1f %% drawn (%.1f Msamples/s really drawn); %.1f walks per wave step, %d wave steps, %d passes" % (
            rows, mn, mx, st["kernel_ms"], size * size * mx / st["kernel_ms"] / 1e3, 100.0 * st["samples"] / (size * size * mx), st["samples"] / st["kernel_ms"] / 1e3,
            (st["node_visits"] + st["leaf_tests"]) / max(st["wave_steps"], 1), st["wave_steps"], st["shading_passes"]), flush=True)
    s.close()
