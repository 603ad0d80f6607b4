"""Adaptive sampling (min < max: the per-pixel estimator stops early, src/worker.cpp:239-259) against the fixed count, on Cornell and the
benchmark scene at 1024 x 1024: Msamples/s on the max-spp basis (how the reference's benchmark counts items, benchmark/main.cpp:30),
samples actually drawn, and how evenly the wavefronts finish (slots free up at different times under early-outs).

    python tools/adaptive_probe.py [mesh_n]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
size = 1024
for name in ("cornell", "dragon"):
    if name == "cornell":
        sc, cam = scenes.cornell_scene(size, size)
    else:
        sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM))
    s = binding.Scene(sc)
    s.process_job(cam, scenes.options(size, size, 4, 4))
    for mn, mx in ((64, 64), (16, 64), (16, 16), (64, 256), (256, 256)):
        best = None
        for _ in range(2):
            img, st = s.process_job(cam, scenes.options(size, size, mn, mx), want_stats=True)
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        st = best
        print("%-8s %3d..%3d spp: kernel %7.1f ms; %7.1f Msamples/s on the max-spp basis; %5.1f %% of the max samples drawn (%.1f Msamples/s really drawn); "
              "%.1f walks per wave step, %d wave steps, %d passes" % (name, mn, mx, st["kernel_ms"], size * size * mx / st["kernel_ms"] / 1e3, 100.0 * st["samples"] / (size * size * mx),
                                                                      st["samples"] / st["kernel_ms"] / 1e3, (st["node_visits"] + st["leaf_tests"]) / max(st["wave_steps"], 1),
                                                                      st["wave_steps"], st["shading_passes"]), flush=True)
    s.close()
