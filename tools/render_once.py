"""One frame of a workload through processJob (the program the profiler runs).
    python tools/render_once.py <mesh_n | 0 = cornell | -1 = box> <spp> [size] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if mesh_n > 0:
    sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM))
else:
    sc, cam = scenes.cornell_scene(size, size) if mesh_n == 0 else scenes.box_scene()
s = binding.Scene(sc)
for _ in range(frames):
    img, st = s.process_job(cam, scenes.options(size, size, spp, spp), want_stats=True)
    print("%.1f Msamples/s, kernel %.1f ms" % (size * size * spp / st["kernel_ms"] / 1e3, st["kernel_ms"]), flush=True)
s.close()
