"""Knob sweep on one scene: PT_STACK_LDS x PT_LDS_PAIRS x PT_REFILL_IDLE (env is read when the Scene is created)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
grid = eval(sys.argv[3]) if len(sys.argv) > 3 else {"PT_STACK_LDS": [8, 16, 24], "PT_LDS_PAIRS": [0, 255, 1023], "PT_REFILL_IDLE": [20]}
if mesh_n > 0:
    pos, nrm = scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM)
    sc, cam = scenes.dragon_box_scene(pos, nrm)
else:
    sc, cam = scenes.cornell_scene(1024, 1024) if mesh_n == 0 else scenes.box_scene()
opt = scenes.options(1024, 1024, spp, spp)
keys = list(grid)
for combo in itertools.product(*[grid[k] for k in keys]):
    for k, v in zip(keys, combo):
        os.environ[k] = str(v)
    s = binding.Scene(sc)
    s.process_job(cam, scenes.options(1024, 1024, 4, 4))
    img, st = s.process_job(cam, opt, want_stats=True)
    s.close()
    print(dict(zip(keys, combo)), "%.1f Msamples/s trace %.0f ms shade %.0f ms launches %d" % (
        1024 * 1024 * spp / st["total_ms"] / 1e3, st["trace_ms"], st["shade_ms"], st["iterations"]), flush=True)
