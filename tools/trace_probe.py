"""Times pt_intersect_batch (pt_closest_kernel: the traversal alone) on random rays; run under rocprofv3 --kernel-trace for kernel durations."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "box"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 20
if which == "box":
    sc, cam = scenes.box_scene()
else:
    m = int(which)
    pos, nrm = scenes.bumpy_sphere_mesh(m, m, scenes.DRAGON_BOX_TRANSFORM)
    sc, cam = scenes.dragon_box_scene(pos, nrm)
s = binding.Scene(sc)
rng = np.random.default_rng(0)
o = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d.astype(np.float32)], axis=1)
for i in range(3):
    t0 = time.time(); t, obj = s.get_intersection(rays); dt = time.time() - t0
    print("%s n=%d call %.1f ms hits %.3f" % (which, n, dt * 1e3, (t >= 0).mean()), flush=True)
