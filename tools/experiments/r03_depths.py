import sys
sys.path.insert(0, ".")
from cpupathtrace_amd import binding, scenes
for n in (9, 13, 40, 100, 200, 300, 600, 1000, 1900):
    sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(n, n, scenes.DRAGON_BOX_TRANSFORM))
    s = binding.Scene(sc)
    print(n, len(sc["obj_kind"]), s.info()["depth"], flush=True)
    s.close()
