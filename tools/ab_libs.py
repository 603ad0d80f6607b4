import os, sys, subprocess
# old library has the old pt_stats layout: time the call on the host instead
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from cpupathtrace_amd import binding, scenes
sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(1900, 1900, scenes.DRAGON_BOX_TRANSFORM))
s = binding.Scene(sc)
opt = scenes.options(1024, 1024, 128, 128)
s.process_job(cam, scenes.options(1024, 1024, 4, 4))
best = 1e9
for _ in range(3):
    t = time.time(); s.process_job(cam, opt); best = min(best, time.time() - t)
print("%s %.1f Msamples/s (host-timed, incl. 16 MB D2H)" % (os.environ.get("PT_LIB_OVERRIDE", "current"), 1024 * 1024 * 128 / best / 1e6))
'''
for lib in [None, "tools/bin/libpt_v1.so", None, "tools/bin/libpt_v1.so"]:
    env = dict(os.environ)
    if lib: env["PT_LIB_OVERRIDE"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", code], env=env)
