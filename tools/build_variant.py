"""Build a variant of libpathtrace_hip.so with extra compiler flags into tools/bin/libpt_<name>.so (load it with PT_LIB_OVERRIDE).

    python tools/build_variant.py timing -DPT_PATH_TIMING
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpupathtrace_amd import build as b

name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
os.makedirs(out_dir, exist_ok=True)
out = os.path.join(out_dir, "libpt_%s.so" % name)
cmd = [b.hipcc()] + b.FLAGS + extra + ["-x", "hip"] + [os.path.join(b.CSRC, f) for f in b.SOURCES] + ["-o", out]
subprocess.run(cmd, check=True)
print(out)
