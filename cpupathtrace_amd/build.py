"""Builds cpupathtrace_amd/libpathtrace_hip.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpathtrace_hip.so")
SOURCES = ["pt_api.cpp", "pt_bvh.cpp", "pt_build.hip", "pt_post.hip", "pt_path.hip"]
HEADERS = ["pt_build.h", "pt_post.h", "pt_types.h", "pt_kernels.h", "pt_device.h", "pt_shading.h", "pt_libm.h", "pt_bvh.h", os.path.join("..", "..", "include", "pt_hip.h")]

# -ffp-contract=off + correctly rounded divide/sqrt: every fp32/fp64 operation is the IEEE operation the reference's
# x86-64 build performs (parity is bit-level); -O3 otherwise.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-pthread"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    """Compile the library unless it is up to date.  Safe when several processes (the ranks of a multi-GPU run) call it at once:
    one of them compiles, into a temporary file that is renamed into place, the others wait for the lock and find the result."""
    if not force and up_to_date():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and up_to_date():
                return LIB
            tmp = "%s.%d.tmp" % (LIB, os.getpid())
            cmd = [hipcc()] + FLAGS + ["-x", "hip"] + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
