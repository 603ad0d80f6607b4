"""Builds cpupathtrace_amd/libPathTrace.so: the reference-compatible C++ API (include/PathTrace) over libpathtrace_hip.so."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "src", "host")
LIB = os.path.join(HERE, "libPathTrace.so")
CXXFLAGS = ["-std=c++20", "-O2", "-fPIC", "-ffp-contract=off", "-Wall", "-I" + os.path.join(ROOT, "include")]
LINK = ["-L" + HERE, "-lpathtrace_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN"]


def sources():
    return sorted(glob.glob(os.path.join(SRC, "*.cpp")))


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True) + [os.path.join(HERE, "libpathtrace_hip.so")]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False):
    from . import build as hip_build
    hip_build.build()
    if not force and up_to_date():
        return LIB
    subprocess.run(["g++"] + CXXFLAGS + ["-shared", "-o", LIB] + sources() + LINK, check=True)
    return LIB


def compile_program(source_files, out_path, extra_includes=(), extra_flags=()):
    """Compile a C++ program against the PathTrace API (used by the tests for their C++ test programs)."""
    build()
    cmd = ["g++"] + CXXFLAGS + ["-I" + i for i in extra_includes] + list(extra_flags) + ["-o", out_path] + list(source_files) + \
        ["-L" + HERE, "-lPathTrace", "-lpathtrace_hip", "-lz", "-pthread", "-Wl,-rpath," + HERE]
    subprocess.run(cmd, check=True)
    return out_path


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from cpupathtrace_amd import build_host
    build_host.build(force=True)
