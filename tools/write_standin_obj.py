"""Write the procedural stand-in for assets/xyzrgb_dragon.obj (absent from the reference mount) as an OBJ file.

    python tools/write_standin_obj.py <path> [nu = nv, default 1900 -> 7,216,200 triangles]

The mesh is cpupathtrace_amd.scenes.bumpy_sphere_vertices: the same vertices and faces bench.py's default workload uses; the reference's
benchmark applies its own transform and smooths the normals when it loads the file (benchmark/main.cpp:80-86)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cpupathtrace_amd import build_host, scenes


def write(path, n=1900):
    verts, faces = scenes.bumpy_sphere_vertices(n, n)
    verts = np.ascontiguousarray(verts, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    lib = C.CDLL(build_host.build())
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    rc = lib.pth_write_obj(C.c_char_p(path.encode()), C.c_void_p(verts.ctypes.data), C.c_uint64(len(verts)), C.c_void_p(faces.ctypes.data), C.c_uint64(len(faces)))
    if rc != 0:
        raise OSError("cannot write " + path)
    return len(faces)


if __name__ == "__main__":
    n = write(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1900)
    print("%s: %d triangles" % (sys.argv[1], n))
