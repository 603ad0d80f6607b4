"""Where the cycles of a traversal step go (pt_steptime_kernel): cycles per step of a walk, and the part of them spent waiting for the
record, for rays of the benchmark scene -- one ray per wavefront, wavefronts alone on their SIMD or four to a SIMD.

    python tools/step_timing.py [mesh_n]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM))
s = binding.Scene(sc)
fn = binding.load().pt_debug_step_timing
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
rng = np.random.default_rng(5)


def rays_through(x0, x1, y0, y1, n):
    """rays from the benchmark camera's position through a window of the image plane (z = 1)"""
    d = np.stack([rng.uniform(x0, x1, n), rng.uniform(y0, y1, n), np.full(n, 1.0)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.ascontiguousarray(np.concatenate([np.tile(np.array([0.0, 0.0, -3.0]), (n, 1)), d], axis=1), dtype=np.float32)


SEGMENTS = ["between two steps", "classification", "waiting for the record", "slab tests, decision, stack", "address + request of the next record",
            "leaving the common step", "the rare step or the test for it", "(two stamps back to back)"]


def run(rays, lanes, flags, segments=False):
    n = len(rays)
    out = np.zeros(20 * n, np.uint32)
    assert fn(s._h, rays.ctypes.data, n, lanes, flags, out.ctypes.data) == 0, binding.load().pt_last_error()
    if segments:
        return out[4 * n:].view(np.uint64).reshape(n, 8).astype(np.float64)
    return out[:4 * n].reshape(n, 4).astype(np.float64)


for label, win in (("window on the mesh", (-0.05, 0.05, -0.21, -0.11)), ("whole frame", (-0.33, 0.33, -0.33, 0.33))):
    for n, lanes in ((256, 1), (4096, 1), (4096, 64), (262144, 64)):
        rays = rays_through(*win, n)
        for pf in (0,):  # (bit 0 of the flags was the children-line prefetch of an experiment: gone)
            run(rays, lanes, pf)
            clean = run(rays, lanes, pf)
            stamped = run(rays, lanes, pf | 2)
            waves = -(-n // lanes)
            # (with 64 rays per wavefront a lane's walk also waits for the other lanes' steps: per-wavefront figures = the longest walk)
            steps = clean[:, 0].reshape(waves, -1).max(axis=1).sum() if lanes > 1 else clean[:, 0].sum()
            total = clean[:, 2].reshape(waves, -1).max(axis=1).sum() if lanes > 1 else clean[:, 2].sum()
            wait = stamped[:, 1].reshape(waves, -1).max(axis=1).sum() if lanes > 1 else stamped[:, 1].sum()
            if lanes == 1 and pf == 0 and os.environ.get("PT_STEP_STAMPS"):
                seg = run(rays, lanes, pf, segments=True)
                st = clean[:, 0].sum()
                price = seg[:, 7].sum() / st
                print("    stamped build, cycles per step (a stamp's own round trip, %.0f, taken off each): " % price +
                      "; ".join("%s %.0f" % (SEGMENTS[k], seg[:, k].sum() / st - price) for k in range(7)), flush=True)
            print("%-18s %6d rays, %2d per wavefront, %5d wavefronts (%4.1f per SIMD): %5.1f steps per walk, %6.0f cycles per step; stamped run: %5.0f of them waiting for the record" % (
                label, n, lanes, waves, min(waves / 1024.0, 8.0), steps / waves, total / steps, wait / steps), flush=True)
s.close()
