"""What the traversal alone delivers on the rays of a real frame, at 4..8 wavefronts per SIMD (the path kernel is held at 4 by the
shading code's registers).  Renders the benchmark scene with rings that never wrap (PT_RING_LOG_RAYS), then replays the logged rays
with pt_debug_replay_rays (same hand-out / burst / leaf-batching loop, no shading).
    python tools/replay_probe.py [mesh_n] [spp]
(Round 2 could first rewrite the pair records in treelets of k levels -- every layout within 1 % of breadth-first,
profiles/r02_layout_probe.txt; the records are one linked array since round 3 and that diagnostic is gone.)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PT_RING_LOG_RAYS"] = os.environ.get("PT_RING_LOG_RAYS", "65536")
from cpupathtrace_amd import binding, scenes

mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM))
s = binding.Scene(sc)
img, st = s.process_job(cam, scenes.options(1024, 1024, spp, spp), want_stats=True)
import hashlib
print("frame sha1", hashlib.sha1(img.tobytes()).hexdigest()[:16], flush=True)
img, st = s.process_job(cam, scenes.options(1024, 1024, spp, spp), want_stats=True)
rays = st["rays_traced"]
print("path kernel: %.1f Msamples/s, %.2f G rays/s (%d rays, %.1f node visits per ray), kernel %.1f ms" % (
    1024 * 1024 * spp / st["kernel_ms"] / 1e3, rays / st["kernel_ms"] / 1e6, rays, st["node_visits"] / rays, st["kernel_ms"]), flush=True)
lib = binding.load()
fn = lib.pt_debug_replay_rays
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
top = 0.0
for parts in (2, 4):
    for waves in (4, 5, 6, 7, 8):
        best = None
        for _ in range(3):
            out = (ctypes.c_ulonglong * 8)()
            ms, blocks = ctypes.c_float(), ctypes.c_int()
            rc = fn(s._h, waves, parts, out, ctypes.byref(ms), ctypes.byref(blocks))
            assert rc == 0, rc
            if best is None or ms.value < best[0]:
                best = (ms.value, list(out), blocks.value)
        ms_v, o, blocks_v = best
        print("replay, %d waves/SIMD asked (%d workgroups per CU resident), rings in %d parts: %.2f ms, %.2f G rays/s, %d rays, %.1f node visits per ray, %.1f walks per wave step, checksum %x" % (
            waves, blocks_v, parts, ms_v, o[0] / ms_v / 1e6, o[0], o[1] / max(o[0], 1), (o[1] + o[2]) / max(o[3], 1), o[4]), flush=True)
        top = max(top, o[0] / ms_v * 1e3)
s.close()
# PT_REPLAY_JSON=<file>: keep the best rate under the workload's key (bench.py's roofline.ceiling.replay reads the newest profiles/r*_replay.json)
if os.environ.get("PT_REPLAY_JSON"):
    import json
    path = os.environ["PT_REPLAY_JSON"]
    try:
        data = json.load(open(path))
    except (OSError, ValueError):
        data = {}
    data["dragon-%d-1024" % mesh_n] = {"rays_per_s": top, "path_kernel_rays_per_s": rays / st["kernel_ms"] * 1e3, "spp": spp,
                                        "note": "best of 4..8 wavefronts per SIMD and rings replayed in 2 or 4 parts (tools/replay_probe.py)"}
    json.dump(data, open(path, "w"), indent=1)
