"""How much does ray ORDER matter to the traversal kernel?  Same rays, different queue order; run under rocprofv3 --kernel-trace --stats
and compare the pt_trace_kernel durations.  usage: coherence_probe.py <mesh_n> <variant>
variants: camera | camera_shuffled | random | random_sorted | bounce | bounce_shuffled | bounce_sorted"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes

m, variant = int(sys.argv[1]), sys.argv[2]
pos, nrm = scenes.bumpy_sphere_mesh(m, m, scenes.DRAGON_BOX_TRANSFORM)
sc, cam = scenes.dragon_box_scene(pos, nrm)
s = binding.Scene(sc)
rng = np.random.default_rng(0)
W = H = 1024


def camera_rays():
    # pinhole at z = -3 looking at the origin, pixels in 32 x 32 tiles (the stream order of pt_render_tiles)
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 32) * (W // 32) + (xs // 32)
    order = np.lexsort((xs.ravel() % 32, ys.ravel() % 32, tile.ravel()))
    px = (xs.ravel()[order] + 0.5) / W - 0.5
    py = (ys.ravel()[order] + 0.5) / H - 0.5
    d = np.stack([px, -py, np.ones_like(px)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.array([[0.0, 0.0, -3.0]]), (len(d), 1))
    return np.concatenate([o, d], axis=1).astype(np.float32)


def morton_key(o, d):
    q = np.clip(((o + 1.0) * 0.5 * 1024).astype(np.int64), 0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
    return (octant << 30) | code


cam_rays = camera_rays()
if variant.startswith("camera"):
    rays = cam_rays
elif variant.startswith("random"):
    n = W * H
    o = rng.uniform(-0.9, 0.9, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
else:
    # first-bounce rays: from the camera rays' hit points into a random direction of the hemisphere (what a diffuse wall emits), in
    # the order of the pixels that produced them
    t, obj = s.get_intersection(cam_rays)
    hit = t >= 0
    p = cam_rays[:, :3] + cam_rays[:, 3:] * np.where(hit, t, 1.0)[:, None]
    p[:, 2] += 1e-3                                     # (the reference's Lambertian bounce leaves along the stored normal: into the box)
    d = rng.normal(size=(len(p), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:, 2] = np.abs(d[:, 2])
    rays = np.concatenate([p, d], axis=1).astype(np.float32)[hit]
if variant.endswith("_shuffled"):
    rays = rays[rng.permutation(len(rays))]
if variant.endswith("_sorted"):
    rays = rays[np.argsort(morton_key(rays[:, :3].astype(np.float64), rays[:, 3:]), kind="stable")]
for i in range(4):
    t0 = time.time(); t, obj = s.get_intersection(rays); dt = time.time() - t0
print("%s: %d rays, last call %.1f ms, hits %.3f, checksum %.6f" % (variant, len(rays), dt * 1e3, (t >= 0).mean(), float(np.where(t >= 0, t, 0).sum())), flush=True)
