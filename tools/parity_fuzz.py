"""Differential test: random scenes (triangles and spheres at inexact coordinates, all BSDF kinds, emitters of both kinds, point lights,
all aperture kinds, with and without a surrounding box) rendered by the HIP path and by the CPU oracle, pixel for pixel and engine
state for engine state.       python tools/parity_fuzz.py [first_seed] [n_scenes] [width] [height] [spp_min] [spp_max] [tri_scale]
(PT_LDS_SMALL=0 in the environment keeps small trees in HBM: the other variant of the kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from cpupathtrace_amd import binding, scenes


def random_scene(seed, tri_scale=1):
    rng = np.random.default_rng(seed)
    sb = scenes.SceneBuilder()
    u = lambda lo, hi, n=None: rng.uniform(lo, hi, n)
    materials = [scenes.NO_MATERIAL]
    for _ in range(int(rng.integers(1, 30))):
        kind = int(rng.integers(0, 3))
        emission = (0, 0, 0, 0)
        if kind == 0 and rng.random() < 0.35:
            emission = tuple(float(v) for v in u(0.0, 4.0, 3)) + (float(u(0.2, 2.0)),)
        materials.append(sb.material(tuple(float(v) for v in u(0.05, 1.0, 3)) + (1.0,), float(u(1.0, 2.2)), emission, bsdf=kind,
                                     one_way=bool(kind == 2 and rng.random() < 0.5), specular=tuple(float(v) for v in u(0.2, 1.0, 3)) + (1.0,)))
    pick = lambda: materials[int(rng.integers(0, len(materials)))]
    if rng.random() < 0.7:
        sb.triangles(scenes.make_box(tuple(float(v) for v in u(-1.6, -0.9, 3)), tuple(float(v) for v in u(0.9, 1.6, 3))), pick())
    for _ in range(int(rng.integers(0, 5))):  # flat axis-aligned quads (lights or not)
        axis = int(rng.integers(0, 3))
        lo, hi = u(-0.8, 0.0, 3), u(0.1, 0.8, 3)
        lo[axis] = hi[axis] = u(-0.85, 0.85)
        try:
            quad = scenes.make_plane(tuple(float(v) for v in lo), tuple(float(v) for v in hi))
        except Exception:
            continue
        if len(np.asarray(quad).reshape(-1)) > 0:
            sb.triangles(quad, pick(), cull=bool(rng.random() < 0.3))
    n_tri = int(rng.integers(0, 60 * tri_scale))
    if n_tri:
        base = u(-0.9, 0.9, (n_tri, 1, 3))
        tri = (base + u(-0.35, 0.35, (n_tri, 3, 3))).astype(np.float32)
        for k in range(0, n_tri, 7):
            normals = None
            if rng.random() < 0.4:
                normals = rng.normal(size=(len(tri[k:k + 7]), 3, 3)).astype(np.float32)
                normals /= np.linalg.norm(normals, axis=2, keepdims=True)
            sb.triangles(tri[k:k + 7], pick(), cull=bool(rng.random() < 0.3), normals=normals)
    for _ in range(int(rng.integers(0, 5))):
        sb.sphere(tuple(float(v) for v in u(-0.8, 0.8, 3)), float(u(0.05, 0.4)), pick())
    for _ in range(int(rng.integers(0, 3))):
        sb.point_light(tuple(float(v) for v in u(-0.8, 0.8, 3)), tuple(float(v) for v in u(0.2, 2.0, 3)) + (1.0,))
    kind = int(rng.integers(0, 3))
    cam = scenes.camera(tuple(float(v) for v in (u(-0.5, 0.5), u(-0.5, 0.5), u(-3.2, -2.2))), tuple(float(v) for v in u(-0.3, 0.3, 3)), (0, 1, 0), float(u(0.8, 1.6)), float(u(0.7, 1.3)),
                        float(rng.choice([-1.0, 1.0, 1.5])), aperture_width=float(u(0.01, 0.1)) if kind else 0.0, aperture_height=float(u(0.01, 0.1)) if kind else 0.0,
                        aperture_kind=kind, hex_ratio=float(u(0.2, 0.9)), focal_plane_dist=float(u(2.0, 4.0)) if rng.random() < 0.7 else 0.0)
    return sb.build(), cam


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    spp = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    spp_max = int(sys.argv[6]) if len(sys.argv) > 6 else spp
    tri_scale = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    ol = oracle.Checker("oracle")
    ys, xs = np.mgrid[0:h, 0:w]
    xs, ys = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32)
    failures = 0
    for seed in range(first, first + n):
        desc, cam = random_scene(seed, tri_scale)
        opt = scenes.options(w, h, spp, spp_max, float(np.random.default_rng(seed + 7).choice([1e-3, 1e-4, 1e-2])))
        states = np.array([binding.seed_to_state(binding.pixel_seed(1000 + seed, int(x), int(y))) for x, y in zip(xs, ys)], np.uint64)
        try:
            s = binding.Scene(desc)
        except binding.PtError as e:
            print("seed %d: scene refused (%s)" % (seed, e), flush=True)
            continue
        img, after = s.process_item(cam, opt, binding.pixel_streams(xs, ys, states))
        # the tree (both builders for scenes the device builder takes) and closest hits on random rays
        ho = ol.scene_create(desc)
        rr = np.random.default_rng(seed + 99)
        d3 = rr.normal(size=(4000, 3)); d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
        rays = np.concatenate([rr.uniform(-1.2, 1.2, (4000, 3)), d3], axis=1).astype(np.float32)
        t_g, o_g = s.get_intersection(rays)
        t_o, o_o = ho.intersect(rays)
        hit = t_o >= 0
        tree_g, box_g = s.bvh_dump()
        tree_o, box_o = ol.bvh_dump(desc)
        s.close()
        if not (np.array_equal(tree_g, tree_o) and np.array_equal(box_g.view(np.uint32), box_o.view(np.uint32))):
            failures += 1
            print("seed %d: BVH differs from the oracle's" % seed, flush=True)
        if not (np.array_equal(t_g.view(np.uint32)[hit], t_o.view(np.uint32)[hit]) and np.array_equal(o_g[hit], o_o[hit]) and (t_g[~hit] < 0).all()):
            failures += 1
            print("seed %d: closest hits differ for %d of %d rays" % (seed, int(((t_g.view(np.uint32) != t_o.view(np.uint32)) & hit).sum() + ((t_g >= 0) & ~hit).sum()), len(rays)), flush=True)
        want, want_after = ho.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=8)
        g, wv = np.ascontiguousarray(img).reshape(-1, 4), want.reshape(-1, 4)
        both_nan = np.isnan(g) & np.isnan(wv)
        bad = ((g.view(np.uint32) != wv.view(np.uint32)) & ~both_nan).any(axis=1)
        bad_state = after != want_after
        if bad.any() or bad_state.any():
            failures += 1
            k = int(np.argwhere(bad | bad_state)[0][0])
            print("seed %d: %d of %d pixels differ, %d engine states; first pixel %d got %s want %s (objects %d, emitters?)" % (seed, bad.sum(), len(bad), bad_state.sum(), k, g[k], wv[k], len(desc["obj_kind"])), flush=True)
        elif seed % 10 == 0:
            print("seed %d ok (%d objects)" % (seed, len(desc["obj_kind"])), flush=True)
    print("%d of %d scenes differ" % (failures, n), flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
