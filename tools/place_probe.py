"""Prototype of the cost-aware first round: measure what every pixel stream costs in a short pilot render, then give the expensive
streams wavefronts with few slots (a lone walk steps in ~0.5 us, one of 40 in ~1.4 us) and pack the cheap ones densely.

    python tools/place_probe.py [--size 1024] [--spp 256] [--n 8] [--rank 0] [--pilot 16]

Uses the diagnostics pt_debug_collect_costs / pt_debug_stream_costs / pt_debug_set_place of the HIP library.  Every variant's frame is
compared with the default placement's: the same pixels bit for bit, whatever the placement."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes, sharding

NONE = 0xFFFFFFFF


def plan(cost, waves, cap, a, b):
    """Streams sorted by falling cost fill one wavefront after another; a wavefront whose first (most expensive) stream costs c takes
    k streams with c * (a + b k) <= T.  The smallest T that needs no more than `waves` wavefronts, by bisection."""
    order = np.argsort(-cost.astype(np.int64), kind="stable")
    c = np.maximum(cost[order].astype(np.float64), 1.0)
    n = len(c)

    def fill(T):
        sizes, at = [], 0
        while at < n:
            k = int(max(1, min(cap, np.floor((T / c[at] - a) / b))))
            k = min(k, n - at)
            sizes.append(k)
            at += k
            if len(sizes) > waves:
                return None
        return sizes
    lo, hi = c[0] * (a + b), c[0] * (a + b * cap) * 4
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if fill(mid) is None:
            lo = mid
        else:
            hi = mid
    sizes = fill(hi)
    table = np.full((waves, cap), NONE, dtype=np.uint32)
    # wavefront j of the plan goes to wavefront (j * stride) % waves of the grid: the sparse wavefronts are spread over the CUs
    at = 0
    slots = np.zeros(waves, np.int64)
    perm = np.arange(waves)
    perm = (perm % 4) * (waves // 4) + perm // 4 if waves % 4 == 0 else perm  # neighbours of the plan -> different workgroups
    for j, k in enumerate(sizes):
        table[perm[j], :k] = order[at:at + k]
        at += k
    return table, sizes, hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--pilot", type=int, default=16)
    ap.add_argument("--mesh-n", type=int, default=1900)
    ap.add_argument("--caps", default="64")
    ap.add_argument("--ab", default="0.45:0.025,0.45:0.05,0.45:0.0125,0.2:0.025")
    args = ap.parse_args()
    lib = binding.load()
    sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(args.mesh_n, args.mesh_n, scenes.DRAGON_BOX_TRANSFORM))
    s = binding.Scene(sc)
    size = args.size
    opt = scenes.options(size, size, args.spp, args.spp)
    tiles = sharding.local_tiles(binding.job_tiles(size, size), args.rank, args.n)
    n = int((tiles["w"].astype(np.int64) * tiles["h"]).sum())
    print("# dragon %d^2 mesh, %dx%d, %d spp, rank %d of %d: %d tiles, %d streams" % (args.mesh_n, size, size, args.spp, args.rank, args.n, len(tiles), n), flush=True)

    def render(o, label):
        t0 = time.perf_counter()
        img, st = s.process_job(cam, o, tiles=tiles, want_stats=True)
        wall = (time.perf_counter() - t0) * 1e3
        print("%-58s kernel %8.2f ms  (%d waves x %d rows, %5.1f walks/step, %6.0f steps/wave, %.2f us/step)" % (
            label, st["kernel_ms"], st["wavefronts"], st["slot_rows"], (st["node_visits"] + st["leaf_tests"]) / max(st["wave_steps"], 1),
            st["wave_steps"] / st["wavefronts"], 1e3 * st["kernel_ms"] * st["wavefronts"] / max(st["wave_steps"], 1)), flush=True)
        return img, st

    def costs():
        out = np.zeros(n, np.uint32)
        binding._check(lib.pt_debug_stream_costs(s._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out

    render(scenes.options(size, size, 2, 2), "(warm-up)")
    base_img, base = render(opt, "default first round")
    binding._check(lib.pt_debug_collect_costs(s._h, 1))
    render(opt, "default first round, costs recorded")
    full_cost = costs()
    _, pilot_st = render(scenes.options(size, size, args.pilot, args.pilot), "pilot, %d spp" % args.pilot)
    pilot_cost = costs()
    binding._check(lib.pt_debug_collect_costs(s._h, 0))
    fc, pc = full_cost / args.spp, pilot_cost / args.pilot
    q = [0, 10, 50, 90, 99, 99.9, 100]
    print("cost per sample (wave steps while a ray of the stream walks), full render: " + ", ".join("%g%%: %.0f" % (p, v) for p, v in zip(q, np.percentile(fc, q))))
    print("cost per sample, pilot:                                                   " + ", ".join("%g%%: %.0f" % (p, v) for p, v in zip(q, np.percentile(pc, q))))
    print("correlation pilot / full: %.3f;  of the 5 %% most expensive streams by the full render the pilot ranks %.0f %% in its top 10 %%" % (
        np.corrcoef(fc, pc)[0, 1], 100.0 * np.isin(np.argsort(-fc)[:n // 20], np.argsort(-pc)[:n // 10]).mean()))
    waves = int(base["wavefronts"])
    max_waves = 4096
    for source, cost in (("pilot", pilot_cost), ("full", full_cost)):
        for cap in [int(v) for v in args.caps.split(",")]:
            for ab in args.ab.split(","):
                a, b = [float(v) for v in ab.split(":")]
                for W in sorted({waves, max_waves}):
                    if W * cap < n:
                        continue
                    table, sizes, T = plan(cost, W, cap, a, b)
                    binding._check(lib.pt_debug_set_place(s._h, C.c_uint32(W), C.c_uint32(cap), table.ctypes.data_as(C.c_void_p)))
                    img, st = render(opt, "%s costs, %d waves, cap %d, a %.2f b %.4f: %d used, sizes %d..%d" % (source, W, cap, a, b, len(sizes), min(sizes), max(sizes)))
                    if not np.array_equal(img.view(np.uint32), base_img.view(np.uint32)):
                        print("   !!! frame differs from the default placement's")
    s.close()


if __name__ == "__main__":
    main()
