"""Does it pay to give every XCD (its own 4 MB L2) a compact part of the job?  First rounds built on the host and handed to the
library with pt_debug_set_place; workgroup b runs on XCD b % 8 (round-robin dispatch: observed, not promised).

    python tools/xcd_probe.py [--size 1024] [--spp 256] [--n 8]
"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes, sharding

NONE = 0xFFFFFFFF


def dealt(n, waves, slots, groups, group_of_wave, index_in_group, piece=1):
    """The job's streams cut into `groups` contiguous ranges; inside a range, piece q of local wavefront v starts on chunk q * waves_in_group + v."""
    per = -(-n // groups)
    wg = waves // groups
    table = np.full((waves, slots), NONE, np.uint32)
    w = np.arange(waves)
    x, v = group_of_wave(w), index_in_group(w)
    for q in range(slots // piece):
        for k in range(piece):
            local = (q * wg + v) * piece + k
            stream = x * per + local
            ok = (local < per) & (stream < n)
            table[ok, q * piece + k] = stream[ok]
    assert (np.sort(table[table != NONE]) == np.arange(n)).all(), "every stream exactly once"
    return table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--mesh-n", type=int, default=1900)
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--interleave", action="store_true")
    args = ap.parse_args()
    lib = binding.load()
    import bench
    sc, cam, label, _ = bench.build_workload(args.workload, args.size, args.size, args.mesh_n)
    s = binding.Scene(sc)
    size = args.size
    opt = scenes.options(size, size, args.spp, args.spp)
    tiles = sharding.local_tiles(binding.job_tiles(size, size), 0, args.n)
    n = int((tiles["w"].astype(np.int64) * tiles["h"]).sum())
    print("# %s, %dx%d, %d spp, rank 0 of %d: %d tiles, %d streams" % (label, size, size, args.spp, args.n, len(tiles), n), flush=True)

    def render(o, label):
        img, st = s.process_job(cam, o, tiles=tiles, want_stats=True)
        img = img.copy()
        print("%-74s kernel %8.2f ms  (%d waves x %d rows, %5.1f walks/step, %6.0f steps/wave, %.2f us/step)" % (
            label, st["kernel_ms"], st["wavefronts"], st["slot_rows"], (st["node_visits"] + st["leaf_tests"]) / max(st["wave_steps"], 1),
            st["wave_steps"] / st["wavefronts"], 1e3 * st["kernel_ms"] * st["wavefronts"] / max(st["wave_steps"], 1)), flush=True)
        return img, st

    render(scenes.options(size, size, 2, 2), "(warm-up)")
    base_img, base = render(opt, "library's own first round")
    waves, slots = int(base["wavefronts"]), -(-n // int(base["wavefronts"]))
    slots = -(-slots // 8) * 8
    wg = lambda w: w // 4
    variants = [
        ("one range, dealt stream by stream (= the library's rule for small jobs)", 1, lambda w: 0 * w, lambda w: w, 1),
        ("8 ranges, one per XCD (workgroup b -> XCD b % 8), dealt stream by stream", 8, lambda w: wg(w) % 8, lambda w: (wg(w) // 8) * 4 + w % 4, 1),
        ("8 ranges, one per 1/8 of the grid in launch order (control: all XCDs mixed)", 8, lambda w: w // (waves // 8), lambda w: w % (waves // 8), 1),
        ("64 ranges: XCD b % 8, then 8 runs of workgroups inside it", 64, lambda w: (wg(w) % 8) * 8 + (wg(w) // 8) * 8 // (waves // 32), lambda w: ((wg(w) // 8) % (waves // 256)) * 4 + w % 4, 1),
        ("8 ranges per XCD, pieces of 8 neighbouring streams", 8, lambda w: wg(w) % 8, lambda w: (wg(w) // 8) * 4 + w % 4, 8),
        ("one range, pieces of 8", 1, lambda w: 0 * w, lambda w: w, 8),
    ]
    if args.interleave:
        # the tile list reordered so that the tiles of one XCD's range lie all over the frame: tile (tx, ty) -> range (tx + ty) % 8
        ts = int(tiles["w"][0])
        key = ((tiles["x"] // ts + tiles["y"] // ts) % 8).astype(np.int64)
        tiles = tiles[np.argsort(key, kind="stable")]
        print("# tile list reordered: range x holds the tiles with (tx + ty) % 8 == x", flush=True)
        base_img, base = render(opt, "library's own first round on the reordered list")
    for name, groups, gfn, ifn, piece in variants:
        if waves % (groups * 4) != 0:
            continue
        table = dealt(n, waves, slots, groups, gfn, ifn, piece)
        binding._check(lib.pt_debug_set_place(s._h, C.c_uint32(waves), C.c_uint32(slots), table.ctypes.data_as(C.c_void_p)))
        img, st = render(opt, name)
        if not np.array_equal(img.view(np.uint32), base_img.view(np.uint32)):
            print("   !!! frame differs from the default placement's")
    s.close()


if __name__ == "__main__":
    main()
