"""Balance the wavefronts of a single-round job with measured stream costs: a pilot render records what every stream costs (wave steps
while one of its rays walks), and the streams -- or small groups of neighbouring ones -- are dealt to the wavefronts in a snake over the
sorted costs, so that every wavefront gets the same load.  Prototype on pt_debug_set_place; frames are compared bit for bit.

    python tools/balance_probe.py [--size 1024] [--spp 256] [--n 8] [--pilot 16]
"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes, sharding

NONE = 0xFFFFFFFF


def snake(cost, n, waves, slots, unit, bin_waves):
    """Units of `unit` neighbouring streams, bins of `bin_waves` wavefronts (unit % bin_waves == 0: a unit's streams are dealt round-robin
    to the bin's wavefronts, unit / bin_waves slots in each)."""
    n_units = -(-n // unit)
    padded = np.zeros(n_units * unit, np.float64)
    padded[:n] = cost
    ucost = padded.reshape(n_units, unit).sum(axis=1)
    order = np.argsort(-ucost, kind="stable")
    bins = waves // bin_waves
    per_wave = unit // bin_waves
    table = np.full((waves, slots), NONE, np.uint32)
    r = np.arange(n_units)
    rnd, pos = r // bins, r % bins
    b = np.where(rnd % 2 == 0, pos, bins - 1 - pos)
    assert (rnd.max() + 1) * per_wave <= slots, "placement needs %d slots per wavefront" % ((rnd.max() + 1) * per_wave)
    for j in range(unit):
        stream = order * unit + j
        ok = stream < n
        table[(b * bin_waves + j % bin_waves)[ok], (rnd * per_wave + j // bin_waves)[ok]] = stream[ok]
    assert (np.sort(table[table != NONE]) == np.arange(n)).all()
    load = np.zeros(waves)
    live = table != NONE
    np.add.at(load, np.nonzero(live)[0], cost[table[live]])
    return table, load.max() / load.mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--pilot", type=int, default=16)
    ap.add_argument("--mesh-n", type=int, default=1900)
    ap.add_argument("--workload", default="dragon")
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
        print("%-70s kernel %8.2f ms  (%d waves x %d rows, %5.1f walks/step, %6.0f steps/wave, %.2f us/step)" % (
            label, st["kernel_ms"], st["wavefronts"], st["slot_rows"], (st["node_visits"] + st["leaf_tests"]) / max(st["wave_steps"], 1),
            st["wave_steps"] / st["wavefronts"], 1e3 * st["kernel_ms"] * st["wavefronts"] / max(st["wave_steps"], 1)), flush=True)
        return img.copy(), st

    def costs():
        out = np.zeros(n, np.uint32)
        binding._check(lib.pt_debug_stream_costs(s._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n)))
        return out.astype(np.float64)

    render(scenes.options(size, size, 2, 2), "(warm-up)")
    base_img, base = render(opt, "library's own first round")
    binding._check(lib.pt_debug_collect_costs(s._h, 1))
    render(opt, "the same, costs recorded")
    full_cost = costs()
    render(scenes.options(size, size, args.pilot, args.pilot), "pilot, %d spp" % args.pilot)
    pilot_cost = costs()
    binding._check(lib.pt_debug_collect_costs(s._h, 0))
    waves = int(base["wavefronts"])
    for source, cost in (("pilot", pilot_cost), ("full", full_cost)):
        for unit, bin_waves, what in ((1, 1, "single streams -> wavefronts"), (4, 4, "4 neighbours -> the 4 wavefronts of a workgroup"), (8, 1, "8 neighbours -> a wavefront"),
                                      (16, 4, "16 neighbours -> a workgroup, 4 per wavefront"), (32, 4, "32 neighbours -> a workgroup, 8 per wavefront")):
            slots = -(-(-(-n // unit)) // (waves // bin_waves)) * (unit // bin_waves)
            if slots > 256:
                continue
            table, imbalance = snake(cost, n, waves, slots, unit, bin_waves)
            binding._check(lib.pt_debug_set_place(s._h, C.c_uint32(waves), C.c_uint32(slots), table.ctypes.data_as(C.c_void_p)))
            img, st = render(opt, "%s costs, %s (planned max/mean %.3f)" % (source, what, imbalance))
            if not np.array_equal(img.view(np.uint32), base_img.view(np.uint32)):
                print("   !!! frame differs from the default placement's")
    s.close()


if __name__ == "__main__":
    main()
