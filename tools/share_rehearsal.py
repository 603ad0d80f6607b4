"""Strong-scaling rehearsal on ONE GPU: what each rank of an N-GPU run of a FIXED frame would take.

    python tools/share_rehearsal.py [--workload dragon] [--mesh-n 1900] [--size 1024] [--spp 1024] [--n 1,2,4,8] [--env "{...}" ...]

bench.py --gpus N gives rank R every N-th tile of processJob's tile list (cpupathtrace_amd/sharding.py `local_tiles`: the reference's tile
queue, src/worker.cpp:398-414, dealt round-robin along the diagonals of the tile grid).  The ranks share nothing while they render, so rank R's time on its own GPU is the time this
GPU needs for that tile set alone; the frame is finished when the slowest rank is (the gather is 16 MB / N per rank, tens of microseconds
over xGMI).  Predicted speed-up of N GPUs = T(all tiles) / max_R T(tiles[R::N]).  Times are wall seconds of the render call with the
output in device memory (what bench.py times) and the launch duration by HIP events beside them.

Every --env adds a variant (a dict of PT_* knobs, read when the scene is created) that is run through the same table.
"""
import argparse, ast, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--mesh-n", type=int, default=1900)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--n", default="1,2,4,8")
    ap.add_argument("--ranks", default="all", help="'all' or how many ranks of each N to run (evenly spaced)")
    ap.add_argument("--env", action="append", default=[], help="dict of PT_* settings for a variant")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--tile", type=int, default=0, help="deal tiles of this many pixels per side instead of processJob's own (0): finer tiles even out what the ranks get")
    args = ap.parse_args()
    import torch
    from cpupathtrace_amd import binding, scenes, sharding
    import bench

    sc, cam, label, gen_s = bench.build_workload(args.workload, args.size, args.size, args.mesh_n)
    opt = scenes.options(args.size, args.size, args.spp, args.spp)
    tiles = binding.job_tiles(args.size, args.size)
    if args.tile > 0:
        t = args.tile
        tiles = np.array([(x, y, min(t, args.size - x), min(t, args.size - y)) for y in range(0, args.size, t) for x in range(0, args.size, t)], dtype=binding.TILE_DTYPE)
    dev = torch.device("cuda", 0)
    image = torch.zeros((args.size, args.size, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    print("# %s, %dx%d, %d spp, %d tiles" % (label, args.size, args.size, args.spp, len(tiles)), flush=True)
    variants = [{}] + [ast.literal_eval(e) for e in args.env]
    touched = set()
    t_full = None  # (kept from the first variant that runs N = 1)
    for env in variants:
        for k in touched:
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(v)
            touched.add(k)
        scene = binding.Scene(sc, device=0)
        print("## knobs: %s" % (env or "defaults"), flush=True)
        print("%3s %5s %8s %10s %10s %9s %7s %5s %8s %9s %9s %8s" % ("N", "rank", "pixels", "wall ms", "kernel ms", "Msamp/s", "waves", "rows", "walks/st", "steps/wv", "passes/wv", "us/step"))
        for n in [int(v) for v in args.n.split(",")]:
            ranks = list(range(n))
            if args.ranks != "all" and int(args.ranks) < n:
                ranks = sorted(set(int(round(i * (n - 1) / max(int(args.ranks) - 1, 1))) for i in range(int(args.ranks))))
            worst = 0.0
            for r in ranks:
                mine = sharding.local_tiles(tiles, r, n)
                pixels = int((mine["w"].astype(np.int64) * mine["h"]).sum())
                scene.process_job_device(cam, scenes.options(args.size, args.size, 2, 2), image.data_ptr(), stream, tiles=mine)  # workspace for this job size
                torch.cuda.synchronize()
                best = None
                for _ in range(args.repeat):
                    t0 = time.perf_counter()
                    st = scene.process_job_device(cam, opt, image.data_ptr(), stream, tiles=mine, want_stats=True)
                    torch.cuda.synchronize()
                    wall = (time.perf_counter() - t0) * 1e3
                    if best is None or wall < best[0]:
                        best = (wall, st)
                wall, st = best
                worst = max(worst, wall)
                wv = max(st["wavefronts"], 1)
                print("%3d %5d %8d %10.1f %10.1f %9.1f %7d %5d %8.1f %9.0f %9.0f %8.2f" % (n, r, pixels, wall, st["kernel_ms"], pixels * args.spp / wall / 1e3, wv, st["slot_rows"],
                                                                                        (st["node_visits"] + st["leaf_tests"]) / max(st["wave_steps"], 1), st["wave_steps"] / wv,
                                                                                        st["shading_passes"] / wv, st["kernel_ms"] * 1e3 / max(st["wave_steps"] / wv, 1)), flush=True)
            if n == 1:
                t_full = worst
            if t_full is not None:
                print("  -> N = %d: slowest rank %.1f ms; frame rate %.1f Msamples/s; predicted speed-up over one GPU %.2fx (efficiency %.0f %%)" % (
                    n, worst, args.size * args.size * args.spp / worst / 1e3, t_full / worst, 100.0 * t_full / worst / n), flush=True)
        scene.close()


if __name__ == "__main__":
    main()
