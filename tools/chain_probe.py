"""The serial chain of a pixel: render ONE tile (1024 streams, one per wavefront: nothing queues, nothing shares a lane) and report the
time per sample and per traversal step of its slowest stream -- the floor under any strong-scaling share that contains that pixel.

    python tools/chain_probe.py [--spp 256] [--tiles 496,528,0] [--env "{...}"]
"""
import argparse, ast, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--mesh-n", type=int, default=1900)
ap.add_argument("--tiles", default="496,400,0")
ap.add_argument("--env", action="append", default=[])
args = ap.parse_args()
from cpupathtrace_amd import binding, scenes
import bench
sc, cam, label, _ = bench.build_workload("dragon", args.size, args.size, args.mesh_n)
opt = scenes.options(args.size, args.size, args.spp, args.spp)
tiles = binding.job_tiles(args.size, args.size)
touched = set()
for env in [{}] + [ast.literal_eval(e) for e in args.env]:
    for k in touched:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
        touched.add(k)
    scene = binding.Scene(sc, device=0)
    print("## knobs: %s" % (env or "defaults"), flush=True)
    for t in [int(v) for v in args.tiles.split(",")]:
        mine = tiles[t:t + 1]
        scene.process_job(cam, scenes.options(args.size, args.size, 2, 2), tiles=mine)
        img, st = scene.process_job(cam, opt, tiles=mine, want_stats=True)
        wv = st["wavefronts"]
        print("tile %4d (x %d, y %d): kernel %.1f ms = %.1f us per sample of the slowest stream; %d wavefronts, mean %.0f steps and %.0f passes per wavefront, "
              "%.2f rays/sample, %.1f nodes/ray, %.2f vertices/sample" % (t, mine["x"][0], mine["y"][0], st["kernel_ms"], st["kernel_ms"] * 1e3 / args.spp, wv, st["wave_steps"] / wv,
                                                                          st["shading_passes"] / wv, st["rays_traced"] / st["samples"], st["node_visits"] / st["rays_traced"],
                                                                          st["vertices"] / st["samples"]), flush=True)
    scene.close()
