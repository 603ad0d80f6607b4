#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path (processJob) on N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dragon|dragons16|cornell|box] [--spp S] [--mesh-n M]

Metric (BASELINE.json / reference benchmark/main.cpp:20,30): Msamples/s = image_width * image_height * spp / wall seconds
of processJob with min_sample_count == max_sample_count == spp; one sample = one camera path with all its bounces and
shadow rays.  A "step" is one whole processJob of the workload's frame.  Default workload (N = 1): the benchmark's
DragonBox scene (benchmark/main.cpp:59-105) with the procedural 7.2 M-triangle stand-in for assets/xyzrgb_dragon.obj
(absent from the reference mount), 1024 x 1024.  For N > 1 the same view is rendered at sqrt(N) times the resolution per side
(1448, 2048, 2896 pixels for 2, 4, 8 GPUs): every GPU renders 1 Mpixel of interleaved 32x32 tiles with the same per-pixel work
("weak" scaling), the scene is replicated, and the tiles are gathered to rank 0 over RCCL at the end of every step.  Inputs are synthetic (procedural mesh), resident in HBM before the timed
region; scene build/upload is reported separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class heartbeat:
    """Prints a line to stderr every 45 s while a long host phase (mesh generation, BVH build) runs."""

    def __init__(self, what):
        import threading
        self.what, self.stop, self.t0 = what, threading.Event(), time.time()
        self.thread = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop.wait(45.0):
            log("... %s (%.0f s)" % (self.what, time.time() - self.t0))

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.thread.join()


def frame_for(n_gpus, base):
    """Weak scaling with the VIEW kept: N GPUs render the same square view at sqrt(N) times the resolution per side (rounded down to a
    multiple of 8), i.e. N x base^2 pixels of the same statistics -- 1024, 1448, 2048, 2896 for 1, 2, 4, 8 GPUs.  (Widening the frame
    instead would add background pixels, whose paths end at once: more pixels but less work per pixel.)"""
    side = int(base * (n_gpus ** 0.5) + 1e-9) // 8 * 8
    return side, side


def build_workload(name, width, height, mesh_n):
    from cpupathtrace_amd import scenes
    aspect = -float(np.float32(width) / np.float32(height))
    t0 = time.time()
    if name == "dragon":
        pos, nrm = scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM)
        sc, cam = scenes.dragon_box_scene(pos, nrm, aspect_ratio=aspect)
        label = "DragonBox (benchmark/main.cpp:59-105), procedural %d-triangle glass mesh standing in for xyzrgb_dragon.obj" % len(pos)
    elif name == "dragons16":
        pos, nrm = scenes.bumpy_sphere_mesh(mesh_n, mesh_n, scenes.DRAGON_BOX_TRANSFORM)
        sc, cam = scenes.dragon_grid_scene(pos, nrm, aspect_ratio=aspect, grid=4)
        label = "16 transformed copies of the %d-triangle stand-in mesh (%d triangles) in an enlarged box (BASELINE.json configs[4])" % (len(pos), 16 * len(pos))
    elif name == "cornell":
        sc, cam = scenes.cornell_scene(width, height)
        label = "Cornell box of demo/main.cpp without the dragon (27 objects), thin-lens camera"
    elif name == "box":
        sc, cam = scenes.box_scene(aspect_ratio=aspect)
        label = "Box (benchmark/main.cpp:34-57), 14 triangles"
    else:
        raise SystemExit("unknown workload " + name)
    return sc, cam, label, time.time() - t0


def ref_formula_bytes_per_sample(R, A, T, V):
    """bytes/sample = R*(A*32 + T*36 + 96) + V*(36 + 64) + 16 with per-sample R, V and per-ray A, T."""
    return R * (A * 32.0 + T * 36.0 + 96.0) + V * 100.0 + 16.0


def measured_traffic(workload_key, samples_per_launch):
    """HBM bytes per launch of the dominant kernel from the newest profiles/r*_traffic.json (written by tools/measure_traffic.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same workload, with the gfx950 correction of
    MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request, calibrated on tools/gather_bench.hip).  The profile stores
    bytes per SAMPLE (the profiler serialises kernels, so its frame runs as one stream group with three times larger launches);
    a launch of this run carries `samples_per_launch` samples.  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    for path in reversed(files):
        try:
            entry = json.load(open(path)).get(workload_key)
        except (OSError, ValueError):
            continue
        if entry and entry.get("bytes_per_sample"):
            return entry["bytes_per_sample"] * samples_per_launch, os.path.relpath(path, ROOT)
        if entry:
            return entry.get("bytes_per_launch"), os.path.relpath(path, ROOT)
    return None, None

def cpu_baseline(sc, cam, opt, seconds_target):
    """The reference itself (oracle/_ref, kind "reference") or, if that was not built, the C restatement (kind "port"),
    timed on this host's cores on a bounded random subset of the SAME frame's pixels (same scene, same spp)."""
    import oracle
    # the reference's default worker count is hardware_concurrency() - 1 (src/worker.cpp:366); a GPU box grants this job
    # a 16-core share of the host, so the count is taken from that share, not from the machine's 256 logical CPUs
    share = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2), int(os.environ.get("PT_CPU_SHARE", "16")))
    threads = max(share - 1, 1)
    try:
        chk, kind = oracle.Checker("ref", ndebug=True), "reference"
    except (FileNotFoundError, OSError):
        chk, kind = oracle.Checker("oracle"), "port"
    t0 = time.time()
    h = chk.scene_create(sc)
    build_s = time.time() - t0
    rng = np.random.default_rng(7)
    w, hgt, spp = opt["image_width"], opt["image_height"], opt["max_sample_count"]

    def run(n_pixels):
        xs = rng.integers(0, w, n_pixels).astype(np.int32)
        ys = rng.integers(0, hgt, n_pixels).astype(np.int32)
        states = rng.integers(1, 2**63, n_pixels).astype(np.uint64)
        t = time.time()
        h.render_streams(cam, opt, oracle.pixel_streams(xs, ys, states), n_threads=threads)
        return time.time() - t

    # grow the sample until it runs for about the requested time (the first, tiny run also pays thread start-up and page faults)
    n1 = max(4 * threads, 64)
    t_run = run(n1)
    for _ in range(4):
        if t_run >= 0.6 * seconds_target:
            break
        n1 = int(min(max(n1 * min(seconds_target / max(t_run, 1e-3), 20.0), n1 + 1), 4_000_000))
        t_run = run(n1)
    counters = None
    if kind == "port":
        counters = h.counters()
    h.close()
    return {"value": n1 * spp / t_run / 1e6, "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": "%d random pixels of the same %dx%d frame at %d spp (%.1f s); scene build %.1f s" % (n1, w, hgt, spp, t_run, build_s)}, counters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--size", type=int, default=1024, help="pixels per side at one GPU (N GPUs: sqrt(N) times as many)")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--mesh-n", type=int, default=1900, help="stand-in mesh resolution (nu = nv); 1900 -> 7.2 M triangles")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        log("note: --gpus %d but WORLD_SIZE %d; using %d" % (args.gpus, world, n_gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from cpupathtrace_amd import binding, scenes, sharding

    width, height = frame_for(n_gpus, args.size)
    with heartbeat("generating the scene"):
        sc, cam, label, gen_s = build_workload(args.workload, width, height, args.mesh_n)
    opt = scenes.options(width, height, args.spp, args.spp)
    t0 = time.time()
    with heartbeat("building the BVH and uploading the scene"):
        scene = binding.Scene(sc, device=local_rank)
    create_s = time.time() - t0
    info = scene.info()
    if rank == 0:
        log("scene: %s; %d objects, BVH depth %d, generated in %.1f s, built+uploaded in %.1f s" % (label, len(sc["obj_kind"]), info["depth"], gen_s, create_s))

    job = sharding.ShardedJob(scene, cam, opt, rank, n_gpus, device, base_seed=args.seed)

    def step(want_stats=False):
        return job.render(want_stats=want_stats)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = None
    for k in range(args.steps):
        stats = step(want_stats=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_samples = float(width) * height * args.spp * args.steps
        value = total_samples / elapsed / 1e6
        # dominant kernel: pt_trace_kernel.  Algorithmic bytes of what the kernel actually did in the last step on this
        # rank (its own counters), by the SURVEY.md 8(d) formula; duration = sum of its launches (HIP events on its stream).
        s = stats
        samples = max(s["samples"], 1)
        R = s["rays_traced"] / samples
        A = (2.0 * s["node_visits"] + s["rays_traced"]) / max(s["rays_traced"], 1)  # two slab tests per inner node + the root test
        T = s["leaf_tests"] / max(s["rays_traced"], 1)
        V = s["vertices"] / samples
        bytes_per_sample = ref_formula_bytes_per_sample(R, A, T, V)
        launches = max(s["iterations"], 1)
        # The streams are rendered as `groups` concurrent groups: `concurrent` launches of the kernel run side by side on average, each
        # on a share of the CUs.  One launch: algorithmic bytes per launch / its duration (HIP events on its stream; this is what
        # rocprofv3's average duration shows).  The chip: that times the launches running at once = all bytes / the time during which
        # at least one launch was running.
        busy_s = max(s.get("trace_busy_ms", s["trace_ms"]), 1e-6) / 1e3
        concurrent = (s["trace_ms"] / 1e3) / busy_s
        achieved = bytes_per_sample * samples / busy_s / 1e9
        traffic, traffic_source = measured_traffic("%s-%d" % (args.workload, args.mesh_n if args.workload.startswith("dragon") else 0), samples / launches)
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_source, "algorithmic_bytes_per_launch": bytes_per_sample * samples / launches,
                    "kernel": "pt_trace_kernel", "avg_launch_ms": s["trace_ms"] / launches, "launches_per_step": launches,
                    "concurrent_launches": concurrent, "stream_groups": s.get("groups", 1), "trace_busy_ms": busy_s * 1e3,
                    "achieved_per_launch": bytes_per_sample * samples / launches / max(s["trace_ms"] / launches / 1e3, 1e-9) / 1e9,
                    "algorithmic_bytes_per_sample": bytes_per_sample,
                    "per_sample": {"rays": R, "aabb_tests_per_ray": A, "leaf_tests_per_ray": T, "vertices": V},
                    "trace_ms": s["trace_ms"], "shade_ms": s["shade_ms"], "step_device_ms": s["total_ms"]}
        cpu = None
        if n_gpus == 1 and args.cpu_seconds > 0:
            cpu, counters = cpu_baseline(sc, cam, opt, args.cpu_seconds)
        out = {
            "metric": "Msamples/s (all bounces)", "value": value, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s, %dx%d, %d spp (min = max), per-pixel engines seeded from base seed %d" % (label, width, height, args.spp, args.seed),
                       "objects": int(len(sc["obj_kind"])), "bvh_depth": int(info["depth"]), "tiles_per_gpu": int(job.n_local_tiles),
                       "parallelism": "tiles interleaved over %d GPU(s), scene replicated, RCCL gather to rank 0" % n_gpus,
                       "scene_build_s": create_s, "scene_generate_s": gen_s},
            "roofline": roofline,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
