#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path (processJob) on N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dragon|dragons16|cornell|box] [--spp S] [--mesh-n M]
                    [--scaling weak|strong] [--size PIXELS]

Metric (BASELINE.json / reference benchmark/main.cpp:20,30): Msamples/s = image_width * image_height * spp / wall seconds
of processJob with min_sample_count == max_sample_count == spp; one sample = one camera path with all its bounces and
shadow rays.  A "step" is one whole processJob of the workload's frame.  Default workload (N = 1): the benchmark's
DragonBox scene (benchmark/main.cpp:59-105) with the procedural 7.2 M-triangle stand-in for assets/xyzrgb_dragon.obj
(absent from the reference mount), 1024 x 1024, 1024 spp.

N > 1.  One process per GPU over RCCL.  Started either by the launcher (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) or directly (`python bench.py --gpus N`):
then this process starts the N ranks itself as child processes BEFORE anything touches a GPU, hands them RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT, passes rank 0's JSON line through and exits non-zero if any rank fails.  The tiles of the
frame (reference src/worker.cpp:398-414) are interleaved over the ranks, the scene is replicated, and the finished tiles are
gathered to rank 0 over RCCL at the end of every step (the multi-GPU form of doWorkParallel, src/worker.cpp:364-387).
  --scaling weak   (default) the same view at sqrt(N) times the resolution per side (1448, 2048, 2896 pixels for 2, 4, 8 GPUs):
                   every GPU renders `size`^2 pixels of interleaved 32x32 tiles with the same per-pixel work;
  --scaling strong the frame stays `size` x `size` for every N -- BASELINE.json configs[3] is
                   `--workload dragon --size 2048 --spp 4096 --scaling strong --gpus 8`.
Inputs are synthetic (procedural mesh), resident in HBM before the timed region; scene build/upload is reported separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth of the 8 XCDs, same guide


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


def frame_for(n_gpus, base, scaling):
    """weak: N GPUs render the same square view at sqrt(N) times the resolution per side (rounded down to a multiple of 8), i.e.
    N x base^2 pixels of the same statistics -- 1024, 1448, 2048, 2896 for 1, 2, 4, 8 GPUs.  (Widening the frame instead would add
    background pixels, whose paths end at once: more pixels but less work per pixel.)  strong: base x base for every N."""
    if scaling == "strong":
        return base, base
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


def measured_traffic(workload_key):
    """HBM bytes per SAMPLE from the newest profiles/r*_traffic.json (written by tools/measure_traffic.py from separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes over the same workload, with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE
    counts 64 B per 128-B request, calibrated on tools/gather_bench.hip).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    for path in reversed(files):
        try:
            entry = json.load(open(path)).get(workload_key)
        except (OSError, ValueError):
            continue
        if entry and entry.get("bytes_per_sample"):
            return entry["bytes_per_sample"], os.path.relpath(path, ROOT)
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
    h.close()
    return {"value": n1 * spp / t_run / 1e6, "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": "%d random pixels of the same %dx%d frame at %d spp (%.1f s); scene build %.1f s" % (n1, w, hgt, spp, t_run, build_s)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--size", type=int, default=1024, help="pixels per side: at one GPU (weak: sqrt(N) times as many at N GPUs) or of the fixed frame (strong)")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--mesh-n", type=int, default=1900, help="stand-in mesh resolution (nu = nv); 1900 -> 7.2 M triangles")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--frame-out", default=None, help="rank 0 writes the last gathered frame here as .npy (tests)")
    return ap.parse_args(argv)


# ---- starting the ranks ---------------------------------------------------------------------------------------------------------------

def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children.  Nothing in this process has touched a GPU (no torch
    import, no HIP call), and no process that has is ever re-executed: the children are fresh interpreters."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    # rank 0's stdout (one JSON line at the very end) is collected by a thread, so that this loop keeps watching every rank: a rank
    # that dies would otherwise leave the others waiting in a collective
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    failed = None
    alive = set(range(n))
    while alive and failed is None:
        for r in list(alive):
            rc = procs[r].poll()
            if rc is not None:
                alive.discard(r)
                if rc != 0 and failed is None:
                    failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:
        for r in alive:  # only the processes started here, by handle
            procs[r].terminate()
        for r in alive:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        log("bench.py: rank %d exited with code %d; the run is void" % failed)
        return failed[1] if failed[1] > 0 else 1
    reader.join(timeout=10)
    sys.stdout.write(b"".join(lines).decode())
    sys.stdout.flush()
    return 0


# ---- one rank -------------------------------------------------------------------------------------------------------------------------

def standin_render_fn(width, seed):
    """Test-only renderer for the CPU rehearsal of the N > 1 path (PT_BENCH_STANDIN=1): every pixel gets a value that depends on
    (x, y, seed) only, so the gathered frame says which pixels arrived and from which tile set."""
    import torch

    def render(tiles, image, want_stats):
        for t in tiles:
            ys = torch.arange(int(t["y"]), int(t["y"]) + int(t["h"]), dtype=torch.float32)[:, None]
            xs = torch.arange(int(t["x"]), int(t["x"]) + int(t["w"]), dtype=torch.float32)[None, :]
            v = ys * float(width) + xs + float(seed)
            image[int(t["y"]):int(t["y"]) + int(t["h"]), int(t["x"]):int(t["x"]) + int(t["w"])] = torch.stack([v, v * 0.5, v * 0.25, torch.ones_like(v)], dim=-1)
        return None

    return render


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench.py: --gpus %d but WORLD_SIZE is %d; refusing to run a mislabelled job" % (args.gpus, world))
        return 2
    if os.environ.get("PT_BENCH_TEST_FAIL_RANK") == str(rank):
        return 7  # tests: a rank that dies must void the whole run
    standin = os.environ.get("PT_BENCH_STANDIN", "0") == "1"      # CPU rehearsal of the sharding with a stand-in renderer (tests)
    share_device = os.environ.get("PT_BENCH_SHARE_DEVICE", "0") == "1"  # rehearsal on a one-GPU box: all ranks on device 0, gloo gather

    # Libraries below (gloo, RCCL, the HIP runtime) sometimes write to stdout; the contract is ONE JSON line there.  Everything else
    # this process prints goes to stderr: fd 1 is pointed at fd 2 and the result line is written to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from cpupathtrace_amd import sharding

    if standin:
        backend, device, dev_index = "gloo", torch.device("cpu"), -1
    else:
        n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if n_dev < 1:
            log("bench.py: no GPU visible")
            return 3
        if share_device:
            backend, dev_index = "gloo", 0
        else:
            if local_rank >= n_dev:
                log("bench.py: rank %d has no GPU (%d visible): start at most one rank per GPU" % (rank, n_dev))
                return 3
            backend, dev_index = "nccl", local_rank
        device = torch.device("cuda", dev_index)
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    width, height = frame_for(world, args.size, args.scaling)
    base_seed = args.seed
    scene = None
    if standin:
        from cpupathtrace_amd import scenes
        opt = scenes.options(width, height, args.spp, args.spp)
        sc, cam, label, gen_s, create_s, info = {"obj_kind": []}, None, "stand-in renderer (test only)", 0.0, 0.0, {"depth": 0}
        job = sharding.ShardedJob(None, None, opt, rank, world, device, base_seed=base_seed, render_fn=standin_render_fn(width, base_seed))
    else:
        from cpupathtrace_amd import binding, scenes
        with heartbeat("generating the scene"):
            sc, cam, label, gen_s = build_workload(args.workload, width, height, args.mesh_n)
        opt = scenes.options(width, height, args.spp, args.spp)
        t0 = time.time()
        with heartbeat("building the BVH and uploading the scene"):
            scene = binding.Scene(sc, device=dev_index)
        create_s = time.time() - t0
        info = scene.info()
        if rank == 0:
            log("scene: %s; %d objects, BVH depth %d, generated in %.1f s, built+uploaded in %.1f s" % (label, len(sc["obj_kind"]), info["depth"], gen_s, create_s))
        job = sharding.ShardedJob(scene, cam, opt, rank, world, device, base_seed=base_seed, staged_gather=(backend == "gloo"))

    def sync():
        if world > 1:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    # A step is one launch of the path kernel per rank.  Asking for statistics does not change the launch: the kernel always counts its
    # work in per-wave registers; the library then reads two HIP events recorded around the launch and the counters (128 KB) back.
    for _ in range(args.warmup):
        job.render()
    sync()
    t0 = time.perf_counter()
    stats = None
    for _ in range(args.steps):
        stats = job.render(want_stats=not standin)
    sync()
    elapsed = time.perf_counter() - t0
    my_pixels = int(sum(int(t["w"]) * int(t["h"]) for t in job.mine))
    mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "tiles": int(job.n_local_tiles), "pixels": my_pixels, "elapsed_s": elapsed,
            "device_name": torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu",
            "device_uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", "")) if device.type == "cuda" else "",
            "samples": int(stats["samples"]) if stats else my_pixels * args.spp,
            "tile_hash": int(np.bitwise_xor.reduce((job.mine["x"].astype(np.int64) * 65537 + job.mine["y"].astype(np.int64)) * 2654435761 % (1 << 61))) if len(job.mine) else 0}
    per_rank = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
        elapsed = max(r["elapsed_s"] for r in per_rank)  # MAX over ranks

    if rank == 0:
        if args.frame_out:
            np.save(args.frame_out, job.image.detach().cpu().numpy())
        total_samples = float(width) * height * args.spp * args.steps
        value = total_samples / elapsed / 1e6
        out = {
            "metric": "Msamples/s (all bounces)" if not standin else "INVALID (stand-in renderer, test only)", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not standin else "stand-in renderer (test only)",
            "config": {"workload": "%s, %dx%d, %d spp (min = max), per-pixel engines seeded from base seed %d" % (label, width, height, args.spp, args.seed),
                       "objects": int(len(sc["obj_kind"])), "bvh_depth": int(info["depth"]), "tiles_per_gpu": int(job.n_local_tiles),
                       "tiles_total": int(len(job.tiles)),
                       "parallelism": "tiles interleaved over %d GPU(s), scene replicated, %s gather to rank 0" % (world, "RCCL" if backend == "nccl" else backend),
                       "scene_build_s": create_s, "scene_generate_s": gen_s},
            "distributed": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else "none"),
                            "gather_bytes_per_step": int(job.gather_bytes), "ranks": per_rank},
        }
        if stats is not None:
            out["roofline"] = roofline(stats, args, value, world)
        if world == 1 and args.cpu_seconds > 0 and not standin:
            out["cpu_baseline"] = cpu_baseline(sc, cam, opt, args.cpu_seconds)
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    if scene is not None:
        scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def roofline(s, args, msamples_per_s, world):
    """Dominant kernel = pt_path_kernel, ONE launch per step.  `achieved` / `frac` are SURVEY.md 8(d)'s figure: ALGORITHMIC bytes per launch
    (the formula, with the kernel's own work counters of that launch) divided by the launch's duration (HIP events on the library's
    stream around the launch, last timed step), against the HBM peak -- a work rate, NOT a bandwidth utilisation.  What the memory
    system actually moves is reported next to it: `traffic` (PMC bytes per launch, from profiles/), `hbm_frac` (PMC bytes per second /
    8 TB/s) and `l2_frac` (algorithmic bytes per second / the L2's 34.5 TB/s: every node record that is not an L1 hit comes from
    there).  `bound` says what the counters say (profiles/): dependent record fetches in SIMT lockstep, not bandwidth."""
    samples = max(s["samples"], 1)
    R = s["rays_traced"] / samples
    A = (2.0 * s["node_visits"] + s["rays_traced"]) / max(s["rays_traced"], 1)  # two slab tests per inner node + the root test
    T = s["leaf_tests"] / max(s["rays_traced"], 1)
    V = s["vertices"] / samples
    bytes_per_sample = ref_formula_bytes_per_sample(R, A, T, V)
    launches = max(s["launches"], 1)
    kernel_s = max(s["kernel_ms"], 1e-6) / 1e3
    achieved = bytes_per_sample * samples / kernel_s / 1e9
    per_sample, source = measured_traffic("%s-%d" % (args.workload, args.mesh_n if args.workload.startswith("dragon") else 0))
    return {"bound": "latency", "bound_note": "dependent record fetches in SIMT lockstep at 4 waves per SIMD; HBM and L2 bandwidth are far from saturated (hbm_frac, l2_frac)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": per_sample * samples / launches if per_sample else None, "traffic_source": source,
            "kernel": "pt_path_kernel", "launches_per_step": launches, "avg_launch_ms": s["kernel_ms"] / launches,
            "algorithmic_bytes_per_launch": bytes_per_sample * samples / launches, "algorithmic_bytes_per_sample": bytes_per_sample,
            "hbm_frac": (per_sample * samples / kernel_s / 1e9 / HBM_PEAK_GBS) if per_sample else None,
            "l2_frac": achieved / L2_PEAK_GBS,
            "per_sample": {"rays": R, "aabb_tests_per_ray": A, "leaf_tests_per_ray": T, "vertices": V},
            "wavefronts": s["wavefronts"], "slot_rows": s["slot_rows"], "wave_steps": s["wave_steps"], "shading_passes": s["shading_passes"],
            "walks_per_wave_step": (s["node_visits"] + s["leaf_tests"]) / max(s["wave_steps"], 1)}


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
