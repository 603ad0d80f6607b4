#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path (processJob) on N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dragon|dragons16|cornell|box] [--spp S] [--mesh-n M]
                    [--scaling weak|strong] [--size PIXELS]

Metric (BASELINE.json / reference benchmark/main.cpp:20,30): Msamples/s = image_width * image_height * spp / wall seconds
of processJob with min_sample_count == max_sample_count == spp; one sample = one camera path with all its bounces and
shadow rays.  A "step" is one whole processJob of the workload's frame.  Default workload (N = 1): the benchmark's
DragonBox scene (benchmark/main.cpp:59-105) with the procedural 7.2 M-triangle stand-in for assets/xyzrgb_dragon.obj
(absent from the reference mount), 1024 x 1024, 1024 spp.

N > 1 renders the SAME frame (strong scaling: the metric is quoted on one frame, BASELINE.json) unless --scaling weak is given, and then
adds one step of the weak-scaling frame (sqrt(N) times the resolution per side) under "weak" in the JSON line.  One process per GPU over RCCL.  Started either by the launcher (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) or directly (`python bench.py --gpus N`):
then this process starts the N ranks itself as child processes BEFORE anything touches a GPU, hands them RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT, passes rank 0's JSON line through and exits non-zero if any rank fails.  The tiles of the
frame (reference src/worker.cpp:398-414) are interleaved over the ranks, the scene is replicated, and the finished tiles are
gathered to rank 0 over RCCL at the end of every step (the multi-GPU form of doWorkParallel, src/worker.cpp:364-387).
  --scaling strong (default) the frame stays `size` x `size` for every N: the default is the metric's frame (1024 x 1024, 1024 spp), and
                   BASELINE.json configs[3] is `--workload dragon --size 2048 --spp 4096 --gpus 8`.  The samples of a pixel are a serial chain
                   (one engine, one estimator: reference src/worker.cpp:149-326), so a share of 1/N of the pixels is NOT 1/N of the time:
                   tools/share_rehearsal.py measures every rank's share on one GPU (profiles/r03_share_rehearsal.txt);
  --scaling weak   the same view at sqrt(N) times the resolution per side (1448, 2048, 2896 pixels for 2, 4, 8 GPUs):
                   every GPU renders `size`^2 pixels of interleaved 32x32 tiles with the same per-pixel work.
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


def cpu_baseline(sc, cam, opt, seconds_target):
    """The reference itself (oracle/_ref, kind "reference") or, if that was not built, the C restatement (kind "port"),
    timed on this host's cores on a bounded random subset of the SAME frame's pixels (same scene, same spp)."""
    import oracle
    # the reference's default worker count is hardware_concurrency() - 1 (src/worker.cpp:366), taken here from the CPUs this job may
    # really use (host_cpus: affinity mask, cgroup quota), not from the machine's logical CPU count
    host = host_cpus()
    threads = max(host["usable"] - 1, 1)
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
            "sample": "%d random pixels of the same %dx%d frame at %d spp (%.1f s); scene build %.1f s" % (n1, w, hgt, spp, t_run, build_s),
            "host": host}


def host_cpus():
    """What this process may use of the host: logical CPUs of the machine, size of the affinity mask, cgroup CPU quota (v2 cpu.max or v1
    cfs quota), and the count the baseline goes by = the smallest of them (PT_CPU_SHARE overrides, and says so).  BASELINE.md asks for
    nproc, the core count used and the CPU model next to the number."""
    nproc = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = min(affinity, int(quota + 0.5) if quota else affinity)
    source = "affinity mask" if usable == affinity else "cgroup quota"
    if os.environ.get("PT_CPU_SHARE"):
        usable, source = max(int(os.environ["PT_CPU_SHARE"]), 1), "PT_CPU_SHARE"
    return {"nproc": nproc, "affinity": affinity, "cgroup_quota": quota, "usable": max(usable, 1), "usable_from": source, "cpu_model": model}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-weak-extra", action="store_true", help="N > 1, strong: skip the extra step of the weak-scaling frame")
    ap.add_argument("--size", type=int, default=1024, help="pixels per side: at one GPU (weak: sqrt(N) times as many at N GPUs) or of the fixed frame (strong)")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--min-spp", type=int, default=0, help="min_sample_count (0 = --spp: the metric's fixed sample count); below --spp the per-pixel estimator may stop early "
                    "(src/worker.cpp:239-259) and the value is on the max-spp basis, as the reference's benchmark counts items (benchmark/main.cpp:30)")
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

    scaling = args.scaling if world > 1 else "none"  # one GPU: nothing scales
    width, height = frame_for(world, args.size, args.scaling)
    base_seed = args.seed
    scene = None
    if standin:
        from cpupathtrace_amd import scenes
        opt = scenes.options(width, height, args.min_spp or args.spp, args.spp)
        sc, cam, label, gen_s, create_s, info = {"obj_kind": []}, None, "stand-in renderer (test only)", 0.0, 0.0, {"depth": 0}
        job = sharding.ShardedJob(None, None, opt, rank, world, device, base_seed=base_seed, render_fn=standin_render_fn(width, base_seed))
    else:
        from cpupathtrace_amd import binding, scenes
        with heartbeat("generating the scene"):
            sc, cam, label, gen_s = build_workload(args.workload, width, height, args.mesh_n)
        opt = scenes.options(width, height, args.min_spp or args.spp, args.spp)
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
    def timed(the_job, warmup, steps):
        for _ in range(warmup):
            the_job.render(want_stats=not standin)  # (the same call as a timed step: the first read-back of the counters sets up the runtime's staging buffers, 6-9 ms once)
        sync()
        t_begin = time.perf_counter()
        st = None
        for _ in range(steps):
            st = the_job.render(want_stats=not standin)
        sync()
        return time.perf_counter() - t_begin, st

    elapsed, stats = timed(job, args.warmup, args.steps)
    # N > 1 on the metric's fixed frame: one more step on the weak-scaling frame (the same view at sqrt(N) times the resolution per side,
    # the same pixels per GPU as the one-GPU run), reported under "weak" -- never as `value`
    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_weak_extra:
        w2, h2 = frame_for(world, args.size, "weak")
        from cpupathtrace_amd import scenes as _scenes
        opt2 = _scenes.options(w2, h2, args.min_spp or args.spp, args.spp)
        if standin:
            job2 = sharding.ShardedJob(None, None, opt2, rank, world, device, base_seed=base_seed, render_fn=standin_render_fn(w2, base_seed))
        else:
            job2 = sharding.ShardedJob(scene, cam, opt2, rank, world, device, base_seed=base_seed, staged_gather=(backend == "gloo"))
        t2, _ = timed(job2, 1, 1)
        times = [None] * world
        dist.all_gather_object(times, t2)
        weak = {"frame": "%dx%d" % (w2, h2), "ms_per_step": max(times) * 1e3, "value": float(w2) * h2 * args.spp / max(times) / 1e6, "unit": "Msamples/s",
                "steps": 1, "warmup": 1, "note": "same view at sqrt(N) times the resolution per side: %d pixels per GPU, as in the one-GPU run" % (w2 * h2 // world)}
        del job2
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
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not standin else "stand-in renderer (test only)",
            "config": {"workload": "%s, %dx%d, %s, per-pixel engines seeded from base seed %d" % (
                label, width, height, ("%d spp (min = max)" % args.spp) if not args.min_spp or args.min_spp == args.spp else
                ("%d..%d spp (adaptive: value on the max-spp basis)" % (args.min_spp, args.spp)), args.seed),
                       "objects": int(len(sc["obj_kind"])), "bvh_depth": int(info["depth"]), "tiles_per_gpu": int(job.n_local_tiles),
                       "tiles_total": int(len(job.tiles)),
                       "parallelism": ("one GPU: one persistent launch per frame, no gather" if world == 1 else
                                       "tiles dealt round-robin to %d GPUs (diagonals of the tile grid), scene replicated, one %s gather of the finished tiles to rank 0 per frame"
                                       % (world, "RCCL" if backend == "nccl" else backend)),
                       "scene_build_s": create_s, "scene_generate_s": gen_s},
            "distributed": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else "none"),
                            "gather_bytes_per_step": int(job.gather_bytes), "ranks": per_rank},
        }
        if stats is not None and args.min_spp and args.min_spp != args.spp:
            drawn = sum(int(r["samples"]) for r in per_rank)
            out["adaptive"] = {"samples_drawn_last_step": drawn, "fraction_of_max": drawn / (float(width) * height * args.spp),
                               "msamples_drawn_per_s": drawn / (elapsed / args.steps) / 1e6}
        if weak is not None:
            out["weak"] = weak
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


def profile_entry(pattern, keys):
    """The newest profiles/<pattern> JSON that has one of `keys`: (entry, file) or (None, None)."""
    import glob
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            continue
        for k in keys:
            if data.get(k):
                return data[k], os.path.relpath(path, ROOT)
    return None, None


L1_LOOKUP_LIMIT = 1.5   # lane look-ups per clock per CU from L1 for scattered 64-byte records (tools/ta_probe.hip, profiles/r02_ta_probe.txt)
CLOCK_HZ = 2.4e9        # MI355X peak shader clock; the chip runs at or below it, so per-clock figures computed with it are lower bounds


def bound_from_counters(d, hbm_frac):
    """What bounds the kernel, from its PMC figures (tools/pmc_summary.py): the most utilised resource if one is at 75 % or more, otherwise
    latency -- the waves spend their cycles waiting (s_waitcnt) while no unit is near its limit."""
    use = {"hbm": hbm_frac or 0.0, "valu": d["valu_busy"], "l1_lookups": d["l1_accesses_per_clk_cu"] / L1_LOOKUP_LIMIT}
    top = max(use, key=use.get)
    if use[top] >= 0.75:
        return top, use
    return ("latency" if d["waves_waiting"] >= 0.4 else "issue"), use


def roofline(s, args, msamples_per_s, world):
    """Dominant kernel = pt_path_kernel, ONE launch per step.

    `achieved` / `frac` are SURVEY.md 8(d)'s figure and nothing else: ALGORITHMIC bytes per launch (the formula, with the kernel's own work
    counters of that launch -- checked against the oracle's counters in tests/test_gpu_parity.py) divided by the launch's duration (HIP
    events on the library's stream around the launch, last timed step), against the HBM peak.  The formula bills every node visit as a
    fetch from HBM, so this is a WORK RATE in bandwidth units: caches make it exceed what HBM moves, and it is not bounded by 1.
    `algorithmic_gbs` repeats it under an honest name.  The figures that ARE bounded by 1 sit in `ceiling`:
      hbm         measured HBM bytes per second (PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/) / 8 TB/s
      l1_lookups  16-byte lane look-ups of record fetches per clock per CU (4 per node or leaf visit, counted by the kernel) / the 1.5 the
                  texture addresser delivers for scattered records (tools/ta_probe.hip); computed with the 2.4 GHz peak clock: a lower bound
      replay      ray rate / the rate of the traversal ALONE on the same frame's rays (pt_replay_kernel, tools/replay_probe.py, profiles/):
                  what a tracer that pays nothing for shading, queues and slot state would deliver
    `bound` is derived from the kernel's PMC figures of the same workload (profiles/r*_pmc_derived.json), not written by hand."""
    samples = max(s["samples"], 1)
    R = s["rays_traced"] / samples
    A = (2.0 * s["node_visits"] + s["rays_traced"]) / max(s["rays_traced"], 1)  # two slab tests per inner node + the root test
    T = s["leaf_tests"] / max(s["rays_traced"], 1)
    V = s["vertices"] / samples
    bytes_per_sample = ref_formula_bytes_per_sample(R, A, T, V)
    launches = max(s["launches"], 1)
    kernel_s = max(s["kernel_ms"], 1e-6) / 1e3
    achieved = bytes_per_sample * samples / kernel_s / 1e9
    mesh = args.mesh_n if args.workload.startswith("dragon") else 0
    keys = ["%s-%d-%d" % (args.workload, mesh, args.size), "%s-%d" % (args.workload, mesh)] if args.size == 1024 else ["%s-%d-%d" % (args.workload, mesh, args.size)]
    traffic, traffic_file = profile_entry("r*_traffic.json", keys)
    per_sample = traffic["bytes_per_sample"] if traffic else None
    hbm_frac = (per_sample * samples / kernel_s / 1e9 / HBM_PEAK_GBS) if per_sample else None
    pmc, pmc_file = profile_entry("r*_pmc_derived.json", keys)
    replay, replay_file = profile_entry("r*_replay.json", keys)
    rays_per_s = s["rays_traced"] / kernel_s
    lookups = 4.0 * (s["node_visits"] + s["leaf_tests"]) / kernel_s / CLOCK_HZ / 256.0
    ceiling = {"hbm": {"frac": hbm_frac, "achieved_gbs": (per_sample * samples / kernel_s / 1e9) if per_sample else None, "peak_gbs": HBM_PEAK_GBS, "source": traffic_file},
               "l1_lookups": {"frac": lookups / L1_LOOKUP_LIMIT, "per_clk_per_cu": lookups, "limit": L1_LOOKUP_LIMIT, "clock_hz_assumed": CLOCK_HZ},
               "replay": {"frac": (rays_per_s / replay["rays_per_s"]) if replay else None, "rays_per_s": rays_per_s,
                          "traversal_alone_rays_per_s": replay["rays_per_s"] if replay else None, "source": replay_file}}
    if pmc:
        bound, use = bound_from_counters(pmc, hbm_frac)
        note = ("from %s: waves waiting %.0f %% of their cycles, vector ALUs %.0f %% busy, L1 %.2f look-ups per clock per CU (limit %.1f), L1 hit rate %.1f %%, L2 hit rate %.0f %%, "
                "HBM %.0f %% of peak" % (pmc_file, 100 * pmc["waves_waiting"], 100 * pmc["valu_busy"], pmc["l1_accesses_per_clk_cu"], L1_LOOKUP_LIMIT, 100 * pmc["l1_hit"],
                                         100 * pmc["l2_hit"], 100 * (hbm_frac or 0.0)))
    else:
        bound, use, note = None, None, "no PMC profile of this workload under profiles/: not derived"
    return {"bound": bound, "bound_note": note, "utilisation": use,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_is": "SURVEY 8(d) work rate (algorithmic bytes / time / HBM peak): not bounded by 1; the bounded figures are in `ceiling`",
            "algorithmic_gbs": achieved,
            "traffic": per_sample * samples / launches if per_sample else None, "traffic_source": traffic_file,
            "kernel": "pt_path_kernel", "launches_per_step": launches, "avg_launch_ms": s["kernel_ms"] / launches,
            "algorithmic_bytes_per_launch": bytes_per_sample * samples / launches, "algorithmic_bytes_per_sample": bytes_per_sample,
            "hbm_frac": hbm_frac, "l2_frac": achieved / L2_PEAK_GBS, "ceiling": ceiling,
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
