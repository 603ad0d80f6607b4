"""bench.py's N > 1 path on CPU: `--gpus 2` must start two ranks (itself, or under torch.distributed.run), shard the frame's tiles
between them without overlap, gather every pixel to rank 0 and label the line with the world size it really ran with.  The renderer
is the stand-in of bench.py (PT_BENCH_STANDIN=1, gloo): what is tested is the launch, the split (reference src/worker.cpp:398-414 tiles,
`tile % world == rank`) and the gather -- the real renderer needs a GPU (tests/test_gpu_parity.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = dict(os.environ, PT_BENCH_STANDIN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def _expected(width, height, seed):
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float32)
    v = ys * np.float32(width) + xs + np.float32(seed)
    return np.stack([v, v * np.float32(0.5), v * np.float32(0.25), np.ones_like(v)], axis=-1)


def _check(line, frame_path, world, side, scaling):
    out = json.loads(line)
    assert out["n_gpus"] == world and out["scaling"] == scaling
    d = out["distributed"]
    assert d["world_size"] == world and d["backend"] == "gloo" and len(d["ranks"]) == world
    assert sorted(r["rank"] for r in d["ranks"]) == list(range(world))
    assert sum(r["tiles"] for r in d["ranks"]) == out["config"]["tiles_total"]
    assert sum(r["pixels"] for r in d["ranks"]) == side * side, "the ranks' tile sets cover the frame exactly once"
    assert len({r["tile_hash"] for r in d["ranks"]}) == world, "ranks rendered different tile sets"
    assert d["gather_bytes_per_step"] > 0
    assert out["metric"].startswith("INVALID"), "a stand-in run must not look like a measurement"
    assert np.array_equal(np.load(frame_path), _expected(side, side, 1234)), "every pixel reached rank 0"


@pytest.mark.parametrize("scaling,side", [("weak", 88), ("strong", 64), (None, 64)])
def test_bench_spawns_its_own_ranks(tmp_path, scaling, side):
    """Without --scaling an N > 1 run is on the metric's fixed frame (strong) and carries the weak-scaling frame as an extra field."""
    frame = str(tmp_path / "frame.npy")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--size", "64", "--spp", "1", "--steps", "1", "--warmup", "1"] + (["--scaling", scaling] if scaling else []) +
                       ["--frame-out", frame], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one JSON line"
    _check(lines[0], frame, 2, side, scaling or "strong")
    out = json.loads(lines[0])
    if scaling == "weak":
        assert "weak" not in out
    else:
        assert out["weak"]["frame"] == "88x88" and out["weak"]["value"] > 0 and out["weak"]["steps"] == 1
        assert "%dx%d" % (side, side) in out["config"]["workload"], "`value` is on the fixed frame"


def test_bench_under_the_launcher(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    frame = str(tmp_path / "frame.npy")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        BENCH, "--gpus", "2", "--size", "64", "--spp", "1", "--steps", "2", "--warmup", "0", "--scaling", "strong", "--frame-out", frame],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _check(lines[0], frame, 2, 64, "strong")


def test_a_dead_rank_voids_the_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--size", "32", "--spp", "1", "--warmup", "0"], env=_env(PT_BENCH_TEST_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no result line from a void run"


def test_mismatched_world_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--size", "32", "--spp", "1"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr
