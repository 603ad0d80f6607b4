"""The N > 1 path on CPU: two gloo ranks shard a frame's tiles, render them with a stand-in renderer and gather to rank 0.
(The real renderer needs a GPU; what is tested here is the tile assignment, the packing and the gather.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpupathtrace_amd import sharding


def _pattern(width, height):
    ys, xs = np.mgrid[0:height, 0:width]
    base = (ys * 131 + xs * 7).astype(np.float32)
    return np.stack([base, base + 0.25, base * 0.5, np.ones_like(base)], axis=-1)


def _worker(rank, world, port, width, height, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    want = torch.from_numpy(_pattern(width, height))
    calls = []

    def render_fn(tiles, image, want_stats):
        calls.append(len(tiles))
        for t in tiles:
            image[t["y"]:t["y"] + t["h"], t["x"]:t["x"] + t["w"]] = want[t["y"]:t["y"] + t["h"], t["x"]:t["x"] + t["w"]]
        return {"tiles": len(tiles)}

    opt = dict(image_width=width, image_height=height, min_sample_count=1, max_sample_count=1, epsilon=1e-3)
    job = sharding.ShardedJob(None, None, opt, rank, world, torch.device("cpu"), render_fn=render_fn)
    all_tiles = sharding._tiles_py(width, height)
    assert job.n_local_tiles == len(sharding.local_tiles(all_tiles, rank, world)) == len(all_tiles[rank::world])
    for _ in range(2):  # two frames through the same buffers
        stats = job.render(want_stats=True)
    assert stats["tiles"] == job.n_local_tiles
    if rank == 0:
        np.save(out_path, job.image.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("width,height", [(132, 68), (64, 64), (40, 9)])
def test_two_rank_gather(tmp_path, width, height):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, width, height, out), nprocs=2, join=True)
    got = np.load(out)
    assert np.array_equal(got, _pattern(width, height))


def test_tile_owner_spreads_rows_and_columns():
    t = sharding._tiles_py(1024, 1024)  # 32 x 32 tiles
    for world in (2, 4, 8):
        owner = sharding.tile_owner(t, world)
        assert (np.bincount(owner, minlength=world) == len(t) // world).all()
        for r in range(world):
            mine = t[owner == r]
            assert len(np.unique(mine["x"])) == 32 and len(np.unique(mine["y"])) == 32, "every rank sees every tile column and row"
        assert len(np.concatenate([sharding.pixel_indices(sharding.local_tiles(t, r, world), 1024) for r in range(world)])) == 1024 * 1024
    t = sharding._tiles_py(40, 9)  # 20 tiles of 2 x 2 per row, 5 rows: 100 tiles
    owner = sharding.tile_owner(t, 3)  # rows are no multiple of 3: plain round-robin
    assert (owner == np.arange(len(t)) % 3).all()
    assert (sharding.tile_owner(t[:7], 1) == 0).all()


def test_tile_list_matches_reference_rule():
    # src/worker.cpp:398-414
    t = sharding._tiles_py(1024, 1024)
    assert len(t) == 1024 and (t["w"] == 32).all()
    t = sharding._tiles_py(132, 68)
    assert len(t) == 8 * 4 and t["w"].max() == 17 and t[-1].tolist() == (119, 51, 13, 17)
    idx = sharding.pixel_indices(t, 132)
    assert len(idx) == 132 * 68 and len(np.unique(idx)) == len(idx)
