"""Tile sharding of one frame over the GPUs of a node: the multi-GPU form of the reference's tile queue.

The reference cuts a frame into <= 32x32 tiles and lets its worker threads pull them from one queue; tiles never talk to
each other (src/worker.cpp:328-424).  Here every rank (one process per GPU, torch.distributed over RCCL) owns every world-th
tile of the same list (`tile_owner`: dealt round-robin along the rows of the tile grid, and along its diagonals when the rows
are a multiple of the world size, so that a rank never ends up with whole columns of the frame), renders them with the
replicated scene, and the only exchange is the gather of finished tiles to rank 0 at the end of a frame.  Because every pixel has its own engine seeded from
(base_seed, x, y), the assembled frame is bit-identical for any number of ranks.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import binding


def tile_owner(tiles, world):
    """Rank of every tile.  Tile k of processJob's row-major list goes to rank k % world; when the grid's rows hold a multiple of
    `world` tiles that would give every rank whole COLUMNS of the frame (with 32 tiles per row and 8 ranks: columns r, r + 8, ...,
    and the ranks' loads differ by what their columns show), so every grid row is then shifted by one more rank: (column + row) % world.
    Same rule in pt_render_tiles_multi (cpupathtrace_amd/csrc/pt_api.cpp)."""
    k = np.arange(len(tiles), dtype=np.int64)
    if world <= 1 or len(tiles) == 0:
        return np.zeros(len(tiles), np.int64)
    per_row = int((tiles["y"] == tiles["y"][0]).sum())
    regular = per_row > 0 and len(tiles) % per_row == 0 and bool((tiles["y"][:per_row] == tiles["y"][0]).all())
    if regular and per_row % world == 0:
        return (k % per_row + k // per_row) % world
    return k % world


def local_tiles(tiles, rank, world):
    return tiles[tile_owner(tiles, world) == rank]


def pixel_indices(tiles, width):
    """Row-major pixel indices (y * width + x) covered by `tiles`, tile after tile."""
    parts = []
    for t in tiles:
        ys = np.arange(t["y"], t["y"] + t["h"], dtype=np.int64)[:, None]
        xs = np.arange(t["x"], t["x"] + t["w"], dtype=np.int64)[None, :]
        parts.append((ys * width + xs).ravel())
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


class ShardedJob:
    """One frame (FrameRenderJob) whose tiles are spread over `world` ranks.

    render_fn(tiles, image_tensor, want_stats) renders the given tiles into the (height, width, 4) float32 tensor on this
    rank's device; the default calls the HIP library.  Tests pass their own to exercise the sharding on CPU/gloo."""

    def __init__(self, scene, camera, options, rank, world, device, base_seed=1234, render_fn=None, staged_gather=False, always_gather=False):
        """staged_gather: move the chunks through host memory around the gather (gloo has no device gather; used to rehearse the
        N > 1 path with several ranks on ONE GPU -- RCCL wants one device per rank).  always_gather: run the pack / gather / scatter
        of the N > 1 path even in a world of one rank (tests/test_gpu_rccl.py: the collective executes on the device through RCCL)."""
        self.scene, self.camera, self.options = scene, camera, options
        self.staged_gather = staged_gather
        self.gather_bytes = 0
        self.rank, self.world, self.device, self.base_seed = rank, world, device, base_seed
        self.width, self.height = options["image_width"], options["image_height"]
        self.tiles = binding.job_tiles(self.width, self.height) if scene is not None else _tiles_py(self.width, self.height)
        self.mine = local_tiles(self.tiles, rank, world)
        self.n_local_tiles = len(self.mine)
        self.image = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=device)
        self.render_fn = render_fn or self._render_hip
        self.gathers = world > 1 or always_gather
        if self.gathers:
            per_rank = [pixel_indices(local_tiles(self.tiles, r, world), self.width) for r in range(world)]
            self.chunk = max(len(p) for p in per_rank)
            self.my_index = torch.from_numpy(per_rank[rank]).to(device)
            self.send = torch.zeros((self.chunk, 4), dtype=torch.float32, device=device)
            self.gather_bytes = self.chunk * 16 * (world - 1)  # what rank 0 receives from the other ranks per frame
            xdev = torch.device("cpu") if staged_gather else device
            if rank == 0:
                self.all_index = [torch.from_numpy(p).to(device) for p in per_rank]
                self.recv = [torch.zeros((self.chunk, 4), dtype=torch.float32, device=xdev) for _ in range(world)]

    def _render_hip(self, tiles, image, want_stats):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        return self.scene.process_job_device(self.camera, self.options, image.data_ptr(), stream, base_seed=self.base_seed, tiles=tiles,
                                             want_stats=want_stats)

    def render(self, want_stats=False):
        """Render this rank's tiles, then gather all tiles into rank 0's image.  Returns the render statistics of this rank."""
        stats = self.render_fn(self.mine, self.image, want_stats)
        if self.gathers:
            flat = self.image.view(-1, 4)
            self.send[: len(self.my_index)] = flat[self.my_index]
            dist.gather(self.send.cpu() if self.staged_gather else self.send, self.recv if self.rank == 0 else None, dst=0)
            if self.rank == 0:
                # (rank 0's own tiles are in place already; with always_gather they are written again from what came back)
                for r in range(0 if self.world == 1 else 1, self.world):
                    idx = self.all_index[r]
                    flat[idx] = self.recv[r][: len(idx)].to(flat.device)
        return stats


def _tiles_py(width, height):
    """processJob's tile list (reference src/worker.cpp:398-414) without loading the HIP library (CPU-only tests)."""
    width, height = max(width, 0), max(height, 0)
    if width == 0 or height == 0:
        return np.zeros(0, dtype=binding.TILE_DTYPE)
    ts = max(min(min(width, height) // 4, 32), 1)
    out = [(tx * ts, ty * ts, min(width - tx * ts, ts), min(height - ty * ts, ts))
           for ty in range((height + ts - 1) // ts) for tx in range((width + ts - 1) // ts)]
    return np.array(out, dtype=binding.TILE_DTYPE)
