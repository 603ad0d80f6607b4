"""Premise check for cost-aware placement: do a few heavy streams, one per wavefront, keep their lonely speed while the rest of the GPU
is busy with dense wavefronts?  Uses explicit streams (processItem per pixel) and zero-area filler streams to shape the wavefronts:
with pieces of 1, stream i of a single-round job is slot i / waves of wavefront i % waves (pt_path.hip, first round).

    python tools/mix_probe.py [spp]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cpupathtrace_amd import binding, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
size, waves, slots = 1024, 2048, 64
sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(1900, 1900, scenes.DRAGON_BOX_TRANSFORM))
s = binding.Scene(sc)
opt = scenes.options(size, size, spp, spp)
tiles = binding.job_tiles(size, size)
t = tiles[400]
hx, hy = np.meshgrid(np.arange(t["x"], t["x"] + 32), np.arange(t["y"], t["y"] + 32))
heavy = np.stack([hx.ravel(), hy.ravel()], axis=1)                      # 1024 pixels of the tile with the slowest chain
rng = np.random.default_rng(1)
allpix = np.stack(np.meshgrid(np.arange(size), np.arange(size)), axis=-1).reshape(-1, 2)
back = allpix[rng.permutation(len(allpix))[:(waves - 1024) * slots]]  # random pixels of the frame for the dense wavefronts


def job(heavy_on, dense_on, heavy_per_wave=1):
    st = np.zeros(waves * slots, dtype=binding.STREAM_DTYPE)            # zero-area streams: finished at once
    grid = np.arange(waves * slots)
    w, q = grid % waves, grid // waves
    if heavy_on:
        n_hw = 1024 // heavy_per_wave
        for k in range(heavy_per_wave):
            sel = (w < n_hw) & (q == k)
            px = heavy[k * n_hw:(k + 1) * n_hw]
            st["x"][sel], st["y"][sel], st["w"][sel], st["h"][sel] = px[:, 0], px[:, 1], 1, 1
    if dense_on:
        sel = w >= 1024
        st["x"][sel], st["y"][sel], st["w"][sel], st["h"][sel] = back[:, 0], back[:, 1], 1, 1
    st["rng_state"] = np.arange(1, len(st) + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    s.process_item(cam, scenes.options(size, size, 2, 2), st)
    img, states, stats = s.process_item(cam, opt, st, want_stats=True)
    return stats


for label, args in (("1024 heavy streams, one per wavefront, nothing else", (True, False)), ("65536 background streams in 1024 dense wavefronts, nothing else", (False, True)),
                    ("both at once", (True, True)), ("heavy streams four per wavefront + background", (True, True, 4)),
                    ("heavy streams sixteen per wavefront + background", (True, True, 16))):
    stt = job(*args)
    print("%-70s kernel %7.1f ms, %d wavefronts x %d rows, %.1f walks per wave step" % (label, stt["kernel_ms"], stt["wavefronts"], stt["slot_rows"],
                                                                                       (stt["node_visits"] + stt["leaf_tests"]) / max(stt["wave_steps"], 1)), flush=True)
s.close()
