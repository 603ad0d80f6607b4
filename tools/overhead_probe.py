"""What a bench step costs outside the launch: wall time of ShardedJob.render (what bench.py times) against the launch duration by HIP events,
for Box, Cornell and a 2 M-triangle mesh at 1024 x 1024 (profiles/r03_host_overhead.txt: 0.2-0.6 ms; the FIRST call that reads the counters back
costs 6 ms more).      python tools/overhead_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cpupathtrace_amd import binding, scenes, sharding
size = 1024
for name in ("box", "cornell", "dragon"):
    if name == "box":
        sc, cam = scenes.box_scene(aspect_ratio=-1.0)
    elif name == "cornell":
        sc, cam = scenes.cornell_scene(size, size)
    else:
        sc, cam = scenes.dragon_box_scene(*scenes.bumpy_sphere_mesh(600, 600, scenes.DRAGON_BOX_TRANSFORM))
    s = binding.Scene(sc)
    dev = torch.device("cuda", 0)
    for spp in (16, 64, 256):
        opt = scenes.options(size, size, spp, spp)
        job = sharding.ShardedJob(s, cam, opt, 0, 1, dev, base_seed=1234)
        job.render(); torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            st = job.render(want_stats=True)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("%-8s spp %4d: render() %.2f ms, + synchronize %.2f ms, kernel (events) %.2f ms -> outside the events %.2f ms" % (name, spp, (t1 - t0) * 1e3, (t2 - t1) * 1e3, st["kernel_ms"], (t2 - t0) * 1e3 - st["kernel_ms"]), flush=True)
    s.close()
