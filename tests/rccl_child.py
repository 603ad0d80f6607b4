"""Child process of tests/test_gpu_rccl.py: one rank, backend nccl (= RCCL on ROCm), the sharded job's gather path on the device.
Started as a fresh interpreter before anything in its parent touched the GPU.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

from cpupathtrace_amd import binding, scenes, sharding


def main():
    port = int(sys.argv[1])
    os.environ["MASTER_PORT"] = str(port)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    sc, cam = scenes.box_scene()
    opt = scenes.options(96, 80, 6, 6)
    scene = binding.Scene(sc, device=0)
    want = scene.process_job(cam, opt, base_seed=99)
    job = sharding.ShardedJob(scene, cam, opt, 0, 1, device, base_seed=99, always_gather=True)
    job.render()
    job.image.zero_()  # what rank 0 shows afterwards must have come back through the collective
    flat = job.image.view(-1, 4)
    dist.gather(job.send, job.recv, dst=0)
    flat[job.all_index[0]] = job.recv[0][: len(job.all_index[0])]
    # an all_reduce too: a checksum of the frame summed over the (one) rank, on the device
    total = job.image.double().sum().reshape(1)
    dist.all_reduce(total)
    torch.cuda.synchronize()
    got = job.image.cpu().numpy()
    maps = open("/proc/self/maps").read()
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "identical": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))),
           "checksum_ok": bool(abs(float(total.item()) - float(want.astype(np.float64).sum())) <= 1e-6 * max(1.0, abs(float(want.sum())))),
           "recv_on_device": bool(job.recv[0].is_cuda), "rccl_mapped": "librccl" in maps, "gather_bytes": int(job.send.numel() * 4),
           "nonzero_pixels": int((got[..., 3] > 0).sum())}
    scene.close()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
