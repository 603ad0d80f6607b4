"""(Run as a script on the GPU box: python tests/post_bench.py [size]; it lives under tests/ because it uses the CPU checker.)
Times postProcess on a rendered frame: device (HBM-resident, pt_post_process_device), device with host buffers (pt_post_process),
and the CPU checker (compiled reference if oracle/_ref is present, else the C restatement)."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from cpupathtrace_amd import binding, scenes

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc, cam = scenes.cornell_scene(size, size)
s = binding.Scene(sc)
frame = s.process_job(cam, scenes.options(size, size, 8, 8), base_seed=1234)
s.close()
lib = binding.load()
t = torch.from_numpy(frame.copy()).cuda()
for rep in range(3):
    work = t.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.pt_post_process_device(C.c_int(0), C.c_void_p(work.data_ptr()), C.c_int32(size), C.c_int32(size), C.c_uint32(3), C.c_float(1.8), C.c_void_p(0))
    dt_dev = time.perf_counter() - t0
    assert rc == 0
t0 = time.perf_counter()
out = binding.post_process(frame, 3, 1.8)
dt_host = time.perf_counter() - t0
try:
    chk, kind = oracle.Checker("ref", ndebug=True), "compiled reference"
except Exception:
    chk, kind = oracle.Checker("oracle"), "C restatement"
t0 = time.perf_counter()
want = chk.post_process(frame, 3, 1.8)
dt_cpu = time.perf_counter() - t0
same = ((out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))).all()
print("postProcess %dx%d: device %.3f ms (frame in HBM), %.3f ms with host buffers, CPU %s %.1f ms; identical: %s" % (size, size, dt_dev * 1e3, dt_host * 1e3, kind, dt_cpu * 1e3, same))
