"""Print VGPR/SGPR/scratch/LDS figures of every kernel in libpathtrace_hip.so (from the code objects' metadata notes)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpupathtrace_amd", "libpathtrace_hip.so")
pat = sys.argv[2] if len(sys.argv) > 2 else "pt_path_kernel"
data = open(so, "rb").read()
# code objects sit in .hip_fatbin as ELF images behind a clang offload bundle header; find each ELF by its magic
starts = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01", data)][1:]
with tempfile.TemporaryDirectory() as tmp:
    for i, s in enumerate(starts):
        path = os.path.join(tmp, "co%d.o" % i)
        open(path, "wb").write(data[s:])
        out = subprocess.run([LLVM + "/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        name = None
        rec = {}
        for line in out.splitlines():
            m = re.match(r"\s+[-\s]*\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "name"):
                rec[k] = v
            if k == "wavefront_size" or k == "vgpr_spill_count":
                if "name" in rec and "vgpr_count" in rec and pat in rec["name"]:
                    print(rec)
                    rec = {}
