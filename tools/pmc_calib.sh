#!/bin/bash
# calibrate FETCH_SIZE on a known random 64-byte-record gather (tools/gather_bench.hip), then measure the bench's kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_calib -- build/gather_bench 461 64 > gpurun_out/pmc_calib.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_calib/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"], "grid", r.get("Grid_Size"))
PY
