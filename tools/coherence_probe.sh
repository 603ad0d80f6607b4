#!/bin/bash
# usage: tools/coherence_probe.sh <mesh_n>   (on the GPU box; writes gpurun_out/coherence.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/coherence.txt
for v in camera camera_shuffled random random_sorted bounce bounce_shuffled bounce_sorted; do
  rm -rf /tmp/coh_$v
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/coh_$v -- python3 tools/coherence_probe.py $1 $v > /tmp/coh_$v.log 2>&1 || { echo "$v failed"; tail -3 /tmp/coh_$v.log; exit 1; }
  grep "rays," /tmp/coh_$v.log >> gpurun_out/coherence.txt
  f=$(find /tmp/coh_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> gpurun_out/coherence.txt <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "pt_trace_kernel" in row["Name"]:
        print("   trace kernel: calls %s avg %.1f us min %.1f us max %.1f us" % (row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
done
cat gpurun_out/coherence.txt
