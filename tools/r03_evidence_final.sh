#!/bin/bash
# Round-3 evidence with the final library (GPU box): part A (traffic passes, default bench line, the same command under rocprofv3 --kernel-trace
# --stats), Cornell (configs[1]) and Box at 256 spp, the adaptive-sampling table -> gpurun_out/r03/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
tools/r03_evidence_a.sh || exit 1
out=gpurun_out/r03
timeout -k 10 200 python3 bench.py --workload cornell --spp 256 --cpu-seconds 0 > $out/cornell_spp256.json 2> $out/cornell.log || { echo "cornell failed"; exit 1; }
timeout -k 10 200 python3 bench.py --workload box --spp 256 --cpu-seconds 0 > $out/box_spp256.json 2> $out/box.log || { echo "box failed"; exit 1; }
echo "cornell / box done"
timeout -k 10 300 python3 tools/adaptive_probe.py 1900 > $out/adaptive.txt 2>&1 || { echo "adaptive probe failed"; tail -3 $out/adaptive.txt; exit 1; }
cat $out/adaptive.txt
