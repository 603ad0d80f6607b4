#!/bin/bash
# Round-3 evidence, part A (GPU box): the default bench line, the same command under rocprofv3 --kernel-trace --stats, and the HBM traffic
# of the same workload from separate FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/r03/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
export PYTHONUNBUFFERED=1
# (the traffic figure first: bench.py's roofline.traffic reads it from profiles/r03_traffic.json)
tools/traffic_pass.sh dragon 1900 1024 64 $out || exit 1
PT_DEBUG=1 timeout -k 10 300 python3 bench.py > $out/bench_default.json 2> $out/bench_default.log || { echo "bench failed"; tail -5 $out/bench_default.log; exit 1; }
echo "bench done"; cut -c1-300 $out/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --cpu-seconds 0 > $out/bench_default_under_rocprof.json 2> $out/under_rocprof.log || { echo "rocprof failed"; tail -5 $out/under_rocprof.log; exit 1; }
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_default_kernel_stats.csv
rm -rf $out/prof
echo "rocprof done"; head -3 $out/bench_default_kernel_stats.csv | cut -c1-200
