#!/bin/bash
# usage (GPU box): tools/final_evidence.sh   -> gpurun_out/final/{bench.json,bench.log,kernel_stats.csv,under_rocprof.json,fetch/,write/}
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
rm -rf $out; mkdir -p $out
export PYTHONUNBUFFERED=1
PT_DEBUG=1 timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.log || { echo "bench failed"; tail -5 $out/bench.log; exit 1; }
echo "bench done"; cat $out/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --cpu-seconds 0 > $out/under_rocprof.json 2> $out/under_rocprof.log || { echo "rocprof failed"; tail -5 $out/under_rocprof.log; exit 1; }
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
echo "rocprof done"; head -4 $out/kernel_stats.csv | cut -c1-60,400-520
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --spp 64 --cpu-seconds 0 --warmup 0 > $out/fetch.log 2>&1 || { echo "fetch pass failed"; tail -3 $out/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --spp 64 --cpu-seconds 0 --warmup 0 > $out/write.log 2>&1 || { echo "write pass failed"; tail -3 $out/write.log; exit 1; }
python3 tools/measure_traffic.py dragon-1900 $out/fetch $out/write $out/traffic.json
rm -rf $out/prof $out/fetch $out/write
