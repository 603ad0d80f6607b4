#!/bin/bash
# usage: tools/traffic_pass.sh <workload> <mesh_n> <size> <spp> <outdir>: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one step of
# bench.py on that workload, merged into <outdir>/traffic.json under the key bench.py looks up (<workload>-<mesh_n or 0>-<size>)
w=$1; mesh=$2; size=$3; spp=$4; out=$5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
key=$w-$mesh-$size
[ "$w" = "cornell" -o "$w" = "box" ] && key=$w-0-$size
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --workload $w --mesh-n $mesh --size $size --spp $spp --cpu-seconds 0 --warmup 0 > $out/pmc_${key}_$c.log 2>&1 || { echo "$c pass of $key failed"; tail -3 $out/pmc_${key}_$c.log; exit 1; }
done
python3 tools/measure_traffic.py $key $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/traffic.json $(python3 -c "print($size * $size * $spp)")
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
# (bench.py looks its traffic figures up under profiles/)
python3 - $out/traffic.json profiles/r03_traffic.json <<'PY'
import json, sys
try:
    data = json.load(open(sys.argv[2]))
except (OSError, ValueError):
    data = {}
data.update(json.load(open(sys.argv[1])))
json.dump(data, open(sys.argv[2], "w"), indent=1)
PY
