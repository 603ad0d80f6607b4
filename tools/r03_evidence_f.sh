#!/bin/bash
# Round-3 evidence, part F (GPU box): the PMC figures of Cornell and Box again with the final library (their leaf batching changed: leaf_min 16)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
export PYTHONUNBUFFERED=1
run() {  # workload mesh size spp
  local key=$1-0-$3
  PMC_SKIP="7 8 10" PMC_TIMEOUT=${5:-150} PMC_SAMPLES=$(($3 * $3 * $4)) PMC_DERIVED=$out/pmc_derived_f.json PMC_KEY=$key \
    tools/pmc2.sh $out/pmc_$key -- python3 bench.py --workload $1 --mesh-n $2 --size $3 --spp $4 --cpu-seconds 0 --warmup 0 > $out/pmc2_$key.log 2>&1
  cp $out/pmc_$key/summary.txt $out/pmc_${key}_summary.txt 2>/dev/null
  rm -rf $out/pmc_$key
  echo "$key done"; tail -3 $out/pmc2_$key.log
}
run cornell 0 1024 64
run box 0 1024 64
