#!/bin/bash
# Round-3 evidence, part E (GPU box): the PMC figures behind `roofline.bound` for the other bench workloads (the benchmark frame's are from part B):
# tools/pmc2.sh over one step of bench.py each, merged into gpurun_out/r03/pmc_derived_more.json under the keys bench.py looks up.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
export PYTHONUNBUFFERED=1
run() {  # workload mesh size spp
  local key=$1-$2-$3
  [ "$1" = "cornell" -o "$1" = "box" ] && key=$1-0-$3
  PMC_SKIP="7 8 10" PMC_TIMEOUT=${5:-150} PMC_SAMPLES=$(($3 * $3 * $4)) PMC_DERIVED=$out/pmc_derived_more.json PMC_KEY=$key \
    tools/pmc2.sh $out/pmc_$key -- python3 bench.py --workload $1 --mesh-n $2 --size $3 --spp $4 --cpu-seconds 0 --warmup 0 > $out/pmc2_$key.log 2>&1
  cp $out/pmc_$key/summary.txt $out/pmc_${key}_summary.txt 2>/dev/null
  rm -rf $out/pmc_$key
  echo "$key done"; tail -3 $out/pmc2_$key.log
}
run cornell 0 1024 64
run box 0 1024 64
run dragon 1900 2048 16
run dragons16 1900 1024 32 240
