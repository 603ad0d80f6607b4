#!/bin/bash
# The reference's benchmark program (128 x 128 x 256 spp per iteration) under a few scheduler settings (GPU box).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
work=/tmp/ref_programs; rm -rf $work; mkdir -p $work/assets
python3 tools/write_standin_obj.py $work/assets/xyzrgb_dragon.obj 1900 > /dev/null 2>&1 || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/cpupathtrace_amd:$LD_LIBRARY_PATH
run() {
  printf "%-60s" "$*"
  (cd $work && env "$@" timeout -k 10 300 $GRAFT_REPO_ROOT/oracle/_ref/ref_benchmark --benchmark_min_time=1 2>&1 | grep "^renderScene" | awk '{printf "%s %s ms   ", $1, $2}')
  echo
}
run PT_NOP=1
run PT_SPREAD_WAVES=2048
run PT_SPREAD_WAVES=4096
run PT_SPREAD_WAVES=4096 PT_LEAF_MIN=1
run PT_SPREAD_WAVES=2048 PT_LEAF_MIN=1
run PT_LEAF_MIN=1
run PT_SPREAD_WAVES=512
run PT_MIN_READY=4
run PT_REFILL_IDLE=32
run PT_REFILL_IDLE=60
run PT_BURST=48
run PT_SPREAD_WAVES=4096 PT_REFILL_IDLE=60
