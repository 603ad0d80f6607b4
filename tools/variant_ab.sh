#!/bin/bash
# usage (GPU box): tools/variant_ab.sh <spp> <frames> <name> [<name> ...]: the benchmark frame through tools/bin/libpt_<name>.so, one process per variant,
# the variants interleaved twice (box-to-box and run-to-run spread is a few per cent: only differences inside one call count)
spp=$1; frames=$2; shift 2
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for v in "$@"; do
    printf "%-8s round %d: " $v $round
    PT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/bin/libpt_$v.so timeout -k 10 300 python3 tools/render_once.py 1900 $spp 1024 $frames 2>/dev/null | tail -n +2 | tr '\n' ' '
    echo
  done
done
