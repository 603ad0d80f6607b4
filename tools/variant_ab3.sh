#!/bin/bash
# usage (GPU box): tools/variant_ab3.sh <name> [<name> ...]: dragon, Cornell and Box frames (1024 x 1024, 128 spp) through tools/bin/libpt_<name>.so, one
# process per variant and workload, the variants interleaved twice (only differences inside one call count)
cd "$GRAFT_REPO_ROOT"
for wl in "1900 128" "0 128" "-1 128"; do
  for round in 1 2; do
    for v in "$@"; do
      printf "%-10s workload %-9s round %d: " $v "$wl" $round
      PT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/bin/libpt_$v.so timeout -k 10 300 python3 tools/render_once.py $wl 1024 3 2>/dev/null | tail -n +2 | tr '\n' ' '
      echo
    done
  done
done
