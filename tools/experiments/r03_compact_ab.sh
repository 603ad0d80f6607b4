#!/bin/bash
# A/B of the compacted shading passes (PT_COMPACT=0|1) on adaptive and fixed sampling
set -o pipefail
spec='[{"PT_COMPACT":0},{"PT_COMPACT":1},{"PT_COMPACT":0},{"PT_COMPACT":1}]'
for args in "1900 64 1024 16" "1900 64 1024 64" "1900 256 1024 64" "0 64 1024 16" "0 64 1024 64" "-1 64 1024 16"; do
  set -- $args
  echo "== mesh $1 spp $4..$2 size $3"
  timeout -k 10 200 python tools/sweep.py $1 $2 "$spec" $3 $4 || exit 1
done
