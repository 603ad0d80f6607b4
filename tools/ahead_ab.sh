#!/bin/bash
# usage (GPU box): tools/ahead_ab.sh <name> [<name> ...]: variants tools/bin/libpt_<name>.so on the three regimes -- a lone chain (one tile), the 1/8 share of
# the benchmark frame (ranks 2 and 6), the full frame
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export PT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/bin/libpt_$v.so
  echo "=== $v"
  timeout -k 10 200 python3 tools/chain_probe.py --spp 128 --tiles 496,400 2>/dev/null | grep "^tile"
  timeout -k 10 300 python3 tools/share_rehearsal.py --spp 256 --n 8 --ranks 3 2>/dev/null | grep "^  8\|->"
  timeout -k 10 300 python3 tools/render_once.py 1900 128 1024 3 2>/dev/null | tail -n 2 | tr '\n' ' '; echo
  timeout -k 10 300 python3 tools/render_once.py 1900 256 128 3 2>/dev/null | tail -n 1
done
