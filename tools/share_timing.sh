#!/bin/bash
# usage (GPU box): tools/share_timing.sh [spp]: where a wavefront's time goes (PT_PATH_TIMING build: shading passes / traversal bursts) on the full benchmark
# frame and on its 1/2, 1/4, 1/8 shares
cd "$GRAFT_REPO_ROOT"
spp=${1:-256}
PT_DEBUG=1 PT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/bin/libpt_timing.so timeout -k 10 600 python3 tools/share_rehearsal.py --size 1024 --spp $spp --n 1,2,4,8 --ranks 1 2>&1 | grep -e "wave time" -e "^ *[1248] " -e "knobs"
