#!/bin/bash
# Round-3 evidence, part D (GPU box): strong-scaling rehearsal, the chain of the slowest pixel, the instruction-cost probe, adaptive sampling,
# the reference's programs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
export PYTHONUNBUFFERED=1
timeout -k 10 300 python3 tools/share_rehearsal.py > $out/share_rehearsal.txt 2> $out/share.err
timeout -k 10 300 python3 tools/share_rehearsal.py --size 2048 --spp 256 --n 1,8 >> $out/share_rehearsal.txt 2>> $out/share.err
timeout -k 10 200 python3 tools/chain_probe.py --spp 256 --tiles 496,400,272,0 > $out/chain_probe.txt 2>&1
timeout -k 10 100 tools/bin/issue_probe > $out/issue_probe.txt 2>&1
timeout -k 10 200 python3 tools/step_timing.py > $out/step_timing.txt 2>&1
timeout -k 10 300 python3 tools/adaptive_probe.py > $out/adaptive.txt 2>&1
tail -3 $out/share_rehearsal.txt; cat $out/adaptive.txt | cut -c1-200
