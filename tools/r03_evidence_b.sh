#!/bin/bash
# Round-3 evidence, part B (GPU box): PMC counter groups of the path kernel on the benchmark scene (derived figures -> pmc_derived.json, which bench.py
# turns into roofline.bound) and the traversal-alone ceiling of the same frame's rays (replay.json)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
export PYTHONUNBUFFERED=1
PMC_SKIP="7 8 10" PMC_TIMEOUT=120 PMC_SAMPLES=33554432 PMC_DERIVED=$out/pmc_derived.json PMC_KEY=dragon-1900-1024 tools/pmc2.sh $out/pmc_dragon -- python3 tools/render_once.py 1900 32 > $out/pmc2.log 2>&1
cp $out/pmc_dragon/summary.txt $out/pmc_path_kernel_spp32.txt
rm -rf $out/pmc_dragon/pass*/
PT_REPLAY_JSON=$out/replay.json timeout -k 10 300 python3 tools/replay_probe.py 1900 16 > $out/replay_probe.txt 2>&1 || { echo "replay failed"; tail -3 $out/replay_probe.txt; }
tail -4 $out/pmc_path_kernel_spp32.txt; grep "path kernel\|rings in 4" $out/replay_probe.txt | cut -c1-140
