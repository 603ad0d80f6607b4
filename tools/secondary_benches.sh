#!/bin/bash
# the other workloads of DESIGN.md's table -> gpurun_out/secondary/*.json
out=gpurun_out/secondary; mkdir -p $out
timeout -k 10 200 python3 bench.py --workload box --spp 256 --cpu-seconds 0 > $out/box_spp256.json 2> $out/box.log || exit 1
timeout -k 10 200 python3 bench.py --workload cornell --spp 256 --cpu-seconds 0 > $out/cornell_spp256.json 2> $out/cornell.log || exit 1
timeout -k 10 300 python3 bench.py --workload dragon --size 2048 --spp 256 --scaling strong --cpu-seconds 0 > $out/dragon_size2048spp256scalingstrong.json 2> $out/dragon2048.log || exit 1
timeout -k 10 500 python3 bench.py --workload dragons16 --spp 64 --cpu-seconds 0 > $out/dragons16_spp64.json 2> $out/dragons16_spp64.log || exit 1
