#!/bin/bash
# Round-3 evidence, part C (GPU box): the other workloads -- bench line + HBM traffic each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
export PYTHONUNBUFFERED=1
case "$1" in
 benches)
  timeout -k 10 200 python3 bench.py --workload box --spp 256 --cpu-seconds 0 > $out/box_spp256.json 2> $out/box.log
  timeout -k 10 200 python3 bench.py --workload cornell --spp 256 --cpu-seconds 0 > $out/cornell_spp256.json 2> $out/cornell.log
  timeout -k 10 300 python3 bench.py --workload dragon --size 2048 --spp 256 --cpu-seconds 0 > $out/dragon_size2048_spp256.json 2> $out/dragon2048.log
  ;;
 small)
  tools/traffic_pass.sh box 0 1024 256 $out && timeout -k 10 200 python3 bench.py --workload box --spp 256 --cpu-seconds 0 > $out/box_spp256.json 2> $out/box.log
  tools/traffic_pass.sh cornell 0 1024 256 $out && timeout -k 10 200 python3 bench.py --workload cornell --spp 256 --cpu-seconds 0 > $out/cornell_spp256.json 2> $out/cornell.log
  tools/traffic_pass.sh dragon 1900 2048 64 $out && timeout -k 10 300 python3 bench.py --workload dragon --size 2048 --spp 256 --cpu-seconds 0 > $out/dragon_size2048_spp256.json 2> $out/dragon2048.log
  ;;
 dragons16)
  tools/traffic_pass.sh dragons16 1900 1024 512 $out && timeout -k 10 500 python3 bench.py --workload dragons16 --spp 512 --cpu-seconds 0 > $out/dragons16_spp512.json 2> $out/dragons16_spp512.log
  ;;
esac
ls -la $out | tail -12
