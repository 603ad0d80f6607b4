#!/bin/bash
# usage (GPU box): tools/variant_sweep.sh "<variant names>" "<mesh_n list>" [spp]: every variant (tools/bin/libpt_<name>.so) on every workload of tools/sweep.py, twice
cd "$GRAFT_REPO_ROOT"
spp=${3:-128}
for n in $2; do
  for round in 1 2; do
    for v in $1; do
      printf "%-6s mesh %-5s round %d: " $v $n $round
      PT_DEBUG=1 PT_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/bin/libpt_$v.so timeout -k 10 300 python3 tools/sweep.py $n $spp "[{}]" 1024 2>&1 | grep -e Msamples -e "path kernel:" | sed -e "s/.*path kernel: 256 CUs x //" -e "s/ rows of slots per wavefront//" | cut -c1-95 | tr '\n' ' '
      echo
    done
  done
done
