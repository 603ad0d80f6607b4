#!/bin/bash
# usage: tools/pmc.sh <outdir> -- <program args...>   ; runs one rocprofv3 --pmc pass per counter group (no tracing domains)
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -- "$@" > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out.pass$i.log"; }
  echo "pass $i done: $group"
done <<'GROUPS'
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE TCC_REQ_sum
GROUPS
