#!/bin/bash
# usage: tools/pmc2.sh <outdir> -- <program args...>   ; cache-path counters of the two kernels, one rocprofv3 --pmc pass per group
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  case " ${PMC_SKIP:-} " in *" $i "*) continue;; esac
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; }
  echo "pass $i done: $group"
done <<'GROUPS'
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAVES
SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TCC_READ_REQ_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
python3 tools/pmc_summary.py "$out" ${PMC_SAMPLES:-0} ${PMC_DERIVED:-} ${PMC_KEY:-} > "$out/summary.txt"
cat "$out/summary.txt"
