set -o pipefail
for env in "PT_ROWS=1" "PT_ROWS=8 PT_STACK_LDS=16" "PT_BURST=1 PT_LEAF_MIN=64" "PT_REFILL_IDLE=64 PT_MIN_READY=512" "PT_SPREAD_WAVES=4 PT_REFILL_IDLE=1 PT_MIN_READY=1" "PT_BLOCKS_PER_CU=1 PT_ROWS=2"; do
  echo "== $env"; env $env timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not configs4 and not mesh7m and not full_size" 2>&1 | tail -1 || exit 1
done
