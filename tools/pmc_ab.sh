#!/bin/bash
# usage (GPU box): tools/pmc_ab.sh <outdir> <lib or ""> ; one rocprofv3 --pmc pass with the SQ counters that split a wavefront's cycles
out=$1; lib=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
[ -n "$lib" ] && export PT_LIB_OVERRIDE=$lib
timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/p1" -- python3 tools/render_once.py 1900 64 > "$out/p1.log" 2>&1 || { echo "pass 1 failed"; tail -3 "$out/p1.log"; }
timeout -k 5 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR --output-format csv -d "$out/p2" -- python3 tools/render_once.py 1900 64 > "$out/p2.log" 2>&1 || { echo "pass 2 failed"; tail -3 "$out/p2.log"; }
timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d "$out/p3" -- python3 tools/render_once.py 1900 64 > "$out/p3.log" 2>&1 || { echo "pass 3 failed"; tail -3 "$out/p3.log"; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_path" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg):
    print("%-28s %.6g" % (k, agg[k]))
PY
grep Msamples "$out/p1.log"
