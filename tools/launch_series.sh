#!/bin/bash
# usage (GPU box): tools/launch_series.sh <mesh_n> <spp>  -> durations of successive pt_trace_kernel / pt_shade_kernel launches of one frame
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/series
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/series -- python3 tools/sweep.py $1 $2 '{"PT_GROUPS":[1]}' > /tmp/series.log 2>&1 || { tail -3 /tmp/series.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/series/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tr = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "pt_trace_kernel" in r["Kernel_Name"]]
sh = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "pt_shade_kernel" in r["Kernel_Name"]]
print("trace launches", len(tr), "shade launches", len(sh))
# the second render of the sweep is the measured one: take the last 75 % of the launches as a rough cut
n = len(tr)
start = n - int(n * 8 / 9)   # warm-up is spp 4 of (4 + spp)
t2, s2 = tr[start:], sh[start:]
print("frame: %d launches, trace total %.1f ms, shade total %.1f ms" % (len(t2), sum(t2) / 1e3, sum(s2) / 1e3))
for i in range(0, len(t2), max(1, len(t2) // 24)):
    print("launch %4d: trace %7.1f us  shade %6.1f us" % (i, t2[i], s2[i]))
PY
