"""Summarise rocprofv3 --pmc csv output per kernel: sum of each counter over all dispatches of a kernel, and per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "pt_path" if "pt_path" in k else "pt_closest" if "pt_closest" in k else "pt_replay" if "pt_replay" in k else None
        if k is None:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print("==", k)
    for c in sorted(agg[k]):
        print("  %-40s total %.6g  per-dispatch %.6g  (n=%d)" % (c, agg[k][c], agg[k][c] / cnt[k][c], cnt[k][c]))
