"""Turn the csv output of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into profiles/rNN_traffic.json.

usage: python tools/measure_traffic.py <workload-key> <fetch_dir> <write_dir> <out.json>

HBM bytes per launch of pt_path_kernel = 2 * FETCH_SIZE[KiB] * 1024 + WRITE_SIZE[KiB] * 1024: on gfx950 FETCH_SIZE counts 64 B per
128-B read request (MI355X_MICROARCH.md, HBM section); the factor was checked on a known access pattern of the same kind
(tools/gather_bench.hip: random 64-byte records from a 461 MB table, profiles/r01_gather_bench_random64B.txt and the
FETCH_SIZE values in profiles/r01_pmc_calibration.txt: 0.87 x 64 B counted per record, i.e. 111 B fetched per record whose
line misses L2 87 % of the time)."""
import csv, glob, json, sys

key, fetch_dir, write_dir, out = sys.argv[1:5]
samples = float(sys.argv[5]) if len(sys.argv) > 5 else 64.0 * 1024 * 1024  # samples of the profiled frame (width * height * spp)


def per_launch(root, counter):
    total, n = 0.0, 0
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pt_path" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                total += float(r["Counter_Value"])
                n += 1
    return total / max(n, 1), n


fetch_kib, n1 = per_launch(fetch_dir, "FETCH_SIZE")
write_kib, n2 = per_launch(write_dir, "WRITE_SIZE")
entry = {"kernel": "pt_path_kernel", "fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib, "launches": n1,
         "bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
         "bytes_per_sample": (2.0 * fetch_kib * 1024.0 + write_kib * 1024.0) * n1 / samples,
         "samples_per_launch": samples,
         "note": "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 64 B per 128-B request), separate rocprofv3 --pmc passes over one step of bench.py on the same workload (tools/traffic_pass.sh); one launch per frame"}
try:
    data = json.load(open(out))
except (OSError, ValueError):
    data = {}
data[key] = entry
json.dump(data, open(out, "w"), indent=1)
print(key, entry)
