"""Summarise rocprofv3 --pmc csv output per kernel: sum of each counter over all dispatches of a kernel, and per dispatch.

    python tools/pmc_summary.py <dir with passN/ subdirectories> [samples of the profiled frame] [derived.json workload-key]

With the last two arguments the derived figures of pt_path_kernel are also merged into derived.json under the workload key
(bench.py reads the newest profiles/r*_pmc_derived.json to say what bounds the kernel)."""
import csv, glob, json, sys, collections
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

# derived figures (gfx950: 8 XCDs, 256 CUs, 1024 SIMDs; FETCH_SIZE counts 64 B per 128-B request -> 2 x FETCH_SIZE + WRITE_SIZE KiB)
samples = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for k in agg:
    v = {c: agg[k][c] / cnt[k][c] for c in agg[k]}
    need = ["GRBM_GUI_ACTIVE", "GRBM_TA_BUSY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "TCP_TCC_READ_REQ_sum",
            "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "FETCH_SIZE", "WRITE_SIZE"]
    if any(c not in v for c in need):
        continue
    dur = v["GRBM_GUI_ACTIVE"] / 8
    print("# derived, per dispatch of", k)
    print("#   kernel duration            = GRBM_GUI_ACTIVE / 8 XCDs = %.3g cycles" % dur)
    print("#   texture addresser busy     = GRBM_TA_BUSY / GRBM_GUI_ACTIVE = %.1f %%;  L1 accesses %.2f per clock per CU" % (100 * v["GRBM_TA_BUSY"] / v["GRBM_GUI_ACTIVE"], v["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / dur))
    print("#   waves waiting              = SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.1f %%   instruction in flight %.1f %%   issue stalls %.1f %%" % (
        100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
    print("#   VALU busy per SIMD         = SQ_INSTS_VALU * 4 / 1024 SIMDs / duration = %.1f %%" % (100 * v["SQ_INSTS_VALU"] * 4 / 1024 / dur))
    print("#   L1 (TCP) hit rate          = %.1f %%   L2 (TCC) hit rate = %.1f %%   mean L1-miss latency = %.0f cycles   L1 stalled on pending misses = %.1f %%" % (
        100 * (1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]), 100 * v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]),
        v["TCP_TCC_READ_REQ_LATENCY_sum"] / v["TCP_TCC_READ_REQ_sum"], 100 * v["TCP_PENDING_STALL_CYCLES_sum"] / v["TCP_GATE_EN1_sum"]))
    if k == "pt_path" and len(sys.argv) > 4:
        hbm_bytes = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        derived = {"cycles": dur, "waves_waiting": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "instruction_in_flight": v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                   "issue_stalls": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                   # a wave64 vector instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md, wave scheduling)
                   "valu_busy": v["SQ_INSTS_VALU"] * 2 / 1024 / dur, "ta_busy": v["GRBM_TA_BUSY"] / v["GRBM_GUI_ACTIVE"],
                   "l1_accesses_per_clk_cu": v["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / dur, "l1_hit": 1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"],
                   "l2_hit": v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), "l1_miss_latency_cycles": v["TCP_TCC_READ_REQ_LATENCY_sum"] / v["TCP_TCC_READ_REQ_sum"],
                   "hbm_bytes": hbm_bytes, "hbm_bytes_per_cycle": hbm_bytes / dur, "l2_requests_per_cycle": v.get("TCC_REQ_sum", 0) / dur,
                   "samples": samples, "valu_per_sample": v["SQ_INSTS_VALU"] / samples if samples else None, "salu_per_sample": v.get("SQ_INSTS_SALU", 0) / samples if samples else None}
        try:
            data = json.load(open(sys.argv[3]))
        except (OSError, ValueError):
            data = {}
        data[sys.argv[4]] = derived
        json.dump(data, open(sys.argv[3], "w"), indent=1)
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    print("#   HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B) = %.3g B%s" % (hbm, (" = %.0f B per sample (%.3g samples)" % (hbm / samples, samples)) if samples else ""))
    if samples:
        print("#   instructions per sample    : VALU %.0f  SALU %.0f  VMEM read %.1f  VMEM write %.1f  LDS %.1f" % (
            v["SQ_INSTS_VALU"] / samples, v.get("SQ_INSTS_SALU", 0) / samples, v.get("SQ_INSTS_VMEM_RD", 0) / samples, v.get("SQ_INSTS_VMEM_WR", 0) / samples, v.get("SQ_INSTS_LDS", 0) / samples))
