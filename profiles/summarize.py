#!/usr/bin/env python3
"""profiles/summarize.py DIR TAG -- condense rocprofv3 CSV output (collect.sh) into two small files:
gpurun_out/TAG_kernel_stats.csv (per-kernel calls / total / average ns) and gpurun_out/TAG_pmc.json
(per-kernel average counter values per launch + the HBM traffic of the dominant kernel, with the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md applied: KiB units, reads doubled)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
out_dir = os.path.dirname(os.path.abspath(d))

# ---- kernel trace -> stats
rows = defaultdict(list)
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns\n")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%d,%.1f,%d,%d\n' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))

# ---- counters
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> counter -> dispatch -> value
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
pmc = {}
for k, cs in acc.items():
    pmc[k] = {c: {"avg_per_launch": sum(v.values()) / len(v), "launches": len(v)} for c, v in cs.items()}
res = {"counters": pmc}
dom = [k for k in pmc if "pileup_tiles_narrow32" in k]
if dom and "FETCH_SIZE" in pmc[dom[0]] and "WRITE_SIZE" in pmc[dom[0]]:
    fe = pmc[dom[0]]["FETCH_SIZE"]["avg_per_launch"] * 1024.0
    wr = pmc[dom[0]]["WRITE_SIZE"]["avg_per_launch"] * 1024.0
    res["kernel"] = "msnv_pileup_tiles_narrow32"
    res["hbm_traffic"] = {"fetch_bytes_raw": fe, "fetch_bytes_corrected_x2": 2 * fe, "write_bytes": wr,
                          "total_bytes_per_launch": 2 * fe + wr,
                          "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream "
                                  "(MI355X_MICROARCH.md section HBM), so reads are doubled; separate --pmc passes with --kernel-trace only"}
# ---- the per-read stage (csrc/devpack.hip): HBM traffic of ONE build of the dataset from resident records = sum over its kernels of
# (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches per build (the bench builds the dataset several times: launches / builds)
PACK = ("msnv_scan_sub", "msnv_scan_seams", "msnv_scan_write", "msnv_scan_segments", "msnv_compact_offsets", "msnv_measure_reads", "msnv_pile_gather", "msnv_depth", "msnv_run_table",
        "msnv_group_pre", "msnv_group_table", "msnv_sample_bases", "msnv_emit_block", "msnv_emit_tail", "msnv_acc_fold", "msnv_acc_init", "msnv_sub_bounds", "msnv_sample_layout",
        "msnv_scan_check", "msnv_scan_fix", "msnv_tables_from_measure", "msnv_group_firsts", "rocprim")      # (substring match: msnv_scan_sub also takes msnv_scan_sub2, msnv_depth takes msnv_depth2 ...)
# (the number of builds of the profiled bench command: one launch of msnv_sample_layout per round, one round per build on the benchmark shape --
# three builds until round 5, seven since round 6)
builds = float(max([cs["FETCH_SIZE"]["launches"] for k, cs in pmc.items() if "msnv_sample_layout" in k and "FETCH_SIZE" in cs] or [3]))
pk = {}
for k, cs in pmc.items():
    if not any(x in k for x in PACK) or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    name = next((x for x in PACK if x in k), k[:40])
    if name == "rocprim":
        name = "rocprim scans"
    n = cs["FETCH_SIZE"]["launches"] / builds
    e = pk.setdefault(name, {"fetch_x2_bytes_per_build": 0.0, "write_bytes_per_build": 0.0})
    e["fetch_x2_bytes_per_build"] += 2 * 1024.0 * cs["FETCH_SIZE"]["avg_per_launch"] * n
    e["write_bytes_per_build"] += 1024.0 * cs["WRITE_SIZE"]["avg_per_launch"] * n
if pk:
    tot = sum(v["fetch_x2_bytes_per_build"] + v["write_bytes_per_build"] for v in pk.values())
    res["pack_traffic"] = {"kernels": pk, "total_bytes_per_build": tot,
                           "builds": builds, "note": "bench.py builds the dataset several times from records resident in HBM (`builds`); rocPRIM scans of any other stage (finalize) are included; "
                                   "compare with roofline_from_records.record_bytes_resident of the bench line"}
json.dump(res, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1)
print(json.dumps(res.get("hbm_traffic", {}), indent=1))
