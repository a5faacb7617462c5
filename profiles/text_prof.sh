#!/bin/bash
# profiles/text_prof.sh -- instruction counters of msnv_parse_pileup_lines (one rocprofv3 --pmc pass of profiles/text_bench.py)
export TMPDIR=/tmp
OUT=gpurun_out/text_prof
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$OUT" -o t --output-format csv -- python3 profiles/text_bench.py ${1:-160} ${2:-10000} > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
agg = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "parse_pileup" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-20s %14.0f per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
tail -2 "$OUT/log"
