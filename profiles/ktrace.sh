#!/bin/bash
# profiles/ktrace.sh TAG [python args...] -- rocprofv3 kernel trace of one bench.py run; prints avg ns per kernel (same-box comparisons of two builds: MSNV_LIBRARY)
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/kt_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT" -o t --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra "$@" > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 8: print("%-44s calls %3d avg %9.1f us" % (k[-44:], len(v), sum(v) / len(v) / 1e3))
PY
