#!/bin/bash
# profiles/ktrace_shape.sh SHAPE... -- rocprofv3 kernel trace of profiles/shape_sweep.py for the named shapes; avg time per kernel
export TMPDIR=/tmp
OUT=gpurun_out/kts
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT" -o t --output-format csv -- python3 profiles/shape_sweep.py "$@" > "$OUT/log" 2>&1
tail -n 3 "$OUT/log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 4: print("%-44s calls %3d avg %9.1f us" % (k[-44:], len(v), sum(v) / len(v) / 1e3))
PY
