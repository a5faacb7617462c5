#!/bin/bash
# profiles/pmc_sq.sh TAG -- SQ busy / wait / issue counters of the dominant kernel (separate --pmc passes, kernel trace only)
set -u
TAG=${1:-sq}
OUT=gpurun_out/pmc_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > "$OUT/sq_counters.txt"
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN" "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES_EQ_64"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "narrow32" -d "$OUT/p$i" -o pmc --output-format csv -- $BENCH > "$OUT/p$i.log" 2>&1 || echo "pass $i ($C) failed" >> "$OUT/errors.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "narrow32" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print("%-28s avg/launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
cat "$OUT/errors.log" 2>/dev/null
