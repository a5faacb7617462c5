#!/bin/bash
# profiles/pmc_shape.sh SHAPE -- counter passes (each in a run of its own, --kernel-trace only) of profiles/shape_sweep.py SHAPE;
# prints the per-launch averages of the pileup kernels
export TMPDIR=/tmp
OUT=gpurun_out/pmc_shape
rm -rf "$OUT"; mkdir -p "$OUT"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE "TCC_ATOMIC_sum TCC_EA_ATOMIC_sum"; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C -d "$OUT/$N" -o pmc --output-format csv -- python3 profiles/shape_sweep.py "$@" > "$OUT/$N.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pileup" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-32s %-24s %14.0f per launch (%d launches)" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
