#!/bin/bash
# profiles/perturb_pmc.sh V1 V2 ... -- VALU / SALU / LDS instruction counts of the dominant kernel under MSNV_PERTURB=V (experiment build of
# profiles/perturb_build.py, selected with MSNV_LIBRARY)
export TMPDIR=/tmp
for V in "$@"; do
  OUT=gpurun_out/pp_$V; mkdir -p $OUT
  export MSNV_PERTURB=$V
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --kernel-include-regex "narrow32" -d $OUT -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra > $OUT.log 2>&1
  python3 - $OUT $V <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "narrow32" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("perturb", sys.argv[2], " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
