#!/bin/bash
# profiles/r06_inflate_pmc.sh -- instruction counters of msnv_inflate_blocks / msnv_crc_blocks on the benchmark's 160 BAMs (the launcher under rocprofv3 --pmc)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=/tmp/e2e_tl
[ -f $W/all_samples ] || bash profiles/r06_inflate_ab.sh > /dev/null 2>&1
rm -rf $W/proj
MSNV_EXIT=normal rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM -d gpurun_out/r06_inflate_pmc -o p --output-format csv -- python3 metaSNV.py $W/proj $W/all_samples $W/ref.fa --threads 32 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for f in glob.glob("gpurun_out/r06_inflate_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate_blocks" in r["Kernel_Name"] or "crc_blocks" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print(k, dict(v))
PY
