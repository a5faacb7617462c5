#!/bin/bash
# profiles/pack_pmc.sh -- counters of the device pack's kernels on the benchmark shape (profiles/pack_prof.py under rocprofv3 --pmc, one pass per counter set)
export TMPDIR=/tmp
for C in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_ATOMIC_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40); OUT=gpurun_out/packpmc_$N; rm -rf $OUT
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT -o p --output-format csv -- python3 profiles/pack_prof.py > $OUT.log 2>&1 || echo "pass $C failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/packpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if any(x in k for x in ("emit_pieces", "emit_headers", "scan_segments", "measure_reads", "msnv_depth", "tile_keys", "gather_pieces")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
