#!/bin/bash
# profiles/collect.sh TAG -- run on the GPU box (through gpurun): kernel trace + PMC passes of bench.py,
# written under gpurun_out/prof_TAG/, summarised into gpurun_out/TAG_kernel_stats.csv and gpurun_out/TAG_pmc.json.
# Every profiler run sits under `timeout 300`: a GPU fault under the profiler leaves rocprofv3 waiting forever (round 3: 40 minutes of box time).
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- $BENCH > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA_ATOMIC_sum TCC_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$N" -o pmc --output-format csv -- $BENCH > "$OUT/pmc_$N.log" 2>&1 || echo "pmc pass $C failed" >> "$OUT/errors.log"
done
python3 profiles/summarize.py "$OUT" "$TAG"
