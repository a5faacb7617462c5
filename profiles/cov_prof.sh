#!/bin/bash
# profiles/cov_prof.sh TAG -- kernel trace + counter passes of the qaCompute coverage kernel alone (profiles/cov_time.py)
set -u
TAG=${1:-cov}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python3 profiles/cov_time.py > "$OUT/trace.log" 2>&1
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCC_EA_ATOMIC_sum TCC_ATOMIC_sum" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$N" -o pmc --output-format csv -- python3 profiles/cov_time.py > "$OUT/pmc_$N.log" 2>&1 || echo "pmc pass $C failed" >> "$OUT/errors.log"
done
python3 profiles/summarize.py "$OUT" "$TAG"
