#!/bin/bash
# profiles/collect_short.sh TAG -- the kernel trace and the three counter passes the bench line's roofline block quotes (collect.sh runs ten)
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- $BENCH > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
    N=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$N" -o pmc --output-format csv -- $BENCH > "$OUT/pmc_$N.log" 2>&1 || echo "pmc pass $C failed" >> "$OUT/errors.log"
done
python3 profiles/summarize.py "$OUT" "$TAG"
