#!/bin/bash
# profiles/ab.sh BASE_SO [ROUNDS] -- same-box A/B of two builds of libmsnv.so (box-to-box variance is +-5 %):
# alternates bench.py between BASE_SO (MSNV_LIBRARY) and the in-tree build and prints the pileup kernel time of each run.
BASE=$1; N=${2:-3}
for i in $(seq $N); do
  for V in base new; do
    if [ $V = base ]; then export MSNV_LIBRARY=$PWD/$BASE; else unset MSNV_LIBRARY; fi
    python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra --steps 30 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['roofline']['kernel_ms_avg'],4), round(d['kernel_ms']['pipeline_total'],4), round(d['value'],1))"
  done
done
