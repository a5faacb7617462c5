#!/bin/bash
# profiles/abn.sh "NAME1 NAME2 ..." [ROUNDS] [extra bench args] -- same-box A/B of several builds (ab/NAME.so; "tree" = the in-tree
# build): round-robin bench.py runs, one line per run: name, pileup kernel ms, pipeline ms, Gbases/s, roofline fraction
NAMES=$1; N=${2:-3}; shift; shift
for i in $(seq $N); do
  for V in $NAMES; do
    if [ $V = tree ]; then unset MSNV_LIBRARY; else export MSNV_LIBRARY=$PWD/ab/$V.so; fi
    python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra --steps 30 "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['roofline']['kernel_ms_avg'],4), round(d['kernel_ms']['pipeline_total'],4), round(d['value'],1), round(d['roofline']['frac'],4), d['config']['called_SNPs_lines_per_rank'])
except Exception as e: print('$V', 'ERR', e)"
  done
done
