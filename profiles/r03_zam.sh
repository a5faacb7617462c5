#!/bin/bash
export TMPDIR=/tmp
for E in 0.003 0.006 0.01 0.015; do
for M in events planes; do
  MSNV_ALLELES=$M timeout 300 python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --steps 10 --warmup 2 --error-rate $E 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E $M', round(d['roofline']['kernel_ms_avg'],4), {k:round(v,3) for k,v in d['kernel_ms'].items()}, round(d['value'],1))"
done; done
