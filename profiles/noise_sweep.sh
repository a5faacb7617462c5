#!/bin/bash
# profiles/noise_sweep.sh -- pileup kernel and whole-pass time over the synthetic sequencing error rate (DESIGN.md section 4,
# "Noisy reads"): prints  error_rate  kernel_ms  {pileup, pipeline_total}  Gbases/s
for E in 0.001 0.01 0.03 0.1; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra --steps 10 --warmup 2 --error-rate $E 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', round(d['roofline']['kernel_ms_avg'],4), {k:round(v,3) for k,v in d['kernel_ms'].items()}, round(d['value'],1))"
done
