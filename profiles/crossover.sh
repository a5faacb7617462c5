#!/bin/bash
# profiles/crossover.sh -- whole-pass ms of the benchmark shape over the synthetic error rate with the allele events and with the allele planes
# (MSNV_ALLELES): where pack.cpp's switch (sampled mismatch rate >= 1.0 %) should sit.  Prints: rate  form  kernel_ms  pass_ms  Gbases/s  sampled ppm
for E in 0.001 0.003 0.006 0.01 0.015 0.03; do for F in events planes; do
  MSNV_ALLELES=$F timeout 300 python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra --steps 10 --warmup 2 --error-rate $E 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E $F', round(d['roofline']['kernel_ms_avg'],4), round(d['kernel_ms']['pipeline_total'],4), round(d['value'],1), d.get('dataset',{}).get('sampled_mismatch_ppm'))"
done; done
