#!/bin/bash
# profiles/modes_probe.sh -- bench.py weak vs strong on the small shape tests/test_gpu_bench_modes.py uses (value, ms per step, kernel ms)
A="--gpus 1 --workload testdata --samples 48 --contig-len 100000 --warmup 3 --no-cpu-baseline --no-annotation --no-overlap-extra"
for S in 10 200; do for M in weak strong; do
  python3 bench.py $A --steps $S --mode $M 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M steps $S', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms_avg'],4))"
done; done
MSNV_PACK=host python3 bench.py $A --steps 10 --mode weak 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('weak hostpack steps 10', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms_avg'],4))"
