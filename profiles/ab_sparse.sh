#!/bin/bash
# profiles/ab_sparse.sh BASE_SO [ROUNDS] -- profiles/ab.sh on the sparse shard (configs[3] shard generator at 0.1)
BASE=$1; N=${2:-3}
for i in $(seq $N); do
  for V in base new; do
    if [ $V = base ]; then export MSNV_LIBRARY=$PWD/$BASE; else unset MSNV_LIBRARY; fi
    python3 bench.py --workload config4shard --scale 0.1 --no-cpu-baseline --no-annotation --no-overlap-extra --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['roofline']['kernel_ms_avg'],4), round(d['ms_per_step'],4), round(d['value'],1))"
  done
done
