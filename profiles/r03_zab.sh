#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zab; mkdir -p $O
( time timeout 1500 python3 bench.py --workload config3 --scale 1.0 --mode weak --steps 5 --warmup 1 --no-cpu-baseline --no-annotation --no-overlap-extra > $O/bench_config3_full.json 2> $O/bench_config3_full.err ) 2>&1 | tail -4
tail -3 $O/bench_config3_full.err
timeout 900 python3 -m pytest tests/test_gpu_full_config3.py -q -m gpu > $O/pytest_full3.log 2>&1; tail -n 3 $O/pytest_full3.log | cut -c1-200
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03zab/bench_config3_full.json").read().strip().splitlines()[-1]); print(round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["host"])
PY
