#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -n 3; tail -n 5 $O/pytest_gpu.log
( time timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -n 3; tail -n 2 $O/bench_default.err
( time timeout 1500 python3 bench.py --workload config4shard --scale 1.0 --mode weak --steps 5 --warmup 1 --no-cpu-baseline --no-annotation --no-overlap-extra > $O/bench_config4shard_full.json 2> $O/c4.err ) 2>&1 | tail -n 3; tail -n 3 $O/c4.err
python3 - <<'PY'
import json
for f in ("bench_default.json", "bench_config4shard_full.json"):
    try:
        d=json.loads(open("gpurun_out/r03w/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_avg"), d.get("host"))
    except Exception as e: print(f, "ERR", e)
PY
