#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
python3 tests/fuzz_parity.py 3000 31337 > $O/fuzz_default.txt 2>&1; tail -n 1 $O/fuzz_default.txt
MSNV_ALLELES=planes python3 tests/fuzz_parity.py 3000 27182 > $O/fuzz_planes.txt 2>&1; tail -n 1 $O/fuzz_planes.txt
( time timeout 1500 python3 bench.py --workload config4shard --scale 1.0 --mode weak --steps 5 --warmup 1 --no-cpu-baseline --no-annotation --no-overlap-extra > $O/bench_config4shard_full.json 2> $O/c4.err ) 2>&1 | tail -n 4; tail -n 3 $O/c4.err
( time timeout 900 python3 bench.py --gpus 2 --dist-backend gloo --workload config3 --scale 0.1 --steps 5 --warmup 1 > $O/bench_strong_2ranks_gloo.json 2> $O/s2.err ) 2>&1 | tail -n 4; tail -n 3 $O/s2.err
python3 - <<'PY'
import json
for f in ("bench_config4shard_full.json","bench_strong_2ranks_gloo.json"):
    try:
        d=json.loads(open("gpurun_out/r03t/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d.get("host"), d.get("imbalance_max_over_mean"), json.dumps(d.get("gather"))[:400], json.dumps(d.get("exchange"))[:500], d["config"].get("positions_per_gpu"), d["config"].get("pileup_bases_per_gpu"))
    except Exception as e: print(f, "ERR", e)
PY
