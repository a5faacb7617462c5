#!/bin/bash
# final measurements of round 3 on one box: bench line, rocprofv3 stats + counters, noise sweep, sigma-2 cohort, sparse shard
export TMPDIR=/tmp
O=gpurun_out/r03zal; mkdir -p $O
( time timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -n 3; tail -n 2 $O/bench_default.err
bash profiles/collect.sh r03zal > $O/collect.log 2>&1; tail -n 8 $O/collect.log; cat gpurun_out/prof_r03zal/errors.log 2>/dev/null
bash profiles/noise_sweep.sh > $O/noise_sweep.txt 2>&1; cat $O/noise_sweep.txt
SIGMA=2 python3 profiles/phase_times.py > $O/phases_sigma2.txt 2>&1; cat $O/phases_sigma2.txt | cut -c1-200
python3 profiles/phase_times.py > $O/phases_default.txt 2>&1; cat $O/phases_default.txt | cut -c1-200
WORKLOAD=config4shard SCALE=0.1 PASSES=10 python3 profiles/phase_times.py > $O/phases_c4_0p1.txt 2>&1; cat $O/phases_c4_0p1.txt | cut -c1-200
( time timeout 1500 python3 bench.py --workload config4shard --scale 1.0 --mode weak --steps 5 --warmup 1 --no-cpu-baseline --no-annotation --no-overlap-extra > $O/bench_config4shard_full.json 2> $O/c4.err ) 2>&1 | tail -n 3; tail -n 2 $O/c4.err
python3 - <<'PY'
import json
for f in ("bench_default.json", "bench_config4shard_full.json"):
    try:
        d=json.loads(open("gpurun_out/r03zal/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_avg"), d.get("host"), (d.get("end_to_end") or {}).get("wall_s"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
