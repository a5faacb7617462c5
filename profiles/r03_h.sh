#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
bash profiles/collect.sh r03 > $O/collect.log 2>&1; tail -12 $O/collect.log
timeout 600 python3 bench.py --workload config4shard --scale 0.1 --mode weak --steps 10 --warmup 2 --no-cpu-baseline --no-annotation --no-overlap-extra > $O/bench_config4shard_0p1.json 2> $O/c4.err
timeout 900 python3 bench.py --workload config3 --scale 0.25 --steps 5 --warmup 1 > $O/bench_strong_config3_quarter_n1.json 2> $O/c3.err
python3 - <<'PY'
import json
for f in ("bench_config4shard_0p1.json","bench_strong_config3_quarter_n1.json"):
    try:
        d=json.loads(open("gpurun_out/r03h/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d.get("kernel_ms"))
    except Exception as e: print(f, "ERR", e)
PY
python3 profiles/shape_sweep.py many_shallow_1600x1 tiny_contigs_3000x300 one_sample_1600x few_deep_16x100 sparse_500x5x_20ofN > $O/shapes.txt 2>&1; cat $O/shapes.txt
