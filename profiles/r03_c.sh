#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
bash profiles/abn.sh "r03_align2 nospill notot noexc nomism noevwrite r03_start" 2 > $O/ab.txt 2>&1; cat $O/ab.txt
( time timeout 1500 python3 bench.py --workload config3 --scale 1.0 --mode weak --steps 5 --warmup 1 --no-cpu-baseline --no-annotation --no-overlap-extra > $O/bench_config3_full.json 2> $O/bench_config3_full.err ) 2>&1 | tail -4
tail -3 $O/bench_config3_full.err
python3 - <<'PY'
import json
for f in ("bench.json","bench_config3_full.json"):
    try:
        d=json.loads(open("gpurun_out/r03c/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d.get("host"), json.dumps(d.get("end_to_end"))[:1500])
    except Exception as e: print(f, "ERR", e)
PY
