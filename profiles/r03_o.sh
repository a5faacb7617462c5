#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03o/bench.json").read().strip().splitlines()[-1]); print(round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["kernel_ms"], d["host"]); print(json.dumps(d.get("end_to_end"))[:900])
PY
bash profiles/abn.sh "r02_head tree" 2 --workload config4shard --scale 0.1 --mode weak > $O/ab_sparse.txt 2>&1; cat $O/ab_sparse.txt
SIGMA=2 python3 profiles/phase_times.py > $O/phases.txt 2>&1; cat $O/phases.txt
