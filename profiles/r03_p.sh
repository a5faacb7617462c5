#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
bash profiles/collect.sh r03 > $O/collect.log 2>&1; tail -8 $O/collect.log
bash profiles/cov_prof.sh r03cov > $O/cov_prof.log 2>&1; tail -3 $O/cov_prof.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03p/bench.json").read().strip().splitlines()[-1]); print(round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["kernel_ms"], d["coverage_pass"]["kernel_ms"])
PY
