#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e/bench.json").read().strip().splitlines()[-1]); print(round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["kernel_ms"], d["host"]); print(json.dumps(d.get("end_to_end"),indent=0)); print(d.get("host_decode"))
PY
bash profiles/noise_sweep.sh > $O/noise.txt 2>&1; cat $O/noise.txt
for E in 0.001 0.03; do ERR=$E python3 profiles/phase_times.py; done > $O/phases.txt 2>&1; cat $O/phases.txt
