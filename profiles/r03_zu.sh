#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zu; mkdir -p $O
( time timeout 1800 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -n 3; tail -n 4 $O/pytest_gpu.log | cut -c1-200
timeout 600 python3 tests/fuzz_parity.py 1500 9001 > $O/fuzz_default.txt 2>&1; tail -n 1 $O/fuzz_default.txt
MSNV_ALLELES=planes timeout 600 python3 tests/fuzz_parity.py 700 9002 > $O/fuzz_planes.txt 2>&1; tail -n 1 $O/fuzz_planes.txt
MSNV_FUSE=1 timeout 600 python3 tests/fuzz_parity.py 700 9003 > $O/fuzz_fuse.txt 2>&1; tail -n 1 $O/fuzz_fuse.txt
( time timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -n 3; tail -n 2 $O/bench_default.err
bash profiles/collect.sh r03zu > $O/collect.log 2>&1; tail -n 8 $O/collect.log; cat gpurun_out/prof_r03zu/errors.log 2>/dev/null
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03zu/bench_default.json").read().strip().splitlines()[-1]); print(round(d["value"],1), d["ms_per_step"], {k:d["roofline"][k] for k in ("frac","kernel_ms_avg","traffic","frac_resident")}, d.get("end_to_end",{}).get("wall_s"))
PY
