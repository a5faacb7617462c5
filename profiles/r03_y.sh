#!/bin/bash
# one-bit quality column: parity first (suite + fuzz in three modes), then the bench line
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -n 3; tail -n 8 $O/pytest_gpu.log
timeout 600 python3 tests/fuzz_parity.py 1500 777 > $O/fuzz_default.txt 2>&1; tail -n 1 $O/fuzz_default.txt
MSNV_ALLELES=planes timeout 600 python3 tests/fuzz_parity.py 800 778 > $O/fuzz_planes.txt 2>&1; tail -n 1 $O/fuzz_planes.txt
MSNV_FUSE=1 timeout 600 python3 tests/fuzz_parity.py 800 779 > $O/fuzz_fuse.txt 2>&1; tail -n 1 $O/fuzz_fuse.txt
MSNV_LAYOUT=dense timeout 600 python3 tests/fuzz_parity.py 600 780 > $O/fuzz_dense.txt 2>&1; tail -n 1 $O/fuzz_dense.txt
( time timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -n 3; tail -n 2 $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03y/bench_default.json").read().strip().splitlines()[-1]); print(round(d["value"],1), d["ms_per_step"], d["roofline"], d.get("host"), d.get("end_to_end"))
PY
