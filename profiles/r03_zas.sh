#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zas; mkdir -p $O
MSNV_LAYOUT=dense timeout 300 python3 tests/fuzz_parity.py 500 9402 > $O/fuzz_dense.txt 2>&1; tail -n 1 $O/fuzz_dense.txt
timeout 300 python3 tests/fuzz_parity.py 700 9403 > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt
timeout 600 python3 -m pytest tests/test_gpu_stress.py tests/test_gpu_guard.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 1 $O/pytest.log
for RL in 50 100; do timeout 300 python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --steps 10 --warmup 2 --read-len $RL 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$RL\", {k:round(v,3) for k,v in d[\"kernel_ms\"].items()}, round(d[\"value\"],1))"; done
