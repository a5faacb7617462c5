#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zr; mkdir -p $O
( time timeout 1800 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -n 3; tail -n 6 $O/pytest_gpu.log | cut -c1-200
MSNV_DEEP=wide timeout 600 python3 tests/fuzz_parity.py 800 4343 > $O/fuzz_wide.txt 2>&1; tail -n 1 $O/fuzz_wide.txt
bash profiles/collect.sh r03zr > $O/collect.log 2>&1; tail -n 12 $O/collect.log; cat gpurun_out/prof_r03zr/errors.log 2>/dev/null
