#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zak; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -n 3; tail -n 4 $O/pytest_gpu.log | cut -c1-200
timeout 900 python3 tests/fuzz_parity.py 3000 9201 > $O/fuzz_default.txt 2>&1; tail -n 1 $O/fuzz_default.txt
MSNV_ALLELES=planes timeout 600 python3 tests/fuzz_parity.py 1000 9202 > $O/fuzz_planes.txt 2>&1; tail -n 1 $O/fuzz_planes.txt
MSNV_FUSE=1 timeout 600 python3 tests/fuzz_parity.py 1000 9203 > $O/fuzz_fuse.txt 2>&1; tail -n 1 $O/fuzz_fuse.txt
MSNV_DEEP=wide timeout 600 python3 tests/fuzz_parity.py 800 9204 > $O/fuzz_wide.txt 2>&1; tail -n 1 $O/fuzz_wide.txt
