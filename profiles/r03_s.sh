#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
( time python3 tests/fuzz_parity.py 300 90210 ) > $O/fuzz_default.txt 2>&1; tail -4 $O/fuzz_default.txt
( time MSNV_ALLELES=planes python3 tests/fuzz_parity.py 300 777 ) > $O/fuzz_planes.txt 2>&1; tail -4 $O/fuzz_planes.txt
( time MSNV_GATHER_SPLIT=1 MSNV_MERGE_ALWAYS=1 python3 tests/fuzz_parity.py 150 4242 ) > $O/fuzz_merge.txt 2>&1; tail -4 $O/fuzz_merge.txt
