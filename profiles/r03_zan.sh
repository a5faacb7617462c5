#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zan; mkdir -p $O
timeout 1500 python3 tests/fuzz_parity.py 7000 9301 > $O/fuzz_default.txt 2>&1; tail -n 1 $O/fuzz_default.txt; grep -c MISMATCH $O/fuzz_default.txt
FUZZ_MANY=overlap timeout 600 python3 tests/fuzz_parity.py 2000 9302 > $O/fuzz_overlap.txt 2>&1; tail -n 1 $O/fuzz_overlap.txt
timeout 600 python3 tests/fuzz_mpileup_text.py 1500 9303 > $O/fuzz_text.txt 2>&1; tail -n 1 $O/fuzz_text.txt
