#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zae; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "token_limit or cigar_ops or wide_kernel or mpileup" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log | cut -c1-200
timeout 300 python3 -m pytest tests -q -m gpu -x -k "mpileup_text or pileup_qualities or token" > $O/pytest2.log 2>&1; tail -n 2 $O/pytest2.log | cut -c1-200
MSNV_DEEP=wide timeout 600 python3 tests/fuzz_parity.py 800 4343 > $O/fuzz_wide.txt 2>&1; tail -n 1 $O/fuzz_wide.txt
timeout 600 python3 tests/fuzz_parity.py 800 4343 > $O/fuzz_split_same_seed.txt 2>&1; tail -n 1 $O/fuzz_split_same_seed.txt
