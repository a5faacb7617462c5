#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zar; mkdir -p $O
bash profiles/abn.sh "tree pair64" 3 > $O/ab.txt 2>&1; cat $O/ab.txt
MSNV_LIBRARY=$PWD/ab/pair64.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "testdata or whole_tile or cigar or planes or many or iupac or edge" > $O/pytest.log 2>&1; tail -n 1 $O/pytest.log
MSNV_LIBRARY=$PWD/ab/pair64.so timeout 300 python3 tests/fuzz_parity.py 600 9401 > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt
