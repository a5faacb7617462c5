#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
bash profiles/abn.sh "r03_start align4 align2 nospill notot" 3 > $O/ab.txt 2>&1
MSNV_LIBRARY=$PWD/ab/align2.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_align2.log 2>&1; echo "rc $?" >> $O/pytest_align2.log
tail -4 $O/pytest.log; cat $O/ab.txt; tail -4 $O/pytest_align2.log
