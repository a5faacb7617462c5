#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zt; mkdir -p $O
bash profiles/abn.sh "tree occ8" 3 > $O/ab.txt 2>&1; cat $O/ab.txt
MSNV_LIBRARY=$PWD/ab/occ8.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "testdata or whole_tile or gate_kernel or cigar or planes or many" > $O/pytest_occ8.log 2>&1; tail -n 1 $O/pytest_occ8.log
for W in config4shard; do
for V in tree occ8; do
if [ $V = tree ]; then unset MSNV_LIBRARY; else export MSNV_LIBRARY=$PWD/ab/$V.so; fi
WORKLOAD=$W SCALE=0.1 PASSES=10 python3 profiles/phase_times.py 2>&1 | cut -c1-120
done; done
