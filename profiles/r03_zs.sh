#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zs; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "testdata or whole_tile or gate_kernel or cigar or planes or many" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
bash profiles/abn.sh "tree aldirty0 spreadalu" 3 > $O/ab.txt 2>&1; cat $O/ab.txt
