#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_testdata and not config4" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for L in ab/r03_align2.so tree; do
  if [ $L = tree ]; then unset MSNV_LIBRARY; else export MSNV_LIBRARY=$PWD/$L; fi
  echo "== $L"; SIGMA=2 python3 profiles/phase_times.py; ERR=0.03 python3 profiles/phase_times.py; python3 profiles/phase_times.py
done > $O/phases.txt 2>&1; cat $O/phases.txt
