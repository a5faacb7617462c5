#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for L in ab/r03_hdr4.so tree; do
  if [ $L = tree ]; then unset MSNV_LIBRARY; else export MSNV_LIBRARY=$PWD/$L; fi
  echo "== $L"; python3 profiles/phase_times.py; SIGMA=2 python3 profiles/phase_times.py; ERR=0.03 python3 profiles/phase_times.py
done > $O/phases.txt 2>&1; cat $O/phases.txt
unset MSNV_LIBRARY
bash profiles/abn.sh "r03_hdr4 tree" 3 > $O/ab.txt 2>&1; cat $O/ab.txt
