#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zz; mkdir -p $O
timeout 300 python3 tests/fuzz_parity.py 230 9001 > $O/cur.txt 2>&1; echo "current: $(tail -n 1 $O/cur.txt)"; grep -c MISMATCH $O/cur.txt
timeout 300 python3 tests/fuzz_parity.py 230 9001 > $O/cur2.txt 2>&1; echo "current again: $(tail -n 1 $O/cur2.txt)"
MSNV_LIBRARY=$PWD/ab/pre_occ8.so timeout 300 python3 tests/fuzz_parity.py 230 9001 > $O/pre.txt 2>&1; echo "pre_occ8: $(tail -n 1 $O/pre.txt)"
MSNV_LIBRARY=$PWD/ab/prev.so timeout 300 python3 tests/fuzz_parity.py 230 9001 > $O/prev.txt 2>&1; echo "prev (ae53a48): $(tail -n 1 $O/prev.txt)"
