#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zaa; mkdir -p $O
FUZZ_MANY=overlap timeout 900 python3 tests/fuzz_parity.py 2500 9101 > $O/overlap.txt 2>&1; echo "overlap: $(tail -n 1 $O/overlap.txt)"; grep MISMATCH $O/overlap.txt | cut -c1-200
FUZZ_MANY=many timeout 900 python3 tests/fuzz_parity.py 2500 9102 > $O/many.txt 2>&1; echo "many: $(tail -n 1 $O/many.txt)"; grep MISMATCH $O/many.txt | cut -c1-200
