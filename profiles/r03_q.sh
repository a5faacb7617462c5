#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
for S in 0 1 2 3; do echo "skip $S"; MSNV_TAIL_SKIP=$S python3 profiles/phase_times.py; done > $O/tail_ablation.txt 2>&1; cat $O/tail_ablation.txt
for S in 0 1 2 3; do echo "r02 skip $S"; MSNV_LIBRARY=$PWD/ab/r02_head.so MSNV_TAIL_SKIP=$S python3 profiles/phase_times.py; done > $O/tail_r02.txt 2>&1; cat $O/tail_r02.txt
