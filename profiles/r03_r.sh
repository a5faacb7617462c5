#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
for R in 1 2; do for B in 32 16 8 4; do echo "scatter_blocks $B"; MSNV_SCATTER_BLOCKS=$B python3 profiles/phase_times.py | cut -c1-110; done; done > $O/tail_tune2.txt 2>&1; cat $O/tail_tune2.txt
for B in 32 8; do echo "sparse scatter_blocks $B"; MSNV_SCATTER_BLOCKS=$B WORKLOAD=config4shard SCALE=0.1 python3 profiles/phase_times.py | cut -c1-130; done >> $O/tail_tune2.txt 2>&1; tail -4 $O/tail_tune2.txt
