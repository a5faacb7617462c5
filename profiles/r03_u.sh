#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
WORKLOAD=config4shard SCALE=1.0 PASSES=5 python3 profiles/phase_times.py > $O/phases_c4_full.txt 2>&1; cat $O/phases_c4_full.txt
WORKLOAD=config4shard SCALE=0.3 PASSES=5 python3 profiles/phase_times.py > $O/phases_c4_0p3.txt 2>&1; cat $O/phases_c4_0p3.txt
