#!/bin/bash
# per-tile fallback of the whole-tile work items: parity first, then the configs[3] shard at scale
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_tile or config3_shape or sparse or gate_kernel" > $O/pytest_fused.log 2>&1; tail -n 6 $O/pytest_fused.log
MSNV_FUSE=1 timeout 900 python3 tests/fuzz_parity.py 1500 4242 > $O/fuzz_fuse1.txt 2>&1; tail -n 1 $O/fuzz_fuse1.txt
WORKLOAD=config4shard SCALE=0.1 PASSES=5 timeout 600 python3 profiles/phase_times.py > $O/phases_c4_0p1.txt 2>&1; cat $O/phases_c4_0p1.txt
WORKLOAD=config4shard SCALE=0.3 PASSES=5 timeout 900 python3 profiles/phase_times.py > $O/phases_c4_0p3.txt 2>&1; cat $O/phases_c4_0p3.txt
WORKLOAD=config4shard SCALE=1.0 PASSES=5 timeout 1500 python3 profiles/phase_times.py > $O/phases_c4_full.txt 2>&1; cat $O/phases_c4_full.txt
