export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "whole_tile or sparse or fused or merged or indiv or config4" 2>&1 | tail -2
MSNV_FUSE=1 timeout 900 python3 tests/fuzz_parity.py 500 103 2>&1 | tail -1
bash profiles/ktrace_shape.sh sparse_500x5x_20ofN 2>&1 | grep -E "gather_scatter|gate_staged|narrow32|pass"
MSNV_PAIR_ONE=0 bash profiles/ktrace_shape.sh sparse_500x5x_20ofN 2>&1 | grep -E "gather_scatter|gate_staged|pass"
