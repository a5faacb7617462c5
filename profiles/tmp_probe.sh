export TMPDIR=/tmp
bash profiles/ktrace_shape.sh sparse_500x5x_20ofN 2>&1 | grep -E "gather_scatter|gate_staged|pass"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "whole_tile or sparse or fused or merged or indiv" 2>&1 | tail -2
MSNV_FUSE=1 timeout 900 python3 tests/fuzz_parity.py 400 83 2>&1 | tail -1
