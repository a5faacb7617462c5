export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "whole_tile or sparse or fused or merged" 2>&1 | tail -2
for i in 1 2 3; do
  for V in base new; do
    if [ $V = base ]; then export MSNV_LIBRARY=$PWD/ab/noplane.so; else unset MSNV_LIBRARY; fi
    echo $V $(python3 profiles/shape_sweep.py sparse_500x5x_20ofN baseline | cut -c60-110)
  done
done
