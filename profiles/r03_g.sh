#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "allele or many_sites or sweep or noisy" > $O/pytest_a.log 2>&1; echo "rc $?" >> $O/pytest_a.log; tail -5 $O/pytest_a.log
MSNV_ALLELES=planes timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "not allele_bookkeeping" > $O/pytest_planes.log 2>&1; echo "rc $?" >> $O/pytest_planes.log; tail -8 $O/pytest_planes.log
for E in 0.001 0.01 0.03; do ERR=$E python3 profiles/phase_times.py; done > $O/phases.txt 2>&1; cat $O/phases.txt
for E in 0.01 0.03; do MSNV_ALLELES=events ERR=$E python3 profiles/phase_times.py; done > $O/phases_events.txt 2>&1; cat $O/phases_events.txt
bash profiles/noise_sweep.sh > $O/noise.txt 2>&1; cat $O/noise.txt
