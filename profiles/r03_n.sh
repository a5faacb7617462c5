#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
MSNV_ALLELES=planes timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not allele_bookkeeping and not noisy_reads_reserve and not event_list_grows" > $O/pytest_planes.log 2>&1; echo "rc $?" >> $O/pytest_planes.log; tail -4 $O/pytest_planes.log
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "allele or many_sites or sweep or noisy" > $O/pytest_a.log 2>&1; echo "rc $?" >> $O/pytest_a.log; tail -3 $O/pytest_a.log
python3 profiles/phase_times.py > $O/phases.txt 2>&1; ERR=0.03 python3 profiles/phase_times.py >> $O/phases.txt 2>&1; ERR=0.1 python3 profiles/phase_times.py >> $O/phases.txt 2>&1; cat $O/phases.txt
bash profiles/noise_sweep.sh > $O/noise.txt 2>&1; cat $O/noise.txt
