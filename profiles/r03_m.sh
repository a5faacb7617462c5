#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
bash profiles/abn.sh "branchy tree" 4 > $O/ab.txt 2>&1; cat $O/ab.txt
python3 profiles/phase_times.py > $O/phases.txt 2>&1; SIGMA=2 python3 profiles/phase_times.py >> $O/phases.txt 2>&1; ERR=0.03 python3 profiles/phase_times.py >> $O/phases.txt 2>&1; cat $O/phases.txt
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not full_testdata and not config4" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
