#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
bash profiles/abn.sh "r02_head hdr8 tree" 4 > $O/ab_main.txt 2>&1; cat $O/ab_main.txt
bash profiles/abn.sh "r02_head tree" 2 --workload config4shard --scale 0.1 --mode weak > $O/ab_sparse.txt 2>&1; cat $O/ab_sparse.txt
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
