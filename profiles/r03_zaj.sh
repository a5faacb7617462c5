#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zaj; mkdir -p $O
MSNV_LAYOUT=dense timeout 600 python3 profiles/stress_case.py 200 run 2>&1 | tail -n 2 | cut -c1-200
MSNV_LAYOUT=dense timeout 600 python3 profiles/stress_case.py 100 overlap 2>&1 | tail -n 2 | cut -c1-200
MSNV_LAYOUT=pieces MSNV_ALLELES=events timeout 600 python3 profiles/stress_case.py 150 run 2>&1 | tail -n 2 | cut -c1-200
bash profiles/abn.sh "tree" 2 > $O/ab.txt 2>&1; cat $O/ab.txt
