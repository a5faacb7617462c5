#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
bash profiles/abn.sh "tree noqual" 3 > $O/ab_noqual.txt 2>&1; cat $O/ab_noqual.txt
