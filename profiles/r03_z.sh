#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
bash profiles/abn.sh "tree nomism noexc nopass noseq" 2 > $O/ab_ablations.txt 2>&1; cat $O/ab_ablations.txt
bash profiles/collect.sh r03z > $O/collect.log 2>&1; tail -n 12 $O/collect.log
