#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
bash profiles/abn.sh "r03_hdr4 spreadalu" 4 > $O/ab_spread.txt 2>&1; cat $O/ab_spread.txt
for R in 1 2; do for P in 700 1000 1400; do echo -n "pieces $P "; MSNV_ITEM_PIECES=$P bash profiles/abn.sh "r03_hdr4" 1; done; done > $O/ab_items.txt 2>&1; cat $O/ab_items.txt
bash profiles/abn.sh "r02_head r03_hdr4" 2 --workload config4shard --scale 0.1 --mode weak > $O/ab_sparse.txt 2>&1; cat $O/ab_sparse.txt
