#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
for R in 1 2 3 4; do for P in 1000 1400 2000 2800; do echo -n "pieces $P "; MSNV_ITEM_PIECES=$P bash profiles/abn.sh "r03_hdr4" 1; done; done > $O/ab_items.txt 2>&1; cat $O/ab_items.txt
for R in 1 2; do for P in 1000 2000; do echo -n "config3 0.1 pieces $P "; MSNV_ITEM_PIECES=$P bash profiles/abn.sh "r03_hdr4" 1 --workload config3 --scale 0.1 --mode weak; done; done > $O/ab_items_c3.txt 2>&1; cat $O/ab_items_c3.txt
