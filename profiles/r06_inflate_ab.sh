#!/bin/bash
# profiles/r06_inflate_ab.sh -- the device inflate of the benchmark's 160 BAMs with the register allocation aimed at 5 (in-tree) / 6 / 7 / 8 wavefronts per SIMD
# (ab/iwN.so: profiles/build_variant.sh iwN -DMSNV_INFLATE_WAVES=N): the launcher's "upload + inflate + check" step and the whole feed
cd $GRAFT_REPO_ROOT
W=/tmp/e2e_tl
[ -f $W/all_samples ] || { mkdir -p $W; python3 - <<PY
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from concurrent.futures import ThreadPoolExecutor
from metasnv_amd import core
sp = core.synth_params(seed=1); syn = core.Synth(sp)
syn.write_fasta("$W/ref.fa")
paths = ["$W/s%04d.bam" % i for i in range(sp.n_samples)]
with ThreadPoolExecutor(32) as ex:
    list(ex.map(lambda i: core.write_bam(paths[i], syn.names, syn.lengths, syn.sample_records(i)), range(sp.n_samples)))
open("$W/all_samples", "w").write("\n".join(paths) + "\n")
PY
}
for rep in 1 2 3; do for V in tree iw6 iw7 iw8; do
  L=""; [ $V = tree ] || L="MSNV_LIBRARY=$PWD/ab/$V.so"
  rm -rf $W/proj
  env $L MSNV_FEED_TRACE=1 MSNV_METRICS=$W/m.jsonl python3 metaSNV.py $W/proj $W/all_samples $W/ref.fa --threads 32 2>&1 | grep "upload + inflate" | awk -v v=$V '{print v, $0}'
done; done
