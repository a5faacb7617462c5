#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zao; mkdir -p $O
for R in 1 2 3; do for V in tree scatu8 scatu2; do
if [ $V = tree ]; then unset MSNV_LIBRARY; else export MSNV_LIBRARY=$PWD/ab/$V.so; fi
echo "$V $(PASSES=30 python3 profiles/phase_times.py 2>&1 | cut -c1-110)"
done; done > $O/scatter_unroll.txt 2>&1; cat $O/scatter_unroll.txt
unset MSNV_LIBRARY
( time timeout 600 python3 bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --samples 24 --contig-len 60000 --no-cpu-baseline --no-annotation --no-overlap-extra --strong-extra-shape 8,40000 > $O/bench2.json 2> $O/bench2.err ) 2>&1 | tail -n 3; tail -n 2 $O/bench2.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r03zao/bench2.json').read().strip().splitlines()[-1]); print(d['n_gpus'], round(d['value'],1), json.dumps(d.get('strong_scaling'))[:600])"
( time MSNV_STRONG_EXTRA_LIMIT_S=1 timeout 600 python3 bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --samples 24 --contig-len 60000 --no-cpu-baseline --no-annotation --no-overlap-extra --strong-extra-shape 8,40000 > $O/bench3.json 2> $O/bench3.err ) 2>&1 | tail -n 3; python3 -c "
import json; d=json.loads(open('gpurun_out/r03zao/bench3.json').read().strip().splitlines()[-1]); print('watchdog:', d['n_gpus'], round(d['value'],1), json.dumps(d.get('strong_scaling'))[:200])"
