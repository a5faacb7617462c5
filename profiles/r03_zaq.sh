#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03zaq; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -n 3; tail -n 4 $O/pytest_gpu.log | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
( time timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -n 3; tail -n 2 $O/bench_default.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r03zaq/bench_default.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], (d.get('end_to_end') or {}).get('wall_s'))"
