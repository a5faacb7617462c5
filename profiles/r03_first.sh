#!/bin/bash
# profiles/r03_first.sh -- first GPU call of round 3: what the box has, the GPU suite, the bench line, the strong mode at N = 1
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
{ nproc; free -g; ulimit -a | head -20; python3 tests/reftools.py; python3 -c "import sys; sys.path.insert(0,'tests'); import reftools, json; print(json.dumps(reftools.report()))"; rocm-smi --showmeminfo vram 2>/dev/null | head; df -h /tmp /dev/shm . | cat; } > $O/box.txt 2>&1
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python3 bench.py --workload config3 --scale 0.1 --steps 10 --warmup 2 > $O/bench_strong_c3_0p1.json 2> $O/bench_strong.err
tail -5 $O/pytest.log; cat $O/box.txt | head -30; python3 - <<'PY'
import json
for f in ("bench.json","bench_strong_c3_0p1.json"):
    try:
        d=json.loads(open("gpurun_out/r03a/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"])
    except Exception as e: print(f, "ERR", e)
PY
