#!/bin/bash
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -8
echo "--- under load"
python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra --steps 6000 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 3.5
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms_avg'], d['ms_per_step'])"
