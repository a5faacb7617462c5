#!/bin/bash
# profiles/final_evidence.sh TAG -- the round's closing evidence in ONE gpurun call: counter passes + kernel trace (collect_short.sh), then the bench
# line that reads them, the sparse-shard and configs[2] lines, a fuzz sweep, the e2e A/B.  Everything lands under gpurun_out/ (copied to profiles/ by hand).
TAG=${1:-r04}
mkdir -p gpurun_out
bash profiles/collect_short.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
cp gpurun_out/${TAG}_pmc.json profiles/${TAG}_pmc.json 2>/dev/null     # (bench.py quotes the traffic of THIS build)
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --workload config4shard --scale 0.1 --no-cpu-baseline --no-annotation > gpurun_out/${TAG}_bench_config4shard_0p1.json 2>/dev/null
python3 bench.py --workload config3 --scale 1.0 --no-cpu-baseline --no-annotation --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_config3_full.json 2>/dev/null
bash profiles/fuzz.sh ${TAG}fin 800 9091 FUZZ_PACK=device > /dev/null 2>&1
bash profiles/e2e_ab.sh "A=1" > gpurun_out/${TAG}_e2e.txt 2>&1
for f in bench bench_config4shard_0p1 bench_config3_full; do python3 - gpurun_out/${TAG}_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["kernel_ms_avg"], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -2 gpurun_out/${TAG}fin_fuzz.txt; tail -5 gpurun_out/${TAG}_e2e.txt
