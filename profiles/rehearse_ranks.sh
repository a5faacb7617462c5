#!/bin/bash
# profiles/rehearse_ranks.sh TAG [SCALE] -- the N-rank product path rehearsed on ONE GPU: bench.py --mode strong on the configs[3] shard
# generator, N = 1, 2, 4, 8 ranks over gloo (the ranks share the GPU; the record exchange and the gathers travel over gloo, so seconds of
# those are host-bound -- what the rehearsal shows is the LPT imbalance, the bytes every rank sends / rank 0 receives, and that the path
# runs at the node's real width).  One JSON line per N under gpurun_out/TAG_strong_gloo_nN.json.
TAG=${1:-r04}
SCALE=${2:-0.05}
mkdir -p gpurun_out
for N in 1 2 4 8; do
    timeout 900 python3 bench.py --gpus $N --dist-backend gloo --mode strong --workload config4shard --scale $SCALE --steps 5 --warmup 1 \
        --no-cpu-baseline --no-annotation --no-overlap-extra > gpurun_out/${TAG}_strong_gloo_n$N.json 2> gpurun_out/${TAG}_strong_gloo_n$N.err
    python3 - <<PY
import json
try:
    d = json.loads([x for x in open("gpurun_out/${TAG}_strong_gloo_n$N.json") if x.startswith("{")][-1])
    print("N=$N", "value", round(d["value"], 1), "imbalance", d["imbalance_max_over_mean"], "feed_s", [round(x, 2) for x in d["exchange"]["feed_s_per_rank"]],
          "gather_bytes", d["gather"]["bytes_received_by_rank0"], "bases_per_rank", d["config"]["pileup_bases_per_rank"])
except Exception as e:
    print("N=$N failed:", e)
PY
done
