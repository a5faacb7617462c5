#!/bin/bash
# profiles/r06_feed.sh -- the N-rank feed's device route with and without the round ahead on a second context (one-rank nccl launcher, 160 BAMs, --threads 7)
cd "$(dirname "$0")/.."
W=/tmp/e2e_ab
[ -f $W/all_samples ] || bash profiles/e2e_ab.sh > /dev/null
for setting in "MSNV_FEED_OVERLAP=1" "MSNV_FEED_OVERLAP=0" "MSNV_FEED_OVERLAP=1" "MSNV_FEED_OVERLAP=0"; do
    rm -rf $W/projn $W/mn.jsonl
    env $setting MSNV_INFLATE=device MSNV_PLAN_MB=${PLAN_MB:-200} MSNV_DIST_FORCE=1 MSNV_METRICS=$W/mn.jsonl python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 \
        metaSNV.py $W/projn $W/all_samples $W/ref.fa --threads 7 > $W/feed.log 2>&1 || tail -20 $W/feed.log
    python3 - "$setting" $W/mn.jsonl <<'PY'
import json, sys
m = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-24s feed %.3f s (decode %.3f, deal+exchange+pack %.3f) finalize %.3f | second context %s, plan rounds %s, called %s" % (
    sys.argv[1], m["feed_s"], m.get("decode_s", 0), m.get("deliver_s", 0), m["finalize_s"], m.get("decode_on_second_context"), m.get("plan_rounds"), m.get("pileup", {}).get("n_called_pop")))
PY
done
