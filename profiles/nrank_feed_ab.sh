#!/bin/bash
# profiles/nrank_feed_ab.sh -- the launcher on the benchmark's 160 BAMs in a ONE-rank nccl process group (MSNV_DIST_FORCE=1; planning budget 200 MB so that most rounds
# stream; --threads 7: with 8 or more the one-rank launcher brings its context up lazily and feeds through the host): BAMs inflated + dealt on the device / inflated by host threads and dealt on the device / inflated and dealt by host threads.  Prints the feed seconds.
cd "$(dirname "$0")/.."
W=/tmp/e2e_ab
[ -f $W/all_samples ] || bash profiles/e2e_ab.sh > /dev/null
for setting in "MSNV_INFLATE=device" "MSNV_INFLATE=host" "MSNV_INFLATE=host MSNV_DEAL=host"; do
  for rep in 1 2; do
    rm -rf $W/projn $W/mn.jsonl
    env $setting MSNV_PLAN_MB=${PLAN_MB:-200} MSNV_DIST_FORCE=1 MSNV_METRICS=$W/mn.jsonl python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 \
        metaSNV.py $W/projn $W/all_samples $W/ref.fa --threads 7 > /dev/null 2>&1
    python3 - "$setting" $W/mn.jsonl <<'PY'
import json, sys
m = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-34s feed %.3f s (decode %.3f, deal+exchange+pack %.3f) finalize %.3f | dealt on device %.2f GB, inflated on device %.2f GB, plan rounds %s" % (
    sys.argv[1], m["feed_s"], m.get("decode_s", 0), m.get("deliver_s", 0), m["finalize_s"], m.get("records_dealt_on_device_bytes", 0) / 1e9, m.get("bams_inflated_on_device_bytes", 0) / 1e9, m.get("plan_rounds")))
PY
  done
done
