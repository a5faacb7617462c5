cd $GRAFT_REPO_ROOT
W=/tmp/e2e_ab
[ -f $W/all_samples ] || bash profiles/e2e_ab.sh > /dev/null
for B in 7 16 32 64; do for O in 1 0; do
    rm -rf $W/projn $W/mn.jsonl
    env MSNV_FEED_BATCH=$B MSNV_FEED_OVERLAP=$O MSNV_INFLATE=device MSNV_PLAN_MB=200 MSNV_DIST_FORCE=1 MSNV_METRICS=$W/mn.jsonl python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 \
        metaSNV.py $W/projn $W/all_samples $W/ref.fa --threads 7 > $W/feed.log 2>&1 || tail -20 $W/feed.log
    python3 - "batch $B overlap $O" $W/mn.jsonl <<'PY'
import json, sys
m = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-24s feed %.3f s (decode %.3f, deliver %.3f) | plan rounds %s called %s" % (sys.argv[1], m["feed_s"], m.get("decode_s", 0), m.get("deliver_s", 0), m.get("plan_rounds"), m.get("pileup", {}).get("n_called_pop")))
PY
done; done
