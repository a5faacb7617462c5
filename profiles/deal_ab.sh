#!/bin/bash
# profiles/deal_ab.sh [SCALE] -- the N-rank feed with the records dealt by kernels (MSNV_DEAL=device) and by host threads (MSNV_DEAL=host): bench.py --mode strong
# on the configs[2] generator at SCALE (default 0.25) in a ONE-rank nccl process group (MSNV_DIST_FORCE=1): seconds of decode (the generator) and of deal + exchange + pack
SCALE=${1:-0.25}
for D in device host; do
  MSNV_DEAL=$D MSNV_DIST_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --workload config3 --scale $SCALE \
      --mode strong --no-cpu-baseline --no-annotation --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
h=d['host']; print('$D', 'value', round(d['value'],1), 'decode_s', round(h['feed_decode_s_rank0'],2), 'deal+exchange+pack_s', round(h['feed_deal_exchange_pack_s_rank0'],2), 'feed_s', round(d['exchange']['feed_s_per_rank'][0],2), 'pack upload_s', round(h['pack_on_device_rank0']['upload_wall_s'],2), 'backend', d['exchange']['backend'])"
done
