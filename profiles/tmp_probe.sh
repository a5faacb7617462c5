export TMPDIR=/tmp
( time RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 MSNV_DIST_FORCE=1 timeout 900 python3 bench.py --workload config4shard --mode strong --steps 10 --warmup 1 > gpurun_out/r05_strong_n1.json 2> gpurun_out/r05_strong_n1.err ) 2>&1 | tail -3
tail -2 gpurun_out/r05_strong_n1.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05_strong_n1.json"))
print(d["value"], d["ms_per_step"], d["scaling"], d["config"]["pileup_bases_total"], d["roofline"]["frac"], d["exchange"]["backend"], d["exchange"]["feed_s_per_rank"], d["host"]["wall_s_whole_run"])
PY
