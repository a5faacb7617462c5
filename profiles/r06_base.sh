#!/bin/bash
# profiles/r06_base.sh TAG -- records -> calls on the benchmark shape: trace marks of one run, rocprofv3 kernel statistics of another
TAG=${1:-r06a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MSNV_FINALIZE_TRACE=1 python3 profiles/pack_resident.py testdata 1 4 > gpurun_out/${TAG}_r2c.json 2> gpurun_out/${TAG}_r2c_trace.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o t --output-format csv -- python3 profiles/pack_resident.py testdata 1 4 > /dev/null 2>&1
F=$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1)
cp $F gpurun_out/${TAG}_r2c_kernel_stats.csv
python3 profiles/kstats.py $F 20
python3 - gpurun_out/${TAG}_r2c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for r in d["reps"][1:]:
    print("pack %.3f finalize %.3f pileup %.3f total %.3f | %s" % (r["pack_wall_ms"], r["finalize_wall_ms"], r["pileup_ms"], r["pack_wall_ms"] + r["finalize_wall_ms"] + r["pileup_ms"], r["pack_kernel_ms"]))
PY
