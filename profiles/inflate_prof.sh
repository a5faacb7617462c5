#!/bin/bash
# profiles/inflate_prof.sh TAG -- device inflate vs host inflate of the benchmark's 160 BAMs (profiles/inflate_prof.py), then the kernel table of one device run
cd "$(dirname "$0")/.."
TAG=${1:-inflate}; export TMPDIR=/tmp
[ -f /tmp/e2e_ab/all_samples ] || bash profiles/e2e_ab.sh > /dev/null
mkdir -p gpurun_out
python3 profiles/inflate_prof.py /tmp/e2e_ab > gpurun_out/${TAG}_inflate_ab.json 2> gpurun_out/${TAG}_inflate_ab.err
tail -c 3000 gpurun_out/${TAG}_inflate_ab.json
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_inflate_prof -o p --output-format csv -- python3 profiles/inflate_prof.py /tmp/e2e_ab device > gpurun_out/${TAG}_inflate_prof.log 2>&1
python3 - gpurun_out/${TAG}_inflate_prof <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        print("%-60s calls %6s total %.3f ms avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
