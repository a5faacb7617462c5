export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2
bash profiles/inflate_prof.sh r05ring 2>&1 | tail -12
for V in ring0 ring2048 ring8192; do
  export MSNV_LIBRARY=$PWD/ab/$V.so
  echo $V
  rocprofv3 --kernel-trace --stats -d gpurun_out/${V}_prof -o p --output-format csv -- python3 profiles/inflate_prof.py /tmp/e2e_ab device > gpurun_out/${V}_prof.log 2>&1
  python3 - gpurun_out/${V}_prof <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:3]:
        print("%-60s calls %6s total %.3f ms avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
done
