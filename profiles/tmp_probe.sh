export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_inflate.py tests/test_inflate.py -x -q 2>&1 | tail -2
bash profiles/e2e_ab.sh > /dev/null 2>&1
for V in none iw5; do
  if [ $V = none ]; then unset MSNV_LIBRARY; else export MSNV_LIBRARY=$PWD/ab/$V.so; fi
  echo $V
  rocprofv3 --kernel-trace --stats -d gpurun_out/${V}_prof -o p --output-format csv -- python3 profiles/inflate_prof.py /tmp/e2e_ab device > gpurun_out/${V}_prof.log 2>&1
  python3 - gpurun_out/${V}_prof <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:2]:
        print("%-60s calls %6s total %.3f ms avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  python3 profiles/inflate_prof.py /tmp/e2e_ab device device device 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print([ (round(r['add_sample_bams_s'],3), r['timers'].get('inflate_device_wall_s')) for r in d['device']])"
done
