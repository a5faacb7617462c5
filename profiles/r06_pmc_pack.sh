#!/bin/bash
# profiles/r06_pmc_pack.sh TAG -- instruction and traffic counters of the per-read stage's kernels (records resident -> calls, 2 builds), separate passes
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_pmcpack_$N -o p --output-format csv -- python3 profiles/pack_resident.py testdata 1 2 > /dev/null 2>&1
done
python3 - $TAG <<'PY'
import csv, glob, sys, re, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/%s_pmcpack_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(msnv_\w+)', r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in acc.items():
    g = lambda c: (sum(cs[c]) / max(1, len(set(range(len(cs[c]))))) ) if c in cs else 0.0
    n = len(cs.get("FETCH_SIZE", cs.get("SQ_WAVES", [1])))
    per = lambda c: sum(cs.get(c, [0.0])) / max(1, len(cs.get(c, [1])))
    rows.append((k, len(cs.get("SQ_WAVES", [])), per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_WAVES"), 2048.0 * per("FETCH_SIZE"), 1024.0 * per("WRITE_SIZE")))
rows.sort(key=lambda r: -r[2])
print("%-28s %5s %12s %12s %12s %10s %14s %14s" % ("kernel", "n", "VALU", "SALU", "LDS", "waves", "fetch_x2_B", "write_B"))
for r in rows[:24]:
    print("%-28s %5d %12.0f %12.0f %12.0f %10.0f %14.0f %14.0f" % r)
PY
