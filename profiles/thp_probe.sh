#!/bin/bash
# profiles/thp_probe.sh -- are the staged record streams of a 160-BAM job backed by huge pages, and what does giving them back cost?
cd "$(dirname "$0")/.."
W=/tmp/e2e_ab
[ -f $W/all_samples ] || bash profiles/e2e_ab.sh >/dev/null
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag /sys/kernel/mm/transparent_hugepage/khugepaged/defrag 2>/dev/null
grep -E "thp_fault_alloc|thp_fault_fallback |compact_stall" /proc/vmstat
python3 - $W <<'PY'
import sys, time
sys.path.insert(0, ".")
from metasnv_amd import core
W = sys.argv[1]
bams = open(W + "/all_samples").read().split()
def roll():
    d = {}
    for l in open("/proc/self/smaps_rollup"):
        w = l.split()
        if w[0] in ("Rss:", "AnonHugePages:", "Anonymous:"): d[w[0]] = int(w[1]) // 1024
    return d
ds = core.Dataset.from_files(None, bams[0], W + "/ref.fa")
t0 = time.perf_counter(); ds.stage_sample_bams(bams, 32); t1 = time.perf_counter()
print("staged in %.3f s; MB:" % (t1 - t0), roll())
t0 = time.perf_counter(); ds.close(); t1 = time.perf_counter()
print("closed (staged streams freed, no device involved) in %.3f s; MB:" % (t1 - t0), roll())
PY
grep -E "thp_fault_alloc|thp_fault_fallback |compact_stall" /proc/vmstat
