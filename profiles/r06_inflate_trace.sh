#!/bin/bash
# profiles/r06_inflate_trace.sh TAG -- the one-shot launcher on the benchmark's 160 BAMs under rocprofv3 --hip-trace --memory-copy-trace --kernel-trace
# (the program directly behind `--`): what fills the step around msnv_inflate_blocks (kernels 78 ms of a 183 ms step).
# Prints, for the window from the first H2D copy of compressed bytes to the end of the last CRC kernel: kernels, copies, and the HIP API
# calls of the calling thread by total time.
cd "$(dirname "$0")/.."
TAG=${1:-r06}; export TMPDIR=/tmp
W=/tmp/e2e_tl; rm -rf $W; mkdir -p $W gpurun_out
python3 - <<PY
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from concurrent.futures import ThreadPoolExecutor
from metasnv_amd import core
sp = core.synth_params(seed=1); syn = core.Synth(sp)
syn.write_fasta("$W/ref.fa")
paths = ["$W/s%04d.bam" % i for i in range(sp.n_samples)]
with ThreadPoolExecutor(32) as ex:
    list(ex.map(lambda i: core.write_bam(paths[i], syn.names, syn.lengths, syn.sample_records(i)), range(sp.n_samples)))
open("$W/all_samples", "w").write("\n".join(paths) + "\n")
PY
MSNV_FEED_TRACE=1 MSNV_METRICS=$W/m0.jsonl python3 metaSNV.py $W/proj0 $W/all_samples $W/ref.fa --threads 32 > gpurun_out/${TAG}_e2e_plain.log 2>&1
MSNV_EXIT=normal MSNV_FEED_TRACE=1 MSNV_METRICS=$W/m.jsonl rocprofv3 --hip-trace --memory-copy-trace --kernel-trace -d gpurun_out/${TAG}_e2e_trace -o t --output-format csv -- python3 metaSNV.py $W/proj $W/all_samples $W/ref.fa --threads 32 > gpurun_out/${TAG}_e2e_traced.log 2>&1
grep -i "feed\|inflate" gpurun_out/${TAG}_e2e_plain.log | head -20
python3 - gpurun_out/${TAG}_e2e_trace <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(d + "/**/*" + pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
k = rows("kernel_trace.csv"); c = rows("memory_copy_trace.csv"); a = rows("hip_api_trace.csv")
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in k]
infl = [x for x in ks if "inflate_blocks" in x[2] or "crc_blocks" in x[2]]
if not infl: print("no inflate kernels in the trace"); sys.exit(0)
t1 = max(x[1] for x in infl)
cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), int(r.get("Bytes", 0) or 0)) for r in c]
first_h2d = min([x[0] for x in cs if x[0] < t1 and x[3] > (1 << 20)] + [min(x[0] for x in infl)])
t0 = first_h2d
print("window: %.1f ms (first large copy -> end of the last inflate / CRC kernel)" % ((t1 - t0) / 1e6))
def busy(iv):
    iv = sorted((max(s, t0), min(e, t1)) for s, e in iv if e > t0 and s < t1)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot / 1e6
print("  kernels busy   %.1f ms" % busy([(s, e) for s, e, _ in ks]))
print("  copies busy    %.1f ms, %.2f GB" % (busy([(s, e) for s, e, _, _ in cs]), sum(b for s, e, _, b in cs if e > t0 and s < t1) / 1e9))
print("  either busy    %.1f ms" % busy([(s, e) for s, e, _ in ks] + [(s, e) for s, e, _, _ in cs]))
kk = collections.defaultdict(lambda: [0, 0])
for s, e, n in ks:
    if e > t0 and s < t1:
        import re
        m = re.search(r"(msnv_\w+|__amd_rocclr_\w+)", n); kk[m.group(1) if m else n[:40]][0] += 1; kk[m.group(1) if m else n[:40]][1] += e - s
for n, (cnt, ns) in sorted(kk.items(), key=lambda x: -x[1][1])[:8]: print("    kernel %-32s %5d calls %8.2f ms" % (n, cnt, ns / 1e6))
cc = collections.defaultdict(lambda: [0, 0, 0])
for s, e, dr, b in cs:
    if e > t0 and s < t1: cc[dr][0] += 1; cc[dr][1] += e - s; cc[dr][2] += b
for n, (cnt, ns, b) in sorted(cc.items(), key=lambda x: -x[1][1]): print("    copy   %-32s %5d calls %8.2f ms %8.3f GB" % (n, cnt, ns / 1e6, b / 1e9))
aa = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for r in a:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > t0 and s < t1: x = aa[r.get("Thread_Id", "?")][r["Function"]]; x[0] += 1; x[1] += min(e, t1) - max(s, t0)
for tid, fs in sorted(aa.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:3]:
    print("  thread %s: HIP API time inside the window %.1f ms" % (tid, sum(v[1] for v in fs.values()) / 1e6))
    for fn, (cnt, ns) in sorted(fs.items(), key=lambda x: -x[1][1])[:8]: print("      %-36s %6d calls %8.2f ms" % (fn, cnt, ns / 1e6))
PY
