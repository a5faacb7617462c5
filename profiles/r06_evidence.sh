#!/bin/bash
# profiles/r06_evidence.sh TAG -- round 6's closing evidence in ONE gpurun call: kernel trace + counter passes of the bench command
# (collect_short.sh -> TAG_kernel_stats.csv, TAG_pmc.json with the pack kernels' HBM traffic per build), the bench line that reads them,
# the sparse-shard and configs[2] lines, the records -> calls split with trace marks, a fuzz sweep.  Everything lands under gpurun_out/.
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
bash profiles/collect_short.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
cp gpurun_out/${TAG}_pmc.json profiles/${TAG}_pmc.json 2>/dev/null     # (bench.py quotes the traffic of THIS build)
cp gpurun_out/prof_$TAG/trace/trace_kernel_stats.csv gpurun_out/${TAG}_rocprofv3_kernel_stats.csv 2>/dev/null
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --workload config4shard --scale 0.1 --no-cpu-baseline --no-annotation > gpurun_out/${TAG}_bench_config4shard_0p1.json 2>/dev/null
MSNV_FINALIZE_TRACE=1 python3 profiles/pack_resident.py testdata 1 4 > gpurun_out/${TAG}_records_to_calls.json 2> gpurun_out/${TAG}_records_to_calls_trace.txt
bash profiles/fuzz.sh ${TAG}fin 3000 9393 > /dev/null 2>&1
bash profiles/fuzz.sh ${TAG}sparse 1500 4343 MSNV_FUSE=1 > /dev/null 2>&1
bash profiles/fuzz.sh ${TAG}careful 800 4444 MSNV_FRONT=careful > /dev/null 2>&1      # the careful route of the per-read stage (the quick one is the default of the sweeps above)      # every tile a whole-tile work item: record lists, msnv_gate_staged, both forms of the merged gather
python3 bench.py --workload config4shard --scale 1.0 --no-cpu-baseline --no-annotation --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_config4shard_full.json 2>/dev/null
python3 bench.py --workload config3 --scale 1.0 --no-cpu-baseline --no-annotation --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_config3_full.json 2>/dev/null
for f in bench bench_config4shard_0p1 bench_config4shard_full bench_config3_full; do python3 - gpurun_out/${TAG}_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["kernel_ms_avg"], 4))
    r = d.get("roofline_from_records")
    if r: print("  records -> calls:", {k: round(r[k], 4) for k in ("frac", "total_ms", "pack_wall_ms", "finalize_ms", "pileup_kernel_ms")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
# the sparse shard's kernels (msnv_pileup_tiles_lean since round 6) with MSNV_LEAN=0 beside them: statistics of a kernel trace, instruction counters in a pass of their own
SB="python3 bench.py --workload config4shard --scale 0.1 --no-cpu-baseline --no-annotation --no-overlap-extra --steps 5 --warmup 2"
for V in lean ordinary; do
  if [ $V = ordinary ]; then export MSNV_LEAN=0; else unset MSNV_LEAN; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_sparse_$V -o t --output-format csv -- $SB > /dev/null 2>&1
  cp $(find gpurun_out/${TAG}_sparse_$V -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_sparse_${V}_kernel_stats.csv 2>/dev/null
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/${TAG}_sparse_pmc_$V -o p --output-format csv -- $SB > /dev/null 2>&1
done
unset MSNV_LEAN
python3 - $TAG > gpurun_out/${TAG}_sparse_counters.txt <<'PY'
import csv, glob, re, collections, sys
tag = sys.argv[1]
for v in ("lean", "ordinary"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/%s_sparse_pmc_%s/**/*counter_collection.csv" % (tag, v), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(msnv_\w+)', r["Kernel_Name"])
            if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
        per = lambda c: sum(cs.get(c, [0.0])) / max(1, len(cs.get(c, [1])))
        print("%-9s %-34s launches %3d  VALU %12.0f  SALU %12.0f  LDS %11.0f  waves %9.0f  VALU/wave %7.1f" % (v, k, len(cs["SQ_WAVES"]), per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_WAVES"), per("SQ_INSTS_VALU") / max(1.0, per("SQ_WAVES"))))
PY
head -n 6 gpurun_out/${TAG}_sparse_counters.txt
bash profiles/fuzz.sh ${TAG}token 1500 7373 > /dev/null 2>&1      # (the sweep shortens snpCall's token in 3 of 7 cases: msnv_cap_reads / msnv_token_cut against the oracle)
tail -n 2 gpurun_out/${TAG}fin_fuzz.txt; tail -n 2 gpurun_out/${TAG}sparse_fuzz.txt; tail -n 2 gpurun_out/${TAG}careful_fuzz.txt; tail -n 2 gpurun_out/${TAG}token_fuzz.txt
bash profiles/r06_base.sh ${TAG} > gpurun_out/${TAG}_r2c_kernels.txt 2>&1; tail -5 gpurun_out/${TAG}_r2c_kernels.txt
