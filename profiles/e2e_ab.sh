#!/bin/bash
# profiles/e2e_ab.sh "ENV=VAL ..." ["ENV=VAL ..." ...] -- the one-shot 160-BAM run of e2e_timeline.sh under several environment settings, 4 runs each
# (wall seconds of the process, and the launcher's own split), on BAMs written once.
cd "$(dirname "$0")/.."
W=/tmp/e2e_ab; rm -rf $W; mkdir -p $W
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
for setting in "$@"; do
  echo "== $setting"
  for rep in 1 2 3 4; do
    rm -rf $W/proj $W/m.jsonl
    env $setting MSNV_METRICS=$W/m.jsonl python3 - $W <<'PY'
import json, subprocess, sys, time
W = sys.argv[1]
t0 = time.perf_counter()
subprocess.run(["python3", "metaSNV.py", W + "/proj", W + "/all_samples", W + "/ref.fa", "--threads", "32"], capture_output=True)
wall = time.perf_counter() - t0
m = json.loads(open(W + "/m.jsonl").read().strip().splitlines()[-1])
c = m["cli_wall"]
print("wall %.3f | start %.2f feed %.3f ctx %.3f finalize %.3f pass %.3f gather %.3f files %.3f | in-process %.3f, ends at %.2f, teardown %.2f" % (
    wall, c["process_age_at_start_s"], m["feed_s"], m.get("wait_for_context_s", 0.0), m["finalize_s"], m["calling_pass_s"], m["gather_sites_s"],
    c["coverage_files_s"] + c["tables_and_splits_s"] + c["calls_text_s"], c["total_s"], c["process_age_at_end_s"], wall - c["process_age_at_end_s"]))
PY
  done
done
