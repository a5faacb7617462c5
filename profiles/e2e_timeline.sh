#!/bin/bash
# profiles/e2e_timeline.sh -- where the wall seconds of a one-shot `metaSNV.py DIR all_samples REF --threads 32` run on the benchmark's 160 BAMs go:
# interpreter / imports / HIP context by themselves, then the launcher with its own metrics (MSNV_METRICS) beside the wall clock.
cd "$(dirname "$0")/.."
t() { python3 - "$@" <<'PY'
import subprocess, sys, time
t0 = time.perf_counter(); subprocess.run(sys.argv[1:], capture_output=True); print("%.3f s : %s" % (time.perf_counter() - t0, " ".join(sys.argv[1:])))
PY
}
t python3 -c "pass"
t python3 -c "import numpy"
t python3 -c "from metasnv_amd import core"
t python3 -c "from metasnv_amd import core; core.Context(0).close()"
W=/tmp/e2e_tl; rm -rf $W; mkdir -p $W
python3 - <<PY
import os, sys, time
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
for rep in 1 2; do
  rm -rf $W/proj
  MSNV_METRICS=$W/m.jsonl t python3 metaSNV.py $W/proj $W/all_samples $W/ref.fa --threads 32
done
python3 - <<PY
import json
m = json.loads(open("$W/m.jsonl").read().strip().splitlines()[-1])
for k in ("open_dataset_s", "feed_s", "wait_for_context_s", "finalize_s", "coverage_run_s", "gather_coverage_s", "calling_pass_s", "gather_sites_s", "close_dataset_s"):
    print(k, m.get(k))
print("pileup", m.get("pileup")); print("cli_wall", m.get("cli_wall"), "(process_age_at_end_s: seconds since exec when the last output file was closed; the rest of the wall clock above is teardown)"); print("host_timers", m.get("host_timers")); print("pack", m.get("pack_on_device"))
PY
