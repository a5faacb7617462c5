#!/bin/bash
# profiles/fuzz.sh TAG CASES SEED [ENV=VAL ...] -- one randomised parity sweep (tests/fuzz_parity.py: HIP path vs oracle) on the GPU box,
# its log under gpurun_out/TAG_fuzz.txt.  Extra arguments are environment settings of the sweep (FUZZ_PACK=host, FUZZ_MANY=overlap,
# MSNV_GUARD_ALLOC=1 ...).  The one-shot launch scripts of round 3 (profiles/r03_*.sh: a test run, an A/B list, a fuzz count each) were
# calls of this script, ab.sh / abn.sh and collect_short.sh with different arguments; their outputs are the r03* files of this directory.
TAG=${1:-fuzz}; N=${2:-1000}; SEED=${3:-1}; shift 3 || true
mkdir -p gpurun_out
env "$@" timeout 3000 python3 tests/fuzz_parity.py "$N" "$SEED" > "gpurun_out/${TAG}_fuzz.txt" 2>&1
tail -3 "gpurun_out/${TAG}_fuzz.txt"
