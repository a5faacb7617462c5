#!/bin/bash
# profiles/perturb.sh V1 V2 ... -- bench.py's pileup kernel time under MSNV_PERTURB=V.  Needs the EXPERIMENT build of profiles/perturb_build.py
# (python3 profiles/perturb_build.py /tmp/msnv_perturb; export MSNV_LIBRARY=/tmp/msnv_perturb/libmsnv.so): the shipped library ignores the word.
for V in "$@"; do
  MSNV_PERTURB=$V python3 bench.py --no-cpu-baseline --no-annotation --no-overlap-extra --no-strong-extra --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('perturb $V', round(d['roofline']['kernel_ms_avg'],4), round(d['kernel_ms']['pipeline_total'],4), round(d['value'],1))"
done
