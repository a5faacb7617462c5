#!/bin/bash
# profiles/r06_variants.sh -- records -> calls, 12 builds per variant of one box: median / min of pack + finalize + pileup kernel
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for V in "default" "MSNV_DEPTH_STREAM=main" "MSNV_COV_THREAD=1" "MSNV_DEPTH_STREAM=main MSNV_COV_THREAD=1"; do
  E=""; [ "$V" = default ] || E="$V"
  env $E python3 profiles/pack_resident.py testdata 1 13 2>/dev/null | python3 -c "
import json,sys,statistics
d=json.loads(sys.stdin.read())
r=d['reps'][1:]
t=[x['pack_wall_ms']+x['finalize_wall_ms']+x['pileup_ms'] for x in r]
p=[x['pack_wall_ms'] for x in r]; f=[x['finalize_wall_ms'] for x in r]
print('%-44s total median %.3f min %.3f | pack median %.3f finalize median %.3f' % ('$V', statistics.median(t), min(t), statistics.median(p), statistics.median(f)))"
done; done
