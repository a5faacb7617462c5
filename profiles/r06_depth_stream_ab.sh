cd $GRAFT_REPO_ROOT
for V in two main two main; do
  E=""; [ $V = main ] && E="MSNV_DEPTH_STREAM=main"
  env $E python3 profiles/pack_resident.py testdata 1 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['reps'][1:]
print('$V', ' | '.join('pack %.3f fin %.3f tot %.3f emit %.3f' % (x['pack_wall_ms'], x['finalize_wall_ms'], x['pack_wall_ms']+x['finalize_wall_ms']+x['pileup_ms'], x['pack_kernel_ms']['emit_ms']) for x in r))"
done
