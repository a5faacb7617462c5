cd $GRAFT_REPO_ROOT
for SB in 2048 4096 6144 8192; do
  MSNV_SCAN_SUB=$SB python3 profiles/pack_resident.py testdata 1 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['reps'][1:]
print('sub $SB', ' | '.join('pack %.3f fin %.3f tot %.3f scan %.3f' % (x['pack_wall_ms'], x['finalize_wall_ms'], x['pack_wall_ms']+x['finalize_wall_ms']+x['pileup_ms'], x['pack_kernel_ms']['scan_ms']) for x in r))"
done
