export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2
bash profiles/e2e_ab.sh > /dev/null 2>&1
for S in par serial par serial; do
if [ $S = serial ]; then export MSNV_UPLOAD_SERIAL=1; else unset MSNV_UPLOAD_SERIAL; fi
echo upload $S $(python3 profiles/inflate_prof.py /tmp/e2e_ab device device device 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print([ (round(r['add_sample_bams_s'],3), r['timers'].get('inflate_device_wall_s')) for r in d['device']])")
done
unset MSNV_UPLOAD_SERIAL
MSNV_FEED_TRACE=1 python3 metaSNV.py /tmp/e2e_ab/projZ /tmp/e2e_ab/all_samples /tmp/e2e_ab/ref.fa --threads 32 2>&1 | grep "^\[feed\]"
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-overlap-extra --no-strong-extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print(round(e['wall_s'],3), round(e['split_wall_s']['decode_and_pack'],3), round(e['split_wall_s']['process_start_hip_runtime_and_context'],3))"; done
