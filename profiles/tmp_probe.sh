export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_devpack.py tests/test_gpu_inflate.py -x -q 2>&1 | tail -6
python3 profiles/pack_resident.py testdata 1 3 | python3 -c "
import json,sys; d=json.load(sys.stdin)
for r in d['reps']: print(r)"
