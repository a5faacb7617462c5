export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_devpack.py tests/test_gpu_guard.py -x -q 2>&1 | tail -2
timeout 900 python3 tests/fuzz_parity.py 400 97 2>&1 | tail -1
python3 profiles/pack_resident.py testdata 1 4 | python3 -c "
import json,sys; d=json.load(sys.stdin)
for r in d['reps']: print({k:r[k] for k in ('pack_wall_ms','finalize_wall_ms')})"
