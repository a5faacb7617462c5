export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_devpack.py tests/test_gpu_pack.py -x -q 2>&1 | tail -3
timeout 900 python3 tests/fuzz_parity.py 400 17 2>&1 | tail -2
python3 profiles/pack_resident.py testdata 1 4 | python3 -c "
import json,sys; d=json.load(sys.stdin)
for r in d['reps']: print({k:r[k] for k in ('pack_wall_ms','finalize_wall_ms','pack_kernel_ms')})"
MSNV_FINALIZE_TRACE=1 python3 profiles/pack_resident.py testdata 1 2 2>&1 | grep -v '^{' | tail -60
