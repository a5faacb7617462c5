export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_devpack.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
MSNV_FINALIZE_TRACE=1 python3 profiles/pack_resident.py testdata 1 3 2>gpurun_out/tr.txt | python3 -c "
import json,sys; d=json.load(sys.stdin)
for r in d['reps']: print({k:r[k] for k in ('pack_wall_ms','finalize_wall_ms','pack_kernel_ms')})"
grep "pack: ref\|reference\|pairs / bases" gpurun_out/tr.txt | tail -4
