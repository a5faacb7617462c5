"""profiles/pack_resident.py [workload] [scale] [reps] -- records -> calls with the RAW RECORDS RESIDENT IN HBM when the clock starts:
the workload's record streams are made on the host, uploaded (hipMalloc + hipMemcpy, outside the timed region), the host copies are
dropped and the device left idle for a moment (freeing uploaded host memory stalls the next GPU operation for milliseconds); then, per
repetition: a fresh dataset, msnv_dataset_add_sample_records_device (per-read stage as kernels) + finalize + one pass, wall-clocked and
split by the library's HIP-event timers.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from metasnv_amd import core  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "testdata"
    argv = ["--workload", wl] + (["--scale", sys.argv[2]] if len(sys.argv) > 2 else [])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    sys.argv = [sys.argv[0]] + argv
    a = bench.parse()
    kw, label = bench.workload_params(a)
    sp = core.synth_params(**kw)
    syn = core.Synth(sp)
    ctx = core.Context(0)
    hip = C.CDLL("libamdhip64.so")
    recs = [syn.sample_records(i) for i in range(sp.n_samples)]
    offs, sizes, o = [], [], 0
    for r in recs:
        offs.append(o); sizes.append(int(r.size)); o += (int(r.size) + 15) & ~15
    cap = o + 256
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), C.c_size_t(cap)) == 0
    for r, off in zip(recs, offs):
        if r.size:
            assert hip.hipMemcpy(C.c_void_p(buf.value + off), C.c_void_p(r.ctypes.data), C.c_size_t(r.size), 1) == 0
    del recs
    hip.hipDeviceSynchronize()
    time.sleep(0.5)
    out = {"workload": label, "record_bytes": int(sum(sizes)), "reps": []}
    for rep in range(reps):
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        t0 = time.perf_counter()
        ds.add_samples_records_resident(buf.value, cap, offs, sizes)
        t1 = time.perf_counter()
        info = ds.finalize()
        t2 = time.perf_counter()
        st = ds.run()
        t3 = time.perf_counter()
        ps = ds.pack_stats()
        out["reps"].append({"pack_wall_ms": 1e3 * (t1 - t0), "finalize_wall_ms": 1e3 * (t2 - t1), "first_pass_wall_ms": 1e3 * (t3 - t2), "pack_kernel_ms": {k: round(v, 3) for k, v in ps.items() if k.endswith("_ms")},
                            "upload_wall_ms": 1e3 * ps["upload_wall_s"], "pileup_bases": info["n_pileup_bases"], "pass_ms": st["ms_total"], "pileup_ms": st.get("ms_pileup"), "called": st["n_called_pop"]})
        ds.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
