"""profiles/inflate_prof.py W -- the 160 BAMs of the benchmark shape (written by profiles/e2e_ab.sh into W) through msnv_dataset_add_sample_bams with the BGZF blocks
inflated on the device (MSNV_INFLATE=device) and on the host (MSNV_INFLATE=host), 32 host threads: wall seconds of the call, the library's timers and the
pack statistics.  Run under rocprofv3 --kernel-trace --stats for the kernel table (msnv_inflate_blocks)."""
import json, os, sys, time
sys.path.insert(0, ".")
from metasnv_amd import core
W = sys.argv[1]
bams = open(W + "/all_samples").read().split()
ctx = core.Context(0)
out = {}
for mode in (sys.argv[2:] or ["device", "host", "device"]):
    os.environ["MSNV_INFLATE"] = mode
    t0 = core.host_timers()
    ds = core.Dataset.from_files(ctx, bams[0], W + "/ref.fa")
    a = time.perf_counter()
    ds.add_sample_bams(bams, 32)
    b = time.perf_counter()
    info = ds.finalize()
    c = time.perf_counter()
    t1 = core.host_timers()
    out.setdefault(mode, []).append({"add_sample_bams_s": b - a, "finalize_s": c - b, "pileup_bases": info["n_pileup_bases"],
                                     "timers": {k: round(t1[k] - t0.get(k, 0.0), 4) for k in t1 if isinstance(t1[k], float) and t1[k] - t0.get(k, 0.0) > 1e-4},
                                     "pack": {k: round(v, 3) for k, v in ds.pack_stats().items() if v}})
    ds.close()
ctx.close()
print(json.dumps(out))
