"""profiles/deal_check.py -- msnv_records_deal_device against msnv_records_partition on streams of tens of megabytes (first difference printed)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from metasnv_amd import core
hip = C.CDLL("libamdhip64.so")
sp = core.synth_params(n_species=int(sys.argv[1]) if len(sys.argv) > 1 else 6, contig_len=400000, n_samples=int(sys.argv[2]) if len(sys.argv) > 2 else 20, mean_cov=10.0, seed=5)
syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
print("streams", len(samples), "bytes", sum(s.size for s in samples))
nc = len(syn.names)
ctx = core.Context(0)
for n_parts in (1, 4):
    owner = np.array([c % n_parts for c in range(nc)], dtype=np.int32)
    cap = sum(int(s.size) for s in samples)
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(cap + 64)) == 0
    pb, st = core.deal_records_device(ctx, samples, owner, n_parts, p.value, cap)
    got = np.zeros(cap, np.uint8); assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), p, C.c_size_t(cap), 2) == 0
    hip.hipFree(p)
    o, bad = 0, 0
    for k in range(n_parts):
        for i, s in enumerate(samples):
            parts, _ = core.partition_records(s, owner, n_parts)
            w = parts[k]
            if int(pb[i, k]) != w.size:
                print("SIZE", n_parts, k, i, int(pb[i, k]), w.size); bad += 1
            g = got[o:o + w.size]
            if g.tobytes() != w.tobytes():
                d = int(np.nonzero(g != w)[0][0]); print("DIFF parts", n_parts, "part", k, "stream", i, "at byte", d, "of", w.size); bad += 1
            o += w.size
    print("n_parts", n_parts, "mismatching (stream, part) pairs:", bad)
