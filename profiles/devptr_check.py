"""profiles/devptr_check.py -- msnv_dataset_add_sample_records_device on tens of megabytes per stream: does the device-pointer path pack what the host-upload path packs?"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from metasnv_amd import core
hip = C.CDLL("libamdhip64.so")
sp = core.synth_params(n_species=3, contig_len=500000, n_samples=int(sys.argv[1]) if len(sys.argv) > 1 else 32, mean_cov=10.0, frac_absent=0.5, seed=5)
syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
print("streams", len(samples), "bytes", sum(s.size for s in samples), [s.size for s in samples][:8])
ctx = core.Context(0)
gapped = len(sys.argv) > 2                  # streams behind one another with 8 * n bytes in front, like the all-to-all's receive buffer
tot = sum(int(s.size) for s in samples) + 8 * len(samples)
p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(tot + 64)) == 0
o = 8 * len(samples); ptrs, sizes = [], []
for s in samples:
    if s.size: assert hip.hipMemcpy(C.c_void_p(p.value + o), C.c_void_p(s.ctypes.data), C.c_size_t(s.size), 1) == 0
    ptrs.append(p.value + o if s.size else 0); sizes.append(int(s.size)); o += int(s.size)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds.add_samples_records_device(ptrs, sizes)
info = ds.finalize(); print("device pointers:", info["n_pileup_bases"])
ds2 = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds2.add_samples_records(samples)
info2 = ds2.finalize(); print("host upload:", info2["n_pileup_bases"])
