import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
from parity import run_oracle, first_diff
import tempfile
t0 = time.time()
sp = core.synth_params(n_species=75, contig_len=30_000_000, n_samples=4, mean_cov=0.01, frac_absent=0.0, snv_density=0.01, seed=99)
syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
print("synth %.1fs, positions %.3g, record bytes %s" % (time.time() - t0, sum(syn.lengths), [s.size for s in samples])); t0 = time.time()
p = core.default_params(min_coverage=1, calling_threshold=1)
ctx = core.Context(0)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
for s in samples: ds.add_sample_records(s)
info = ds.finalize(); st = ds.run()
print("gpu %.1fs" % (time.time() - t0), {k: info[k] for k in ("n_positions", "n_tiles", "n_pairs", "n_work", "device_bytes", "n_pileup_bases")}, st["ms_total"], st["n_sites"]); t0 = time.time()
with tempfile.TemporaryDirectory() as td:
    ds.write_calls(td + "/c", td + "/i"); pop, ind = open(td + "/c").read(), open(td + "/i").read()
o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
print("oracle %.1fs" % (time.time() - t0))
print("pop equal", pop == o[0], "ind equal", ind == o[1], pop.count("\n"), "lines; last:", pop.splitlines()[-1][:60] if pop else "")
if pop != o[0]: print(first_diff(pop, o[0]))
