"""One-off: bit-exact parity at 5x the benchmark size (160 samples x 15 refGenomes x 300 kb, ~7.4 G pileup bases)."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
from parity import run_oracle, first_diff
t0 = time.time()
sp = core.synth_params(n_species=15, contig_len=300000, n_samples=160, mean_cov=10.0, seed=77)
syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
print("synth %.1fs, %.1f GB of records" % (time.time() - t0, sum(s.size for s in samples) / 1e9)); t0 = time.time()
ctx = core.Context(0)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds.add_synth_samples(sp, 0, sp.n_samples, 0)
info = ds.finalize(); st = ds.run()
with tempfile.TemporaryDirectory() as td:
    ds.write_calls(td + "/c", td + "/i"); pop, ind = open(td + "/c").read(), open(td + "/i").read()
print("gpu path %.1fs: %.3g bases, pass %.2f ms, %d lines" % (time.time() - t0, info["n_pileup_bases"], st["ms_total"], pop.count("\n"))); t0 = time.time()
o = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
print("oracle %.1fs; bases equal %s; called_SNPs equal %s; indiv_called equal %s" % (time.time() - t0, info["n_pileup_bases"] == o[3], pop == o[0], ind == o[1]))
if pop != o[0]: print(first_diff(pop, o[0]))
