import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
from parity import run_oracle, first_diff
cases = {
 141: (dict(n_species=1, contig_len=1500, n_samples=2, mean_cov=200, read_len=250, sigma_cov=1.0, frac_absent=0.1, snv_density=0.05, error_rate=0.001, frac_lowq=0.5, frac_indel_reads=0.0, frac_clip_reads=0.0, frac_flagged=0.0, lowercase_ref=1, seed=816661297), dict(min_coverage=4, calling_threshold=1, min_fraction=0.2, min_baseq=13)),
 0: (dict(n_species=2, contig_len=1500, n_samples=1, mean_cov=300, read_len=50, sigma_cov=0.5, frac_absent=0.1, snv_density=0.05, error_rate=0.001, frac_lowq=0.0, frac_indel_reads=0.0, frac_clip_reads=0.03, frac_flagged=0.0, lowercase_ref=1, seed=929360195), dict(min_coverage=1, calling_threshold=4, min_fraction=0.2, min_baseq=13)),
 31: (dict(n_species=1, contig_len=2048, n_samples=16, mean_cov=200, read_len=100, sigma_cov=0.5, frac_absent=0.1, snv_density=0.0, error_rate=0.001, frac_lowq=0.0, frac_indel_reads=0.0, frac_clip_reads=0.3, frac_flagged=0.0, lowercase_ref=1, seed=318214282), dict(min_coverage=4, calling_threshold=4, min_fraction=0.01, min_baseq=13)),
}
ctx = core.Context(0)
for cid, (kw, pk) in cases.items():
    sp = core.synth_params(**kw); syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    p = core.default_params(**pk)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
    for s in samples: ds.add_sample_records(s)
    info = ds.finalize(); st = ds.run()
    with tempfile.TemporaryDirectory() as td:
        ds.write_calls(td + "/c", td + "/i"); pop, ind = open(td + "/c").read(), open(td + "/i").read()
    o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    print("case", cid, "equal", pop == o[0], ind == o[1], {k: info[k] for k in ("n_pairs", "n_work", "n_pileup_bases")}, "overflow", st["n_overflow"], "lines", pop.count("\n"))
    if pop != o[0]:
        d = first_diff(pop, o[0]); print(d[:700])
    ds.close()
