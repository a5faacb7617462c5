"""profiles/repro_wide.py -- the MSNV_DEEP=wide fuzz mismatch (profiles/r03zr_fuzz_deep_wide.txt): every differing called_SNPs line."""
import os, sys, tempfile
os.environ["MSNV_DEEP"] = "wide"; os.environ.setdefault("MSNV_LAYOUT", "dense")
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from metasnv_amd import core
from parity import run_oracle
kw = {'n_species': 3, 'contig_len': 1500, 'n_samples': 2, 'mean_cov': 300, 'read_len': 100, 'sigma_cov': 1.0, 'frac_absent': 0.1, 'snv_density': 0.0, 'error_rate': 0.02, 'frac_lowq': 0.5, 'frac_indel_reads': 0.0, 'frac_clip_reads': 0.3, 'frac_flagged': 0.0, 'lowercase_ref': 1, 'frac_paired': 0.5, 'seed': 710363175}
pk = {'min_coverage': 10, 'calling_threshold': 4, 'min_fraction': 0.0, 'min_baseq': 0, 'max_depth': 8000, 'min_mapq': 0, 'count_orphans': 1, 'flag_filter': 1796, 'ignore_overlaps': 1}
for k, v in [a.split("=") for a in sys.argv[1:]]:
    (pk if k in pk else kw)[k] = type((pk if k in pk else kw)[k])(v)
sp = core.synth_params(**kw); syn = core.Synth(sp)
samples = [syn.sample_records(i) for i in range(sp.n_samples)]
p = core.default_params(**pk)
o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
ctx = core.Context(0)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, p)
for s in samples: ds.add_sample_records(s)
info = ds.finalize(); st = ds.run()
with tempfile.TemporaryDirectory() as td:
    ds.write_calls(td + "/c", td + "/i", None, None); pop = open(td + "/c").read()
g = {tuple(l.split("\t")[:3]): l for l in pop.splitlines()}; e = {tuple(l.split("\t")[:3]): l for l in o[0].splitlines()}
bad = [k for k in sorted(set(g) | set(e), key=lambda k: (k[0], int(k[2]))) if g.get(k) != e.get(k)]
print("lines", len(g), len(e), "differing", len(bad), {k: st[k] for k in ("n_overflow", "n_events", "n_sites")}, {k: info[k] for k in ("n_pairs", "n_work", "n_reads_pileup")})
for k in bad[:12]:
    print(k, "\n  got ", (g.get(k) or "-")[:160], "\n  want", (e.get(k) or "-")[:160])
ds.close(); ctx.close()
