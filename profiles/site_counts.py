import os, sys
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
for ns in (160, 320):
    sp = core.synth_params(n_species=3, contig_len=300000, n_samples=ns, mean_cov=10.0, seed=1)
    syn = core.Synth(sp); ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    ds.add_synth_samples(sp, 0, ns, 0); ds.finalize()
    st = ds.run()
    print(ns, {k: st[k] for k in ("n_sites", "n_events", "n_called_pop", "n_called_indiv")})
    ds.close(); ctx.close()
