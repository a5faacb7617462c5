import sys, os
os.environ["MSNV_PHASE_TIMES"] = "1"
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
sp = core.synth_params(n_species=3, contig_len=300000, n_samples=int(os.environ.get("NS", "160")), mean_cov=float(os.environ.get("COV", "10")), seed=1,
                       **({"error_rate": float(os.environ["ERR"])} if "ERR" in os.environ else {}))
syn = core.Synth(sp); ctx = core.Context(0)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds.add_synth_samples(sp, 0, sp.n_samples, 0); ds.finalize()
for _ in range(3): ds.run()
acc = {}
for _ in range(20):
    st = ds.run()
    for k in ("ms_total", "ms_pileup", "ms_gate", "ms_gather", "ms_decide"):
        acc[k] = acc.get(k, 0) + st[k] / 20
print({k: round(v, 4) for k, v in acc.items()}, {k: st[k] for k in ("n_sites", "n_events", "n_called_pop", "n_called_indiv")})
