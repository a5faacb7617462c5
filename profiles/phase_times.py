"""profiles/phase_times.py -- per-phase times of a pass (MSNV_PHASE_TIMES=1: HIP events between the tail kernels, ~6 us each).
  NS / COV / ERR / SIGMA: testdata shape with that many samples / mean coverage / error rate / log-normal sigma of the coverages;  WORKLOAD=config3|config4shard SCALE=0.1: bench.py's shapes."""
import sys, os
os.environ["MSNV_PHASE_TIMES"] = "1"
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
if "WORKLOAD" in os.environ:
    import argparse, bench
    a = argparse.Namespace(workload=os.environ["WORKLOAD"], scale=float(os.environ.get("SCALE", "0.1")), samples=None, contig_len=None, species=None,
                           mean_cov=None, read_len=100, error_rate=None)
    kw, label = bench.workload_params(a)
    sp = core.synth_params(**kw)
else:
    sp = core.synth_params(n_species=3, contig_len=300000, n_samples=int(os.environ.get("NS", "160")), mean_cov=float(os.environ.get("COV", "10")), seed=1,
                           **({"error_rate": float(os.environ["ERR"])} if "ERR" in os.environ else {}), **({"sigma_cov": float(os.environ["SIGMA"])} if "SIGMA" in os.environ else {}))
syn = core.Synth(sp); ctx = core.Context(0)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds.add_synth_samples(sp, 0, sp.n_samples, 0); info = ds.finalize()
for _ in range(3): ds.run()
acc = {}
N = int(os.environ.get("PASSES", "20"))
for _ in range(N):
    st = ds.run()
    for k in ("ms_total", "ms_pileup", "ms_gate", "ms_gather", "ms_decide"):
        acc[k] = acc.get(k, 0) + st[k] / N
print({k: round(v, 4) for k, v in acc.items()}, {k: st[k] for k in ("n_sites", "n_events", "n_called_pop", "n_called_indiv")},
      {k: info[k] for k in ("n_pileup_bases", "n_positions", "n_tiles", "n_pairs", "n_work")}, {k: ds.info()[k] for k in ("n_whole_tile_items", "n_listed_tiles")})
