"""profiles/sparse_items.py -- what a sparse cohort's work items look like (chunks and pieces per whole-tile item, fill of the 16-piece rounds)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from metasnv_amd import core
sp = core.synth_params(n_species=150, contig_len=2070000, n_samples=500, mean_cov=5.0, sigma_cov=0.3, contigs_per_species_max=20, species_per_sample=1, frac_absent=0.75, seed=1)
ctx = core.Context(0)
syn = core.Synth(sp)
ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
ds.add_synth_samples(sp, 0, sp.n_samples, 0)
info = ds.finalize()
print({k: info[k] for k in ("n_tiles", "n_pairs", "n_work", "n_whole_tile_items", "n_listed_tiles", "n_reads_pileup")})
w = ds.column("work").view(np.uint32).reshape(-1, 16)
ch = ds.column("chunks").view(np.uint32).reshape(-1, 8)
nch = (w[:, 4] - w[:, 3]).astype(np.int64)
npairs = (w[:, 2] - w[:, 1]).astype(np.int64)
nrd = ch[:, 6] & 0xffff
print("items", len(w), "chunks", len(ch), "chunks/item histogram", np.bincount(np.minimum(nch, 8)))
print("pairs/item histogram", np.bincount(np.minimum(npairs, 8)))
cs = np.concatenate([[0], np.cumsum(nrd)])
pieces = cs[w[:, 4]] - cs[w[:, 3]]
print("pieces/item: mean %.1f  percentiles 10/50/90/99: %s" % (pieces.mean(), np.percentile(pieces, [10, 50, 90, 99])))
rounds16 = (nrd + 15) // 16
print("16-piece wavefront rounds with pieces: %d; if dense: %d; slots of 4 x 2 per chunk: %d" % (rounds16.sum(), (nrd.sum() + 15) // 16, 8 * len(ch)))
