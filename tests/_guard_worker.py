"""Worker of tests/test_gpu_guard.py: cohorts of every kernel family through the product with MSNV_GUARD_ALLOC=1 (set by the caller) --
every device buffer then ends at the end of its mapping with unmapped addresses behind it, so a kernel that reads or writes past the
end of a buffer dies with a GPU memory fault instead of landing in a neighbour."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from metasnv_amd import core  # noqa: E402
from parity import run_product, run_oracle, synth_case, first_diff  # noqa: E402

CASES = {
    "narrow": (dict(n_species=3, contig_len=30000, n_samples=24, mean_cov=10.0, seed=71), dict()),
    "short_reads_dense_layout": (dict(n_species=2, contig_len=20000, n_samples=10, mean_cov=8.0, read_len=50, seed=72), dict()),
    "deep_wide": (dict(n_species=2, contig_len=9000, n_samples=6, mean_cov=400.0, sigma_cov=1.2, seed=73), dict()),
    "sparse_whole_tile": (dict(n_species=12, contig_len=5000, n_samples=16, mean_cov=5.0, sigma_cov=0.5, snv_density=0.012, error_rate=0.004, frac_absent=0.6, lowercase_ref=1, seed=74),
                          dict(min_coverage=2, calling_threshold=2)),
    "merged_and_split": (dict(n_species=4, contig_len=5000, n_samples=40, mean_cov=2.0, sigma_cov=2.2, snv_density=0.03, error_rate=0.01, frac_absent=0.2, read_len=60, lowercase_ref=1, seed=75),
                         dict(min_coverage=3, calling_threshold=3)),
    "noisy_planes": (dict(n_species=2, contig_len=12000, n_samples=12, mean_cov=12.0, error_rate=0.04, seed=76), dict()),
    "many_sites": (dict(n_species=2, contig_len=8000, n_samples=20, mean_cov=10.0, snv_density=0.3, seed=77), dict(min_coverage=2, calling_threshold=2)),
    # round 4: overlapping mates, aux tags, SEQ `*` and CG-tag CIGARs through the device pack's kernels (devpack.hip: msnv_ovl_*, rec_load)
    "paired_aux_records": (dict(n_species=2, contig_len=15000, n_samples=10, mean_cov=14.0, frac_paired=0.6, frac_aux=0.5, frac_noseq=0.05, read_len=90, seed=78), dict()),
    # ... and BAM FILES through the resident device inflate (inflate_k.hip: msnv_inflate_blocks + msnv_crc_blocks) into the device pack
    "bam_files_device_inflate": (dict(n_species=2, contig_len=25000, n_samples=6, mean_cov=12.0, frac_paired=0.3, seed=79), dict()),
}


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        sk, pk = CASES[name]
        syn, samples = synth_case(**sk)
        p = core.default_params(**pk)
        if name == "bam_files_device_inflate":
            import tempfile
            os.environ["MSNV_INFLATE"] = "device"
            with tempfile.TemporaryDirectory() as td:
                fa = os.path.join(td, "ref.fa"); syn.write_fasta(fa)
                paths = []
                for i, rec in enumerate(samples):
                    paths.append(os.path.join(td, "s%d.bam" % i)); core.write_bam(paths[-1], syn.names, syn.lengths, rec, level=[6, 1, 0, 9, 4, 2][i % 6])
                ctx = core.Context(0)
                ds = core.Dataset.from_files(ctx, paths[0], fa, p)
                ds.add_sample_bams(paths, 3)
                info = ds.finalize(); st = ds.run()
                ds.write_calls(os.path.join(td, "c"), os.path.join(td, "i"), None, None)
                pop, ind = open(os.path.join(td, "c")).read(), open(os.path.join(td, "i")).read()
        else:
            pop, ind, info, st, ds, ctx = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p, return_ds=True)
        if os.environ.get("MSNV_GUARD_DEBUG"):
            print("first pass", {k: st[k] for k in ("n_sites", "n_events", "n_overflow", "n_called_pop", "n_called_indiv")}, {k: info[k] for k in ("n_tiles", "n_pairs", "n_work", "allele_planes")}, flush=True)
        ds.run_many(2, overlap=True)
        cov = ds.fused_run()
        o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        assert pop == o[0], (name, first_diff(pop, o[0]))
        assert ind == o[1], (name, first_diff(ind, o[1]))
        ds.close(); ctx.close()
        print("ok", name, pop.count("\n"), ind.count("\n"), flush=True)


if __name__ == "__main__":
    main()
