"""Worker of the 2-rank product-path test: every rank packs its contig shard (contig mask), runs the kernels on the
GPU, the site and annotation records are gathered and rank 0 writes called_SNPs / indiv_called."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from metasnv_amd import core, parallel  # noqa: E402


def main():
    work = sys.argv[1]
    rank, world, local = parallel.init_from_env()
    sp = core.synth_params(n_species=5, contig_len=4000, n_samples=6, mean_cov=11.0, snv_density=0.03, frac_absent=0.2, seed=55)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    ctx = core.Context(local)

    def add(ds):
        for s in samples:
            ds.add_sample_records(s)

    ann, fa = os.path.join(work, "ann.tsv"), os.path.join(work, "ref.fa")
    sites, smp, info, st = parallel.sharded_call(ctx, syn.names, syn.lengths, syn.seqs, add,
                                                 called_path=os.path.join(work, "called_SNPs"), indiv_path=os.path.join(work, "indiv_called"),
                                                 ann_path=ann if os.path.exists(ann) else None, fasta_path=fa)
    open(os.path.join(work, "rank%d.info" % rank), "w").write("%d %d" % (info["n_positions"], info["n_pileup_bases"]))
    parallel.barrier()
    parallel.finalize()


if __name__ == "__main__":
    main()
