"""metaSNV.py-compatible driver on top of libmsnv.so.

Same argv surface and project-directory layout as the reference driver (metaSNV.py:224-292), so
that metaSNV_Filtering.py / metaSNV_DistDiv.py consume the results unchanged:

    DIR/cov/<bam>.cov, .cov.detail, .cov.summary      <- qaCompute + computeGenomeCoverage.py
    DIR/<name>.all_cov.tab, <name>.all_perc.tab        <- collapse_coverages.py
    DIR/bed_header                                     <- samtools view -H
    DIR/bestsplits/best_split_K                        <- createOptimumSplit.py
    DIR/snpCaller/called_SNPs[.best_split_K], indiv_called[.best_split_K]   <- mpileup | snpCall
    DIR/all_samples

Where the reference forks one process per BAM / per split (multiprocessing.Pool, metaSNV.py:58,197),
this driver makes one library call per step and the GPU does the work.  Under torch.distributed
(one rank per GPU) the splits / BAMs are dealt to the ranks (metasnv_amd/parallel.py).
"""
import argparse
import glob
import os
import shutil
import sys

from . import tables


def mkdir_p(d):
    os.makedirs(d, exist_ok=True)


def create_directories(base):                                  # metaSNV.py:26-35
    mkdir_p(base)
    for sub in ('cov', 'bestsplits', 'snpCaller', 'filtered', 'filtered/pop', 'filtered/ind', 'distances'):
        mkdir_p(os.path.join(base, sub))


def read_sample_list(path):
    with open(path) as f:
        return [line.rstrip() for line in f if line.rstrip()]


def compute_opt(args, ctx, rank=0, world=1):                    # metaSNV.py:55-78 (qaCompute -c 10 -d -i per BAM)
    from . import core
    out_dir = os.path.join(args.project_dir, 'cov')
    mkdir_p(out_dir)
    bams = read_sample_list(args.all_samples)
    mine = [b for i, b in enumerate(bams) if i % world == rank]
    if args.print_commands:
        for b in bams:
            print("msnv_coverage -c 10 -d -i {} {}/{}.cov".format(b, out_dir, os.path.basename(b)))
        return
    if not mine:
        return
    # all BAMs of this rank share one header (metaSNV.py:82-83): one dataset, one device pass
    ds = core.Dataset.from_files(ctx, mine[0], None, core.default_params(cov_max=10, cov_min_mapq=1))
    try:
        ds.add_sample_bams(mine, args.threads)
        ds.finalize()
        ds.coverage_run()
        for i, b in enumerate(mine):
            out = os.path.join(out_dir, os.path.basename(b) + '.cov')
            ds.write_coverage(i, out, out + '.detail')
            print("Printing details in {}!".format(out + '.detail'))      # qaCompute.cpp:387
    except core._lib.MsnvError as e:
        sys.stderr.write("Failure in sample set starting at {}\n{}\n".format(mine[0], e))
        sys.exit(1)
    finally:
        ds.close()


def compute_summary(args):                                      # metaSNV.py:97-125
    name = os.path.basename(args.project_dir)
    cov_files = glob.glob(os.path.join(args.project_dir, 'cov', '*.cov'))
    if not cov_files:
        sys.stderr.write("Coverage files not found.\n")
        if args.print_commands:
            sys.stderr.write("Finish running the commands printed above and then run this command again.\n")
        sys.exit(1)
    for f in cov_files:
        tables.species_summary(f, f + '.detail', f + '.summary')
    print("\nCoverage summary here: {}".format(args.project_dir))
    print("	Average vertical genome coverage: '{}/{}.all_cov.tab'".format(args.project_dir, name))
    print("	Horizontal genome coverage (1X): '{}/{}.all_perc.tab'".format(args.project_dir, name))
    print("")
    tables.collapse_tables(args.project_dir)


def get_header(args):                                           # metaSNV.py:81-94 (samtools view -H)
    from . import core
    first = open(args.all_samples).readline().rstrip()
    core.write_bed_header(first, os.path.join(args.project_dir, 'bed_header'))
    args.ctg_len = os.path.join(args.project_dir, 'bed_header')


def split_opt(args):                                            # metaSNV.py:128-150
    if args.n_splits > 100:
        sys.stderr.write("Maximum number of splits is 100.\n")
        args.n_splits = 100
    older = glob.glob(args.project_dir + '/bestsplits/*')
    if older:
        sys.stderr.write("\nremoving old splits.\n")
        for f in older:
            os.unlink(f)
    name = os.path.basename(args.project_dir)
    print("\nCalculating best database split:")
    tables.plan_splits("{}/{}.all_cov.tab".format(args.project_dir, name), "{}/{}.all_perc.tab".format(args.project_dir, name),
                       args.ctg_len, args.n_splits, os.path.join(args.project_dir, "bestsplits", "best_split"))


def execute_snp_call(args, ctx, ifile, ofile, split):           # metaSNV.py:153-176 (mpileup | snpCall)
    from . import core
    bams = read_sample_list(args.all_samples)
    if args.print_commands:
        print("msnv_call -f {} {}{}-b {} -i {} -c {} -t {} > {}".format(
            args.ref_db, "-g {} ".format(args.db_ann) if args.db_ann else "", "-l {} ".format(split) if split else "",
            args.all_samples, ifile, args.min_pos_cov, args.min_pos_snvs, ofile))
        return 0
    params = core.default_params(min_coverage=args.min_pos_cov, calling_threshold=args.min_pos_snvs)
    ds = core.Dataset.from_files(ctx, bams[0], args.ref_db, params)
    try:
        if split:
            ds.set_bed_file(split)
        ds.add_sample_bams(bams, args.threads)
        ds.finalize()
        st = ds.run()
        _write_metrics({"mode": "call", "split": split or "", "pileup": st, "dataset": ds.info()})
        ds.write_calls(ofile, ifile, args.db_ann or None, args.ref_db)
        return 0
    except core._lib.MsnvError as e:
        sys.stderr.write(str(e) + "\n")
        return e.code
    finally:
        ds.close()


def _write_metrics(obj):
    """Kernel timings / counts of the run as JSON when MSNV_METRICS names a file (the reference has no metrics output;
    nothing is written into the project directory unless asked)."""
    path = os.environ.get("MSNV_METRICS")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(obj) + "\n")


def fused_cov_and_call(args, ctx):
    """Single-GPU, unsplit run: qaCompute (metaSNV.py:55-78) and mpileup | snpCall (:153-221) from ONE dataset, so
    every BAM is inflated, packed and uploaded once.  Writes the same files in the same places as the two steps."""
    from . import core
    bams = read_sample_list(args.all_samples)
    cov_dir, snp_dir = os.path.join(args.project_dir, 'cov'), os.path.join(args.project_dir, 'snpCaller')
    mkdir_p(cov_dir); mkdir_p(snp_dir)
    params = core.default_params(min_coverage=args.min_pos_cov, calling_threshold=args.min_pos_snvs, cov_max=10, cov_min_mapq=1)
    ds = core.Dataset.from_files(ctx, bams[0], args.ref_db, params)
    try:
        ds.add_sample_bams(bams, args.threads)
        ds.finalize()
        st_p, st_c = ds.fused_run()
        _write_metrics({"mode": "fused", "pileup": st_p, "coverage": st_c, "dataset": ds.info()})
        for i, b in enumerate(bams):
            out = os.path.join(cov_dir, os.path.basename(b) + '.cov')
            ds.write_coverage(i, out, out + '.detail')
            print("Printing details in {}!".format(out + '.detail'))      # qaCompute.cpp:387
        compute_summary(args)
        get_header(args)
        shutil.copy(args.all_samples, args.project_dir + '/all_samples')
        ds.write_calls(os.path.join(snp_dir, "called_SNPs"), os.path.join(snp_dir, "indiv_called"), args.db_ann or None, args.ref_db)
    except core._lib.MsnvError as e:
        sys.stderr.write(str(e) + "\n")
        sys.stderr.write("SNV calling failed")
        sys.exit(1)
    finally:
        ds.close()


def snp_call(args, ctx, rank=0, world=1):                       # metaSNV.py:179-221
    out_dir = os.path.join(args.project_dir, 'snpCaller')
    mkdir_p(out_dir)
    if rank == 0:
        shutil.copy(args.all_samples, args.project_dir + '/all_samples')
    indiv_out = os.path.join(out_dir, "indiv_called")
    called = os.path.join(out_dir, "called_SNPs")
    if args.n_splits > 1:
        splits = sorted(glob.glob('{}/bestsplits/best_split_*'.format(args.project_dir)))
        for i, split in enumerate(splits):
            if i % world != rank:
                continue                                        # splits are whole species: shards need no exchange
            v = execute_snp_call(args, ctx, '{}.{}'.format(indiv_out, os.path.basename(split)),
                                 '{}.{}'.format(called, os.path.basename(split)), split)
            if v:
                sys.stderr.write("SNV calling failed")
                sys.exit(1)
    elif rank == 0:
        v = execute_snp_call(args, ctx, indiv_out, called, None)
        if v:
            sys.stderr.write("SNV calling failed")
            sys.exit(1)


def build_parser():                                             # metaSNV.py:225-247
    p = argparse.ArgumentParser(description='Compute SNV profiles')
    p.add_argument('project_dir', metavar='DIR', help='The output directory that metaSNV will create e.g. "outputs". Can be a path.')
    p.add_argument('all_samples', metavar='FILE', help='File with an input list of bam files, one file per line')
    p.add_argument("ref_db", metavar='REF_DB_FILE', help='reference multi-sequence FASTA file used for the alignments.')
    p.add_argument('--db_ann', metavar='DB_ANN_FILE', default='', help='Database gene annotation.')
    p.add_argument('--print-commands', default=False, action='store_true', help='Instead of executing the commands, simply print them out')
    p.add_argument('--threads', metavar='INT', default=1, type=int,
                   help='Host threads for BAM decoding. Will create same number of splits, unless n_splits set differently.')
    p.add_argument('--n_splits', metavar='INT', default=1, type=int, help='Number of bins to split ref into')
    p.add_argument('--use_prev_cov', default=False, action="store_true",
                   help='Use "cov/" and "outputs.all_cov.tab" and "outputs.all_perc.tab" data produced by previous metaSNV run')
    p.add_argument('--min_pos_cov', metavar='INT', default=4, type=int, help='minimum coverage (mapped reads) per position for snpCall.')
    p.add_argument('--min_pos_snvs', metavar='INT', default=4, type=int, help='minimum number of non-reference nucleotides per position for snpCall.')
    return p


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    args.project_dir = args.project_dir.rstrip('/')
    if not os.path.isfile(args.ref_db):
        sys.stderr.write("\nERROR:	No reference database or annotation file found!\"\nERROR:	'{}' is not a file.\"\n\n".format(args.ref_db))
        parser.print_help()
        sys.exit(1)
    if args.threads > 1 and args.n_splits == 1:                 # metaSNV.py:275-276
        args.n_splits = args.threads

    from . import parallel
    rank, world, local = parallel.init_from_env()
    if rank == 0 and os.path.exists(args.project_dir) and not args.print_commands and not args.use_prev_cov:
        sys.stderr.write("Project directory '{}' already exists\n\n\n".format(args.project_dir))
        parallel.abort(1)
    parallel.barrier()
    if rank == 0:
        create_directories(args.project_dir)
    parallel.barrier()

    ctx = None
    if not args.print_commands:
        from . import core
        try:
            ctx = core.Context(local)
        except core._lib.MsnvError as e:
            sys.stderr.write("\nERROR:  {}\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n\n".format(e))
            parallel.abort(1)

    if world == 1 and args.n_splits <= 1 and not args.use_prev_cov and not args.print_commands:
        fused_cov_and_call(args, ctx)                           # BASELINE configs[2]: one decode, both passes on the device
        parallel.finalize()
        return
    if not args.use_prev_cov:
        compute_opt(args, ctx, rank, world)
        parallel.barrier()
        if rank == 0:
            compute_summary(args)
    parallel.barrier()
    if rank == 0:
        get_header(args)
        if args.n_splits > 1:
            split_opt(args)
    args.ctg_len = os.path.join(args.project_dir, 'bed_header')
    parallel.barrier()
    snp_call(args, ctx, rank, world)
    parallel.barrier()
    parallel.finalize()


if __name__ == '__main__':
    main()
