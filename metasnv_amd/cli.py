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
this driver builds ONE resident dataset per rank -- every BAM is decoded once, by one rank -- and writes
every cov/ file and every split's output from it.  Under torch.distributed (one rank per GPU) the contigs
are sharded over the ranks and the tables gathered to rank 0 (metasnv_amd/parallel.py).
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


def print_coverage_commands(args):                              # metaSNV.py:55-78 with --print-commands
    out_dir = os.path.join(args.project_dir, 'cov')
    for b in read_sample_list(args.all_samples):
        print("msnv_coverage -c 10 -d -i {} {}/{}.cov".format(b, out_dir, os.path.basename(b)))


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


def print_call_commands(args):                                  # metaSNV.py:153-221 with --print-commands
    out_dir = os.path.join(args.project_dir, 'snpCaller')
    splits = sorted(glob.glob('{}/bestsplits/best_split_*'.format(args.project_dir))) if args.n_splits > 1 else [None]
    for split in splits:
        sfx = '.' + os.path.basename(split) if split else ''
        print("msnv_snpcall -f {} {}{}-b {} -i {}/indiv_called{} -c {} -t {} > {}/called_SNPs{}".format(
            args.ref_db, "-g {} ".format(args.db_ann) if args.db_ann else "", "-l {} ".format(split) if split else "",
            args.all_samples, out_dir, sfx, args.min_pos_cov, args.min_pos_snvs, out_dir, sfx))


def _since_process_start():
    """Seconds since the kernel started this process (10 ms steps): what interpreter start-up and the imports cost before main()."""
    try:
        with open('/proc/self/stat') as f:
            ticks = int(f.read().rsplit(')', 1)[1].split()[19])
        with open('/proc/uptime') as f:
            return float(f.read().split()[0]) - ticks / os.sysconf('SC_CLK_TCK')
    except (OSError, ValueError, IndexError):
        return None


def _write_metrics(obj):
    """Kernel timings / counts of the run as JSON when MSNV_METRICS names a file (the reference has no metrics output;
    nothing is written into the project directory unless asked)."""
    path = os.environ.get("MSNV_METRICS")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(obj) + "\n")


def read_split_file(path):
    """Contig names of a best_split_K file (createOptimumSplit.py:54-62 copies bed_header lines `SN\t1\tLN`)."""
    names = []
    for line in open(path):
        w = line.rstrip("\n").split("\t")
        if len(w) >= 3:
            if w[1] != "1":
                raise ValueError("{}: split regions other than `name 1 LEN` are not what metaSNV writes (metaSNV.py:92)".format(path))
            names.append(w[0])
    return names


def resident_run(args, ctx, rank, world):
    """metaSNV.py:55-78 (qaCompute per BAM), :97-150 (summary, tables, split plan) and :153-221 (mpileup | snpCall per split)
    from ONE resident dataset per rank: every BAM is inflated, packed and uploaded once in the whole job -- by one rank --
    whatever --threads / --n_splits / the number of ranks are (the reference decodes every BAM 1 + n_splits times).
    Same files in the same places; rank 0 writes them."""
    import time
    from . import core, parallel
    t_start = time.perf_counter()
    wall = {"coverage_files_s": 0.0, "tables_and_splits_s": 0.0, "calls_text_s": 0.0, "process_age_at_start_s": _since_process_start(), "process_age_at_main_s": args.age_at_main}
    bams = read_sample_list(args.all_samples)
    cov_dir, snp_dir = os.path.join(args.project_dir, 'cov'), os.path.join(args.project_dir, 'snpCaller')
    params = core.default_params(min_coverage=args.min_pos_cov, calling_threshold=args.min_pos_snvs, cov_max=10, cov_min_mapq=1)

    def between_passes(res):
        if rank == 0:
            mkdir_p(cov_dir); mkdir_p(snp_dir)
            t0 = time.perf_counter()
            if not args.use_prev_cov:
                for i, b in enumerate(bams):
                    out = os.path.join(cov_dir, os.path.basename(b) + '.cov')
                    core.write_coverage_records(res["names"], res["lengths"], params.cov_max, res["stats"][i], res["acc"][i], out, out + '.detail')
                    print("Printing details in {}!".format(out + '.detail'))      # qaCompute.cpp:387
                wall["coverage_files_s"] = time.perf_counter() - t0
                t0 = time.perf_counter()
                compute_summary(args)
            get_header(args)
            if args.n_splits > 1:
                split_opt(args)
            shutil.copy(args.all_samples, args.project_dir + '/all_samples')
            wall["tables_and_splits_s"] = time.perf_counter() - t0
        parallel.barrier()

    # contig -> rank by the reference's split rule, genome length x summed coverage (createOptimumSplit.py:46-62): with --use_prev_cov
    # the previous run's all_cov.tab has the coverages, as the reference's split planner reads them; else the ranks take them from
    # their first round of decoded BAMs (parallel.feed_sharded)
    species_weight = None
    if args.use_prev_cov and world > 1:
        tab = "{}/{}.all_cov.tab".format(args.project_dir, os.path.basename(args.project_dir))
        if os.path.isfile(tab):
            species_weight = tables.species_coverage(tab)
    try:
        res = parallel.resident_project_run(ctx, bams[0], args.ref_db, bams, params, batch=max(1, args.threads),
                                            want_coverage=not args.use_prev_cov, ann_path=args.db_ann or None, after_coverage=between_passes,
                                            species_weight=species_weight, inflate_after_context=getattr(args, "inflate_after_context", False))
    except core._lib.MsnvError as e:
        sys.stderr.write(str(e) + "\n")
        sys.stderr.write("SNV calling failed")
        parallel.abort(1)
    def finish():
        res["metrics"]["host_timers"] = core.host_timers()
        res["metrics"]["cli_wall"] = dict(wall, total_s=time.perf_counter() - t_start, process_age_at_end_s=_since_process_start())
        _write_metrics(res["metrics"])

    if rank != 0:
        finish()
        return
    t0 = time.perf_counter()
    try:
        if args.n_splits > 1:
            for split in sorted(glob.glob('{}/bestsplits/best_split_*'.format(args.project_dir))):
                sites, row_off, cell_sample, cells, ann = parallel.split_view(res, read_split_file(split))
                sfx = '.' + os.path.basename(split)
                core.write_calls_cells(res["names"], res["n_samples"], sites, row_off, cell_sample, cells, os.path.join(snp_dir, "called_SNPs" + sfx),
                                       os.path.join(snp_dir, "indiv_called" + sfx), args.db_ann or None, args.ref_db, ann)
        else:
            core.write_calls_cells(res["names"], res["n_samples"], res["sites"], res["row_off"], res["cell_sample"], res["cells"],
                                   os.path.join(snp_dir, "called_SNPs"), os.path.join(snp_dir, "indiv_called"), args.db_ann or None, args.ref_db, res["ann"])
    except (core._lib.MsnvError, ValueError) as e:
        sys.stderr.write(str(e) + "\n")
        sys.stderr.write("SNV calling failed")
        parallel.abort(1)
    wall["calls_text_s"] = time.perf_counter() - t0
    finish()


def build_parser():                                             # metaSNV.py:225-247
    p = argparse.ArgumentParser(description='Compute SNV profiles')
    p.add_argument('project_dir', metavar='DIR', help='The output directory that metaSNV will create e.g. "outputs". Can be a path.')
    p.add_argument('all_samples', metavar='FILE', help='File with an input list of bam files, one file per line')
    p.add_argument("ref_db", metavar='REF_DB_FILE', help='reference multi-sequence FASTA file used for the alignments.')
    p.add_argument('--db_ann', metavar='DB_ANN_FILE', default='', help='Database gene annotation.')
    p.add_argument('--print-commands', default=False, action='store_true', help='Instead of executing the commands, simply print them out')
    p.add_argument('--threads', metavar='INT', default=1, type=int,
                   help='Host threads for BAM decoding (per rank). Will create same number of splits, unless n_splits set differently.')
    p.add_argument('--n_splits', metavar='INT', default=1, type=int, help='Number of bins to split ref into')
    p.add_argument('--use_prev_cov', default=False, action="store_true",
                   help='Use "cov/" and "outputs.all_cov.tab" and "outputs.all_perc.tab" data produced by previous metaSNV run')
    p.add_argument('--min_pos_cov', metavar='INT', default=4, type=int, help='minimum coverage (mapped reads) per position for snpCall.')
    p.add_argument('--min_pos_snvs', metavar='INT', default=4, type=int, help='minimum number of non-reference nucleotides per position for snpCall.')
    return p


def main(argv=None):
    age = _since_process_start()
    parser = build_parser()
    args = parser.parse_args(argv)
    args.age_at_main = age
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
        no_device = "\nERROR:  no HIP device is available\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n\n"
        if world == 1 and args.threads >= 8:
            # one process, many host threads: the BAMs are inflated by the host decoder (csrc/api.cpp: want_device_inflate), so the device
            # is first needed when the packed columns go up -- the runtime (counting the devices alone takes 0.1 s) and the context come
            # up on a thread of their own meanwhile; a node without a GPU is reported when that thread is asked for the context
            from concurrent.futures import ThreadPoolExecutor
            # ... unless the host has few cores for the job (a container's CPU quota counts: 16 cores inflate the benchmark's 160 BAMs in 0.34 s
            # at best, the device in 0.25 s once its context stands): then the feed waits for the context and inflates on the device
            # (parallel.feed_sharded: MSNV_ONESHOT=device | host overrides)
            how = os.environ.get("MSNV_ONESHOT", "")[:1]
            # ... and only when the library WILL inflate there (csrc/api.cpp: want_device_inflate -- MSNV_INFLATE=host, or fewer than 64 MB of
            # BAMs, keeps the host decoder): waiting for the context first and then inflating on the host threads anyway would lose the overlap
            def device_would_inflate():
                e = os.environ.get("MSNV_INFLATE", "")[:1]
                if e:
                    return e == "d"
                total = 0
                for b in read_sample_list(args.all_samples):
                    try:
                        total += os.path.getsize(b)
                    except OSError:
                        pass
                return total >= (64 << 20)
            args.inflate_after_context = how == "d" or (how != "h" and min(args.threads, core.host_cores()) <= 16 and device_would_inflate())

            def bring_up():
                return core.Context(local) if core.device_count() >= 1 else None
            fut = ThreadPoolExecutor(max_workers=1).submit(bring_up)

            def ctx():
                try:
                    c = fut.result()
                except core._lib.MsnvError as e:
                    sys.stderr.write("\nERROR:  {}\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n\n".format(e))
                    parallel.abort(1)
                if c is None:
                    sys.stderr.write(no_device)
                    parallel.abort(1)
                return c
        else:
            if core.device_count() < 1:
                sys.stderr.write(no_device)
                parallel.abort(1)
            try:
                ctx = core.Context(local)
            except core._lib.MsnvError as e:
                sys.stderr.write("\nERROR:  {}\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n\n".format(e))
                parallel.abort(1)

    if args.print_commands:                                     # metaSNV.py --print-commands: nothing runs
        if rank == 0:
            if not args.use_prev_cov:
                print_coverage_commands(args)
                compute_summary(args)                           # exits like the reference when cov/ is still empty
            get_header(args)
            if args.n_splits > 1:
                split_opt(args)
            print_call_commands(args)
        parallel.finalize()
        return
    resident_run(args, ctx, rank, world)
    parallel.barrier()
    parallel.finalize()


if __name__ == '__main__':
    main()
