"""metaSNV_Filtering.py-compatible driver (reference: /metaSNV_Filtering.py) with filter_two on the GPU.

Same argv, same project-directory inputs (`<proj>.all_cov.tab`, `<proj>.all_perc.tab`, `all_samples`,
`snpCaller/called*`, `snpCaller/indiv*`) and the same outputs (`filtered/pop/<species>.filtered.freq`,
`filtered/ind/...` with --ind).  FILTER I (samples of interest per species) is table logic and stays in
Python like the reference's; FILTER II (position filter + allele frequencies, metaSNV_Filtering.py:156-242)
is one library call for all species (msnv_filter_files): the files are parsed once, not once per species."""
import argparse
import ctypes as C
import glob
import os
import shutil
import sys


def build_parser():                                            # metaSNV_Filtering.py:18-50
    p = argparse.ArgumentParser(prog='metaSNV_filtering.py', description='metaSNV filtering step', epilog='''Note:''',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--version', action='version', version='%(prog)s 2.0', help=argparse.SUPPRESS)
    p.add_argument("--debug", action="store_true", help=argparse.SUPPRESS)
    p.add_argument('projdir', help='project name', metavar='Proj')
    p.add_argument('-b', metavar='FLOAT', type=float, default=40.0,
                   help="Coverage breadth: minimal horizontal genome coverage percentage per sample per species")
    p.add_argument('-d', metavar='FLOAT', type=float, default=5.0,
                   help="Coverage depth: minimal average vertical genome coverage per sample per species")
    p.add_argument('-m', metavar='INT', type=int, help="Minimum number of samples per species", default=2)
    p.add_argument('-c', metavar='FLOAT', type=float, help="FILTERING STEP II:minimum coverage per position per sample per species", default=5.0)
    p.add_argument('-p', metavar='FLOAT', type=float,
                   help="FILTERING STEP II:required proportion of informative samples (coverage non-zero) per position", default=0.50)
    p.add_argument('--ind', action='store_true', help="Compute individual SNVs")
    p.add_argument('--n_threads', metavar=': Number of Processes', default=1, type=int, help="Number of jobs to run simultaneously.")
    return p


def file_check(args):                                          # metaSNV_Filtering.py:57-75 (same messages, same exits)
    args.projdir = args.projdir.rstrip('/')
    stem = args.projdir + '/' + args.projdir.split('/')[-1]
    args.coverage_file, args.percentage_file = stem + '.all_cov.tab', stem + '.all_perc.tab'
    args.all_samples = args.projdir + '/' + 'all_samples'
    print("Checking for necessary input files...")
    tables = (args.coverage_file, args.percentage_file)
    if not all(os.path.isfile(t) for t in tables):
        sys.exit("\nERROR: No such file '{}',\nERROR: No such file '{}'".format(*tables))
    print("found: '{}' \nfound:'{}'".format(*tables))
    if not os.path.isfile(args.all_samples):
        sys.exit("\nERROR: No such file '{}'".format(args.all_samples))
    print("found: '{}'\n".format(args.all_samples))


_OPTION_LABELS = (('b', "threshold: percentage covered (breadth) {}"), ('d', "threshold: average coverage (depth) {}"),
                  ('m', "threshold: Min. number samples_of_interest per taxid_of_interest {}"),
                  ('c', "threshold: Min. position coverage per sample within samples_of_interest {}"),
                  ('p', "threshold: Min. proportion of covered samples in samples_of_interest {}"),
                  ('ind', "Compute indiv SNVs : {}"), ('n_threads', "Number of parallel processes : {}"))


def print_arguments(args):                                     # metaSNV_Filtering.py:78-95: falsy options are not echoed
    print("Options:")
    for attr, label in _OPTION_LABELS:
        value = getattr(args, attr)
        if value:
            print(label.format(value))
    print("")


def relevant_taxa(coverage_file, percentage_file, b, d, m):    # metaSNV_Filtering.py:111-145 (FILTER I)
    """Species -> samples of interest (depth >= d and breadth >= b), kept when at least m samples qualify.
    A species row shorter than the header is never kept: the reference only tests m at the header's last column."""
    soi = {}
    with open(coverage_file) as cov, open(percentage_file) as per:
        header_cov, header_per = cov.readline().split(), per.readline().split()
        cov.readline(); per.readline()
        if header_cov != header_per:
            sys.exit("ERROR: Coverage file headers do not match!")
        for cl, pl in zip(cov, per):
            cs, ps = cl.split(), pl.split()
            ctax, ptax = cs.pop(0), ps.pop(0)
            coverage, percentage = list(map(float, cs)), list(map(float, ps))
            if ctax != ptax:
                sys.exit("ERROR: TaxIDs in the coverage files are not in the same order!")
            names = []
            for k, (c, p) in enumerate(zip(coverage, percentage), 1):
                if c >= d and p >= b:
                    names.append(header_cov[k - 1])
                if k == len(header_cov) and len(names) >= m:
                    soi[ctax] = names
    return {'SoI': soi, 'h': header_cov}


def filter_two_all(ctx, all_samples, snp_files, outdir, samples_of_interest, c, p):
    """FILTER II for every species at once (metaSNV_Filtering.py:156-242) on the device."""
    from ._lib import lib, check, FilterSpecies
    snp_header = [l.split('/')[-1] for l in open(all_samples).read().splitlines()]      # :163-165
    species = list(samples_of_interest.keys())
    arr = (FilterSpecies * max(1, len(species)))()
    keep = []
    for i, sp in enumerate(species):
        names = samples_of_interest[sp]
        idx = [snp_header.index(n) for n in names]                                         # :180-181 (ValueError like the reference)
        ia = (C.c_int32 * len(idx))(*idx)
        na = (C.c_char_p * len(names))(*[n.encode() for n in names])
        keep += [ia, na]
        arr[i] = FilterSpecies(sp.encode(), len(idx), ia, na)
    paths = (C.c_char_p * max(1, len(snp_files)))(*[f.encode() for f in snp_files])
    kept, ms = C.c_uint64(), C.c_double()
    check(lib.msnv_filter_files(ctx._h, paths, len(snp_files), len(snp_header), arr, len(species), c, p, outdir.encode(),
                                C.byref(kept), C.byref(ms)))
    for sp in species:
        if os.path.isfile(outdir + '/' + '%s.filtered.freq' % sp):
            print("Generating: {}".format(outdir + '/' + '%s.filtered.freq' % sp))
    return kept.value, ms.value


def main(argv=None):                                           # metaSNV_Filtering.py:248-301
    args = build_parser().parse_args(argv)
    print_arguments(args)
    file_check(args)
    samples_of_interest = relevant_taxa(args.coverage_file, args.percentage_file, args.b, args.d, args.m)['SoI']
    print(samples_of_interest.keys())
    filt_folder = args.projdir + '/filtered' + '/'
    if os.path.exists(filt_folder):
        shutil.rmtree(filt_folder)
    os.makedirs(filt_folder)
    os.makedirs(filt_folder + '/pop/')
    from . import core
    try:
        ctx = core.Context(0)
    except core._lib.MsnvError as e:
        sys.exit("\nERROR:  {}\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n".format(e))
    try:
        filter_two_all(ctx, args.all_samples, glob.glob(args.projdir + '/snpCaller/called*'), filt_folder + '/pop',
                       samples_of_interest, args.c, args.p)
        if args.ind:
            if not os.path.exists(filt_folder + '/ind/'):
                os.makedirs(filt_folder + '/ind/')
            filter_two_all(ctx, args.all_samples, glob.glob(args.projdir + '/snpCaller/indiv*'), filt_folder + '/ind',
                           samples_of_interest, args.c, args.p)
    except core._lib.MsnvError as e:
        sys.exit("ERROR: {}".format(e))
    finally:
        ctx.close()


if __name__ == '__main__':
    main()
