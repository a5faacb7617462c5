"""metaSNV_DistDiv.py-compatible driver for `--dist` (reference: /metaSNV_DistDiv.py:105-139, 355-384) with the
pairwise distances computed on the GPU (msnv_dist_file, one call per species table).

Same argv and the same files: `--filt <proj>/filtered/pop` -> `<proj>/distances/<species>.filtered.mann.dist` and
`.allele.dist`.  The diversity / FST options (--div, --divNS, --matched) are not built (SURVEY.md section 8 lists only
the distances as row f3); asking for them is refused instead of silently skipped."""
import argparse
import ctypes as C
import glob
import os
import sys
from datetime import datetime


def build_parser():                                            # metaSNV_DistDiv.py:31-57
    p = argparse.ArgumentParser(prog='metaSNV_DistDiv.py', description='metaSNV distances and diversity computation',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--version', action='version', version='%(prog)s 2.0', help=argparse.SUPPRESS)
    p.add_argument("--debug", action="store_true", help=argparse.SUPPRESS)
    p.add_argument('--filt', metavar=': Filtered frequency files', help="Folder containing /pop/*.filtered.freq", required=True)
    p.add_argument('--dist', action='store_true', help="Compute distances")
    p.add_argument('--div', action='store_true', help="Compute Diversity and FST")
    p.add_argument('--divNS', action='store_true', help="Computing piN and piS")
    p.add_argument('--matched', action='store_true', help="Computing on matched positions only")
    p.add_argument('--n_threads', metavar=': Number of Processes', default=1, type=int, help="Number of jobs to run simultaneously.")
    return p


def file_check(args):                                          # metaSNV_DistDiv.py:62-78 (same messages, same exit)
    parts = args.filt.rstrip('/').split('/')
    args.projdir = '/'.join(parts[:-2])                        # .../<proj>/filtered<pars>/pop -> <proj>
    args.pars = parts[-2].strip('filtered')
    stem = args.projdir + '/' + args.projdir.split('/')[-1]
    args.coverage_file, args.percentage_file = stem + '.all_cov.tab', stem + '.all_perc.tab'
    args.bedfile = args.projdir + '/' + 'bed_header'
    needed = (args.coverage_file, args.percentage_file, args.bedfile)
    print("Checking for necessary input files...")
    if not all(os.path.isfile(n) for n in needed):
        sys.exit("\nERROR: No such file '{}',\nERROR: No such file '{}',\nERROR: No such file '{}'".format(*needed))
    print("found: '{}' \nfound:'{}' \nfound:'{}'".format(*needed))


def compute_dist(ctx, filt_file, outdir, threshold=.6):        # metaSNV_DistDiv.py:113-124
    from ._lib import lib, check
    species = filt_file.split('/')[-1].replace('.freq', '')
    ns, npos, ms = C.c_int32(), C.c_uint64(), C.c_double()
    check(lib.msnv_dist_file(ctx._h, filt_file.encode(), (outdir + '/' + '%s.mann.dist' % species).encode(),
                             (outdir + '/' + '%s.allele.dist' % species).encode(), threshold, C.byref(ns), C.byref(npos), C.byref(ms)))
    return ns.value, npos.value, ms.value


def main(argv=None):                                           # metaSNV_DistDiv.py:355-384
    args = build_parser().parse_args(argv)
    file_check(args)
    if args.div or args.divNS or args.matched:
        sys.exit("ERROR: --div / --divNS / --matched are not built in this GPU port (only --dist is); run the reference script for them")
    outdir = args.projdir + '/distances' + args.pars + '/'
    if not os.path.exists(outdir):
        os.makedirs(outdir)
    print("Starting computations: ", datetime.now())
    if args.dist:
        print("Computing distances")
        from . import core
        try:
            ctx = core.Context(0)
        except core._lib.MsnvError as e:
            sys.exit("\nERROR:  {}\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n".format(e))
        try:
            for f in glob.glob(args.filt + '/*.freq'):
                compute_dist(ctx, f, outdir)
        except core._lib.MsnvError as e:
            sys.exit("ERROR: {}".format(e))
        finally:
            ctx.close()
    print("Computations complete: ", datetime.now())


if __name__ == '__main__':
    main()
