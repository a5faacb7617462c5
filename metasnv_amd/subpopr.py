"""subpopr's two raw-SNV consumers, same argv and the same files as the reference scripts:

  getGenotypingSNVSubset.py <hapDir> <metaSNVdir>     (src/subpopr/inst/getGenotypingSNVSubset.py)
      <hapDir>/*hap_positions.tab + <metaSNVdir>/snpCaller/called_SNPs*  ->  <hapDir>/<species>.pos
  convertSNVtoAlleleFreq.py <file.pos> <minDepth>     (src/subpopr/inst/convertSNVtoAlleleFreq.py)
      <file.pos>  ->  <file.pos>.freq

The first is a text filter (native, no device work); the frequencies of the second are computed on the GPU
(msnv_snv_allele_freq) and there is no CPU fallback for them."""
import ctypes as C
import glob
import sys


def _paths(lst):
    arr = (C.c_char_p * len(lst))(*[p.encode() for p in lst])
    return arr, len(lst)


def genotyping_subset(hap_paths, snp_paths, out_dir):
    """The lists are read in the order given; returns (#distinct positions wanted, #lines written)."""
    from ._lib import lib, check
    h, nh = _paths(hap_paths)
    s, ns = _paths(snp_paths)
    npos, nl = C.c_uint64(), C.c_uint64()
    check(lib.msnv_genotyping_subset(h, nh, s, ns, out_dir.encode(), C.byref(npos), C.byref(nl)))
    return npos.value, nl.value


def get_genotyping_snv_subset_main(argv=None):                 # getGenotypingSNVSubset.py:5-48
    argv = sys.argv[1:] if argv is None else argv
    hap_dir, metasnv_dir = argv[0], argv[1]
    print("Getting subspecies genotyping info from: " + hap_dir + '/*hap_positions.tab')
    print("Getting SNV for all data from raw SNV calls: " + metasnv_dir + '/snpCaller/called_SNPs*')
    haps = glob.glob(hap_dir + '/*hap_positions.tab')          # directory order, like the reference
    snps = glob.glob(metasnv_dir + '/snpCaller/called_SNPs*')
    if len(haps) < 1:
        sys.exit("Error: no *hap_positions.tab files")
    if len(snps) < 1:
        sys.exit("Error: no /snpCaller/called_SNPs* files in metaSNV output directory")
    from ._lib import MsnvError, EDOMAIN
    try:
        genotyping_subset(haps, snps, hap_dir)
    except MsnvError as e:
        if e.code == EDOMAIN:
            sys.exit("Error: no parse-able data in " + hap_dir + "/*hap_positions.tab files")
        sys.exit("Error: {}".format(e))


def snv_allele_freq(ctx, pos_path, min_depth):
    """Writes pos_path + '.freq'; returns (#allele rows, kernel ms)."""
    from ._lib import lib, check
    n, ms = C.c_uint64(), C.c_double()
    check(lib.msnv_snv_allele_freq(ctx._h, pos_path.encode(), int(min_depth), C.byref(n), C.byref(ms)))
    return n.value, ms.value


def convert_snv_to_allele_freq_main(argv=None):                # convertSNVtoAlleleFreq.py:3-27
    argv = sys.argv[1:] if argv is None else argv
    pos_path, min_depth = argv[0], int(argv[1])
    from . import core
    try:
        ctx = core.Context(0)
    except core._lib.MsnvError as e:
        sys.exit("\nERROR:  {}\n\nSOLUTION: run on a node with an AMD Instinct GPU (there is no CPU fallback)\n".format(e))
    try:
        snv_allele_freq(ctx, pos_path, min_depth)
    except core._lib.MsnvError as e:
        sys.exit("ERROR: {}".format(e))
    finally:
        ctx.close()
