"""ctypes binding of libmsnv.so (include/msnv.h).

The library is the product: HIP kernels for gfx950 behind a C ABI.  There is no Python or
CPU fallback -- if the shared object is missing this module raises at import time, and if no
HIP device is usable every compute entry point returns MSNV_ENODEV, surfaced as MsnvError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSNV_LIBRARY") or os.path.join(_HERE, "csrc", "libmsnv.so")   # MSNV_LIBRARY: developer A/B of two builds


class MsnvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmsnv error %d: %s" % (code, msg))
        self.code = code


OK, EINVAL, EIO, EFORMAT, ENODEV, EHIP, ENOMEM, EDOMAIN, ECAPACITY = range(9)


class Params(C.Structure):
    _fields_ = [("min_coverage", C.c_int32), ("calling_threshold", C.c_int32), ("min_fraction", C.c_double),
                ("min_baseq", C.c_int32), ("flag_filter", C.c_int32), ("count_orphans", C.c_int32),
                ("max_depth", C.c_int32), ("min_mapq", C.c_int32), ("drop_first_line", C.c_int32),
                ("cov_max", C.c_int32), ("cov_min_mapq", C.c_int32), ("ignore_overlaps", C.c_int32), ("token_limit", C.c_int32)]


class CovArgs(C.Structure):
    _fields_ = [("bam_path", C.c_char_p), ("max_cov", C.c_int32), ("min_mapq", C.c_int32),
                ("out_cov_path", C.c_char_p), ("out_detail_path", C.c_char_p)]


class CallArgs(C.Structure):
    _fields_ = [("bam_paths", C.POINTER(C.c_char_p)), ("n_bams", C.c_int32), ("ref_fasta", C.c_char_p),
                ("ann_path", C.c_char_p), ("bed_split_path", C.c_char_p), ("out_called_path", C.c_char_p),
                ("out_indiv_path", C.c_char_p), ("contig_rank_mask", C.POINTER(C.c_uint8)),
                ("n_contig_rank_mask", C.c_int32), ("host_threads", C.c_int32), ("params", Params)]


class MpileupArgs(C.Structure):
    _fields_ = [("text", C.c_char_p), ("text_bytes", C.c_uint64), ("mpileup_path", C.c_char_p), ("ref_fasta", C.c_char_p),
                ("ann_path", C.c_char_p), ("out_called_path", C.c_char_p), ("out_indiv_path", C.c_char_p), ("params", Params)]


class RefDesc(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("names", C.POINTER(C.c_char_p)), ("lengths", C.POINTER(C.c_int64)),
                ("seqs", C.POINTER(C.c_char_p)), ("seq_lens", C.POINTER(C.c_int64))]


class DatasetInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_samples", "n_contigs", "n_positions", "n_reads", "n_reads_pileup", "n_pileup_bases",
        "bytes_headers", "bytes_cigar", "bytes_seq", "bytes_qual", "bytes_ref", "bytes_index",
        "n_tiles", "n_pairs", "n_work", "device_bytes", "allele_planes", "sampled_mismatch_ppm", "n_whole_tile_items", "n_listed_tiles")]


class RunStats(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_pileup", C.c_float), ("ms_gate", C.c_float), ("ms_gather", C.c_float),
                ("ms_decide", C.c_float), ("ms_coverage", C.c_float), ("n_sites", C.c_uint64),
                ("n_called_pop", C.c_uint64), ("n_called_indiv", C.c_uint64), ("n_events", C.c_uint64),
                ("n_overflow", C.c_uint64), ("algorithmic_bytes", C.c_uint64)]


class Site(C.Structure):
    _fields_ = [("tid", C.c_int32), ("pos", C.c_int32), ("cov", C.c_uint32), ("n", C.c_uint32 * 4),
                ("pop_mask", C.c_uint8), ("ind_mask", C.c_uint8), ("refchar", C.c_uint8), ("dropped", C.c_uint8)]


class SiteSample(C.Structure):
    _fields_ = [("cov", C.c_uint16), ("n", C.c_uint16 * 4)]


class SiteAnn(C.Structure):
    _fields_ = [("gene", C.c_int32), ("codon", (C.c_uint8 * 8) * 4)]


class SampleStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("total_reads", "unmapped", "zero_quality", "proper_pairs", "duplicates", "any_mapped")]


COV_WORDS = 17


class FilterSpecies(C.Structure):
    _fields_ = [("species", C.c_char_p), ("n_soi", C.c_int32), ("soi", C.POINTER(C.c_int32)), ("soi_names", C.POINTER(C.c_char_p))]


class BamData(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("names", C.POINTER(C.c_char_p)), ("lengths", C.POINTER(C.c_int64)),
                ("records", C.POINTER(C.c_uint8)), ("n_record_bytes", C.c_uint64), ("header_text", C.c_char_p)]


class SynthParams(C.Structure):
    _fields_ = [("n_species", C.c_int32), ("contig_len", C.c_int64), ("n_samples", C.c_int32), ("read_len", C.c_int32),
                ("mean_cov", C.c_double), ("sigma_cov", C.c_double), ("frac_absent", C.c_double),
                ("snv_density", C.c_double), ("error_rate", C.c_double), ("frac_lowq", C.c_double),
                ("frac_indel_reads", C.c_double), ("frac_clip_reads", C.c_double), ("frac_flagged", C.c_double),
                ("lowercase_ref", C.c_int32), ("seed", C.c_uint64), ("frac_paired", C.c_double),
                ("contigs_per_species_max", C.c_int32), ("species_per_sample", C.c_int32),
                ("frac_aux", C.c_double), ("frac_noseq", C.c_double)]


# every symbol include/msnv.h declares: (name, restype, argtypes)
P = C.POINTER
_vp = C.c_void_p
SYMBOLS = [
    ("msnv_abi_version", C.c_int, []),
    ("msnv_last_error", C.c_char_p, []),
    ("msnv_device_count", C.c_int, []),
    ("msnv_ctx_create", C.c_int, [C.c_int, P(_vp)]),
    ("msnv_ctx_destroy", None, [_vp]),
    ("msnv_params_default", None, [P(Params)]),
    ("msnv_coverage", C.c_int, [_vp, P(CovArgs)]),
    ("msnv_call", C.c_int, [_vp, P(CallArgs)]),
    ("msnv_call_from_mpileup", C.c_int, [_vp, P(MpileupArgs), P(C.c_uint64)]),
    ("msnv_bam_records_many", C.c_int, [_vp, P(C.c_char_p), C.c_int32, C.c_int32, P(P(C.c_uint8)), P(C.c_uint64)]),
    ("msnv_bgzf_inflate", C.c_int, [_vp, C.c_char_p, C.c_int32, P(P(C.c_uint8)), P(C.c_uint64), P(C.c_uint64)]),
    ("msnv_dataset_create", C.c_int, [_vp, P(RefDesc), P(Params), P(_vp)]),
    ("msnv_dataset_create_from_files", C.c_int, [_vp, C.c_char_p, C.c_char_p, P(Params), P(_vp)]),
    ("msnv_dataset_attach_ctx", C.c_int, [_vp, _vp]),
    ("msnv_dataset_set_feed_ctx", C.c_int, [_vp, _vp]),
    ("msnv_dataset_destroy", None, [_vp]),
    ("msnv_dataset_set_bed", C.c_int, [_vp, C.c_int32, P(C.c_int32), P(C.c_int64), P(C.c_int64)]),
    ("msnv_dataset_set_bed_file", C.c_int, [_vp, C.c_char_p]),
    ("msnv_dataset_set_contig_mask", C.c_int, [_vp, P(C.c_uint8), C.c_int32]),
    ("msnv_dataset_add_sample_records", C.c_int, [_vp, _vp, C.c_uint64]),
    ("msnv_dataset_add_sample_records_many", C.c_int, [_vp, P(_vp), P(C.c_uint64), C.c_int32, C.c_int32]),
    ("msnv_dataset_stage_sample_bams", C.c_int, [_vp, P(C.c_char_p), C.c_int32, C.c_int32]),
    ("msnv_dataset_add_sample_records_device", C.c_int, [_vp, P(_vp), P(C.c_uint64), C.c_int32]),
    ("msnv_dataset_add_sample_records_resident", C.c_int, [_vp, _vp, C.c_uint64, P(C.c_uint64), P(C.c_uint64), C.c_int32]),
    ("msnv_dataset_pack_stats", C.c_int, [_vp, P(C.c_double), C.c_int32]),
    ("msnv_dataset_fetch_column", C.c_int, [_vp, C.c_char_p, _vp, C.c_uint64, P(C.c_uint64)]),
    ("msnv_dataset_add_sample_bam", C.c_int, [_vp, C.c_char_p]),
    ("msnv_dataset_add_sample_bams", C.c_int, [_vp, P(C.c_char_p), C.c_int32, C.c_int32]),
    ("msnv_dataset_pileup_qualities", C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    ("msnv_dataset_finalize", C.c_int, [_vp]),
    ("msnv_dataset_info_get", C.c_int, [_vp, P(DatasetInfo)]),
    ("msnv_pileup_run", C.c_int, [_vp, P(RunStats)]),
    ("msnv_pileup_run_many", C.c_int, [_vp, C.c_int32, C.c_int32, P(RunStats)]),
    ("msnv_pileup_reserve", C.c_int, [_vp, C.c_int32]),
    ("msnv_host_cores", C.c_int32, []),
    ("msnv_dataset_deal_bams_device", C.c_int, [_vp, P(C.c_char_p), C.c_int32, C.c_int32, P(C.c_int32), C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64,
                                                P(C.c_uint64), P(SampleStats), P(C.c_uint64)]),
    ("msnv_dataset_inflate_bams_device", C.c_int, [_vp, P(C.c_char_p), C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, P(C.c_uint64), P(C.c_uint64), P(SampleStats), P(C.c_uint64)]),
    ("msnv_records_deal_device", C.c_int, [_vp, P(C.c_void_p), P(C.c_uint64), C.c_int32, C.c_int32, P(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64,
                                           P(C.c_uint64), P(SampleStats), P(C.c_uint64)]),
    ("msnv_coverage_run", C.c_int, [_vp, P(RunStats)]),
    ("msnv_fused_run", C.c_int, [_vp, P(RunStats), P(RunStats)]),
    ("msnv_write_coverage", C.c_int, [_vp, C.c_int32, C.c_char_p, C.c_char_p]),
    ("msnv_write_calls", C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    ("msnv_dataset_first_line", C.c_int, [_vp, P(C.c_int32), P(C.c_int32)]),
    ("msnv_dataset_first_lines", C.c_int, [_vp, P(C.c_int32), P(C.c_int32), C.c_int32]),
    ("msnv_write_calls_records", C.c_int, [P(RefDesc), C.c_int32, P(Site), P(SiteSample), C.c_uint64, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, P(SiteAnn)]),
    ("msnv_records_partition", C.c_int, [_vp, C.c_uint64, P(C.c_int32), C.c_int32, C.c_int32, C.c_int32, _vp, P(C.c_uint64), P(SampleStats)]),
    ("msnv_dataset_sample_stats", C.c_int, [_vp, C.c_int32, P(SampleStats)]),
    ("msnv_coverage_fetch", C.c_int, [_vp, P(C.c_uint64), C.c_uint64]),
    ("msnv_coverage_rows_count", C.c_int, [_vp, P(C.c_uint64)]),
    ("msnv_coverage_fetch_rows", C.c_int, [_vp, P(C.c_uint32), P(C.c_uint32), P(C.c_uint64), C.c_uint64]),
    ("msnv_write_coverage_records", C.c_int, [P(RefDesc), C.c_int32, P(SampleStats), P(C.c_uint64), C.c_char_p, C.c_char_p]),
    ("msnv_annotate_run", C.c_int, [_vp, C.c_char_p, C.c_char_p, P(C.c_double)]),
    ("msnv_results_fetch_ann", C.c_int, [_vp, P(SiteAnn), C.c_uint64]),
    ("msnv_results_count", C.c_int, [_vp, P(C.c_uint64)]),
    ("msnv_results_fetch", C.c_int, [_vp, P(Site), P(SiteSample), C.c_uint64]),
    ("msnv_results_cells_count", C.c_int, [_vp, P(C.c_uint64), P(C.c_uint64)]),
    ("msnv_results_fetch_cells", C.c_int, [_vp, P(Site), P(C.c_uint64), P(C.c_uint32), P(SiteSample), C.c_uint64, C.c_uint64]),
    ("msnv_write_calls_cells", C.c_int, [P(RefDesc), C.c_int32, P(Site), P(C.c_uint64), P(C.c_uint32), P(SiteSample), C.c_uint64, C.c_char_p, C.c_char_p,
                                        C.c_char_p, C.c_char_p, P(SiteAnn)]),
    ("msnv_records_contig_bases", C.c_int, [_vp, C.c_uint64, C.c_int32, P(C.c_uint64)]),
    ("msnv_parse_float", C.c_int, [C.c_char_p, P(C.c_double)]),
    ("msnv_dist_file", C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, P(C.c_int32), P(C.c_uint64), P(C.c_double)]),
    ("msnv_genotyping_subset", C.c_int, [P(C.c_char_p), C.c_int32, P(C.c_char_p), C.c_int32, C.c_char_p, P(C.c_uint64), P(C.c_uint64)]),
    ("msnv_snv_allele_freq", C.c_int, [_vp, C.c_char_p, C.c_int32, P(C.c_uint64), P(C.c_double)]),
    ("msnv_format_float", C.c_int, [C.c_double, C.c_char_p, C.c_int32]),
    ("msnv_filter_files", C.c_int, [_vp, P(C.c_char_p), C.c_int32, C.c_int32, P(FilterSpecies), C.c_int32, C.c_double, C.c_double,
                                   C.c_char_p, P(C.c_uint64), P(C.c_double)]),
    ("msnv_host_stats", C.c_int, [P(C.c_uint64)]),
    ("msnv_host_timers", C.c_int, [P(C.c_double), C.c_int32, C.c_int32]),
    ("msnv_filter_resident", C.c_int, [_vp, C.c_int32, P(FilterSpecies), C.c_int32, C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_char_p,
                                      P(C.c_uint64), P(C.c_double)]),
    ("msnv_bam_write_bed_header", C.c_int, [C.c_char_p, C.c_char_p]),
    ("msnv_bam_read", C.c_int, [C.c_char_p, P(BamData)]),
    ("msnv_bam_read_header", C.c_int, [C.c_char_p, P(BamData)]),
    ("msnv_bam_data_free", None, [P(BamData)]),
    ("msnv_bam_write", C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, P(C.c_char_p), P(C.c_int64), _vp, C.c_uint64, C.c_int32]),
    ("msnv_synth_params_default", None, [P(SynthParams)]),
    ("msnv_synth_contig_count", C.c_int, [P(SynthParams)]),
    ("msnv_synth_reference", C.c_int, [P(SynthParams), P(P(C.c_char_p)), P(P(C.c_int64)), P(P(C.c_char_p))]),
    ("msnv_synth_sample", C.c_int, [P(SynthParams), C.c_int32, P(C.c_char_p), P(P(C.c_uint8)), P(C.c_uint64)]),
    ("msnv_dataset_add_synth_samples", C.c_int, [_vp, P(SynthParams), C.c_int32, C.c_int32, C.c_int32]),
    ("msnv_free", None, [_vp]),
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "metasnv_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("MSNV_DIST_FORCE") == "1":
        # multi-rank runs exchange tables through torch.distributed: torch ships its own libamdhip64, and two HIP
        # runtimes in one process do not both see the GPU.  Importing torch first makes libmsnv.so bind to the
        # runtime torch already loaded (same SONAME).
        import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)      # AttributeError here = the .so does not match include/msnv.h
        except AttributeError:
            if os.environ.get("MSNV_LIBRARY"):
                continue                 # developer A/B against an older build (profiles/ab.sh): newer entry points are simply absent
            raise
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise MsnvError(rc, (lib.msnv_last_error() or b"").decode("utf-8", "replace"))
