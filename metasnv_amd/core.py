"""Thin object layer over the C ABI (metasnv_amd/_lib.py).  No compute happens in Python."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check, Params, SynthParams, DatasetInfo, RunStats, Site, SiteSample, SiteAnn, RefDesc, SampleStats, COV_WORDS

SITE_DTYPE = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("cov", "<u4"), ("n", "<u4", (4,)),
                       ("pop_mask", "u1"), ("ind_mask", "u1"), ("refchar", "u1"), ("dropped", "u1")])
SAMPLE_DTYPE = np.dtype([("cov", "<u2"), ("n", "<u2", (4,))])
ANN_DTYPE = np.dtype([("gene", "<i4"), ("codon", "u1", (4, 8))])
STATS_FIELDS = [n for n, _ in SampleStats._fields_]
assert SITE_DTYPE.itemsize == C.sizeof(Site) and SAMPLE_DTYPE.itemsize == C.sizeof(SiteSample) and ANN_DTYPE.itemsize == C.sizeof(SiteAnn)


def default_params(**kw):
    p = Params()
    lib.msnv_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError("msnv_params has no field %r" % k)
        setattr(p, k, v)
    return p


def host_cores():
    """Cores' worth of CPU time the process may use (msnv_host_cores: hardware threads or the container's cgroup quota)."""
    return int(lib.msnv_host_cores())


def device_count():
    return lib.msnv_device_count()


class Context:
    """One per GPU (one per rank)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib.msnv_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            lib.msnv_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _cstr_array(strings):
    arr = (C.c_char_p * len(strings))()
    arr[:] = [s if isinstance(s, bytes) else s.encode() for s in strings]
    return arr


class Dataset:
    """Packed read columns of one shard, resident in HBM after finalize()."""

    def __init__(self, ctx, names, lengths, seqs=None, params=None):
        self.ctx = ctx
        self.names = [n.decode() if isinstance(n, bytes) else n for n in names]
        self.lengths = [int(x) for x in lengths]
        n = len(names)
        self._names = _cstr_array(names)
        self._lengths = (C.c_int64 * n)(*lengths)
        seqs = seqs if seqs is not None else [None] * n
        self._seqs_keep = [None if s is None else (s if isinstance(s, bytes) else s.encode()) for s in seqs]
        self._seqs = (C.c_char_p * n)(*self._seqs_keep)
        self._seq_lens = (C.c_int64 * n)(*[0 if s is None else len(s) for s in self._seqs_keep])
        rd = RefDesc(n, self._names, self._lengths, self._seqs, self._seq_lens)
        self.params = params or default_params()
        self._h = C.c_void_p()
        self.n_samples = 0
        check(lib.msnv_dataset_create(ctx._h if ctx is not None else None, C.byref(rd), C.byref(self.params), C.byref(self._h)))

    @classmethod
    def from_files(cls, ctx, first_bam, fasta, params=None):
        self = cls.__new__(cls)
        self.ctx = ctx
        self.params = params or default_params()
        self._h = C.c_void_p()
        self.n_samples = 0
        check(lib.msnv_dataset_create_from_files(ctx._h if ctx is not None else None, first_bam.encode(), fasta.encode() if fasta else None,
                                                 C.byref(self.params), C.byref(self._h)))
        hdr = read_bam(first_bam, records=False)
        self.names, self.lengths = hdr["names"], hdr["lengths"]
        return self

    def attach_context(self, ctx):
        """Gives a dataset created with ctx=None its device context (before finalize)."""
        check(lib.msnv_dataset_attach_ctx(self._h, ctx._h))
        self.ctx = ctx

    def set_feed_context(self, ctx):
        """A second context of the dataset's device for deal_bams_device / inflate_bams_device (None: back to the dataset's own): the N-rank
        feed decodes round k + 1 through it on a thread of its own while this dataset's context packs round k."""
        check(lib.msnv_dataset_set_feed_ctx(self._h, ctx._h if ctx is not None else None))

    def set_bed(self, regions):
        """regions: iterable of (tid, beg, end), 0-based half-open (mpileup -l)."""
        regions = list(regions)
        n = len(regions)
        t = (C.c_int32 * n)(*[r[0] for r in regions])
        b = (C.c_int64 * n)(*[r[1] for r in regions])
        e = (C.c_int64 * n)(*[r[2] for r in regions])
        check(lib.msnv_dataset_set_bed(self._h, n, t, b, e))

    def set_bed_file(self, path):
        check(lib.msnv_dataset_set_bed_file(self._h, path.encode()))

    def set_contig_mask(self, mask):
        m = (C.c_uint8 * len(mask))(*[1 if x else 0 for x in mask])
        check(lib.msnv_dataset_set_contig_mask(self._h, m, len(mask)))

    def add_sample_records(self, records):
        """records: bytes / numpy uint8 array of raw BAM alignment records."""
        if isinstance(records, np.ndarray):
            buf = np.ascontiguousarray(records, dtype=np.uint8)
            ptr, n = buf.ctypes.data, buf.size
        else:
            buf = bytes(records)
            ptr, n = C.cast(C.c_char_p(buf), C.c_void_p), len(buf)
        check(lib.msnv_dataset_add_sample_records(self._h, ptr, n))
        self.n_samples += 1

    def add_samples_records(self, streams, host_threads=0):
        """Several record streams (numpy uint8 arrays) as consecutive samples, packed by a pool of host threads."""
        bufs = [np.ascontiguousarray(r, dtype=np.uint8) for r in streams]
        n = len(bufs)
        if n == 0:
            return
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_uint64 * n)(*[b.size for b in bufs])
        check(lib.msnv_dataset_add_sample_records_many(self._h, ptrs, sizes, n, host_threads))
        self.n_samples += n

    def deal_bams_device(self, paths, contig_owner, n_parts, out_ptr, capacity, gap=0, cov_min_mapq=1, host_threads=0):
        """BAM files inflated on the device and their records dealt from there into out_ptr (DEVICE memory; msnv_dataset_deal_bams_device).
        Returns (part_bytes[n][n_parts] int64, stats[n][6] uint32, record_bytes[n] int64)."""
        n = len(paths)
        owner = np.ascontiguousarray(contig_owner, dtype=np.int32)
        if owner.size != len(self.names):
            raise ValueError("contig_owner has %d entries, the dataset %d contigs" % (owner.size, len(self.names)))
        pb = (C.c_uint64 * max(1, n * n_parts))()
        st = (SampleStats * max(1, n))()
        rb = (C.c_uint64 * max(1, n))()
        check(lib.msnv_dataset_deal_bams_device(self._h, _cstr_array(paths), n, int(host_threads), owner.ctypes.data_as(C.POINTER(C.c_int32)), int(n_parts), int(cov_min_mapq),
                                                C.c_void_p(int(out_ptr)), int(capacity), int(gap), pb, st, rb))
        parts = np.array(list(pb)[:n * n_parts], dtype=np.int64).reshape(n, n_parts) if n else np.zeros((0, n_parts), np.int64)
        stats = np.array([[getattr(st[i], k) for k in STATS_FIELDS] for i in range(n)], dtype=np.uint32).reshape(n, len(STATS_FIELDS))
        return parts, stats, np.array(list(rb)[:n], dtype=np.int64)

    def inflate_bams_device(self, paths, out_ptr, capacity, contig_bases=None, host_threads=0):
        """BAM files inflated and checked on the device, their record streams left in out_ptr (DEVICE memory; msnv_dataset_inflate_bams_device).
        Returns (offsets[n], sizes[n] as int64 arrays, stats[n][6] uint32); contig_bases (uint64[n_contigs]) += aligned bases per contig."""
        n = len(paths)
        ro = (C.c_uint64 * max(1, n))(); rb = (C.c_uint64 * max(1, n))()
        st = (SampleStats * max(1, n))()
        cb = contig_bases.ctypes.data_as(C.POINTER(C.c_uint64)) if contig_bases is not None else None
        check(lib.msnv_dataset_inflate_bams_device(self._h, _cstr_array(paths), n, int(host_threads), C.c_void_p(int(out_ptr)), int(capacity), ro, rb, st, cb))
        stats = np.array([[getattr(st[i], k) for k in STATS_FIELDS] for i in range(n)], dtype=np.uint32).reshape(n, len(STATS_FIELDS))
        return np.array(list(ro)[:n], dtype=np.int64), np.array(list(rb)[:n], dtype=np.int64), stats

    def add_samples_records_device(self, ptrs, sizes):
        """Record streams that lie in HBM of this dataset's device (device addresses + byte counts, e.g. slices of a torch tensor
        an all-to-all has just filled) as consecutive samples: parsed, filtered and packed by kernels (msnv_dataset_add_sample_records_device)."""
        n = len(ptrs)
        if n == 0:
            return
        pa = (C.c_void_p * n)(*[int(p) for p in ptrs])
        sa = (C.c_uint64 * n)(*[int(x) for x in sizes])
        check(lib.msnv_dataset_add_sample_records_device(self._h, pa, sa, n))
        self.n_samples += n

    def add_samples_records_resident(self, dev_buffer, capacity, offsets, sizes):
        """Record streams inside ONE device buffer (16-byte aligned, 256 readable bytes behind the last stream), read where they lie --
        no copy; qualities may be edited in place (msnv_dataset_add_sample_records_resident)."""
        n = len(offsets)
        if n == 0:
            return
        oa = (C.c_uint64 * n)(*[int(x) for x in offsets])
        sa = (C.c_uint64 * n)(*[int(x) for x in sizes])
        check(lib.msnv_dataset_add_sample_records_resident(self._h, C.c_void_p(int(dev_buffer)), int(capacity), oa, sa, n))
        self.n_samples += n

    def pileup_qualities(self, records):
        """The record stream with the base qualities as the pileup engine sees them (overlap tweak, token limit)."""
        rec = np.ascontiguousarray(records, dtype=np.uint8)
        out = np.empty_like(rec)
        check(lib.msnv_dataset_pileup_qualities(self._h, rec.ctypes.data, rec.size, out.ctypes.data))
        return out

    def add_sample_bam(self, path):
        check(lib.msnv_dataset_add_sample_bam(self._h, path.encode()))
        self.n_samples += 1

    def add_sample_bams(self, paths, host_threads=0):
        arr = _cstr_array(paths)
        check(lib.msnv_dataset_add_sample_bams(self._h, arr, len(paths), host_threads))
        self.n_samples += len(paths)

    def stage_sample_bams(self, paths, host_threads=0):
        """Reads and inflates the BAMs now, packs them in finalize() (msnv_dataset_stage_sample_bams): on the device when the dataset has
        its context by then (attach_context)."""
        arr = _cstr_array(paths)
        check(lib.msnv_dataset_stage_sample_bams(self._h, arr, len(paths), host_threads))
        self.n_samples += len(paths)

    def add_synth_samples(self, synth_p, first, count, host_threads=0):
        check(lib.msnv_dataset_add_synth_samples(self._h, C.byref(synth_p), first, count, host_threads))
        self.n_samples += count

    def finalize(self):
        check(lib.msnv_dataset_finalize(self._h))
        return self.info()

    def info(self):
        i = DatasetInfo()
        check(lib.msnv_dataset_info_get(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in DatasetInfo._fields_}

    PACK_STATS = ["scan_ms", "measure_ms", "depth_ms", "emit_ms", "tile_sort_ms", "upload_wall_s", "download_wall_s", "host_prepass_wall_s",
                  "record_bytes", "records", "pieces", "prepass_samples", "scan_segments_redone", "deep_runs_split", "dense_samples", "quick_rounds_redone", "device_edit_samples"]

    def pack_stats(self):
        """Cost of the per-read stage on the device so far (msnv_dataset_pack_stats); all zero for a host-packed dataset."""
        a = (C.c_double * len(self.PACK_STATS))()
        check(lib.msnv_dataset_pack_stats(self._h, a, len(self.PACK_STATS)))
        return dict(zip(self.PACK_STATS, [float(x) for x in a]))

    def column(self, name):
        """Bytes of a device column / index table of the finalized dataset (msnv_dataset_fetch_column) as a uint8 array."""
        n = C.c_uint64()
        check(lib.msnv_dataset_fetch_column(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        if n.value:
            check(lib.msnv_dataset_fetch_column(self._h, name.encode(), out.ctypes.data, out.size, C.byref(n)))
        return out

    def run(self):
        st = RunStats()
        check(lib.msnv_pileup_run(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in RunStats._fields_}

    def reserve_passes(self, n):
        """Creates what a batch of n passes needs besides the passes (events, pinned counters: msnv_pileup_reserve)."""
        check(lib.msnv_pileup_reserve(self._h, int(n)))

    def run_many(self, n, overlap=False):
        """n passes back to back, one host synchronisation; returns the list of per-pass stats.  overlap=True runs
        consecutive passes on two streams (the tail kernels of one pass under the pileup kernel of the next)."""
        arr = (RunStats * n)()
        check(lib.msnv_pileup_run_many(self._h, n, 1 if overlap else 0, arr))
        return [{k: getattr(s, k) for k, _ in RunStats._fields_} for s in arr]

    def coverage_run(self):
        st = RunStats()
        check(lib.msnv_coverage_run(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in RunStats._fields_}

    def fused_run(self):
        """Coverage (qaCompute) and SNV calling over the same resident columns; returns (pileup stats, coverage stats)."""
        sp, sc = RunStats(), RunStats()
        check(lib.msnv_fused_run(self._h, C.byref(sp), C.byref(sc)))
        return ({k: getattr(sp, k) for k, _ in RunStats._fields_}, {k: getattr(sc, k) for k, _ in RunStats._fields_})

    def coverage_accumulators(self):
        """[n_samples][n_contigs][COV_WORDS] uint64 of the last coverage run (zeros for contigs outside the shard)."""
        acc = np.zeros((max(1, self.n_samples), len(self.names), COV_WORDS), dtype=np.uint64)
        check(lib.msnv_coverage_fetch(self._h, acc.ctypes.data_as(C.POINTER(C.c_uint64)), acc.size))
        return acc

    def coverage_rows(self):
        """The accumulators of the last coverage run as rows (msnv_coverage_fetch_rows): (sample[n], contig[n], acc[n][COV_WORDS]),
        one row per (sample, contig) with reads on this rank, in (sample, contig) order."""
        n = C.c_uint64()
        check(lib.msnv_coverage_rows_count(self._h, C.byref(n)))
        smp, ctg = np.zeros(n.value, dtype=np.uint32), np.zeros(n.value, dtype=np.uint32)
        acc = np.zeros((n.value, COV_WORDS), dtype=np.uint64)
        check(lib.msnv_coverage_fetch_rows(self._h, smp.ctypes.data_as(C.POINTER(C.c_uint32)), ctg.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           acc.ctypes.data_as(C.POINTER(C.c_uint64)), n.value))
        return smp, ctg, acc

    def sample_stats(self, sample_idx):
        st = SampleStats()
        check(lib.msnv_dataset_sample_stats(self._h, sample_idx, C.byref(st)))
        return np.array([getattr(st, k) for k in STATS_FIELDS], dtype=np.uint32)

    def write_coverage(self, sample_idx, cov_path, detail_path):
        check(lib.msnv_write_coverage(self._h, sample_idx, cov_path.encode(), detail_path.encode()))

    def results(self):
        n = C.c_uint64()
        check(lib.msnv_results_count(self._h, C.byref(n)))
        sites = np.zeros(n.value, dtype=SITE_DTYPE)
        samples = np.zeros((n.value, max(1, self.n_samples)), dtype=SAMPLE_DTYPE)
        check(lib.msnv_results_fetch(self._h, sites.ctypes.data_as(C.POINTER(Site)),
                                     samples.ctypes.data_as(C.POINTER(SiteSample)), n.value))
        return sites, samples

    def results_cells(self):
        """The records of the last run with the per-sample part as rows of cells (msnv_results_fetch_cells):
        (sites, row_off[n_sites + 1], cell_sample, cells) -- site i owns cells row_off[i]:row_off[i + 1], samples without a
        cell hold zeros."""
        ns, nc = C.c_uint64(), C.c_uint64()
        check(lib.msnv_results_cells_count(self._h, C.byref(ns), C.byref(nc)))
        sites = np.zeros(ns.value, dtype=SITE_DTYPE)
        row_off = np.zeros(ns.value + 1, dtype=np.uint64)
        cell_sample = np.zeros(nc.value, dtype=np.uint32)
        cells = np.zeros(nc.value, dtype=SAMPLE_DTYPE)
        check(lib.msnv_results_fetch_cells(self._h, sites.ctypes.data_as(C.POINTER(Site)), row_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           cell_sample.ctypes.data_as(C.POINTER(C.c_uint32)), cells.ctypes.data_as(C.POINTER(SiteSample)), ns.value, nc.value))
        return sites, row_off, cell_sample, cells

    def annotate(self, ann_path, fasta_path):
        """Gene / codon annotation of the last run's sites on the device; returns (records, kernel ms)."""
        ms = C.c_double()
        check(lib.msnv_annotate_run(self._h, ann_path.encode(), fasta_path.encode(), C.byref(ms)))
        n = C.c_uint64()
        check(lib.msnv_results_count(self._h, C.byref(n)))
        ann = np.zeros(n.value, dtype=ANN_DTYPE)
        check(lib.msnv_results_fetch_ann(self._h, ann.ctypes.data_as(C.POINTER(SiteAnn)), n.value))
        return ann, ms.value

    def first_line(self):
        t, p = C.c_int32(), C.c_int32()
        check(lib.msnv_dataset_first_line(self._h, C.byref(t), C.byref(p)))
        return t.value, p.value

    def filter_resident(self, species, out_dir, min_cov_c=5.0, min_prop_p=0.5, ind=False, ann_path=None, fasta_path=None):
        """metaSNV_Filtering.py filter_two straight from the records of the last run (no called_SNPs text round trip).
        species: list of (taxid, [sample indices of interest], [their names]).  Returns (positions kept, kernel ms)."""
        from ._lib import FilterSpecies
        keep = []
        arr = (FilterSpecies * len(species))()
        for i, (name, idx, names) in enumerate(species):
            ia = (C.c_int32 * len(idx))(*idx)
            na = _cstr_array(names)
            keep += [ia, na]
            arr[i] = FilterSpecies(name.encode(), len(idx), ia, na)
        n, ms = C.c_uint64(), C.c_double()
        check(lib.msnv_filter_resident(self._h, 1 if ind else 0, arr, len(species), float(min_cov_c), float(min_prop_p), out_dir.encode(),
                                       ann_path.encode() if ann_path else None, fasta_path.encode() if fasta_path else None, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def first_lines(self):
        """Per contig: first pileup line without -l / under metaSNV's `name 1 LEN` split BED (-1 = none); int32 arrays."""
        n = len(self.names)
        a, b = np.zeros(n, np.int32), np.zeros(n, np.int32)
        check(lib.msnv_dataset_first_lines(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), b.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return a, b

    def write_calls(self, called_path, indiv_path=None, ann_path=None, fasta_path=None):
        check(lib.msnv_write_calls(self._h, called_path.encode(), indiv_path.encode() if indiv_path else None,
                                   ann_path.encode() if ann_path else None, fasta_path.encode() if fasta_path else None))

    def close(self):
        if self._h:
            lib.msnv_dataset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_calls_records(names, n_samples, sites, samples, called_path, indiv_path=None, ann_path=None, fasta_path=None, ann=None):
    """Format gathered site records (numpy arrays of SITE_DTYPE / SAMPLE_DTYPE) as called_SNPs / indiv_called.
    With ann_path the gathered device annotation records (ANN_DTYPE, Dataset.annotate) are required."""
    n = len(names)
    arr = _cstr_array(names)
    lens = (C.c_int64 * n)(*([0] * n))
    rd = RefDesc(n, arr, lens, None, None)
    sites = np.ascontiguousarray(sites, dtype=SITE_DTYPE)
    samples = np.ascontiguousarray(samples, dtype=SAMPLE_DTYPE).reshape(len(sites), max(1, n_samples) if len(sites) else 0)
    check(lib.msnv_write_calls_records(C.byref(rd), n_samples, sites.ctypes.data_as(C.POINTER(Site)),
                                       samples.ctypes.data_as(C.POINTER(SiteSample)), len(sites), called_path.encode(),
                                       indiv_path.encode() if indiv_path else None, ann_path.encode() if ann_path else None,
                                       fasta_path.encode() if fasta_path else None,
                                       np.ascontiguousarray(ann, dtype=ANN_DTYPE).ctypes.data_as(C.POINTER(SiteAnn)) if ann is not None else None))


def write_calls_cells(names, n_samples, sites, row_off, cell_sample, cells, called_path, indiv_path=None, ann_path=None, fasta_path=None, ann=None):
    """write_calls_records for records in the cell form (Dataset.results_cells / parallel.gather_sites_root)."""
    n = len(names)
    rd = RefDesc(n, _cstr_array(names), (C.c_int64 * n)(*([0] * n)), None, None)
    sites = np.ascontiguousarray(sites, dtype=SITE_DTYPE)
    row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
    cell_sample = np.ascontiguousarray(cell_sample, dtype=np.uint32)
    cells = np.ascontiguousarray(cells, dtype=SAMPLE_DTYPE)
    assert row_off.size == len(sites) + 1
    check(lib.msnv_write_calls_cells(C.byref(rd), n_samples, sites.ctypes.data_as(C.POINTER(Site)), row_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     cell_sample.ctypes.data_as(C.POINTER(C.c_uint32)), cells.ctypes.data_as(C.POINTER(SiteSample)), len(sites),
                                     called_path.encode(), indiv_path.encode() if indiv_path else None, ann_path.encode() if ann_path else None,
                                     fasta_path.encode() if fasta_path else None,
                                     np.ascontiguousarray(ann, dtype=ANN_DTYPE).ctypes.data_as(C.POINTER(SiteAnn)) if ann is not None else None))


def cells_to_dense(n_samples, row_off, cell_sample, cells):
    """[sites][n_samples] array of SAMPLE_DTYPE from the cell form (tests, small results)."""
    n = len(row_off) - 1
    out = np.zeros((n, max(1, n_samples)), dtype=SAMPLE_DTYPE)
    site_of = np.repeat(np.arange(n), np.diff(row_off.astype(np.int64)))
    out[site_of, cell_sample] = cells
    return out


def dense_to_cells(samples):
    """(row_off, cell_sample, cells) of a dense [sites][n_samples] SAMPLE_DTYPE array: the entries that hold anything."""
    samples = np.asarray(samples)
    if samples.size == 0:
        return np.zeros(len(samples) + 1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, SAMPLE_DTYPE)
    nz = (samples["cov"] != 0) | (samples["n"] != 0).any(axis=-1)
    site, smp = np.nonzero(nz)
    row_off = np.concatenate([[0], np.cumsum(nz.sum(axis=1))]).astype(np.uint64)
    return row_off, smp.astype(np.uint32), np.ascontiguousarray(samples[site, smp])


def deal_records_device(ctx, streams, contig_owner, n_parts, out_ptr, capacity, gap=0, cov_min_mapq=1, contig_bases=None, on_device=False, sizes=None):
    """core.partition_records for several streams at once, on the device (msnv_records_deal_device): streams = uint8 arrays (or device addresses
    with sizes= and on_device=True); out_ptr = DEVICE address of `capacity` bytes that receives the parts destination-major with `gap` free bytes
    in front of every part.  Returns (part_bytes[n_streams][n_parts] as int64 array, stats[n_streams][6] as uint32 array)."""
    n = len(streams)
    owner = np.ascontiguousarray(contig_owner, dtype=np.int32)
    if on_device:
        keep, ptrs, nb = None, [int(p) for p in streams], [int(x) for x in sizes]
    else:
        keep = [np.ascontiguousarray(r, dtype=np.uint8) for r in streams]
        ptrs, nb = [b.ctypes.data for b in keep], [b.size for b in keep]
    pa = (C.c_void_p * max(1, n))(*ptrs)
    sa = (C.c_uint64 * max(1, n))(*nb)
    pb = (C.c_uint64 * max(1, n * n_parts))()
    st = (SampleStats * max(1, n))()
    cb = contig_bases.ctypes.data_as(C.POINTER(C.c_uint64)) if contig_bases is not None else None
    check(lib.msnv_records_deal_device(ctx._h, pa, sa, n, 1 if on_device else 0, owner.ctypes.data_as(C.POINTER(C.c_int32)), owner.size, int(n_parts), int(cov_min_mapq),
                                       C.c_void_p(int(out_ptr)), int(capacity), int(gap), pb, st, cb))
    parts = np.array(list(pb)[:n * n_parts], dtype=np.int64).reshape(n, n_parts) if n else np.zeros((0, n_parts), np.int64)
    stats = np.array([[getattr(st[i], k) for k in STATS_FIELDS] for i in range(n)], dtype=np.uint32).reshape(n, len(STATS_FIELDS))
    return parts, stats


def contig_bases(records, n_contigs, into=None):
    """Aligned bases per contig of a raw record stream (msnv_records_contig_bases), added to `into` (uint64[n_contigs])."""
    rec = np.ascontiguousarray(records, dtype=np.uint8)
    out = into if into is not None else np.zeros(n_contigs, dtype=np.uint64)
    check(lib.msnv_records_contig_bases(rec.ctypes.data, rec.size, n_contigs, out.ctypes.data_as(C.POINTER(C.c_uint64))))
    return out


def write_coverage_records(names, lengths, max_cov, stats, acc, cov_path, detail_path):
    """OUT / OUT.detail of one sample from gathered accumulators acc[n_contigs][COV_WORDS] and its statistics (STATS_FIELDS order)."""
    n = len(names)
    rd = RefDesc(n, _cstr_array(names), (C.c_int64 * n)(*[int(x) for x in lengths]), None, None)
    st = SampleStats(*[int(x) for x in stats])
    a = np.ascontiguousarray(acc, dtype=np.uint64).reshape(n, COV_WORDS)
    check(lib.msnv_write_coverage_records(C.byref(rd), int(max_cov), C.byref(st), a.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          cov_path.encode(), detail_path.encode()))


def partition_records(records, contig_owner, n_parts, cov_min_mapq=1):
    """Deals a sample's raw record stream (uint8 array) to n_parts parts by contig owner.  Returns (list of uint8 arrays, stats)."""
    rec = np.ascontiguousarray(records, dtype=np.uint8)
    owner = np.ascontiguousarray(contig_owner, dtype=np.int32)
    out = np.empty(rec.size, dtype=np.uint8)
    sizes = (C.c_uint64 * n_parts)()
    st = SampleStats()
    check(lib.msnv_records_partition(rec.ctypes.data, rec.size, owner.ctypes.data_as(C.POINTER(C.c_int32)), owner.size, n_parts,
                                     int(cov_min_mapq), out.ctypes.data, sizes, C.byref(st)))
    parts, o = [], 0
    for k in range(n_parts):
        parts.append(out[o:o + sizes[k]])
        o += sizes[k]
    return parts, np.array([getattr(st, k) for k in STATS_FIELDS], dtype=np.uint32)


# ------------------------------------------------------------------------------------ host I/O helpers
def read_bam(path, records=True):
    d = _lib.BamData()
    check((lib.msnv_bam_read if records else lib.msnv_bam_read_header)(path.encode(), C.byref(d)))
    try:
        out = {"names": [d.names[i].decode() for i in range(d.n_contigs)],
               "lengths": [int(d.lengths[i]) for i in range(d.n_contigs)],
               "header_text": (d.header_text or b"").decode()}
        if records:
            out["records"] = np.ctypeslib.as_array(d.records, shape=(d.n_record_bytes,)).copy() if d.n_record_bytes else np.zeros(0, np.uint8)
        return out
    finally:
        lib.msnv_bam_data_free(C.byref(d))


def call_from_mpileup(ctx, called_path, indiv_path=None, text=None, mpileup_path=None, fasta=None, ann=None, params=None):
    """snpCall on mpileup TEXT (msnv_call_from_mpileup: `snpCall -f fasta [-g ann] -i indiv_path ... < text > called_path`),
    parsed and called on the device.  Returns the stats of include/msnv.h as a dict."""
    a = _lib.MpileupArgs()
    if text is not None:
        raw = text if isinstance(text, (bytes, bytearray)) else text.encode("latin-1")
        a.text = C.c_char_p(bytes(raw)); a.text_bytes = len(raw)
    elif mpileup_path is not None:
        a.mpileup_path = mpileup_path.encode()
    a.ref_fasta = fasta.encode() if fasta else None
    a.ann_path = ann.encode() if ann else None
    a.out_called_path = called_path.encode()
    a.out_indiv_path = indiv_path.encode() if indiv_path else None
    a.params = params or default_params()
    st = (C.c_uint64 * 8)()
    check(lib.msnv_call_from_mpileup(ctx._h, C.byref(a), st))
    return {"lines": int(st[0]), "samples": int(st[1]), "called_lines": int(st[2]), "indiv_lines": int(st[3]), "kernel_ms": st[4] / 1000.0,
            "text_bytes": int(st[5]), "base_chars": int(st[6])}


HOST_TIMERS = ["read_s", "inflate_host_s", "inflate_device_wall_s", "pack_s", "finalize_upload_wall_s", "format_wall_s", "add_bams_wall_s", "pack_device_wall_s", "synth_wall_s"]


def host_timers(reset=False):
    """Cumulative host-stage seconds (msnv_host_timers): thread-seconds for read / host inflate / pack, wall for the others."""
    a = (C.c_double * len(HOST_TIMERS))()
    check(lib.msnv_host_timers(a, len(HOST_TIMERS), 1 if reset else 0))
    return dict(zip(HOST_TIMERS, [float(x) for x in a]))


def read_bam_records(paths, ctx=None, threads=0):
    """Record streams (uint8 arrays) of several BAM files in one call: BGZF blocks inflated on the device when ctx is given and the
    files are large enough (msnv_bam_records_many), else one host thread per file."""
    n = len(paths)
    if n == 0:
        return []
    recs = (C.POINTER(C.c_uint8) * n)()
    sizes = (C.c_uint64 * n)()
    check(lib.msnv_bam_records_many(ctx._h if ctx is not None else None, _cstr_array(paths), n, threads, recs, sizes))
    out = []
    try:
        for i in range(n):
            out.append(np.ctypeslib.as_array(recs[i], shape=(sizes[i],)).copy() if sizes[i] else np.zeros(0, np.uint8))
    finally:
        for i in range(n):
            lib.msnv_free(recs[i])
    return out


def bgzf_inflate(path, ctx=None):
    """Inflated bytes of a BGZF file through the device (ctx given) or the host decoder; returns (bytes as a numpy array, counters)."""
    out = C.POINTER(C.c_uint8)()
    n = C.c_uint64()
    cnt = (C.c_uint64 * 4)()
    check(lib.msnv_bgzf_inflate(ctx._h if ctx is not None else None, path.encode(), 1 if ctx is not None else 0, C.byref(out), C.byref(n), cnt))
    try:
        data = np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)
    finally:
        lib.msnv_free(out)
    return data, {"blocks": int(cnt[0]), "host_blocks": int(cnt[1]), "kernel_ms": cnt[2] / 1000.0, "bytes": int(cnt[3])}


def write_bam(path, names, lengths, records, header_text=None, level=1):
    rec = np.ascontiguousarray(records, dtype=np.uint8)
    n = len(names)
    check(lib.msnv_bam_write(path.encode(), header_text.encode() if header_text else None, n, _cstr_array(names),
                             (C.c_int64 * n)(*lengths), rec.ctypes.data, rec.size, level))


def write_bed_header(bam_path, out_path):
    check(lib.msnv_bam_write_bed_header(bam_path.encode(), out_path.encode()))


# ------------------------------------------------------------------------------------ synthetic workload
def synth_params(**kw):
    p = SynthParams()
    lib.msnv_synth_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError("msnv_synth_params has no field %r" % k)
        setattr(p, k, v)
    return p


class Synth:
    """Deterministic synthetic reference + per-sample raw BAM record streams."""

    def __init__(self, params):
        self.p = params
        names = C.POINTER(C.c_char_p)()
        lengths = C.POINTER(C.c_int64)()
        seqs = C.POINTER(C.c_char_p)()
        check(lib.msnv_synth_reference(C.byref(params), C.byref(names), C.byref(lengths), C.byref(seqs)))
        try:
            n = lib.msnv_synth_contig_count(C.byref(params))
        except AttributeError:                            # an older build under MSNV_LIBRARY (profiles/ab.sh): one contig per species
            n = params.n_species
        self.names = [names[i].decode() for i in range(n)]
        self.lengths = [int(lengths[i]) for i in range(n)]
        self.seqs = [seqs[i] for i in range(n)]          # bytes copies
        self._seq_arr = (C.c_char_p * n)(*self.seqs)
        # the library allocated with malloc/strdup
        vp = C.cast(names, C.POINTER(C.c_void_p)), C.cast(seqs, C.POINTER(C.c_void_p))
        for i in range(n):
            lib.msnv_free(vp[0][i]); lib.msnv_free(vp[1][i])
        lib.msnv_free(C.cast(names, C.c_void_p)); lib.msnv_free(C.cast(seqs, C.c_void_p)); lib.msnv_free(C.cast(lengths, C.c_void_p))

    def sample_records(self, idx):
        rec = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        check(lib.msnv_synth_sample(C.byref(self.p), idx, self._seq_arr, C.byref(rec), C.byref(n)))
        try:
            return np.ctypeslib.as_array(rec, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)
        finally:
            lib.msnv_free(C.cast(rec, C.c_void_p))

    def write_fasta(self, path, width=60):
        with open(path, "w") as f:
            for name, seq in zip(self.names, self.seqs):
                f.write(">%s\n" % name)
                s = seq.decode()
                for i in range(0, len(s), width):
                    f.write(s[i:i + width] + "\n")
