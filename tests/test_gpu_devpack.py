"""The per-read stage on the device (csrc/devpack.hip) builds the same dataset as the host stage (csrc/pack.cpp), byte for byte:
piece headers, 4-bit bases, quality flags, qaCompute intervals, and the whole tile index derived from them -- and the same
per-sample summaries (statistics, first lines, pileup base counts).  Then both are checked against the oracle's text."""
import ctypes as C
import os

import numpy as np
import pytest

import bamtools as bt
from metasnv_amd import core
from parity import run_oracle, synth_case, first_diff

pytestmark = pytest.mark.gpu

COLUMNS = ["hdr", "hdr4", "hdr8m", "blk", "seq", "qual", "s_read_base", "s_seq_base", "ref4", "pairs", "work", "chunks", "cov_iv", "cov_pairs", "cov_work"]


class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _build(where, names, lengths, seqs, samples, bed=None, params=None, many=False, device_ptrs=False, finalize=None):
    with _env(MSNV_PACK=where, MSNV_FINALIZE=finalize):
        ctx = core.Context(0)
        ds = core.Dataset(ctx, names, lengths, seqs, params)
        if bed:
            ds.set_bed(bed)
        if device_ptrs:
            # record streams already in HBM (what an all-to-all over RCCL leaves there): plain hipMalloc + hipMemcpy through the
            # runtime the library itself is linked to -- torch in this process would bring a second HIP runtime
            hip = C.CDLL("libamdhip64.so")
            ptrs, sizes = [], []
            for smp in samples:
                a = np.ascontiguousarray(smp, dtype=np.uint8)
                p = C.c_void_p()
                assert hip.hipMalloc(C.byref(p), C.c_size_t(max(1, a.size))) == 0
                if a.size:
                    assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.size), 1) == 0
                ptrs.append(p.value); sizes.append(a.size)
            ds.add_samples_records_device(ptrs, sizes)
            for p in ptrs:
                hip.hipFree(C.c_void_p(p))
        elif many:
            ds.add_samples_records(samples)
        else:
            for s in samples:
                ds.add_sample_records(s)
        info = ds.finalize()
    return ctx, ds, info


def _same_dataset(names, lengths, seqs, samples, bed=None, params=None, many=False, device_ptrs=False, check_oracle=True, finalize=None):
    """finalize: None = the tile index of the device-packed dataset is built in HBM where it can be (devfin_*), "host" = its headers come
    down and the host loops of finalize_dataset build it."""
    ch, dh, ih = _build("host", names, lengths, seqs, samples, bed, params)
    cd, dd, idv = _build("device", names, lengths, seqs, samples, bed, params, many=many, device_ptrs=device_ptrs, finalize=finalize)
    try:
        for k in ("n_reads", "n_reads_pileup", "n_pileup_bases", "bytes_headers", "bytes_cigar", "bytes_seq", "bytes_qual", "n_tiles", "n_pairs", "n_work",
                  "allele_planes", "sampled_mismatch_ppm"):
            assert ih[k] == idv[k], (k, ih[k], idv[k])
        for col in COLUMNS:
            a, b = dh.column(col), dd.column(col)
            assert a.size == b.size, (col, a.size, b.size)
            if not np.array_equal(a, b):
                i = int(np.flatnonzero(a != b)[0])
                raise AssertionError("column %s differs at byte %d of %d: host %s device %s" % (col, i, a.size, a[max(0, i - 4):i + 12].tolist(), b[max(0, i - 4):i + 12].tolist()))
        for s in range(len(samples)):
            assert np.array_equal(dh.sample_stats(s), dd.sample_stats(s)), s
        assert dh.first_line() == dd.first_line()
        if not bed:                                               # (the per-contig first lines are defined for a whole-BAM dataset only)
            fa, fb = dh.first_lines(), dd.first_lines()
            assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1])
        dd.run(); dh.run()
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            out = []
            for tag, ds in (("h", dh), ("d", dd)):
                pp, ip = os.path.join(td, "c" + tag), os.path.join(td, "i" + tag)
                ds.write_calls(pp, ip)
                out.append((open(pp).read(), open(ip).read()))
        assert out[0] == out[1]
        if check_oracle:
            orac = run_oracle(names, lengths, seqs, samples, bed=bed, params=params)
            assert out[1][0] == orac[0], first_diff(out[1][0], orac[0])
            assert out[1][1] == orac[1], first_diff(out[1][1], orac[1])
        return idv
    finally:
        dh.close(); dd.close(); ch.close(); cd.close()


def test_synthetic_cohort_same_columns():
    syn, samples = synth_case(n_species=3, contig_len=9000, n_samples=10, mean_cov=12.0, snv_density=0.02, error_rate=0.004, lowercase_ref=1, seed=21)
    info = _same_dataset(syn.names, syn.lengths, syn.seqs, samples)
    assert info["n_pileup_bases"] > 100000
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, finalize="host", check_oracle=False)


def test_sparse_cohort_and_merged_groups_same_columns():
    """Whole-tile work items and merged groups of shallow pairs (their headers are written by devfin_merged_headers), many contigs."""
    syn, samples = synth_case(n_species=12, contig_len=5000, n_samples=24, mean_cov=2.0, sigma_cov=0.3, species_per_sample=3, contigs_per_species_max=3,
                              snv_density=0.02, seed=26)
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True)
    with _env(MSNV_FUSE="0", MSNV_MERGE_ALWAYS="1"):
        _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True, check_oracle=False)


def test_deep_runs_are_dealt_into_groups_and_short_reads_into_block_streams_on_the_device():
    """(sample, tile) runs deeper than the byte bins hold (pack.cpp: split_deep_runs): exact sweep, round-robin groups, header permutation and
    the re-layout of the sample's columns run as kernels (devpack.hip: devfin_deep_runs) and give the host stage's bytes; so does the dense
    block layout of short reads (pack.cpp: relayout_dense; devpack.hip: devfin_dense) -- descriptors, columns and run tables."""
    syn, samples = synth_case(n_species=1, contig_len=4000, n_samples=3, mean_cov=300.0, sigma_cov=0.3, snv_density=0.02, seed=27)      # runs deeper than 192: dealt into groups
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples)
    with _env(MSNV_PACK="device"):
        ctx = core.Context(0)
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        ds.add_samples_records(samples)
        ds.finalize()
        assert ds.pack_stats()["deep_runs_split"] >= 2
        ds.close(); ctx.close()
    with _env(MSNV_SPLIT_AT="40", MSNV_GROUP_DEPTH="24"):                                                        # (many groups per run; some runs only look deep)
        syn2, samples2 = synth_case(n_species=2, contig_len=5000, n_samples=4, mean_cov=45.0, sigma_cov=0.6, snv_density=0.02, seed=29)
        _same_dataset(syn2.names, syn2.lengths, syn2.seqs, samples2)
    # short reads: dense block layout (odd and even lengths, indels cut the reads into pieces of every length)
    for read_len, seed, kw in ((36, 28, {}), (35, 30, dict(min_baseq=0)), (51, 31, dict(min_baseq=30)), (20, 32, {})):
        syn, samples = synth_case(n_species=2, contig_len=5000, n_samples=4, mean_cov=12.0, read_len=read_len, snv_density=0.02, frac_indel_reads=0.3, seed=seed)
        _same_dataset(syn.names, syn.lengths, syn.seqs, samples, params=core.default_params(**kw) if kw else None)
    with _env(MSNV_PACK="device"):
        ctx = core.Context(0)
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        ds.add_samples_records(samples)
        ds.finalize()
        assert ds.pack_stats()["dense_samples"] == len(samples) and ds.column("blk").size > 0
        ds.close(); ctx.close()
    with _env(MSNV_SPLIT_AT="40", MSNV_GROUP_DEPTH="24"):                                                        # deep runs of short reads: dealt first, then laid out
        syn3, samples3 = synth_case(n_species=1, contig_len=4000, n_samples=3, mean_cov=60.0, read_len=40, sigma_cov=0.4, snv_density=0.02, seed=33)
        _same_dataset(syn3.names, syn3.lengths, syn3.seqs, samples3)
    with _env(MSNV_LAYOUT="dense"):                                                                              # long reads forced into blocks: pieces of up to four blocks
        syn4, samples4 = synth_case(n_species=2, contig_len=6000, n_samples=3, mean_cov=15.0, snv_density=0.02, seed=34)
        _same_dataset(syn4.names, syn4.lengths, syn4.seqs, samples4)


def test_many_streams_one_round_and_device_pointers():
    syn, samples = synth_case(n_species=2, contig_len=7000, n_samples=12, mean_cov=8.0, snv_density=0.02, seed=22)
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True)
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, device_ptrs=True, check_oracle=False)


def test_bed_split_and_thresholds():
    syn, samples = synth_case(n_species=3, contig_len=5000, n_samples=5, mean_cov=15.0, snv_density=0.03, frac_absent=0.0, seed=23)
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, bed=[(0, 1, 5000), (2, 1, 5000)])
    for kw in (dict(min_baseq=0), dict(min_baseq=35), dict(flag_filter=0, min_mapq=1), dict(count_orphans=1), dict(min_baseq=200)):
        _same_dataset(syn.names, syn.lengths, syn.seqs, samples, params=core.default_params(**kw), check_oracle=kw.get("min_baseq", 0) < 128)


def _edge_reads():
    ref = ("ACGTTGCAAGGCTTAACCGGTTAACGTAGCTAGCTAGGATCCGATTACAGATTACAGGCATTACGGATCACGATCGACTAGCTAGCATCGACTGACTAGC" * 60)[:5200]

    def sub(i, n, mut=None):
        s = list(ref[i:i + n])
        for k, b in (mut or {}).items():
            s[k] = b
        return "".join(s)
    other = lambda c: "A" if c != "A" else "C"
    s1, s2, s3 = [], [], []
    for k in range(6):                                            # reads straddling the tile boundary at 2048, and one that spans two boundaries
        s1.append(bt.make_record(0, 2000 + k, "100M", sub(2000 + k, 100, {50 - k: other(ref[2050])}), name="a%d" % k))
    s1.append(bt.make_record(0, 2040, "30M2100D40M", sub(2040, 30) + sub(4170, 40), name="span"))
    s1.append(bt.make_record(0, 4090, "100M", sub(4090, 100), name="late"))
    # '=' / X ops, soft and hard clips, insertion, deletion, padding, odd starts, odd lengths, '=' in the SEQ field
    s2.append(bt.make_record(0, 100, "10=5X10=", sub(100, 25, {10: other(ref[110]), 11: other(ref[111])}), name="b0"))
    s2.append(bt.make_record(0, 100, "5S20M", "NNNNN" + sub(100, 20, {10: other(ref[110])}), name="b1"))
    s2.append(bt.make_record(0, 100, "3H21M2H", sub(100, 21, {10: other(ref[110])}), name="b2"))
    s2.append(bt.make_record(0, 100, "8M3I12M", sub(100, 8) + "GGG" + sub(108, 12, {2: other(ref[110])}), name="b3"))
    s2.append(bt.make_record(0, 100, "8M2D13M", sub(100, 8) + sub(110, 13, {0: other(ref[110])}), name="b4"))
    s2.append(bt.make_record(0, 100, "8M1P12M", sub(100, 20, {10: other(ref[110])}), name="b5"))
    s2.append(bt.make_record(0, 101, "1S7M", "T" + "=" * 3 + sub(104, 4), name="eq"))
    for k in range(5):
        s2.append(bt.make_record(0, 105, "11M", sub(105, 11), qual=[5] * 11, name="lowq%d" % k))
    s2.append(bt.make_record(0, 300, "300M", sub(300, 300), name="long"))                     # three pieces
    s2.append(bt.make_record(0, 5150, "60M", sub(5150, 50) + "ACGTACGTAC", name="overhang"))    # runs past the contig end (5200)
    # filtered reads: duplicate, secondary, QC fail, MAPQ 0, orphan, unmapped with a position
    s3.append(bt.make_record(0, 500, "50M", sub(500, 50), flag=0x400, name="dup"))
    s3.append(bt.make_record(0, 500, "50M", sub(500, 50), flag=0x100, name="sec"))
    s3.append(bt.make_record(0, 500, "50M", sub(500, 50), flag=0x200, name="qcf"))
    s3.append(bt.make_record(0, 501, "50M", sub(501, 50), mapq=0, name="mq0"))
    s3.append(bt.make_record(0, 502, "50M", sub(502, 50), flag=0x1, name="orphan"))
    s3.append(bt.make_record(0, 503, "*", "", flag=0x4, name="unm"))
    s3.append(bt.make_record(0, 504, "50M", sub(504, 50, {7: other(ref[511])}), name="plain"))
    s3.append(bt.make_record(1, 10, "40M", "ACGT" * 10, name="othercontig"))
    s3.append(bt.make_record(-1, -1, "*", "", flag=0x4, name="unplaced"))
    return ["edge", "tiny"], [len(ref), 60], [ref, "ACGT" * 15], [bt.records(*s1), bt.records(*s2), bt.records(*s3), bt.records()]


def test_edge_case_records_same_columns():
    names, lengths, seqs, samples = _edge_reads()
    _same_dataset(names, lengths, seqs, samples, params=core.default_params(min_coverage=1, calling_threshold=1))


def test_overlapping_mates_are_edited_on_the_device():
    """Proper pairs whose mates overlap: htslib's quality tweak (sam.c tweak_overlap_quality [EXT]) runs as a kernel over the candidates grouped
    by read name (devpack.hip: msnv_ovl_groups) -- no sample takes the host pre-pass -- and the columns equal the host stage's, which walks
    the reads one by one with a hash of waiting mates (pack.cpp); MSNV_OVERLAP=host sends them through that walk instead."""
    syn, samples = synth_case(n_species=2, contig_len=6000, n_samples=6, mean_cov=14.0, snv_density=0.02, frac_paired=0.6, frac_indel_reads=0.2, frac_clip_reads=0.1, seed=24)
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples)
    ctx = core.Context(0)
    try:
        for where, want in (("device", 0), ("host", len([s for s in samples if len(s)]))):
            with _env(MSNV_PACK="device", MSNV_OVERLAP=where):
                ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
                ds.add_samples_records(samples)
                assert ds.pack_stats()["prepass_samples"] == want, where
                ds.close()
    finally:
        ctx.close()
    with _env(MSNV_OVERLAP="host"):
        _same_dataset(syn.names, syn.lengths, syn.seqs, samples, check_oracle=False)
    # a depth cap in reach sends the sample through the host pre-pass whatever its pairs do
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, params=core.default_params(max_depth=9))


def _edit_stats(names, lengths, seqs, samples, params, **env):
    ctx = core.Context(0)
    try:
        with _env(MSNV_PACK="device", **env):
            ds = core.Dataset(ctx, names, lengths, seqs, params)
            ds.add_samples_records(samples)
            st = ds.pack_stats()
            ds.close()
        return st
    finally:
        ctx.close()


@pytest.mark.parametrize("kw", [dict(max_depth=9), dict(max_depth=3, count_orphans=1), dict(token_limit=60), dict(token_limit=45, min_baseq=0), dict(max_depth=12, token_limit=40),
                                dict(max_depth=1, token_limit=41, ignore_overlaps=1)])
def test_depth_cap_and_token_limit_run_as_kernels(kw):
    """The two sequential edits that were the host pre-pass's alone -- mpileup's depth cap (-d) and snpCall's 10000-character token -- as kernels
    (devpack.hip: msnv_cap_reads, msnv_token_cut): no sample takes the host pre-pass, the columns equal the host stage's byte for byte and the
    calls equal the oracle's (with a token of the same length: oracle/orc.h token_cap); MSNV_PREPASS=host sends the samples through the host
    pre-pass instead.  Paired reads (the overlapping-mate edits come in front of the token count; a capped read is no candidate), indels
    (their suffixes count towards the token), clips, low qualities (elements below -Q are not printed)."""
    syn, samples = synth_case(n_species=2, contig_len=5000, n_samples=5, mean_cov=30.0, sigma_cov=0.6, snv_density=0.03, frac_paired=0.6, frac_indel_reads=0.3, frac_clip_reads=0.1,
                              frac_lowq=0.2, read_len=75, seed=606)
    p = core.default_params(min_coverage=2, calling_threshold=2, **kw)
    _same_dataset(syn.names, syn.lengths, syn.seqs, samples, params=p, many=True)
    dev = _edit_stats(syn.names, syn.lengths, syn.seqs, samples, p)
    assert dev["prepass_samples"] == 0 and dev["device_edit_samples"] >= 3, dev
    host = _edit_stats(syn.names, syn.lengths, syn.seqs, samples, p, MSNV_PREPASS="host")
    assert host["prepass_samples"] == dev["device_edit_samples"] and host["device_edit_samples"] == 0, host
    with _env(MSNV_PREPASS="host"):
        _same_dataset(syn.names, syn.lengths, syn.seqs, samples, params=p, many=True, check_oracle=False)


def test_a_deep_stack_is_capped_and_cut_on_the_device():
    """One position where 9000 reads start (test_overlap_host's stack: 3 characters per read start pass snpCall's REAL 10000-character token) beside
    a deep plain pileup: the cap drops every read behind the 8000th, the token cuts the string -- both as kernels, against the oracle."""
    import random
    rnd = random.Random(5)
    ref = "".join(rnd.choice("ACGT") for _ in range(1200))
    recs = []
    for k in range(9000):
        q = list(ref[300:360])
        if k % 3 == 0:
            q[20] = "A" if ref[320] != "A" else "C"
        recs.append(bt.make_record(0, 300, "60M", "".join(q), name="s%d" % k, qual=[rnd.choice([10, 20, 30, 40]) for _ in range(60)]))
    for k in range(3000):                                           # reads that arrive while the stack is alive: all dropped by the cap, or cut
        st = 301 + k % 50
        q = list(ref[st:st + 50]); q[5] = "T" if ref[st + 5] != "T" else "G"
        recs.append((st, bt.make_record(0, st, "20M2D30M", "".join(q[:20] + q[22:] + ["A", "C"]), name="t%d" % k)))
    stack = [r for r in recs if not isinstance(r, tuple)] + [r for _, r in sorted([r for r in recs if isinstance(r, tuple)], key=lambda t: t[0])]
    other = bt.records(bt.make_record(0, 290, "60M", ref[290:350], name="o1"), bt.make_record(0, 310, "60M", ref[310:370], name="o2"))
    p = core.default_params(min_coverage=1, calling_threshold=2)
    samples = [other, bt.records(*stack)]
    _same_dataset(["c1"], [len(ref)], [ref], samples, params=p, many=True)
    st = _edit_stats(["c1"], [len(ref)], [ref], samples, p)
    assert st["prepass_samples"] == 0 and st["device_edit_samples"] == 1, st


def test_samples_of_one_round_take_different_routes_through_the_sequential_edits():
    """One round, four samples: (a) plain; (b) deeper than the cap AND beyond the (shortened) token: msnv_cap_reads + msnv_token_cut; (c) a template
    with a dozen alignments under one name -- more than the overlap kernel's slots: the host pre-pass takes the whole sample, its depth cap and
    token cut included; (d) proper pairs below every limit: the overlap kernel.  The verdicts of the host's samples and of the kernels' samples
    share one table, the second pass measures with both; columns byte for byte against the host stage, calls against the oracle."""
    import random
    rnd = random.Random(77)
    L = 3000
    ref = "".join(rnd.choice("ACGT") for _ in range(L))

    def read(pos, n, name, flag=0, mpos=-1, tlen=0, cigar=None, mut=0.02):
        q = [c if rnd.random() > mut else rnd.choice("ACGT") for c in ref[pos:pos + n]]
        return (pos, bt.make_record(0, pos, cigar or "%dM" % n, "".join(q), name=name, flag=flag, mtid=0 if mpos >= 0 else -1, mpos=mpos, tlen=tlen,
                                    qual=[rnd.choice([5, 20, 30, 40]) for _ in range(n)]))

    plain = [read(rnd.randrange(0, L - 80), 70, "a%d" % k) for k in range(300)]
    deep = [read(500 + k // 6, 60, "b%d" % k, cigar="30M2D30M" if k % 5 == 0 else None) for k in range(400)] + [read(rnd.randrange(0, L - 80), 70, "bb%d" % k) for k in range(200)]
    many = [read(900 + 3 * k, 60, "tmpl", flag=99 if k % 2 == 0 else 147, mpos=905, tlen=200) for k in range(12)] + \
           [read(880 + k // 5, 50, "c%d" % k) for k in range(300)]
    pairs = []
    for k in range(150):
        a = rnd.randrange(0, L - 200)
        pairs.append(read(a, 70, "p%d" % k, flag=99, mpos=a + 40, tlen=110)); pairs.append(read(a + 40, 70, "p%d" % k, flag=147, mpos=a, tlen=-110))
    samples = [bt.records(*[r for _, r in sorted(x, key=lambda t: t[0])]) for x in (plain, deep, many, pairs)]
    p = core.default_params(min_coverage=2, calling_threshold=2, max_depth=25, token_limit=70)
    _same_dataset(["c1"], [L], [ref], samples, params=p, many=True)
    st = _edit_stats(["c1"], [L], [ref], samples, p)
    assert st["prepass_samples"] == 1 and st["device_edit_samples"] >= 1, st


def test_errors_carry_the_host_stage_codes():
    from metasnv_amd import _lib
    ref = "ACGT" * 100
    good = bt.make_record(0, 10, "20M", ref[10:30], name="g")
    unsorted_stream = bt.records(bt.make_record(0, 50, "20M", ref[50:70], name="x"), good)
    truncated = bt.records(good)[:-5]
    ctx = core.Context(0)
    try:
        for bad in (unsorted_stream, truncated):
            for where in ("host", "device"):
                with _env(MSNV_PACK=where):
                    ds = core.Dataset(ctx, ["c"], [400], [ref])
                    with pytest.raises(_lib.MsnvError) as e:
                        ds.add_sample_records(bad)
                    assert e.value.code == _lib.EFORMAT, where
                    ds.close()
    finally:
        ctx.close()


def test_record_scan_across_segment_seams():
    """The record chain is walked piecewise with GUESSED entry points checked at every seam -- round 5: a lane per sub-segment, seams checked
    on the device, wrong guesses walked again there (devpack.hip: msnv_scan_sub / msnv_scan_repair; MSNV_SCAN_SUB bytes); the careful form (msnv_scan_segments, MSNV_SCAN=segments,
    MSNV_SCAN_SEG_KB) repairs wrong guesses on the host and takes over when the quick form meets a chain that breaks.  Tiny pieces put a seam into almost
    every record (and records longer than a sub-segment next to it); a read name that holds plausible record headers makes guesses go wrong."""
    syn, samples = synth_case(n_species=2, contig_len=6000, n_samples=4, mean_cov=9.0, snv_density=0.02, seed=25)
    for sub in ("64", "100", "333", "4096"):
        with _env(MSNV_SCAN_SUB=sub):
            _same_dataset(syn.names, syn.lengths, syn.seqs, samples, check_oracle=(sub == "100"))
    for kb in ("1", "2", "64"):
        with _env(MSNV_SCAN_SEG_KB=kb, MSNV_SCAN="segments"):
            _same_dataset(syn.names, syn.lengths, syn.seqs, samples, check_oracle=(kb == "1"))
    # a decoy: inside a long read name, the bytes of three consecutive plausible record headers
    import struct
    ref = "ACGT" * 1500

    def fake(nxt):                      # 36 header bytes of a "record" with block_size nxt - 4 whose fields pass every plausibility test (name: 2 bytes, NUL-terminated)
        return struct.pack("<iiiBBHHHiiii", nxt - 4, 0, 5, 2, 60, 4680, 0, 0, 0, -1, -1, 0)
    decoy = (fake(40) + b"a\0\0\0") * 6
    recs = []
    for k in range(40):
        recs.append(bt.make_record(0, 10 + 20 * k, "50M", ref[10 + 20 * k:60 + 20 * k], name="r%d" % k, aux=b"ZZZ" + decoy[:200] if k % 3 == 0 else b""))
    stream = bt.records(*recs)
    ctx = core.Context(0)
    try:
        for env in (dict(MSNV_SCAN_SEG_KB="1", MSNV_SCAN="segments"), dict(MSNV_SCAN_SEG_KB="256", MSNV_SCAN="segments"), dict(MSNV_SCAN_SUB="64"), dict(MSNV_SCAN_SUB="128"), dict(MSNV_SCAN_SUB="4096")):
            with _env(MSNV_PACK="device", **env):
                ds = core.Dataset(ctx, ["c"], [6000], [ref])
                ds.add_sample_records(stream)
                ds.finalize()
                assert ds.info()["n_reads_pileup"] == 40
                if env.get("MSNV_SCAN_SUB") in ("64", "128"):
                    # the quick form fell for the decoy and walked those sub-segments again from the true entry (msnv_scan_repair), on the device
                    assert ds.pack_stats()["scan_segments_redone"] >= 1
                ds.close()
        for sub in ("64", "128"):                                    # ... and what it builds is the host pack's dataset, byte for byte
            with _env(MSNV_SCAN_SUB=sub):
                _same_dataset(["c"], [6000], [ref], [stream])
    finally:
        ctx.close()


def test_aux_fields_and_cigars_in_the_cg_field():
    """What real aligners write: auxiliary fields behind the qualities (every bwa / ngless record), and -- for more than 65535 operations --
    the CIGAR in CG:B,I behind a placeholder.  Both packers read such records like htslib does; same calls as the plain records."""
    ref = ("ACGTTGCAAGGCTTAACCGGTTAACGTAGCTAGCTAGGATCCGATTACAGATTACAGGCATTACGGATCACGATCGACTAGCTAGCATCGACTGACTAGC" * 30)[:2600]
    other = lambda c: "A" if c != "A" else "C"
    streams = {False: [], True: []}
    for cg in (False, True):
        for smp in range(3):
            recs = []
            for k in range(40):
                pos = 20 + 31 * k + smp
                seq = list(ref[pos:pos + 20] + ref[pos + 22:pos + 62])
                seq[30] = other(ref[pos + 32]) if k % 3 == 0 else seq[30]
                recs.append(bt.make_record(0, pos, "20M2D40M", "".join(seq), name="s%dr%d" % (smp, k), aux=bt.aux_fields(nm=k % 5, md="20^AC40", score=55), cg_form=cg and k % 2 == 0))
            streams[cg].append(bt.records(*recs))
    p = core.default_params(min_coverage=1, calling_threshold=1)
    _same_dataset(["c"], [len(ref)], [ref], streams[True], params=p)
    a = run_oracle(["c"], [len(ref)], [ref], streams[True], params=p)
    b = run_oracle(["c"], [len(ref)], [ref], streams[False], params=p)
    assert a[0] == b[0] and a[1] == b[1] and a[0].count("\n") > 10


def test_records_dealt_on_the_device_equal_the_host_partition():
    """msnv_records_deal_device (the N-rank feed's dealing step as kernels) against msnv_records_partition / msnv_records_contig_bases on host
    threads: the bytes of every (stream, part), their sizes, qaCompute's statistics and the aligned bases per contig -- paired reads, unmapped
    records, aux tags, SEQ `*`, contigs owned by nobody, empty streams, one-kilobyte scan segments, a gap in front of every part."""
    hip = C.CDLL("libamdhip64.so")
    syn, samples = synth_case(n_species=5, contig_len=6000, n_samples=9, mean_cov=9.0, frac_paired=0.4, frac_aux=0.4, frac_noseq=0.05, frac_absent=0.3, seed=41)
    unm = bt.records(bt.make_record(-1, -1, "*", "ACGT", name="u1", flag=4), bt.make_record(-1, -1, "*", "ACGTAC", name="u2", flag=4))
    samples = [np.concatenate([s, np.frombuffer(unm.tobytes(), np.uint8)]) if i % 3 == 0 else s for i, s in enumerate(samples)] + [np.zeros(0, np.uint8)]
    nc = len(syn.names)
    ctx = core.Context(0)
    for n_parts, seg_kb, gap in ((1, None, 0), (3, "1", 72), (8, None, 16)):
        owner = np.array([(c * 7 + 1) % n_parts if c % 4 != 3 else -1 for c in range(nc)], dtype=np.int32)
        want_parts, want_stats, want_cb = [], [], np.zeros(nc, np.uint64)
        for s in samples:
            parts, st = core.partition_records(s, owner, n_parts)
            want_parts.append(parts); want_stats.append(st); core.contig_bases(s, nc, into=want_cb)
        cap = sum(int(s.size) for s in samples) + n_parts * gap
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(max(16, cap))) == 0
        assert hip.hipMemset(p, 0xEE, C.c_size_t(max(16, cap))) == 0
        cb = np.zeros(nc, np.uint64)
        with _env(MSNV_SCAN_SEG_KB=seg_kb):
            pb, stats = core.deal_records_device(ctx, samples, owner, n_parts, p.value, cap, gap=gap, contig_bases=cb)
        got = np.zeros(max(16, cap), np.uint8)
        assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), p, C.c_size_t(max(16, cap)), 2) == 0
        hip.hipFree(p)
        assert np.array_equal(cb, want_cb) and np.array_equal(stats, np.stack(want_stats))
        assert np.array_equal(pb, np.array([[q.size for q in parts] for parts in want_parts], dtype=np.int64))
        o = 0
        for k in range(n_parts):
            assert (got[o:o + gap] == 0xEE).all()                   # the caller's gap is left alone
            o += gap
            for i in range(len(samples)):
                w = want_parts[i][k]
                assert got[o:o + w.size].tobytes() == w.tobytes(), (n_parts, k, i)
                o += w.size
        assert (got[o:cap] == 0xEE).all()
    # a record of a contig the header does not have / an owner beyond the parts: the host partition's error codes
    bad = bt.records(bt.make_record(nc + 2, 5, "10M", "ACGTACGTAC", name="b"))
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(4096)) == 0
    with pytest.raises(core._lib.MsnvError) as e:
        core.deal_records_device(ctx, [np.frombuffer(bad.tobytes(), np.uint8)], np.zeros(nc, np.int32), 2, p.value, 4096)
    assert e.value.code == core._lib.EFORMAT
    p2 = C.c_void_p(); assert hip.hipMalloc(C.byref(p2), C.c_size_t(int(samples[1].size) + 64)) == 0
    with pytest.raises(core._lib.MsnvError) as e:
        core.deal_records_device(ctx, [samples[1]], np.full(nc, 5, np.int32), 2, p2.value, int(samples[1].size) + 64)
    assert e.value.code == core._lib.EINVAL
    with pytest.raises(core._lib.MsnvError) as e:                   # an output that cannot hold every record
        core.deal_records_device(ctx, [samples[1]], np.zeros(nc, np.int32), 2, p.value, 64)
    assert e.value.code == core._lib.ECAPACITY
    # a record whose CIGAR cannot fit its block_size (n_cigar_op patched to 65535): refused like the host partition refuses it -- not walked
    # (256 KB of "CIGAR" behind a 60-byte record), not forwarded
    broken = bytearray(bt.records(bt.make_record(0, 5, "10M", "ACGTACGTAC", name="b"), bt.make_record(0, 9, "10M", "ACGTACGTAC", name="c")).tobytes())
    broken[16:18] = b"\xff\xff"
    with pytest.raises(core._lib.MsnvError) as e:
        core.deal_records_device(ctx, [np.frombuffer(bytes(broken), np.uint8)], np.zeros(nc, np.int32), 2, p.value, 4096, contig_bases=np.zeros(nc, np.uint64))
    assert e.value.code == core._lib.EFORMAT and "malformed" in str(e.value)
    with pytest.raises(core._lib.MsnvError) as e:
        core.partition_records(np.frombuffer(bytes(broken), np.uint8), np.zeros(nc, np.int32), 2)
    assert e.value.code == core._lib.EFORMAT
    hip.hipFree(p); hip.hipFree(p2); ctx.close()


def test_streams_read_in_place_from_one_device_buffer():
    """msnv_dataset_add_sample_records_resident: the streams lie at odd offsets of ONE padded device buffer and are packed where they lie
    (no copy into a round buffer) -- the same dataset as the host pack builds; arguments that break the contract are refused."""
    from metasnv_amd import _lib
    syn, samples = synth_case(n_species=2, contig_len=7000, n_samples=6, mean_cov=9.0, snv_density=0.02, frac_paired=0.5, seed=33)
    hip = C.CDLL("libamdhip64.so")
    offs, sizes, o = [], [], 16
    for smp in samples:
        offs.append(o); sizes.append(int(smp.size)); o += int(smp.size) + 37          # (odd gaps: nothing of a stream is aligned)
    cap = o + 256
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), C.c_size_t(cap)) == 0
    for smp, off in zip(samples, offs):
        a = np.ascontiguousarray(smp, dtype=np.uint8)
        if a.size:
            assert hip.hipMemcpy(C.c_void_p(buf.value + off), C.c_void_p(a.ctypes.data), C.c_size_t(a.size), 1) == 0
    with _env(MSNV_PACK="host"):
        ch = core.Context(0); dh = core.Dataset(ch, syn.names, syn.lengths, syn.seqs)
        for smp in samples:
            dh.add_sample_records(smp)
        ih = dh.finalize()
    cd = core.Context(0); dd = core.Dataset(cd, syn.names, syn.lengths, syn.seqs)
    dd.add_samples_records_resident(buf.value, cap, offs, sizes)
    idv = dd.finalize()
    assert dd.pack_stats()["upload_wall_s"] == 0.0
    for k in ("n_reads", "n_pileup_bases", "n_pairs", "n_work"):
        assert ih[k] == idv[k], k
    for col in COLUMNS:
        assert np.array_equal(dh.column(col), dd.column(col)), col
    dd.close()
    d2 = core.Dataset(cd, syn.names, syn.lengths, syn.seqs)
    with pytest.raises(_lib.MsnvError):
        d2.add_samples_records_resident(buf.value + 4, cap - 4, offs, sizes)                 # base not on 16 bytes
    with pytest.raises(_lib.MsnvError):
        d2.add_samples_records_resident(buf.value, offs[-1] + sizes[-1] + 8, offs, sizes)   # no room behind the last stream
    with pytest.raises(_lib.MsnvError):
        d2.add_samples_records_resident(buf.value, cap, list(reversed(offs)), list(reversed(sizes)))   # not ascending
    d2.close(); dh.close(); ch.close(); cd.close()
    hip.hipFree(buf)


def test_a_call_that_fails_after_a_round_leaves_the_dataset_unusable():
    """msnv_dataset_add_sample_records_many packs in rounds (MSNV_PACK_ROUND_MB); when a LATER round fails, the rounds that went through have
    left their tables behind: the samples of the call are taken back, and the dataset answers MSNV_EINVAL to every further add_* and to
    finalize instead of indexing with stale rounds.  A call that fails in its FIRST round leaves the dataset usable."""
    syn, samples = synth_case(n_species=1, contig_len=30000, n_samples=3, mean_cov=60.0, seed=5)
    assert all(s.size > (1 << 20) // 2 for s in samples)            # (more than half a megabyte each: a round of its own below)
    bad = samples[2].copy()
    bad[16:18] = 0xff                                               # n_cigar_op of the first record: 65535 operations cannot fit its block_size
    ctx = core.Context(0)
    try:
        with _env(MSNV_PACK="device", MSNV_PACK_ROUND_MB="1"):
            ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
            with pytest.raises(core._lib.MsnvError) as e:
                ds.add_samples_records([samples[0], samples[1], bad])
            assert e.value.code == core._lib.EFORMAT
            for call in (lambda: ds.add_sample_records(samples[0]), lambda: ds.finalize()):
                with pytest.raises(core._lib.MsnvError) as e:
                    call()
                assert e.value.code == core._lib.EINVAL and "cannot be used further" in str(e.value)
            ds.close()
            ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
            with pytest.raises(core._lib.MsnvError):
                ds.add_samples_records([bad])                       # (the first round of the call: nothing was left behind)
            ds.add_samples_records([samples[0], samples[1]])
            assert ds.finalize()["n_samples"] == 2
            ds.close()
    finally:
        ctx.close()


def test_a_bam_group_that_fails_behind_a_packed_group_poisons_the_dataset(tmp_path):
    """msnv_dataset_add_sample_bams with the host inflate packs the files group by group (max(threads, 16) files a group: one
    add_streams_device call each).  A corrupt file in the SECOND group: the first group's samples are dropped with the rest -- a failed
    add_* call adds nothing -- and, their rounds' tables being left behind, the dataset refuses everything further (the
    poison flag was set inside one add_streams_device call only, and finalize would have indexed with the stale rounds)."""
    syn, samples = synth_case(n_species=1, contig_len=4000, n_samples=18, mean_cov=6.0, seed=21)
    paths = []
    for i, smp in enumerate(samples):
        p = str(tmp_path / ("s%02d.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, smp)
        paths.append(p)
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    bad = samples[17].copy()
    bad[16:18] = 0xff                                               # n_cigar_op of the first record: cannot fit its block_size
    core.write_bam(paths[17], syn.names, syn.lengths, bad)
    ctx = core.Context(0)
    try:
        with _env(MSNV_PACK="device", MSNV_INFLATE="host"):
            ds = core.Dataset.from_files(ctx, paths[0], fa)
            with pytest.raises(core._lib.MsnvError) as e:
                ds.add_sample_bams(paths, 4)                        # groups of 16: files 0-15 pack, the group of files 16-17 fails
            assert e.value.code == core._lib.EFORMAT
            for call in (lambda: ds.add_sample_bams(paths[:2], 2), lambda: ds.finalize()):
                with pytest.raises(core._lib.MsnvError) as e:
                    call()
                assert e.value.code == core._lib.EINVAL and "cannot be used further" in str(e.value)
            ds.close()
            ds = core.Dataset.from_files(ctx, paths[0], fa)         # the same files without the corrupt one: a usable dataset
            ds.add_sample_bams(paths[:17], 4)
            assert ds.finalize()["n_samples"] == 17
            ds.close()
    finally:
        ctx.close()


def test_the_emit_kernels_two_routes_write_the_same_columns():
    """msnv_emit_block (one lane per record cuts the pieces, the workgroup moves them through an LDS image) leaves the blocks that do not fit
    that form to msnv_emit_block_slow (four lanes per record from global memory); MSNV_EMIT=slow sends every block there.  Both against
    the host pack, column by column -- with indels, clips and tile crossings (further pieces), reads of 180 bases (blocks beyond the
    window) and a cohort whose samples are smaller than a block (two samples in one block)."""
    for kw in (dict(n_species=2, contig_len=9000, n_samples=6, mean_cov=14.0, snv_density=0.02, error_rate=0.01, seed=31),
               dict(n_species=1, contig_len=7000, n_samples=3, mean_cov=9.0, read_len=180, seed=32),
               dict(n_species=3, contig_len=2500, n_samples=12, mean_cov=1.5, seed=33)):
        syn, samples = synth_case(**kw)
        _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True, check_oracle=False)
        with _env(MSNV_EMIT="slow"):
            _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True, check_oracle=False)
        with _env(MSNV_FILL_PADDING="1"):
            _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True, check_oracle=False)


def test_the_quick_and_the_careful_front_build_the_same_dataset():
    """Round 6: records' boundaries and everything a record decides by itself come from ONE walk (msnv_scan_sub2, the quick route); the
    careful route -- msnv_scan_sub / msnv_scan_segments, msnv_measure_reads, a wait between the stages -- takes what the quick one leaves
    (paired reads, the host pre-pass, malformed input) and MSNV_FRONT=careful sends everything there.  Both against the host pack, column by
    column; small sub-segments put many seams, run and group boundaries between the lanes of the walk."""
    for kw in (dict(n_species=3, contig_len=9000, n_samples=8, mean_cov=12.0, snv_density=0.02, error_rate=0.004, seed=41),
               dict(n_species=6, contig_len=2600, n_samples=20, mean_cov=1.2, species_per_sample=2, seed=42),
               dict(n_species=1, contig_len=5000, n_samples=2, mean_cov=40.0, read_len=150, seed=43)):
        syn, samples = synth_case(**kw)
        for env in (dict(), dict(MSNV_FRONT="careful"), dict(MSNV_SCAN_SUB="256"), dict(MSNV_SCAN_SUB="8192"), dict(MSNV_SCAN="segments")):
            with _env(**env):
                _same_dataset(syn.names, syn.lengths, syn.seqs, samples, many=True, check_oracle=False)
