"""Parity of the HIP path (called through the C ABI) with the oracle: bit-exact text.

Integer counting => the bar is byte-identical called_SNPs / indiv_called."""
import os
import tempfile

import numpy as np
import pytest

import bamtools as bt
import orc
from metasnv_amd import core
from parity import run_product, run_oracle, synth_case, first_diff

pytestmark = pytest.mark.gpu


def _assert_same(prod, orac):
    assert prod[0] == orac[0], "called_SNPs differs, " + first_diff(prod[0], orac[0])
    assert prod[1] == orac[1], "indiv_called differs, " + first_diff(prod[1], orac[1])


def test_synthetic_small_multi_contig():
    syn, samples = synth_case(n_species=3, contig_len=5000, n_samples=8, mean_cov=12.0, snv_density=0.02, error_rate=0.004,
                              lowercase_ref=1, seed=11)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 50
    assert prod[2]["n_pileup_bases"] == orac[3]          # the metric's unit of work agrees too


def test_synthetic_testdata_shape_reduced():
    # the BASELINE "testdata" shape at 1/10 contig length and 40 samples (oracle finishes in seconds)
    syn, samples = synth_case(n_species=3, contig_len=30000, n_samples=40, seed=3)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 300


def test_thresholds_and_quality_cutoffs():
    syn, samples = synth_case(n_species=1, contig_len=6000, n_samples=5, mean_cov=9.0, snv_density=0.02, seed=5)
    for kw in (dict(min_coverage=1, calling_threshold=1), dict(min_coverage=10, calling_threshold=2, min_fraction=0.2),
               dict(min_baseq=0), dict(min_baseq=35), dict(flag_filter=0, min_mapq=1), dict(calling_threshold=3, min_fraction=0.0)):
        p = core.default_params(**kw)
        _assert_same(run_product(syn.names, syn.lengths, syn.seqs, samples, params=p),
                     run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p))


def test_bed_split_excludes_position_one_and_other_contigs():
    syn, samples = synth_case(n_species=3, contig_len=4000, n_samples=4, mean_cov=15.0, snv_density=0.03, frac_absent=0.0, seed=9)
    bed = [(0, 1, 4000), (2, 1, 4000)]                   # best_split file: `name\t1\tLEN` (metaSNV.py:92)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
    _assert_same(prod, orac)
    assert "refGenome2clus" not in prod[0] and "refGenome3clus" in prod[0]


def _reads_edge():
    ref = ("ACGTTGCAAGGCTTAACCGGTTAACGTAGCTAGCTAGGATCCGATTACAGATTACAGGCATTACGGATCACGATCGACTAGCTAGCATCGACTGACTAGC" * 30)[:2600]
    def sub(i, n, mut=None):
        s = list(ref[i:i + n])
        for k, b in (mut or {}).items():
            s[k] = b
        return "".join(s)
    s1, s2, s3 = [], [], []
    # a pile of mismatching reads around the tile boundary at 2048 (reads straddle two tiles)
    for k in range(6):
        s1.append(bt.make_record(0, 2000 + k, "100M", sub(2000 + k, 100, {50 - k: "A" if ref[2050] != "A" else "C"}), name="a%d" % k))
    # '=' and X ops, N (ref skip: not counted), soft clip, hard clip, insertion, deletion, P
    s2.append(bt.make_record(0, 100, "10=5X10=", sub(100, 25, {10: "T" if ref[110] != "T" else "G", 11: "T" if ref[111] != "T" else "G"}), name="b0"))
    s2.append(bt.make_record(0, 100, "5S20M", "NNNNN" + sub(100, 20, {10: "T" if ref[110] != "T" else "G"}), name="b1"))
    s2.append(bt.make_record(0, 100, "3H20M2H", sub(100, 20, {10: "T" if ref[110] != "T" else "G"}), name="b2"))
    s2.append(bt.make_record(0, 100, "8M3I12M", sub(100, 8) + "GGG" + sub(108, 12, {2: "T" if ref[110] != "T" else "G"}), name="b3"))
    s2.append(bt.make_record(0, 100, "8M2D12M", sub(100, 8) + sub(110, 12, {0: "T" if ref[110] != "T" else "G"}), name="b4"))
    s2.append(bt.make_record(0, 100, "8M1P12M", sub(100, 20, {10: "T" if ref[110] != "T" else "G"}), name="b5"))
    for k in range(5):
        s2.append(bt.make_record(0, 105, "10M", sub(105, 10), qual=[5] * 10, name="lowq%d" % k))      # all below BQ 13
    # per-sample depth above 255 (coverage byte saturates; overflow list) with an individual-only allele
    for k in range(500):
        mut = {20: "G" if ref[520] != "G" else "T"} if k < 4 else None
        s3.append(bt.make_record(0, 500, "40M", sub(500, 40, mut), name="d%d" % k, flag=16 if k % 2 else 0))
    # read N bases and odd read lengths
    s3.append(bt.make_record(0, 700, "7M", "NNNN" + sub(704, 3), name="n0"))
    # reads that end exactly at the contig end and one flagged read of every kind
    s3.append(bt.make_record(0, 2590, "10M", sub(2590, 10), name="e0"))
    for fl in (0x400, 0x100, 0x200, 0x1, 0x4):
        s3.append(bt.make_record(0, 2590, "10M", "TTTTTTTTTT", flag=fl, name="f%x" % fl))
    key = lambda r: (int.from_bytes(r[4:8], "little"), int.from_bytes(r[8:12], "little", signed=True))
    return ref, [bt.records(*sorted(s, key=key)) for s in (s1, s2, s3)] + [np.zeros(0, np.uint8)]


@pytest.mark.parametrize("deep", ["split", "wide"])
def test_cigar_ops_tile_boundary_overflow_and_empty_sample(deep, monkeypatch):
    """deep = how the 500-read pile of sample 3 (depth above the byte bins) is handled: split into groups of reads that each
    stay below 255 and summed per sample on the device (default), or kept whole for the 16-bit kernel (MSNV_DEEP=wide)."""
    monkeypatch.setenv("MSNV_DEEP", deep)
    ref, samples = _reads_edge()
    p = core.default_params(min_coverage=2, calling_threshold=2)
    prod = run_product(["ctg"], [len(ref)], [ref], samples, params=p)
    orac = run_oracle(["ctg"], [len(ref)], [ref], samples, params=p)
    _assert_same(prod, orac)
    if deep == "wide":
        assert prod[3]["n_overflow"] > 0                  # the >= 255 path of the wide kernel really ran
    else:
        assert prod[3]["n_overflow"] == 0 and prod[2]["n_pairs"] >= 6   # 2 tiles x 2 shallow samples + >= 3 groups of the deep one
    assert "\t2051\t" in prod[0] or "\t2051\t" in prod[1]
    assert "\t521\t" in prod[1] and "\t521\t" not in prod[0]   # 4 of 500 reads (< 1 %): individual, not population


def test_wide_kernel_on_tiles_that_some_samples_have_no_reads_in(monkeypatch):
    """MSNV_DEEP=wide keeps deep (sample, tile) runs whole for msnv_pileup_tiles_wide.  A tile numbers its slots without the samples
    that have no reads in it, so a pair's slot is not its sample: the kernel took the sample's columns by the SLOT until round 3 and
    piled up somebody else's bases whenever a sample was absent (found with the guarded allocator, tests/test_gpu_guard.py)."""
    monkeypatch.setenv("MSNV_DEEP", "wide")
    for kw in (dict(n_species=2, contig_len=9000, n_samples=6, mean_cov=400.0, sigma_cov=1.2, seed=73),
               dict(n_species=3, contig_len=5000, n_samples=9, mean_cov=300.0, sigma_cov=0.8, frac_absent=0.4, snv_density=0.02, seed=74)):
        syn, samples = synth_case(**kw)
        prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
        _assert_same(prod, orac)
        assert prod[3]["n_overflow"] > 0                  # the >= 255 path of the wide kernel really ran


@pytest.mark.parametrize("deep", ["split", "wide"])
def test_token_limit_cuts_bases_whatever_the_quality_cutoff(deep, monkeypatch):
    """snpCall keeps 10 000 characters of a sample's base string (call_vC.cpp:48,92-111): behind a stack of reads that start at one
    position (`^]` costs two characters per start) the last bases of a 6 600-deep position are cut.  The host marks them; until round 3
    the mark was "quality 0", which a cutoff of 0 (mpileup -Q 0) does not drop -- 34 bases too many at one position of this cohort
    (found by the fuzz sweep, profiles/r03zr_fuzz_deep_wide.txt).  -Q 0 and the default -Q 13, both deep modes."""
    monkeypatch.setenv("MSNV_DEEP", deep)
    kw = dict(n_species=3, contig_len=1500, n_samples=2, mean_cov=300, read_len=100, sigma_cov=1.0, frac_absent=0.1, snv_density=0.0, error_rate=0.02,
              frac_lowq=0.5, frac_indel_reads=0.0, frac_clip_reads=0.3, frac_flagged=0.0, lowercase_ref=1, frac_paired=0.5, seed=710363175)
    syn, samples = synth_case(**kw)
    for q in (0, 13):
        p = core.default_params(min_coverage=10, calling_threshold=4, min_fraction=0.0, min_baseq=q, count_orphans=1, ignore_overlaps=1)
        prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same(prod, orac)
        assert "\t1397\t" in prod[0]


def test_many_samples_more_than_one_wave_of_columns():
    """300 samples (more than 256 columns: every 64-lane sample loop wraps several times), shallow coverage."""
    syn, samples = synth_case(n_species=1, contig_len=2500, n_samples=300, mean_cov=3.0, snv_density=0.05, frac_absent=0.3, seed=300)
    p = core.default_params(min_coverage=4, calling_threshold=3)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 20


def test_long_reads_span_pieces_and_tiles():
    """Reads of several thousand bases: each is cut into many 128-base pieces, crosses tile boundaries (2048) and carries
    indels, soft clips and '=' / 'X' ops; a second sample holds short reads on the same contig."""
    import random
    rnd = random.Random(9)
    L = 9000
    ref = "".join(rnd.choice("ACGT") for _ in range(L))
    def long_read(start, n_ref, name):
        cigar, seq, pos = [], [], start
        cigar.append("7S"); seq.append("".join(rnd.choice("ACGT") for _ in range(7)))
        left = n_ref
        while left > 0:
            m = min(left, rnd.randrange(150, 900))
            op = rnd.choice("MMM=X") if m < 400 else "M"
            chunk = list(ref[pos:pos + m])
            if op != "=":
                for _ in range(max(1, m // 60)):
                    k = rnd.randrange(m)
                    chunk[k] = rnd.choice("ACGT")
            cigar.append("%d%s" % (m, op)); seq.append("".join(chunk)); pos += m; left -= m
            if left > 0:
                if rnd.random() < 0.5:
                    n = rnd.randrange(1, 6); cigar.append("%dI" % n); seq.append("".join(rnd.choice("ACGT") for _ in range(n)))
                else:
                    n = min(left, rnd.randrange(1, 9)); cigar.append("%dD" % n); pos += n; left -= n
        return bt.make_record(0, start, "".join(cigar), "".join(seq), qual=[rnd.choice([5, 20, 30, 38]) for _ in range(sum(len(s) for s in seq))], name=name)
    long_sample = sorted([long_read(rnd.randrange(0, 3000), rnd.randrange(2500, 5500), "L%d" % k) for k in range(12)],
                         key=lambda r: int.from_bytes(r[8:12], "little", signed=True))
    short = []
    for k in range(400):
        s = rnd.randrange(0, L - 100)
        q = list(ref[s:s + 100])
        if k % 3 != 1:
            gpos = (s + 50) // 40 * 40                   # shared mutation positions -> many called sites
            q[gpos - s] = "A" if ref[gpos] != "A" else "C"
        short.append(bt.make_record(0, s, "100M", "".join(q), name="s%d" % k))
    short.sort(key=lambda r: int.from_bytes(r[8:12], "little", signed=True))
    samples = [bt.records(*long_sample), bt.records(*short)]
    p = core.default_params(min_coverage=3, calling_threshold=2)
    prod = run_product(["ctg"], [L], [ref], samples, params=p)
    orac = run_oracle(["ctg"], [L], [ref], samples, params=p)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 30, prod[0].count("\n")


@pytest.mark.parametrize("read_len", [36, 50, 75])
def test_short_reads_use_the_dense_block_layout(read_len):
    """Mean piece length below 72 bases -> finalize packs the dense 32-base block stream and msnv_pileup_tiles_dense
    runs (two segments per block, pieces that start mid-block, odd lengths with a pad nibble, fresh-block rule)."""
    syn, samples = synth_case(n_species=2, contig_len=6000, n_samples=7, mean_cov=14.0, snv_density=0.03, read_len=read_len,
                              lowercase_ref=1, seed=500 + read_len)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 50
    assert prod[2]["bytes_headers"] * 8 < prod[2]["n_pileup_bases"] * 2     # 4 B per 32-base block, not 8 B per short piece


@pytest.mark.parametrize("layout", ["pieces", "dense"])
def test_proper_pairs_with_overlapping_mates(layout, monkeypatch):
    """Paired-end input, most mates overlapping on the reference: mpileup (no -x, metaSNV.py:160-165) counts a template once
    per position (sam.c tweak_overlap_quality).  The host stage edits the qualities before packing; called_SNPs and
    indiv_called equal the oracle's with the tweak on (default) and with -x, and the two differ."""
    monkeypatch.setenv("MSNV_LAYOUT", layout)
    syn, samples = synth_case(n_species=2, contig_len=9000, n_samples=6, mean_cov=16.0, snv_density=0.03, frac_paired=0.8,
                              read_len=100 if layout == "pieces" else 50, seed=77)
    texts = []
    for x in (0, 1):
        p = core.default_params(ignore_overlaps=x)
        prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same(prod, orac)
        assert prod[2]["n_pileup_bases"] == orac[3]
        texts.append(prod[0])
    assert texts[0] != texts[1] and texts[0].count("\n") > 100


def test_token_limit_of_snpcall_cuts_a_deep_stack_like_the_reference():
    """call_vC.cpp:481-483: a sample's base string is cut at 10000 characters.  A stack of ~5600 reads starting at one position
    (3 characters per read start) passes the limit; the host stage takes the bases behind the cut out of the pileup and the
    device counts (deep pair, split into byte-bin groups) equal the oracle's, which builds and cuts the text like snpCall."""
    from test_overlap_host import _stack
    ref, s = _stack(11)
    other = bt.records(bt.make_record(0, 90, "60M", ref[90:150], name="o1"), bt.make_record(0, 95, "60M", ref[95:155], name="o2"))
    p = core.default_params(min_coverage=1, calling_threshold=2)
    prod = run_product(["c1"], [len(ref)], [ref], [other, s], params=p)
    orac = run_oracle(["c1"], [len(ref)], [ref], [other, s], params=p)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 30
    if os.environ.get("MSNV_PACK", "d")[0] != "h" and os.environ.get("MSNV_PREPASS", "d")[0] != "h":      # round 6: the cut is a kernel's (devpack.hip: msnv_token_cut), no sample goes through the host pre-pass
        assert prod[2]["pack_stats"]["device_edit_samples"] == 1 and prod[2]["pack_stats"]["prepass_samples"] == 0, prod[2]["pack_stats"]
    p0 = core.default_params(min_coverage=1, calling_threshold=2, token_limit=0)         # every base counted: not what snpCall does
    assert run_product(["c1"], [len(ref)], [ref], [other, s], params=p0)[0] != prod[0]


def _random_flagged_samples(seed, L=3000, n_reads=700, n_samples=3):
    import random
    rnd = random.Random(seed)
    ref = "".join(rnd.choice("ACGT") for _ in range(L))
    samples = []
    for s in range(n_samples):
        recs = []
        for k in range(n_reads):
            st = rnd.randrange(0, L - 80)
            n = rnd.randrange(30, 80)
            q = list(ref[st:st + n])
            g = (st + n // 2) // 25 * 25
            if st <= g < st + n and k % 2 == 0:
                q[g - st] = "T" if ref[g] != "T" else "G"
            flag = rnd.choice([0, 0, 16, 0x1 | 0x2 | 0x40, 0x1 | 0x2 | 0x80 | 16, 0x1 | 0x40, 0x1 | 0x80 | 0x8, 0x400, 0x100, 0x800, 0x200])
            recs.append((st, bt.make_record(0, st, "%dM" % n, "".join(q), flag=flag, mapq=rnd.choice([0, 1, 5, 20, 60, 60]),
                                            qual=[rnd.choice([2, 12, 13, 14, 30, 40]) for _ in range(n)], name="r%d_%d" % (s, k))))
        recs.sort(key=lambda t: t[0])
        samples.append(bt.records(*[r for _, r in recs]))
    return ref, samples


def test_read_filters_orphans_mapq_and_depth_cap():
    """samtools' read-level filters as mpileup applies them before the pileup (Appendix C): flag filter 0x704, orphans
    (paired but not proper) dropped unless count_orphans, min MAPQ, and the per-file depth cap (-d) in several settings."""
    ref, samples = _random_flagged_samples(71)
    for kw in (dict(), dict(count_orphans=1), dict(min_mapq=2), dict(min_mapq=21, count_orphans=1), dict(flag_filter=0x400),
               dict(max_depth=9), dict(max_depth=3, count_orphans=1), dict(max_depth=1)):
        p = core.default_params(min_coverage=3, calling_threshold=2, **kw)
        prod = run_product(["ctg"], [len(ref)], [ref], samples, params=p)
        orac = run_oracle(["ctg"], [len(ref)], [ref], samples, params=p)
        _assert_same(prod, orac)
        assert prod[2]["n_pileup_bases"] == orac[3], kw
        if "max_depth" in kw and os.environ.get("MSNV_PACK", "d")[0] != "h" and os.environ.get("MSNV_PREPASS", "d")[0] != "h":      # round 6: the cap is a kernel's (msnv_cap_reads)
            assert prod[2]["pack_stats"]["device_edit_samples"] >= 1 and prod[2]["pack_stats"]["prepass_samples"] == 0, (kw, prod[2]["pack_stats"])
    assert prod[0].count("\n") + prod[1].count("\n") > 0


@pytest.mark.parametrize("layout", ["pieces", "dense"])
def test_deep_pairs_with_more_chunks_than_the_descriptor_ring(layout, monkeypatch):
    """A (sample, tile) pair just below the byte-bin depth limit with short reads holds far more 128-piece chunks than the
    32 chunk descriptors a workgroup keeps in LDS (they go through a ring); mixed with pairs above the limit (wide kernel)."""
    monkeypatch.setenv("MSNV_LAYOUT", layout)
    syn, samples = synth_case(n_species=1, contig_len=5000, n_samples=5, mean_cov=170.0, sigma_cov=0.25, read_len=40, snv_density=0.02,
                              frac_absent=0.0, seed=808)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[2]["n_pileup_bases"] == orac[3] and prod[0].count("\n") > 20


@pytest.mark.parametrize("layout", ["pieces", "dense"])
def test_noisy_reads_reserve_their_event_ranges_per_pass(layout, monkeypatch):
    """Several per cent of mismatching bases: one per-sample pass of a tile emits more allele events (2048 positions x
    depth x error rate) than the 256-entry LDS staging buffer holds, so the pass reserves its range of the event list per
    wave and writes at prefix-sum offsets; passes that fit (shallow samples) keep staging -- both paths in one run."""
    monkeypatch.setenv("MSNV_LAYOUT", layout)
    monkeypatch.setenv("MSNV_ALLELES", "events")                      # (reads this noisy would be switched to allele planes: this test is about the event path)
    syn, samples = synth_case(n_species=2, contig_len=9000, n_samples=9, mean_cov=25.0, sigma_cov=1.0, snv_density=0.02,
                              error_rate=0.06, frac_absent=0.0, seed=4242)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[2]["n_pileup_bases"] == orac[3]
    assert prod[3]["n_events"] > 9 * 9 * 256                          # well beyond what staging alone could take per pass
    assert prod[0].count("\n") > 200


@pytest.mark.parametrize("n_samples,mean_cov,sigma,gather_split", [(21, 12.0, 0.4, None), (70, 6.0, 1.2, "1"), (33, 9.0, 0.3, "4"), (16, 14.0, 0.2, None)])
def test_many_sites_per_tile_cells_written_with_wide_stores(n_samples, mean_cov, sigma, gather_split, monkeypatch):
    """Several per cent of mismatches and low thresholds: most positions of a tile are called, so the coverage gather takes its
    dense form, and with >= 16 samples in a tile (rows padded to multiples of 8 cells, first cell a multiple of 8) the one that
    assembles 64 sites x 64 slots in LDS and writes 16 bytes per lane (kernels.hip: gather_cov_wide): slot counts that are and are
    not multiples of 8 / 64, one and several workgroups per tile, tiles whose shallow pairs are merged (their slots are left to the
    merged gather), samples absent from a contig.  Same bytes as the oracle."""
    if gather_split:
        monkeypatch.setenv("MSNV_GATHER_SPLIT", gather_split)
    monkeypatch.setenv("MSNV_MERGE_ALWAYS", "1")
    syn, samples = synth_case(n_species=3, contig_len=5200, n_samples=n_samples, mean_cov=mean_cov, sigma_cov=sigma, snv_density=0.05,
                              error_rate=0.04, frac_absent=0.1, seed=7300 + n_samples)
    p = core.default_params(min_coverage=2, calling_threshold=2)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    _assert_same(prod, orac)
    assert prod[2]["n_pileup_bases"] == orac[3]
    assert prod[0].count("\n") + prod[1].count("\n") > 3 * 2048          # well over 32 sites per gather workgroup in every tile


@pytest.mark.parametrize("layout", ["pieces", "dense"])
def test_shallow_cohort_and_contigs_shorter_than_a_tile(layout, monkeypatch):
    """Many samples at ~1x over contigs of a few hundred bases: every (sample, tile) pair is a single short chunk, so each
    chunk closes its pair and the next pair's column loads are issued under the per-sample pass (narrow32), work items hold
    the maximum of 32 pairs, most tiles are mostly empty, and some samples have no read at all on a contig."""
    monkeypatch.setenv("MSNV_LAYOUT", layout)
    syn, samples = synth_case(n_species=40, contig_len=420, n_samples=120, mean_cov=1.2, sigma_cov=0.8, snv_density=0.05,
                              frac_absent=0.2, seed=1601)
    p = core.default_params(min_coverage=3, calling_threshold=2)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    _assert_same(prod, orac)
    assert prod[2]["n_pileup_bases"] == orac[3]
    assert prod[2]["n_pairs"] > 2000 and prod[0].count("\n") > 100


@pytest.mark.parametrize("shallow_pieces", ["0", "48", "400"])
def test_merged_groups_of_shallow_pairs(shallow_pieces, monkeypatch):
    """Cohort of many shallow samples: the (sample, tile) pairs whose depth bound is at most MSNV_SHALLOW_MAX are merged into
    groups that share one set of LDS bins and one pass (msnv_pileup_tiles_merged); their per-sample allele counts travel as one
    event per mismatching base and their per-sample coverage at the called positions is recomputed from the pieces.  Same bytes
    as the oracle with merging off (0), with the default bound (48 pieces: the ~1x samples merge, the deeper ones do not) and with
    every pair of the cohort merged (400), population and individual calls (threshold 2 is reached inside single samples)."""
    monkeypatch.setenv("MSNV_SHALLOW_PIECES", shallow_pieces)
    monkeypatch.setenv("MSNV_LAYOUT", "pieces")
    sp = core.synth_params(n_species=3, contig_len=7000, n_samples=150, mean_cov=1.5, sigma_cov=1.0, snv_density=0.04, error_rate=0.01,
                           frac_absent=0.15, lowercase_ref=1, seed=1700)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    for kw in (dict(min_coverage=3, calling_threshold=2), dict(min_coverage=4, calling_threshold=4, min_fraction=0.2)):
        p = core.default_params(**kw)
        prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same(prod, orac)
        assert prod[2]["n_pileup_bases"] == orac[3]
    assert prod[0].count("\n") + prod[1].count("\n") > 50


@pytest.mark.gpu
@pytest.mark.parametrize("fuse,lean", [("1", "1"), ("1", "0"), ("0", "1")])
def test_whole_tile_work_items(fuse, lean, monkeypatch):
    """Sparse cohorts: a tile whose pairs fit one merged group is piled up by ONE workgroup, which applies the gates and the calling
    rule itself and leaves a record list for the gate kernel (kernels.hip: fused_tile_gate; pack.cpp: fuse_tile).  MSNV_FUSE=1 forces
    the path on cohorts of any shape, 0 switches it off.  Round 6: a whole-tile item of one chunk costs what its pieces cost, not what 2 048
    positions cost (msnv_pileup_tiles_lean: allele bins only, the candidates' coverage counted from the pieces in the registers); MSNV_LEAN=0
    sends the items through the ordinary body.  Same bytes as the oracle for (a) a sparse cohort with a lower-case
    reference and a BED split, several thresholds; (b) SNVs so dense that tiles hold more candidates than a record list (those tiles
    go through the ordinary gate kernel: test_whole_tile_items_with_more_candidates_than_a_record_list); (c) a cohort where some tiles
    are fused and others hold deep / split pairs."""
    monkeypatch.setenv("MSNV_FUSE", fuse)
    monkeypatch.setenv("MSNV_LEAN", lean)
    monkeypatch.setenv("MSNV_LAYOUT", "pieces")
    syn, samples = synth_case(n_species=9, contig_len=5000, n_samples=40, mean_cov=4.0, sigma_cov=0.6, snv_density=0.004, error_rate=0.004,
                              frac_absent=0.85, lowercase_ref=1, seed=6100)
    for kw in (dict(), dict(min_coverage=1, calling_threshold=1), dict(min_coverage=3, calling_threshold=2, min_fraction=0.4), dict(calling_threshold=1, min_coverage=2, min_fraction=0.0)):
        p = core.default_params(**kw)
        pop, ind, info, st, ds, ctx = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p, return_ds=True)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same((pop, ind, info, st), orac)
        ds.run()                                                    # a second pass over the resident dataset
        with tempfile.TemporaryDirectory() as td:
            ds.write_calls(os.path.join(td, "p"), os.path.join(td, "i"), None, None)
            assert open(os.path.join(td, "p")).read() == pop and open(os.path.join(td, "i")).read() == ind
        ds.close(); ctx.close()
    bed = [(t, 1, syn.lengths[t]) for t in (1, 4, 7)]
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
    _assert_same(prod, orac)
    # (b) more candidates per tile than a record list holds
    syn2, samples2 = synth_case(n_species=3, contig_len=4500, n_samples=5, mean_cov=6.0, sigma_cov=0.3, snv_density=0.2, error_rate=0.02, frac_absent=0.3, seed=6101)
    p = core.default_params(min_coverage=2, calling_threshold=2)
    prod = run_product(syn2.names, syn2.lengths, syn2.seqs, samples2, params=p)
    orac = run_oracle(syn2.names, syn2.lengths, syn2.seqs, samples2, params=p)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 100
    # (c) fused tiles next to tiles with a deep sample
    syn3, samples3 = synth_case(n_species=6, contig_len=4200, n_samples=12, mean_cov=5.0, sigma_cov=2.0, snv_density=0.01, frac_absent=0.5, read_len=50, seed=6102)
    prod = run_product(syn3.names, syn3.lengths, syn3.seqs, samples3)
    orac = run_oracle(syn3.names, syn3.lengths, syn3.seqs, samples3)
    _assert_same(prod, orac)


@pytest.mark.gpu
def test_whole_tile_items_with_more_candidates_than_a_record_list(monkeypatch):
    """A whole-tile work item that finds more than STAGE_CAP (24) candidate positions does not fit its record list: THAT tile leaves
    the workgroup the unfused way (partial row, allele totals, marks: kernels.hip fused_tile_spill), goes on a device-side list and
    through msnv_gate_sites behind msnv_gate_staged; the other tiles keep their record lists (until round 3 one such tile sent the
    whole dataset back to the unfused pass).  Cohorts where (a) some tiles are listed and most are not, (b) tiles with ONE pair are
    listed (their cells are written by the gate kernel: the merged gather is not launched for them), (c) every tile is listed
    (threshold 1: every error is a candidate); lower-case reference; repeated and overlapped passes over the resident dataset (the
    list lives in both sets of intermediates).  Same bytes as the oracle, and the info block says how many tiles were listed."""
    monkeypatch.setenv("MSNV_FUSE", "1")
    monkeypatch.setenv("MSNV_LAYOUT", "pieces")
    cases = [
        (dict(n_species=12, contig_len=5000, n_samples=16, mean_cov=5.0, sigma_cov=0.5, snv_density=0.012, error_rate=0.004, frac_absent=0.6, lowercase_ref=1, seed=6301),
         dict(min_coverage=2, calling_threshold=2), "some"),
        (dict(n_species=10, contig_len=4500, n_samples=3, mean_cov=7.0, sigma_cov=0.4, snv_density=0.03, error_rate=0.004, frac_absent=0.6, lowercase_ref=1, seed=6302),
         dict(min_coverage=2, calling_threshold=2), "some"),
        (dict(n_species=5, contig_len=4200, n_samples=8, mean_cov=4.0, sigma_cov=0.5, snv_density=0.01, error_rate=0.02, frac_absent=0.5, seed=6303),
         dict(min_coverage=1, calling_threshold=1, min_fraction=0.0), "all"),
    ]
    for sk, pk, how in cases:
        syn, samples = synth_case(**sk)
        p = core.default_params(**pk)
        pop, ind, info, st, ds, ctx = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p, return_ds=True)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same((pop, ind, info, st), orac)
        i2 = ds.info()
        assert i2["n_whole_tile_items"] > 0
        if how == "some":
            assert 0 < i2["n_listed_tiles"] < i2["n_whole_tile_items"], i2
        else:
            assert i2["n_listed_tiles"] > 0.5 * i2["n_whole_tile_items"], i2      # (the short last tile of a contig may hold few)
        for again in ("run", "many", "overlap"):
            if again == "run": ds.run()
            else: ds.run_many(3, overlap=again == "overlap")
            assert ds.info()["n_listed_tiles"] == i2["n_listed_tiles"]
            with tempfile.TemporaryDirectory() as td:
                ds.write_calls(os.path.join(td, "p"), os.path.join(td, "i"), None, None)
                assert open(os.path.join(td, "p")).read() == pop and open(os.path.join(td, "i")).read() == ind, again
        ds.close(); ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gate_tiles", ["1", "3", "8"])
def test_gate_kernel_with_several_tiles_per_workgroup(gate_tiles, monkeypatch):
    """msnv_gate_sites hands out site slots and per-sample cells with one reservation per workgroup; with many active tiles a
    workgroup takes several consecutive tiles and stages their sites in LDS (MSNV_GATE_TILES sets how many).  Same bytes as the
    oracle when (a) most tiles hold a handful of sites, some none (staged, several tiles per reservation), (b) SNVs are so dense
    that tiles hold more sites than the stage (flushed early; tiles above 384 sites written directly), (c) a deep sample is split
    into several pairs of a tile and shallow ones are merged (positions decided behind the scatter go through the stage too)."""
    monkeypatch.setenv("MSNV_GATE_TILES", gate_tiles)
    cases = [
        (dict(n_species=14, contig_len=4500, n_samples=10, mean_cov=6.0, sigma_cov=1.0, snv_density=0.004, error_rate=0.004, frac_absent=0.5,
              lowercase_ref=1, seed=515), dict()),
        (dict(n_species=3, contig_len=6500, n_samples=6, mean_cov=12.0, sigma_cov=0.3, snv_density=0.30, error_rate=0.01, frac_absent=0.0,
              seed=516), dict(min_coverage=2, calling_threshold=2)),
        (dict(n_species=4, contig_len=5000, n_samples=40, mean_cov=2.0, sigma_cov=2.2, snv_density=0.03, error_rate=0.01, frac_absent=0.2,
              read_len=60, lowercase_ref=1, seed=517), dict(min_coverage=3, calling_threshold=3)),
    ]
    n_lines = []
    for sk, pk in cases:
        syn, samples = synth_case(**sk)
        p = core.default_params(**pk)
        prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same(prod, orac)
        n_lines.append(prod[0].count("\n") + prod[1].count("\n"))
    assert n_lines[0] > 10 and n_lines[1] > 3 * 400 and n_lines[2] > 10, n_lines


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["pieces", "dense"])
@pytest.mark.parametrize("tot_mode", ["0", "1", "2"])
def test_allele_total_modes(tot_mode, layout, monkeypatch):
    """The allele totals of a tile are as narrow as its summed depth bound allows (pack.cpp: 4 bytes in one word per position,
    2 x u16 in two words, or four words; kernels.hip tot_add / msnv_gate_sites).  An uneven cohort has tiles of the first two
    kinds (a few shallow samples here, a deep one there, one sample deep enough for the wide kernel); MSNV_TOT_MODE raises the
    narrowest mode allowed so that every width runs on every tile.  Same bytes as the oracle each time, two passes per dataset
    (the gate kernel leaves the totals zero for the next pass in every mode)."""
    monkeypatch.setenv("MSNV_TOT_MODE", tot_mode)
    monkeypatch.setenv("MSNV_LAYOUT", layout)
    sp = core.synth_params(n_species=6, contig_len=5000, n_samples=14, mean_cov=9.0, sigma_cov=1.6, snv_density=0.03, error_rate=0.01,
                           frac_absent=0.4, lowercase_ref=1, seed=4242)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    for kw in (dict(min_coverage=3, calling_threshold=2), dict()):
        p = core.default_params(**kw)
        pop, ind, info, st, ds, ctx = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p, return_ds=True)
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
        _assert_same((pop, ind, info, st), orac)
        st2 = ds.run()                                              # second pass over the resident dataset
        with tempfile.TemporaryDirectory() as td:
            ds.write_calls(os.path.join(td, "p"), os.path.join(td, "i"), None, None)
            assert open(os.path.join(td, "p")).read() == pop and open(os.path.join(td, "i")).read() == ind
        ds.close(); ctx.close()
    assert pop.count("\n") + ind.count("\n") > 20


def test_event_list_grows_when_a_sub_list_overflows(monkeypatch):
    """The allele-event list is 32 sub-lists with their own counters; a pass that overflows one reports the capacity the
    fullest asked for, the host grows the list and runs the pass again (msnv_pileup_run) -- same records, same event count."""
    syn, samples = synth_case(n_species=2, contig_len=7000, n_samples=8, mean_cov=14.0, snv_density=0.03, error_rate=0.01, seed=909)
    monkeypatch.setenv("MSNV_ALLELES", "events")                      # (whatever the mismatch rate says: this test is about the event list)
    ref_run = run_product(syn.names, syn.lengths, syn.seqs, samples)
    monkeypatch.setenv("MSNV_CAP_EVENTS", "1024")                    # 32 events per sub-list: every list overflows
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[3]["n_events"] == ref_run[3]["n_events"] > 4096


@pytest.mark.parametrize("knobs", [dict(MSNV_ITEM_PIECES="64"), dict(MSNV_ITEM_PIECES="300", MSNV_ITEM_TAPER="0"),
                                   dict(MSNV_ITEM_PIECES="5000"), dict(MSNV_TAPER_AT="900,300,100"), dict(MSNV_COV_ITEM="7")])
def test_results_do_not_depend_on_the_work_decomposition(knobs, monkeypatch, tmp_path):
    """Work items of 64 ... 5000 pieces, with and without the taper of the last tiles, tiny coverage items: the cut of the
    (tile, sample) pairs into workgroups changes partial rows, event sub-lists and row types (u8 / u16), never the output."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    syn, samples = synth_case(n_species=3, contig_len=7000, n_samples=24, mean_cov=11.0, sigma_cov=0.9, snv_density=0.02, seed=606)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same(prod, orac)
    assert prod[0].count("\n") > 100
    for (got, exp) in _coverage_both(syn.names, syn.lengths, samples[:4], tmp_path)[0]:
        assert got == exp


def test_annotation_codon_path(tmp_path):
    syn, samples = synth_case(n_species=2, contig_len=3000, n_samples=4, mean_cov=14.0, snv_density=0.03, frac_absent=0.0, seed=21)
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    ann = str(tmp_path / "ann.tsv")
    with open(ann, "w") as f:
        f.write("gene_id\texternal_id\tsequence_id\ttype\tinfo\tlength\tstart\tend\tstrand\tsc\tstop\tgc\n")
        rows = [("g1", syn.names[0], 10, 900, "+"), ("g2", syn.names[0], 600, 1500, "-"), ("g3", syn.names[0], 2000, 2000, "+"),
                ("g4", syn.names[0], 2500, 2400, "+"), ("h1", syn.names[1], 1, 2997, "-")]
        for i, (g, c, s, e, st) in enumerate(rows):
            f.write("%d\t%s\t%s\tCDS\tx\t%d\t%d\t%d\t%s\tATG\tTAG\t0.4\n" % (i, g, c, e - s + 1, s, e, st))
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa)
    _assert_same(prod, orac)
    assert "\tg1\t" in prod[0] and "\tg2\t" in prod[0] and "\th1\t" in prod[0] and "|S[" in prod[0] and "|N[" in prod[0]


def _write_ann(path, rows):
    with open(path, "w") as f:
        f.write("gene_id\texternal_id\tsequence_id\ttype\tinfo\tlength\tstart\tend\tstrand\tsc\tstop\tgc\n")
        for i, (g, c, s, e, st) in enumerate(rows):
            f.write("%d\t%s\t%s\tCDS\tx\t%d\t%d\t%d\t%s\tATG\tTAG\t0.4\n" % (i, g, c, e - s + 1, s, e, st))


def _write_fasta(path, names, seqs, width=60):
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n" % n)
            s = s.decode()
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def test_annotation_random_gene_tables_and_odd_codons(tmp_path):
    """Device gene lookup + codon arithmetic against the oracle: overlapping / nested genes in random file
    order, both strands, start column 0 (start = -1), N and lower-case letters inside codons (gene.h stores
    lower case as 'A'; the reverse complement drops N), genes on a contig listed after a gene-less one."""
    import random
    rnd = random.Random(5)
    syn, samples = synth_case(n_species=3, contig_len=4000, n_samples=5, mean_cov=12.0, snv_density=0.04, frac_absent=0.0, seed=33)
    seqs = []
    for s in syn.seqs:
        b = bytearray(s)
        for _ in range(150):
            b[rnd.randrange(len(b))] = ord("N")
        for _ in range(300):
            k = rnd.randrange(len(b))
            b[k] = ord(chr(b[k]).lower())
        seqs.append(bytes(b))
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, syn.names, seqs)
    rows = [("z0", syn.names[0], 0, 700, "-")]
    for c in (0, 2):                                       # contig 1 has no genes
        for k in range(40):
            a = rnd.randrange(1, 3900)
            rows.append(("g%d_%d" % (c, k), syn.names[c], a, min(3990, a + rnd.randrange(0, 600)), rnd.choice("+-")))
    ann = str(tmp_path / "ann.tsv")
    _write_ann(ann, rows)
    prod = run_product(syn.names, syn.lengths, seqs, samples, ann=ann, fasta=fa)
    orac = run_oracle(syn.names, syn.lengths, seqs, samples, ann=ann, fasta=fa)
    _assert_same(prod, orac)
    import re
    assert "\tz0\t" in prod[0]
    assert re.search(r"\[[ACGT]{0,2}-", prod[0] + prod[1])      # a codon that lost a letter in the reverse complement
    assert re.search(r"\[[ACGT]*N", prod[0] + prod[1])          # an N printed on the + strand


def test_annotation_records_path_equals_file_path(tmp_path):
    """Multi-GPU formatter path: annotation records fetched from the device and handed to
    msnv_write_calls_records give the same text as msnv_write_calls."""
    syn, samples = synth_case(n_species=2, contig_len=3000, n_samples=4, mean_cov=14.0, snv_density=0.03, frac_absent=0.0, seed=21)
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    ann = str(tmp_path / "ann.tsv")
    _write_ann(ann, [("g1", syn.names[0], 10, 900, "+"), ("g2", syn.names[0], 600, 1500, "-"), ("h1", syn.names[1], 1, 2997, "-")])
    pop, ind, info, st, ds, ctx = run_product(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa, return_ds=True)
    sites, smp = ds.results()
    recs, ms = ds.annotate(ann, fa)
    assert len(recs) == len(sites) and (recs["gene"] >= 0).any() and ms >= 0
    core.write_calls_records(syn.names, ds.n_samples, sites, smp, str(tmp_path / "p"), str(tmp_path / "i"), ann, fa, recs)
    assert open(tmp_path / "p").read() == pop and open(tmp_path / "i").read() == ind
    with pytest.raises(Exception):                          # the annotation is never recomputed on the host
        core.write_calls_records(syn.names, ds.n_samples, sites, smp, str(tmp_path / "p2"), None, ann, fa, None)


def test_annotation_domain_errors_match_the_oracle(tmp_path):
    """A contig with gene rows but no FASTA record makes the reference dereference map::end():
    both the oracle and the device path refuse (MSNV_EDOMAIN)."""
    import orc
    syn, samples = synth_case(n_species=2, contig_len=3000, n_samples=4, mean_cov=14.0, snv_density=0.03, frac_absent=0.0, seed=21)
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, syn.names[:1], syn.seqs[:1])
    ann = str(tmp_path / "ann.tsv")
    _write_ann(ann, [("g1", syn.names[0], 10, 900, "+"), ("h1", syn.names[1], 1, 2997, "-")])
    with pytest.raises(orc.OrcError):
        run_oracle(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa)
    with pytest.raises(Exception) as ei:
        run_product(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa)
    assert "FASTA" in str(ei.value)


def test_one_call_entry_point_from_bam_files(tmp_path):
    import ctypes as C
    from metasnv_amd import _lib
    syn, samples = synth_case(n_species=2, contig_len=4000, n_samples=3, mean_cov=10.0, snv_density=0.02, seed=13)
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    paths = []
    for i, s in enumerate(samples):
        p = str(tmp_path / ("s%04d.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, s)
        paths.append(p)
    ctx = core.Context(0)
    a = _lib.CallArgs()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    a.bam_paths, a.n_bams, a.ref_fasta = arr, len(paths), fa.encode()
    a.out_called_path = str(tmp_path / "called_SNPs").encode()
    a.out_indiv_path = str(tmp_path / "indiv_called").encode()
    _lib.lib.msnv_params_default(C.byref(a.params))
    _lib.check(_lib.lib.msnv_call(ctx._h, C.byref(a)))
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    assert open(tmp_path / "called_SNPs").read() == orac[0]
    assert open(tmp_path / "indiv_called").read() == orac[1]


def test_repeated_runs_are_idempotent():
    syn, samples = synth_case(n_species=1, contig_len=8000, n_samples=6, seed=2)
    pop, ind, info, st, ds, ctx = run_product(syn.names, syn.lengths, syn.seqs, samples, return_ds=True)
    s1, m1 = ds.results()
    for _ in range(3):
        ds.run()
    s2, m2 = ds.results()
    assert s1.tobytes() == s2.tobytes() and m1.tobytes() == m2.tobytes()
    for overlap, n in ((False, 5), (True, 4), (True, 5), (False, 2), (True, 1)):
        stats = ds.run_many(n, overlap)                     # batched forms: n passes, one host sync; two streams when overlapped
        s3, m3 = ds.results()
        assert s1.tobytes() == s3.tobytes() and m1.tobytes() == m3.tobytes(), (overlap, n)
        assert len(stats) == n and all(x["n_sites"] == st["n_sites"] and x["ms_pileup"] > 0 for x in stats)
    ds.run()                                                # the single-pass form still works on whichever set is primary
    s4, m4 = ds.results()
    assert s1.tobytes() == s4.tobytes() and m1.tobytes() == m4.tobytes()


def _coverage_both(names, lengths, samples, tmp_path, params=None):
    ctx = core.Context(0)
    ds = core.Dataset(ctx, names, lengths, None, params)
    for s in samples:
        ds.add_sample_records(s)
    ds.finalize()
    st = ds.coverage_run()
    out = []
    for i, s in enumerate(samples):
        cp, dp = str(tmp_path / ("s%d.cov" % i)), str(tmp_path / ("s%d.cov.detail" % i))
        ds.write_coverage(i, cp, dp)
        p = params or core.default_params()
        out.append(((open(cp).read(), open(dp).read()), orc.qacompute(names, lengths, s, max_cov=p.cov_max, min_mapq=p.cov_min_mapq)))
    ds.close(); ctx.close()
    return out, st


def test_coverage_matches_qacompute_restatement(tmp_path):
    syn, samples = synth_case(n_species=3, contig_len=7000, n_samples=5, mean_cov=6.0, frac_absent=0.3, seed=17)
    # header contigs before, between and after the covered ones print zero rows (printSkipped)
    names = ["empty.a"] + syn.names[:2] + ["empty.b", syn.names[2], "empty.c"]
    lengths = [1234] + syn.lengths[:2] + [99, syn.lengths[2], 5000]
    remap = {0: 1, 1: 2, 2: 4}
    fixed = []
    for s in samples:
        b = bytearray(s.tobytes())
        off = 0
        while off < len(b):
            bs = int.from_bytes(b[off:off + 4], "little")
            tid = int.from_bytes(b[off + 4:off + 8], "little", signed=True)
            b[off + 4:off + 8] = remap[tid].to_bytes(4, "little", signed=True)
            off += bs + 4
        fixed.append(np.frombuffer(bytes(b), dtype=np.uint8))
    fixed = [s for s in fixed if s.size]
    res, st = _coverage_both(names, lengths, fixed, tmp_path)
    for (got, want) in res:
        assert got[0] == want[0]
        assert got[1] == want[1]
    assert "empty.a\t1234\t0.00000" in res[0][0][0]


def test_coverage_cigar_quirks_and_filters(tmp_path):
    L = 300
    ref = "ACGT" * 75
    rec = [
        bt.make_record(0, 10, "5S20M", "N" * 5 + ref[10:30]),                       # leading clip skipped without advancing
        bt.make_record(0, 10, "10M5I10M", ref[10:20] + "GGGGG" + ref[20:30]),       # insertion advances the cursor
        bt.make_record(0, 12, "10M4D10M", ref[12:22] + ref[26:36]),
        bt.make_record(0, 15, "8=4X8=", ref[15:35]),                                # '=' / 'X' blocks are never counted
        bt.make_record(0, 20, "10M3S", ref[20:30] + "NNN"),
        bt.make_record(0, 30, "20M", ref[30:50], mapq=0),                           # sub-par mapping quality
        bt.make_record(0, 30, "20M", ref[30:50], flag=0x400),                       # duplicate
        bt.make_record(0, 30, "20M", ref[30:50], flag=0x100 | 0x200 | 0x1),         # secondary/QC-fail/orphan ARE counted
        bt.make_record(0, 280, "19M", ref[280:299]),                                # hangs into the clamp at L-1
        bt.make_record(0, 285, "14M", ref[285:299]),
        bt.make_record(-1, -1, "*", "ACGT", flag=4),
    ]
    s1 = bt.records(*rec)
    s2 = bt.records(*[bt.make_record(1, 5 * k, "50M", "A" * 50, name="k%d" % k) for k in range(60)])     # depth well above -c 10
    res, st = _coverage_both(["c1", "c2"], [L, 400], [s1, s2], tmp_path)
    for (got, want) in res:
        assert got[0] == want[0]
        assert got[1] == want[1]
    for p in (core.default_params(cov_max=3, cov_min_mapq=0), core.default_params(cov_max=15)):
        for (got, want) in _coverage_both(["c1", "c2"], [L, 400], [s1, s2], tmp_path, p)[0]:
            assert got == want


def test_coverage_reads_at_the_contig_end_do_not_fail_the_run(tmp_path):
    """qaCompute.cpp:542-549: an M op whose cursor (pos + 1, advanced by EVERY earlier op) is at or beyond the contig length only
    decrements the last position, which no read can cover -- its coverage becomes -1 and the reference increments
    coverageHist[-1] (undefined).  One such read must not fail a whole metaSNV run: the library warns and computes the
    reference's arithmetic short of the out-of-bounds write (covSum takes the -1, the position lands in no bin), and so does
    the oracle."""
    L = 100
    ref = "ACGT" * 25
    s = bt.records(*[bt.make_record(0, 60, "40M", ref[60:100], name="c%d" % i) for i in range(3)],
                   bt.make_record(0, 90, "5M8I2M", ref[90:95] + "G" * 8 + "AC", name="t2"),      # second M op at cursor 104 > L
                   bt.make_record(0, 97, "2M", "CG", name="t0"),
                   bt.make_record(0, 99, "1M4S", "ACGTA", name="t1"))                              # cursor == L
    res, st = _coverage_both(["c1", "c2"], [L, 50], [s], tmp_path)
    assert res[0][0] == res[0][1]
    assert res[0][0][1].startswith("c1\t100\t38\t")           # indices 61..98 covered, index 99 (coverage -2) in no bin


def test_coverage_kernel_variants_and_a_pile_of_reads_with_one_start(tmp_path, monkeypatch):
    """msnv_coverage_tiles keeps two positions per LDS word (16-bit differences) for pairs of at most 32 767 intervals and one
    word per position beyond: (a) both variants on the same ordinary data, (b) 40 000 reads that start on one position (the
    16-bit form would wrap) next to ordinary samples."""
    syn, samples = synth_case(n_species=2, contig_len=5000, n_samples=4, mean_cov=8.0, seed=23)
    samples = [s for s in samples if s.size]
    want, _ = _coverage_both(syn.names, syn.lengths, samples, tmp_path)
    monkeypatch.setenv("MSNV_COV_NARROW_MAX", "1")                   # every work item goes to the one-word-per-position variant
    got, _ = _coverage_both(syn.names, syn.lengths, samples, tmp_path)
    monkeypatch.delenv("MSNV_COV_NARROW_MAX")
    for (g, w) in zip(got, want):
        assert g[0] == g[1] and g[0] == w[0]
    L = 5000
    ref = "ACGT" * (L // 4)
    pile = [bt.make_record(0, 2100, "60M", ref[2100:2160], name="p%d" % i) for i in range(40000)]
    rest = [bt.make_record(0, 2090 + 7 * k, "50M", ref[2090 + 7 * k:2140 + 7 * k], name="r%d" % k) for k in range(40)]
    order = sorted(pile + rest, key=lambda r: int.from_bytes(r[8:12], "little", signed=True))
    s1 = bt.records(*order)
    s2 = bt.records(*[bt.make_record(0, 13 * k, "70M", ref[13 * k:13 * k + 70], name="q%d" % k) for k in range(300)])
    res, _ = _coverage_both(["c1"], [L], [s1, s2], tmp_path)
    for (g, w) in res:
        assert g == w


def _write_inputs(tmp_path, syn, samples):
    fa = str(tmp_path / "ref.fa")
    syn.write_fasta(fa)
    paths = []
    for i, s in enumerate(samples):
        p = str(tmp_path / ("s%04d.insilico.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, s)
        paths.append(p)
    lst = str(tmp_path / "all_samples")
    open(lst, "w").write("\n".join(paths) + "\n")
    return fa, paths, lst


def test_config3_shape_fused_coverage_and_calls(tmp_path):
    """BASELINE configs[2] at reduced size: many species, every sample carries a random subset of them,
    qaCompute + snpCall from ONE resident dataset (msnv_fused_run), both checked against the oracle."""
    syn, samples = synth_case(n_species=24, contig_len=3000, n_samples=12, mean_cov=9.0, snv_density=0.02, frac_absent=0.6,
                              lowercase_ref=1, seed=303)
    ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    for s in samples:
        ds.add_sample_records(s)
    info = ds.finalize()
    st_p, st_c = ds.fused_run()
    pp, ip = str(tmp_path / "called"), str(tmp_path / "indiv")
    ds.write_calls(pp, ip)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same((open(pp).read(), open(ip).read()), orac)
    for i, s in enumerate(samples):
        cp, dp = str(tmp_path / ("s%d.cov" % i)), str(tmp_path / ("s%d.cov.detail" % i))
        if s.size == 0:
            continue
        ds.write_coverage(i, cp, dp)
        want = orc.qacompute(syn.names, syn.lengths, s)
        assert open(cp).read() == want[0] and open(dp).read() == want[1]
    assert st_p["n_called_pop"] == orac[0].count("\n") and info["n_contigs"] == 24
    ds.close(); ctx.close()


def test_process_level_drop_ins(tmp_path):
    """msnv_qacompute (argv of qaCompute as metaSNV.py:63-65 passes it) and msnv_snpcall (the mpileup | snpCall pipe
    of metaSNV.py:160-176 as one process, population lines on stdout) against the oracle."""
    import subprocess
    tools = os.path.join(os.path.dirname(core._lib.LIB_PATH), "tools")
    syn, samples = synth_case(n_species=2, contig_len=4000, n_samples=3, mean_cov=10.0, snv_density=0.02, seed=13)
    fa, paths, lst = _write_inputs(tmp_path, syn, samples)
    out = str(tmp_path / "s0.cov")
    r = subprocess.run([os.path.join(tools, "msnv_qacompute"), "-c", "10", "-d", "-i", paths[0], out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "Printing details in %s.detail!\n" % out
    want = orc.qacompute(syn.names, syn.lengths, samples[0])
    assert open(out).read() == want[0] and open(out + ".detail").read() == want[1]
    indiv = str(tmp_path / "indiv_called")
    r = subprocess.run([os.path.join(tools, "msnv_snpcall"), "-f", fa, "-b", lst, "-i", indiv, "-c", "4", "-t", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    assert r.stdout == orac[0] and open(indiv).read() == orac[1]
    r = subprocess.run([os.path.join(tools, "msnv_qacompute"), "-c", "10", "-d", "-i", str(tmp_path / "missing.bam"), out], capture_output=True, text=True)
    assert r.returncode == 1                              # qaCompute.cpp:376-379


@pytest.mark.parametrize("case,argv", [("filtering", []), ("filtering2", ["-m", "2", "-d", "1", "-b", "10", "-c", "3", "-p", "0.4", "--ind"])])
def test_filtering_on_device_matches_reference_outputs(tmp_path, golden_dir, case, argv):
    """metaSNV_Filtering.py filter_two on the device: the project directories under
    tests/golden/python_callers/filtering* hold the inputs and the outputs the reference script produced."""
    import filecmp
    import shutil
    from metasnv_amd import filtering
    src = os.path.join(golden_dir, "python_callers", case, "proj")
    proj = str(tmp_path / "proj")
    shutil.copytree(src, proj)
    shutil.rmtree(os.path.join(proj, "filtered"))
    filtering.main([proj] + argv)
    for sub in ("pop", "ind"):
        want_dir, got_dir = os.path.join(src, "filtered", sub), os.path.join(proj, "filtered", sub)
        want = sorted(os.listdir(want_dir)) if os.path.isdir(want_dir) else []
        got = sorted(os.listdir(got_dir)) if os.path.isdir(got_dir) else []
        assert got == want
        for f in want:
            assert open(os.path.join(got_dir, f)).read() == open(os.path.join(want_dir, f)).read(), f


def test_filtering_of_pipeline_output_matches_python_restatement(tmp_path):
    """Filtering of OUR called_SNPs (160-column-style lines from the device) against a plain-Python evaluation of the
    reference's formulas (metaSNV_Filtering.py:183-231) on the same text."""
    from metasnv_amd import cli, filtering
    syn, samples = synth_case(n_species=3, contig_len=5000, n_samples=6, mean_cov=12.0, snv_density=0.03, frac_absent=0.1, seed=77)
    fa, paths, lst = _write_inputs(tmp_path, syn, samples)
    proj = str(tmp_path / "outf")
    cli.main([proj, lst, fa])
    filtering.main([proj, "-m", "2", "-d", "2", "-b", "20", "-c", "4", "-p", "0.5", "--ind"])
    soi = filtering.relevant_taxa(os.path.join(proj, "outf.all_cov.tab"), os.path.join(proj, "outf.all_perc.tab"), 20.0, 2.0, 2)["SoI"]
    header = [p.split("/")[-1] for p in open(os.path.join(proj, "all_samples")).read().splitlines()]
    assert soi
    for sub, fname in (("pop", "called_SNPs"), ("ind", "indiv_called")):
        for sp, names in soi.items():
            idx = [header.index(n) for n in names]
            want = ""
            for line in open(os.path.join(proj, "snpCaller", fname)):
                w = line.split()
                if w[0].split(".")[0] != sp:
                    continue
                cov = list(map(int, w[4].split("|")))
                good = sum(1 for i in idx if not (cov[i] < 4.0 or cov[i] == 0))
                if float(good) / len(idx) < 0.5:
                    continue
                for snp in w[5].split(","):
                    x = snp.split("|")
                    c = list(map(float, x[3:]))
                    fr = [c[i] / cov[i] if (cov[i] >= 4.0 and cov[i] != 0) else -1 for i in idx]
                    want += ":".join(w[:4]) + ">" + x[1] + ":" + x[2] + "\t" + "\t".join(str(v) for v in fr) + "\n"
            path = os.path.join(proj, "filtered", sub, sp + ".filtered.freq")
            if want:
                assert open(path).read() == "\t" + "\t".join(names) + "\n" + want
            else:
                assert not os.path.exists(path)


def test_filter_from_resident_records_equals_the_file_path(tmp_path):
    """filter_two fed from the records of the last pass (msnv_filter_resident) writes the same
    <species>.filtered.freq bytes as the path that parses the called_SNPs / indiv_called text the same dataset wrote
    (msnv_filter_files, itself pinned by the reference script's outputs), population and --ind, with codon annotation."""
    from metasnv_amd import filtering
    sp = core.synth_params(n_species=6, contig_len=5000, n_samples=9, mean_cov=9.0, sigma_cov=0.8, snv_density=0.04, frac_absent=0.2,
                           contigs_per_species_max=3, seed=808)
    syn = core.Synth(sp)
    fa, ann = str(tmp_path / "ref.fa"), str(tmp_path / "ann.tsv")
    syn.write_fasta(fa)
    rows = []
    for i, (n, L) in enumerate(zip(syn.names, syn.lengths)):
        if L > 900:
            rows.append(("g%da" % i, n, 10, min(L - 20, 700), "+-"[i % 2]))
    _write_ann(ann, rows)
    ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs, core.default_params(min_coverage=3, calling_threshold=2))
    for i in range(sp.n_samples):
        ds.add_sample_records(syn.sample_records(i))
    ds.finalize(); ds.run()
    called, indiv = str(tmp_path / "called_SNPs"), str(tmp_path / "indiv_called")
    ds.write_calls(called, indiv, ann, fa)
    names = ["s%d.bam" % i for i in range(sp.n_samples)]
    open(tmp_path / "all_samples", "w").write("\n".join(names) + "\n")
    species = sorted({n.split(".")[0] for n in syn.names})
    soi = {s: [names[k] for k in range(sp.n_samples) if (k + j) % 4 != 0] for j, s in enumerate(species)}
    for ind, src in ((False, called), (True, indiv)):
        d_file, d_res = str(tmp_path / ("file%d" % ind)), str(tmp_path / ("res%d" % ind))
        os.makedirs(d_file); os.makedirs(d_res)
        filtering.filter_two_all(ctx, str(tmp_path / "all_samples"), [src], d_file, soi, 3.0, 0.4)
        n_kept, ms = ds.filter_resident([(s, [names.index(x) for x in soi[s]], soi[s]) for s in species], d_res, 3.0, 0.4, ind=ind, ann_path=ann, fasta_path=fa)
        got, want = sorted(os.listdir(d_res)), sorted(os.listdir(d_file))
        assert got == want and (ind or len(want) >= 3)
        for f in want:
            assert open(os.path.join(d_res, f)).read() == open(os.path.join(d_file, f)).read(), (ind, f)
        assert n_kept > 0 or ind
    ds.close(); ctx.close()


def test_two_rank_sharded_call_with_annotation(tmp_path):
    """The N-rank product path (contig mask per rank, kernels on the GPU, gather of site + annotation records, rank 0
    writes the files) rehearsed with two ranks sharing this GPU (tables over gloo): output equals the oracle's
    single-process text, and the shards partition the positions."""
    import subprocess
    import sys
    sp = core.synth_params(n_species=5, contig_len=4000, n_samples=6, mean_cov=11.0, snv_density=0.03, frac_absent=0.2, seed=55)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    fa, ann = str(tmp_path / "ref.fa"), str(tmp_path / "ann.tsv")
    syn.write_fasta(fa)
    _write_ann(ann, [("a1", syn.names[0], 5, 1800, "+"), ("a2", syn.names[0], 1500, 3600, "-"), ("c1", syn.names[2], 100, 3999, "-"),
                     ("e1", syn.names[4], 1, 900, "+"), ("e2", syn.names[4], 2000, 2000, "+")])
    env = dict(os.environ, MSNV_DIST_BACKEND="gloo")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shard_worker.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", worker, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa)
    assert open(tmp_path / "called_SNPs").read() == orac[0]
    assert open(tmp_path / "indiv_called").read() == orac[1]
    info = [list(map(int, open(tmp_path / ("rank%d.info" % k)).read().split())) for k in (0, 1)]
    assert info[0][0] + info[1][0] == sum(syn.lengths) and info[0][0] > 0 and info[1][0] > 0
    assert info[0][1] + info[1][1] == orac[3]


def _torchrun(n, script_args, env=None, timeout=900):
    import socket
    import subprocess
    import sys
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1", **(env or {})), capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("extra", [[], ["--n_splits", "3"], ["--threads", "2", "--db_ann", "ANN"]])
def test_launcher_metasnv_py_under_two_ranks_matches_the_oracle(tmp_path, extra):
    """The LAUNCHER (metaSNV.py, reference argv) under torchrun with two ranks sharing this GPU (tables over gloo): contigs
    are sharded over the ranks, every BAM is decoded by ONE rank and its records exchanged, rank 0 writes the project.
    called_SNPs / indiv_called (unsplit, per best_split_K, with --db_ann) and every cov/ file equal the oracle's bytes, and
    each rank inflated about half of the job's record bytes (metaSNV.py:179-221 is the pool this replaces)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    syn, samples = synth_case(n_species=6, contig_len=3500, n_samples=8, mean_cov=11.0, snv_density=0.03, frac_absent=0.15, seed=91)
    fa, paths, lst = _write_inputs(tmp_path, syn, samples)
    ann = None
    if "ANN" in extra:
        ann = str(tmp_path / "ann.tsv")
        _write_ann(ann, [("a1", syn.names[0], 5, 1800, "+"), ("a2", syn.names[0], 1500, 3300, "-"), ("c1", syn.names[2], 100, 3400, "-"),
                         ("e1", syn.names[4], 1, 900, "+"), ("f1", syn.names[5], 10, 3000, "-")])
        extra = [ann if x == "ANN" else x for x in extra]
    proj, met = str(tmp_path / "proj"), str(tmp_path / "metrics.jsonl")
    r = _torchrun(2, [os.path.join(root, "metaSNV.py"), proj, lst, fa] + extra, env=dict(MSNV_DIST_BACKEND="gloo", MSNV_METRICS=met))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    n_splits = 3 if "--n_splits" in extra else (2 if "--threads" in extra else 1)
    if n_splits == 1:
        orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, ann=ann, fasta=fa)
        assert open(os.path.join(proj, "snpCaller", "called_SNPs")).read() == orac[0]
        assert open(os.path.join(proj, "snpCaller", "indiv_called")).read() == orac[1]
        assert orac[0].count("\n") > 50
    else:
        splits = sorted(os.listdir(os.path.join(proj, "bestsplits")))
        assert splits == ["best_split_%d" % k for k in range(n_splits)]
        seen, total = [], 0
        for sp in splits:
            bed = [(syn.names.index(l.split()[0]), int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(proj, "bestsplits", sp))]
            seen += [b[0] for b in bed]
            o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed, ann=ann, fasta=fa)
            assert open(os.path.join(proj, "snpCaller", "called_SNPs." + sp)).read() == o[0], sp
            assert open(os.path.join(proj, "snpCaller", "indiv_called." + sp)).read() == o[1], sp
            total += o[0].count("\n")
        assert sorted(seen) == list(range(6)) and total > 50
    for i, p in enumerate(paths):
        want = orc.qacompute(syn.names, syn.lengths, samples[i])
        base = os.path.join(proj, "cov", os.path.basename(p) + ".cov")
        assert open(base).read() == want[0] and open(base + ".detail").read() == want[1]
        assert os.path.exists(base + ".summary")
    assert open(os.path.join(proj, "all_samples")).read() == open(lst).read()
    m = [json.loads(l) for l in open(met)]
    assert sorted(x["rank"] for x in m) == [0, 1] and all(x["world"] == 2 and x["contigs"] == 3 for x in m)
    per_rank = m[0]["inflated_record_bytes_per_rank"]
    total_bytes = sum(s.size for s in samples)
    assert sum(per_rank) == total_bytes and max(per_rank) <= 0.75 * total_bytes      # 8 BAMs of unequal size over 2 ranks
    assert sum(x["dataset"]["n_positions"] for x in m) == sum(syn.lengths)


def test_launcher_under_two_ranks_on_a_sparse_cohort(tmp_path):
    """The same launcher on a SPARSE cohort (30 samples, each species carried by a few: the configs[3] shape in small): the ranks'
    datasets run on whole-tile work items (the workgroup that piles a tile up applies the gates; kernels.hip: fused_tile_gate,
    msnv_gate_staged), with --n_splits 2 so that every split's own first line is dropped.  Same bytes as the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    syn, samples = synth_case(n_species=8, contig_len=4300, n_samples=36, mean_cov=5.0, sigma_cov=0.4, snv_density=0.02, frac_absent=0.85, lowercase_ref=1, seed=93)
    samples = [s for s in samples if s.size]                      # (a BAM without mapped reads is undefined in qaCompute: MSNV_EDOMAIN)
    assert len(samples) > 20
    fa, paths, lst = _write_inputs(tmp_path, syn, samples)
    proj = str(tmp_path / "proj")
    r = _torchrun(2, [os.path.join(root, "metaSNV.py"), proj, lst, fa, "--n_splits", "2", "--min_pos_cov", "2", "--min_pos_snvs", "2"], env=dict(MSNV_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    p = core.default_params(min_coverage=2, calling_threshold=2)
    total = 0
    for sp in sorted(os.listdir(os.path.join(proj, "bestsplits"))):
        bed = [(syn.names.index(l.split()[0]), int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(proj, "bestsplits", sp))]
        o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed, params=p)
        assert open(os.path.join(proj, "snpCaller", "called_SNPs." + sp)).read() == o[0], sp
        assert open(os.path.join(proj, "snpCaller", "indiv_called." + sp)).read() == o[1], sp
        total += o[0].count("\n") + o[1].count("\n")
    assert total > 30
    for i, pth in enumerate(paths):
        want = orc.qacompute(syn.names, syn.lengths, samples[i])
        base = os.path.join(proj, "cov", os.path.basename(pth) + ".cov")
        assert open(base).read() == want[0] and open(base + ".detail").read() == want[1]


def test_launcher_under_eight_ranks_writes_the_one_rank_project(tmp_path):
    """The N-rank path at the node's real width: metaSNV.py under torchrun with EIGHT ranks (sharing this GPU, tables over gloo) on a
    sparse cohort of 20 species -- species LPT over eight owners (createOptimumSplit.py:46-62), every BAM decoded by one rank, records
    exchanged, eight datasets, gather to rank 0 -- writes the same called_SNPs / indiv_called / cov files, byte for byte, as one rank."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    syn, samples = synth_case(n_species=20, contig_len=4100, n_samples=24, mean_cov=6.0, sigma_cov=0.4, snv_density=0.02, species_per_sample=5, lowercase_ref=1, seed=97)
    samples = [s for s in samples if s.size]
    fa, paths, lst = _write_inputs(tmp_path, syn, samples)
    out = {}
    for n in (1, 8):
        proj, met = str(tmp_path / ("proj%d" % n)), str(tmp_path / ("metrics%d.jsonl" % n))
        r = _torchrun(n, [os.path.join(root, "metaSNV.py"), proj, lst, fa, "--min_pos_cov", "2", "--min_pos_snvs", "2"], env=dict(MSNV_DIST_BACKEND="gloo", MSNV_METRICS=met), timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        files = {"called_SNPs": open(os.path.join(proj, "snpCaller", "called_SNPs")).read(), "indiv_called": open(os.path.join(proj, "snpCaller", "indiv_called")).read()}
        for pth in paths:
            base = os.path.join(proj, "cov", os.path.basename(pth) + ".cov")
            files[os.path.basename(base)] = open(base).read()
            files[os.path.basename(base) + ".detail"] = open(base + ".detail").read()
        out[n] = (files, [json.loads(l) for l in open(met)])
    assert out[1][0].keys() == out[8][0].keys()
    for k in out[1][0]:
        assert out[1][0][k] == out[8][0][k], k
    assert out[1][0]["called_SNPs"].count("\n") > 100
    m8 = out[8][1]
    assert sorted(x["rank"] for x in m8) == list(range(8)) and all(x["contigs"] >= 1 for x in m8)      # 20 species over 8 owners
    bases = [x["dataset"]["n_pileup_bases"] for x in m8]
    assert max(bases) <= 2.0 * sum(bases) / 8                      # LPT on length x first-round coverage


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` outside torchrun spawns the two rank processes itself (before any GPU call) and the
    rank-0 line reports n_gpus 2, both ranks' line counts and the slowest rank's roofline (gloo rehearsal: one GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--samples", "24", "--contig-len", "60000", "--no-cpu-baseline", "--no-annotation", "--no-overlap-extra", "--no-strong-extra"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(line["config"]["called_SNPs_lines_per_rank"]) == 2
    assert len(line["roofline"]["achieved_per_rank"]) == 2 and line["roofline"]["achieved"] == min(line["roofline"]["achieved_per_rank"])
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_distances_on_device_match_reference_outputs(tmp_path, golden_dir):
    """metaSNV_DistDiv.py --dist on the device: tests/golden/python_callers/distdiv holds the
    *.filtered.freq inputs and the .mann.dist / .allele.dist files the reference script (pandas) wrote for them."""
    import shutil
    from metasnv_amd import distdiv
    src = os.path.join(golden_dir, "python_callers", "distdiv", "proj")
    proj = str(tmp_path / "proj")
    shutil.copytree(src, proj)
    shutil.rmtree(os.path.join(proj, "distances"))
    distdiv.main(["--filt", os.path.join(proj, "filtered", "pop"), "--dist"])
    want = sorted(os.listdir(os.path.join(src, "distances")))
    assert sorted(os.listdir(os.path.join(proj, "distances"))) == want and len(want) == 8
    for f in want:
        assert open(os.path.join(proj, "distances", f)).read() == open(os.path.join(src, "distances", f)).read(), f


def test_distances_random_tables_against_pandas(tmp_path):
    """The same arithmetic evaluated live with pandas (metaSNV_DistDiv.py:105-124 restated in four lines): table lengths
    around numpy's pairwise-summation block boundaries, NaN-heavy columns, an all-NaN sample."""
    import random
    import numpy as np
    pd = pytest.importorskip("pandas")
    from metasnv_amd import _lib
    ctx = core.Context(0)
    rnd = random.Random(4)
    for n_pos, S in [(1, 3), (7, 4), (8, 4), (9, 5), (127, 6), (128, 6), (129, 6), (136, 3), (257, 5), (1000, 9), (2049, 4), (5003, 12)]:
        names = ["smp%d.bam" % i for i in range(S)]
        path = str(tmp_path / ("t%d.filtered.freq" % n_pos))
        with open(path, "w") as f:
            f.write("\t" + "\t".join(names) + "\n")
            for k in range(n_pos):
                vals = []
                for s in range(S):
                    c = rnd.choice([1, 3, 7, 40, 97, 1000, 29989, 200003])
                    vals.append("-1" if (rnd.random() < (0.6 if s == 1 else 0.1) or (s == 2 and S > 4)) else repr(rnd.randint(0, c) / c))
                f.write("c:-:%d:A>T:.\t%s\n" % (k + 1, "\t".join(vals)))
        mp, ap = path + ".mann", path + ".allele"
        _lib.check(_lib.lib.msnv_dist_file(ctx._h, path.encode(), mp.encode(), ap.encode(), 0.6, None, None, None))
        data = pd.read_table(path, index_col=0, na_values=['-1']).T
        dist = pd.DataFrame([[np.abs(data.iloc[i] - data.iloc[j]).mean() for i in range(len(data))] for j in range(len(data))], index=data.index, columns=data.index)
        dist.to_csv(mp + ".want", sep='\t')
        dist = pd.DataFrame([[(np.abs(data.iloc[i] - data.iloc[j]) > .6).mean() for i in range(len(data))] for j in range(len(data))], index=data.index, columns=data.index)
        dist.to_csv(ap + ".want", sep='\t')
        assert open(mp).read() == open(mp + ".want").read(), (n_pos, S)
        assert open(ap).read() == open(ap + ".want").read(), (n_pos, S)
    ctx.close()


def test_cli_project_layout_and_contents(tmp_path, capsys):
    from metasnv_amd import cli, tables
    syn, samples = synth_case(n_species=3, contig_len=4000, n_samples=4, mean_cov=12.0, snv_density=0.03, frac_absent=0.0, seed=31)
    fa, paths, lst = _write_inputs(tmp_path, syn, samples)
    # ---- one split (metaSNV.py ... --threads 1)
    proj = str(tmp_path / "out1")
    cli.main([proj, lst, fa])
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    assert open(os.path.join(proj, "snpCaller", "called_SNPs")).read() == orac[0]
    assert open(os.path.join(proj, "snpCaller", "indiv_called")).read() == orac[1]
    assert open(os.path.join(proj, "bed_header")).read() == "".join("%s\t1\t%d\n" % (n, l) for n, l in zip(syn.names, syn.lengths))
    assert open(os.path.join(proj, "all_samples")).read() == open(lst).read()
    for i, p in enumerate(paths):
        want = orc.qacompute(syn.names, syn.lengths, samples[i])
        base = os.path.join(proj, "cov", os.path.basename(p) + ".cov")
        assert open(base).read() == want[0] and open(base + ".detail").read() == want[1]
        assert os.path.exists(base + ".summary")
    assert open(os.path.join(proj, "out1.all_cov.tab")).read().startswith("\t" + "\t".join(os.path.basename(p) for p in sorted(paths)))
    for sub in ("filtered/pop", "filtered/ind", "distances", "bestsplits"):
        assert os.path.isdir(os.path.join(proj, sub))
    # ---- three splits (metaSNV.py ... --threads 3): one file pair per best_split_K, BED semantics per split
    proj3 = str(tmp_path / "out3")
    cli.main([proj3, lst, fa, "--threads", "3"])
    splits = sorted(os.listdir(os.path.join(proj3, "bestsplits")))
    assert splits == ["best_split_0", "best_split_1", "best_split_2"]
    total = 0
    for sp in splits:
        bed = []
        for line in open(os.path.join(proj3, "bestsplits", sp)):
            n, b, e = line.split()
            bed.append((syn.names.index(n), int(b), int(e)))
        o = run_oracle(syn.names, syn.lengths, syn.seqs, samples, bed=bed)
        got = open(os.path.join(proj3, "snpCaller", "called_SNPs." + sp)).read()
        assert got == o[0]
        assert open(os.path.join(proj3, "snpCaller", "indiv_called." + sp)).read() == o[1]
        total += got.count("\n")
    assert total > 0
    # ---- eight threads, one split: the launcher brings the HIP context up on a thread of its own while the host threads read and pack
    # (cli.py: lazy context; msnv_dataset_attach_ctx gives the dataset its device before finalize)
    proj8 = str(tmp_path / "out8")
    cli.main([proj8, lst, fa, "--threads", "8", "--n_splits", "8"])
    got8 = "".join(open(os.path.join(proj8, "snpCaller", f)).read() for f in sorted(os.listdir(os.path.join(proj8, "snpCaller"))) if f.startswith("called_SNPs"))
    assert got8.count("\n") == total                        # (the same three species, dealt to eight split files: five of them empty)
    for i, p in enumerate(paths):
        want = orc.qacompute(syn.names, syn.lengths, samples[i])
        base = os.path.join(proj8, "cov", os.path.basename(p) + ".cov")
        assert open(base).read() == want[0] and open(base + ".detail").read() == want[1]
    # an existing project directory is refused exactly like the reference (metaSNV.py:278-280)
    with pytest.raises(SystemExit):
        cli.main([proj, lst, fa])


def test_full_testdata_shape_bit_exact(tmp_path):
    """BASELINE configs[1] at FULL size (160 samples x 3 refGenomes x 300 kb, 1.49 G pileup bases): called_SNPs and
    indiv_called of the HIP path are byte-identical to the oracle's (the single-threaded oracle needs ~35 s for it)."""
    sp = core.synth_params(seed=1)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    ds.add_synth_samples(sp, 0, sp.n_samples, 0)           # the same generator as sample_records, packed inside the library
    info = ds.finalize()
    st = ds.run()
    pp, ip = str(tmp_path / "called_SNPs"), str(tmp_path / "indiv_called")
    ds.write_calls(pp, ip)
    ds.close(); ctx.close()
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    assert info["n_pileup_bases"] == orac[3] and info["n_samples"] == 160
    _assert_same((open(pp).read(), open(ip).read()), orac)
    assert st["n_called_pop"] == orac[0].count("\n") > 5000


def test_full_testdata_shape_with_gene_annotation_bit_exact(tmp_path):
    """BASELINE configs[4] at its own size: the 160-sample testdata shape with --db_ann -- the SURVEY 8d gene table (CDS of 300-3000 bp,
    ~85 % coding density, 5 % overlapping their predecessor, half on the '-' strand; bench.synth_annotation) -- called_SNPs and
    indiv_called with the gene column and the codon tags (call_vC.cpp:567-633) byte-identical to the oracle's."""
    import bench
    sp = core.synth_params(seed=1)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    fa, an = str(tmp_path / "ref.fa"), str(tmp_path / "ann.tsv")
    syn.write_fasta(fa)
    n_genes = bench.synth_annotation(syn, an)
    ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    ds.add_synth_samples(sp, 0, sp.n_samples, 0)
    ds.finalize()
    ds.run()
    pp, ip = str(tmp_path / "called_SNPs"), str(tmp_path / "indiv_called")
    ds.write_calls(pp, ip, an, fa)
    ds.close(); ctx.close()
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, ann=an, fasta=fa)
    pop, ind = open(pp).read(), open(ip).read()
    _assert_same((pop, ind), orac)
    assert n_genes > 300 and pop.count("\n") > 5000
    assert sum(1 for l in pop.split("\n") if l and l.split("\t")[1] != "-") > 3000        # most called sites lie in a gene
    assert "S[" in pop and "N[" in pop


def test_full_testdata_shape_properties():
    """BASELINE configs[1] at full size (160 samples x 3 x 300 kb, 1.5 G pileup bases): too big for the
    oracle to finish in seconds, so the HIP path is checked through size-independent properties."""
    sp = core.synth_params(seed=1)
    syn = core.Synth(sp)
    ctx = core.Context(0)

    def run(sample_ids):
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        for i in sample_ids:
            ds.add_synth_samples(sp, i, 1, 0)
        info = ds.finalize()
        st = ds.run()
        sites, samples = ds.results()
        st2 = ds.run()                                     # idempotence: a second pass over resident columns
        s2, m2 = ds.results()
        assert s2.tobytes() == sites.tobytes() and m2.tobytes() == samples.tobytes()
        ds.close()
        return info, st, sites, samples

    ids = list(range(sp.n_samples))
    info, st, sites, samples = run(ids)
    assert info["n_samples"] == 160 and info["n_positions"] == 900000 and info["n_pileup_bases"] > 1.2e9
    assert 3000 < st["n_called_pop"] < 20000
    # conservation: the per-sample columns add up to the totals the calling rule saw
    assert (samples["cov"].astype(np.int64).sum(axis=1) == sites["cov"]).all()
    for x in range(4):
        called = ((sites["pop_mask"] | sites["ind_mask"]) >> x) & 1 == 1
        assert (samples["n"][:, :, x].astype(np.int64).sum(axis=1)[called] == sites["n"][called, x]).all()
    # sites come out in (contig, position) order without duplicates
    key = sites["tid"].astype(np.int64) << 32 | sites["pos"]
    assert (np.diff(key) > 0).all()
    # per-sample independence: a run over a subset of the samples reproduces those samples' columns
    sub = ids[10:30]
    _, _, s_sub, m_sub = run(sub)
    pos_full = {(int(t), int(p)): i for i, (t, p) in enumerate(zip(sites["tid"], sites["pos"]))}
    hits = 0
    for j, (t, p) in enumerate(zip(s_sub["tid"], s_sub["pos"])):
        i = pos_full.get((int(t), int(p)))
        if i is None:
            continue
        hits += 1
        assert (m_sub["cov"][j] == samples["cov"][i, 10:30]).all()
        both = (int(s_sub["pop_mask"][j]) | int(s_sub["ind_mask"][j])) & (int(sites["pop_mask"][i]) | int(sites["ind_mask"][i]))
        for x in range(4):
            if (both >> x) & 1:
                assert (m_sub["n"][j, :, x] == samples["n"][i, 10:30, x]).all()
    assert hits > 500
    ctx.close()


@pytest.mark.parametrize("shape", ["config3", "config4shard"])
def test_config3_and_config4_shapes_against_the_oracle_at_reduced_size(shape, tmp_path):
    """The SURVEY.md section 8d generators of BASELINE configs[2] (species of 1-50 contigs, every sample carries a few random
    species) and of one GPU's shard of configs[3] (500-sample cohort shape: many samples, a species carried by a handful of
    them at ~5x -- per-tile slots, sparse tiles, merged groups at the contig ends) at a size the oracle finishes in seconds:
    called_SNPs / indiv_called and the coverage files byte for byte, from ONE resident dataset (fused coverage + calls)."""
    if shape == "config3":
        kw = dict(n_species=12, contig_len=24000, n_samples=24, mean_cov=10.0, sigma_cov=0.7, contigs_per_species_max=10, species_per_sample=3, snv_density=0.02, seed=2003)
    else:
        kw = dict(n_species=40, contig_len=9000, n_samples=60, mean_cov=5.0, sigma_cov=0.3, contigs_per_species_max=5, species_per_sample=3, frac_absent=0.5, snv_density=0.02, seed=2004)
    sp = core.synth_params(**kw)
    syn = core.Synth(sp)
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    assert len(syn.names) > sp.n_species and len({n.split(".")[0] for n in syn.names}) == sp.n_species
    ctx = core.Context(0)
    ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
    for s in samples:
        ds.add_sample_records(s)
    info = ds.finalize()
    st_p, st_c = ds.fused_run()
    pp, ip = str(tmp_path / "called"), str(tmp_path / "indiv")
    ds.write_calls(pp, ip)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    _assert_same((open(pp).read(), open(ip).read()), orac)
    assert info["n_pileup_bases"] == orac[3] and orac[0].count("\n") + orac[1].count("\n") > 30
    n_cov = 0
    for i, s in enumerate(samples):
        if s.size == 0:
            continue
        cp, dp = str(tmp_path / "c.cov"), str(tmp_path / "c.detail")
        ds.write_coverage(i, cp, dp)
        want = orc.qacompute(syn.names, syn.lengths, s)
        assert open(cp).read() == want[0] and open(dp).read() == want[1]
        n_cov += 1
    assert n_cov >= sp.n_samples // 2
    ds.close(); ctx.close()


def test_config4_shard_shape_properties_at_scale():
    """One GPU's shard of BASELINE configs[3] at 2 % of its species (30 species x ~2 Mbp = 62 M positions, 500 samples, 5x, every
    species carried by a handful of samples): too big for the oracle, checked through size-independent properties --
    conservation of the per-sample columns, (contig, position) order, idempotence of a second pass, and independence of a
    sample's columns from the other samples."""
    kw = dict(n_species=30, contig_len=2070000, n_samples=500, mean_cov=5.0, sigma_cov=0.3, contigs_per_species_max=20, species_per_sample=1, frac_absent=0.95, seed=77)
    sp = core.synth_params(**kw)
    syn = core.Synth(sp)
    ctx = core.Context(0)

    def run(sample_ids):
        ds = core.Dataset(ctx, syn.names, syn.lengths, syn.seqs)
        for i in sample_ids:
            ds.add_synth_samples(sp, i, 1, 0)
        info = ds.finalize()
        st = ds.run()
        sites, samples = ds.results()
        ds.run()
        s2, m2 = ds.results()
        assert s2.tobytes() == sites.tobytes() and m2.tobytes() == samples.tobytes()
        ds.close()
        return info, st, sites, samples

    ids = list(range(sp.n_samples))
    info, st, sites, samples = run(ids)
    assert info["n_positions"] > 5e7 and info["n_pileup_bases"] > 1e8 and len(sites) > 1000
    assert (samples["cov"].astype(np.int64).sum(axis=1) == sites["cov"]).all()
    for x in range(4):
        called = ((sites["pop_mask"] | sites["ind_mask"]) >> x) & 1 == 1
        assert (samples["n"][:, :, x].astype(np.int64).sum(axis=1)[called] == sites["n"][called, x]).all()
    key = sites["tid"].astype(np.int64) << 32 | sites["pos"]
    assert (np.diff(key) > 0).all()
    # the samples that carry anything, alone: their columns do not change
    carriers = [i for i in ids if samples["cov"][:, i].any()][:40]
    _, _, s_sub, m_sub = run(carriers)
    pos_full = {(int(t), int(p)): i for i, (t, p) in enumerate(zip(sites["tid"], sites["pos"]))}
    hits = 0
    for j, (t, p) in enumerate(zip(s_sub["tid"], s_sub["pos"])):
        i = pos_full.get((int(t), int(p)))
        if i is None:
            continue
        hits += 1
        assert (m_sub["cov"][j] == samples["cov"][i, carriers]).all()
    assert hits > 100
    ctx.close()


def test_randomised_parity_sweep(monkeypatch):
    """A slice of the randomised sweep (tests/fuzz_parity.py: generator and caller parameters, read lengths 20-400, coverages
    0.5-300x, both data layouts, BED splits, fused coverage, device annotation, batched / overlapped passes).  The full
    sweep (thousands of cases) found the descriptor-ring and wide-kernel bugs this file now has regression tests for."""
    import fuzz_parity
    monkeypatch.delenv("MSNV_LAYOUT", raising=False)
    try:
        assert fuzz_parity.sweep(70, 2024, verbose=False) == 0
    finally:
        os.environ.pop("MSNV_LAYOUT", None)


def test_randomised_parity_sweep_with_allele_planes(monkeypatch):
    """The same sweep with the allele bookkeeping of noisy reads forced on (MSNV_ALLELES=planes, pack.cpp): every (sample, tile) pair of
    an ordinary work item writes its mismatching A / C / G / T counts as four byte planes, the gate kernel sums the planes, the gather
    transposes them into the cells -- no total atomics, no events.  Other seeds than the sweep above."""
    import fuzz_parity
    monkeypatch.setenv("MSNV_ALLELES", "planes")
    monkeypatch.delenv("MSNV_LAYOUT", raising=False)
    try:
        assert fuzz_parity.sweep(70, 4711, verbose=False) == 0
    finally:
        os.environ.pop("MSNV_LAYOUT", None)


@pytest.mark.parametrize("error_rate,expect_planes", [(0.001, 0), (0.04, 1)])
def test_allele_bookkeeping_follows_the_sampled_mismatch_rate(error_rate, expect_planes):
    """finalize samples every 16th piece against the reference: clean reads keep the sparse events, a few per cent of mismatches switch
    the dataset to allele planes (reported in the dataset info).  Same bytes as the oracle either way, with many sites per tile, split
    samples (one deep sample), merged groups in other tiles and both gather forms."""
    syn, samples = synth_case(n_species=3, contig_len=5200, n_samples=24, mean_cov=10.0, sigma_cov=1.6, snv_density=0.03 if expect_planes else 0.004,
                              error_rate=error_rate, frac_absent=0.1, seed=8800 + expect_planes)
    p = core.default_params(min_coverage=3, calling_threshold=2)
    prod = run_product(syn.names, syn.lengths, syn.seqs, samples, params=p)
    orac = run_oracle(syn.names, syn.lengths, syn.seqs, samples, params=p)
    _assert_same(prod, orac)
    assert prod[2]["allele_planes"] == expect_planes
    assert (prod[2]["sampled_mismatch_ppm"] > 20000) == bool(expect_planes)
    assert prod[3]["n_events"] == 0 if expect_planes else prod[3]["n_events"] > 0


def test_filtering_random_tables_against_python_formulas(tmp_path):
    """FILTER II on random called_SNPs-like text (zero coverages, huge coverages -> exponent-form repr, several alleles per
    line, species not of interest, thresholds on both sides of every gate) against the reference's formulas
    (metaSNV_Filtering.py:183-231) evaluated in Python on the same text."""
    import random
    from metasnv_amd import filtering
    rnd = random.Random(12)
    for trial in range(6):
        S = rnd.choice([2, 5, 9])
        names = ["s%d.bam" % i for i in range(S)]
        species = ["spA", "spB", "spC"]
        proj = str(tmp_path / ("t%d" % trial) / "proj")
        os.makedirs(os.path.join(proj, "snpCaller"))
        open(os.path.join(proj, "all_samples"), "w").write("".join("/d/%s\n" % n for n in names))
        cov = "\t" + "\t".join(names) + "\nTaxId\t" + "\t".join(["Average_cov"] * S) + "\n"
        per = "\t" + "\t".join(names) + "\nTaxId\t" + "\t".join(["Percentage_1x"] * S) + "\n"
        for sp in species:
            cov += sp + "\t" + "\t".join("%f" % rnd.choice([0.0, 1.5, 6.0, 20.0]) for _ in range(S)) + "\n"
            per += sp + "\t" + "\t".join("%f" % rnd.choice([5.0, 45.0, 90.0]) for _ in range(S)) + "\n"
        open(os.path.join(proj, "proj.all_cov.tab"), "w").write(cov)
        open(os.path.join(proj, "proj.all_perc.tab"), "w").write(per)
        lines = ""
        for k in range(150):
            ctg = rnd.choice(["spA.c1", "spA.c2", "spB", "spC.x.y", "spD.z"])
            c = [rnd.choice([0, 0, 1, 3, 4, 5, 6, 9, 40, 1000, 200003]) for _ in range(S)]
            ents = []
            for alt in rnd.sample("ACGT", rnd.choice([1, 1, 2, 3])):
                n = [rnd.randint(0, x) if x else 0 for x in c]
                ents.append("%d|%s|%s|%s" % (sum(n), alt, rnd.choice([".", "S[GCT-GCC]", "N[TA-TC]"]), "|".join(map(str, n))))
            lines += "%s\t%s\t%d\t%s\t%s\t%s\n" % (ctg, rnd.choice(["-", "g%d" % k]), 5 + 2 * k, rnd.choice("ACGTn"), "|".join(map(str, c)), ",".join(ents))
        open(os.path.join(proj, "snpCaller", "called_SNPs"), "w").write(lines)
        open(os.path.join(proj, "snpCaller", "indiv_called"), "w").write("")
        b, d, m, cc, p = rnd.choice([10.0, 40.0]), rnd.choice([1.0, 5.0]), rnd.choice([1, 2]), rnd.choice([1.0, 5.0]), rnd.choice([0.2, 0.5, 0.9])
        filtering.main([proj, "-b", str(b), "-d", str(d), "-m", str(m), "-c", str(cc), "-p", str(p)])
        soi = filtering.relevant_taxa(os.path.join(proj, "proj.all_cov.tab"), os.path.join(proj, "proj.all_perc.tab"), b, d, m)["SoI"]
        for sp in species:
            path = os.path.join(proj, "filtered", "pop", sp + ".filtered.freq")
            if sp not in soi:
                assert not os.path.exists(path)
                continue
            idx = [names.index(n) for n in soi[sp]]
            want = ""
            for line in lines.splitlines():
                w = line.split()
                if w[0].split(".")[0] != sp:
                    continue
                c = list(map(int, w[4].split("|")))
                good = sum(1 for i in idx if not (c[i] < cc or c[i] == 0))
                if float(good) / len(idx) < p:
                    continue
                for snp in w[5].split(","):
                    x = snp.split("|")
                    n = list(map(float, x[3:]))
                    fr = [n[i] / c[i] if (c[i] >= cc and c[i] != 0) else -1 for i in idx]
                    want += ":".join(w[:4]) + ">" + x[1] + ":" + x[2] + "\t" + "\t".join(str(v) for v in fr) + "\n"
            if want:
                assert open(path).read() == "\t" + "\t".join(soi[sp]) + "\n" + want, (trial, sp)
            else:
                assert not os.path.exists(path)
