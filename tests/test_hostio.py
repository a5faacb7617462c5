"""Host I/O of libmsnv.so (BGZF/BAM on zlib) against the standard library."""
import gzip
import os

import numpy as np
import pytest

import bamtools as bt
from metasnv_amd import core


def _records():
    return bt.records(
        bt.make_record(0, 2, "5M", "GTACG", name="a"),
        bt.make_record(0, 4, "3M1I2M", "ACGTTA", flag=16, name="b"),
        bt.make_record(1, 0, "2S3M", "NNACG", qual=[1, 2, 3, 4, 5], name="c", mapq=7),
        bt.make_record(-1, -1, "*", "ACGT", flag=4, name="u"),
    )


def test_bam_write_is_readable_by_gzip_and_roundtrips(tmp_path):
    p = str(tmp_path / "x.bam")
    rec = _records()
    core.write_bam(p, ["c1", "c2"], [20, 30], rec)
    text, names, lengths, raw = bt.read_bam_py(p)          # independent parser
    assert names == ["c1", "c2"] and lengths == [20, 30]
    assert "@SQ\tSN:c1\tLN:20" in text
    assert raw == rec.tobytes()
    back = core.read_bam(p)                                 # library reader
    assert back["names"] == names and back["lengths"] == lengths
    assert back["records"].tobytes() == rec.tobytes()
    assert [r["name"] for r in bt.iter_records(back["records"])] == ["a", "b", "c", "u"]
    # BGZF EOF marker (SAMv1 4.1.2)
    assert open(p, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_reader_accepts_multi_block_files(tmp_path):
    p = str(tmp_path / "big.bam")
    rng = np.random.default_rng(0)
    recs = [bt.make_record(0, i * 3, "100M", "".join("ACGT"[k] for k in rng.integers(0, 4, 100)), name="r%d" % i) for i in range(3000)]
    rec = bt.records(*recs)
    assert rec.size > 3 * 65536
    core.write_bam(p, ["c1"], [100000], rec, level=6)
    assert core.read_bam(p)["records"].tobytes() == rec.tobytes()
    assert bt.read_bam_py(p)[3] == rec.tobytes()


def test_bed_header_matches_metaSNV_get_header(tmp_path):
    # metaSNV.py:81-94: SN\t1\tLN per @SQ line with exactly three fields; first header line skipped
    p = str(tmp_path / "x.bam")
    core.write_bam(p, ["spA.p.c1", "spB.q.c1"], [1000, 2000], _records()[:0],
                   header_text="@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:spA.p.c1\tLN:1000\n@SQ\tSN:spB.q.c1\tLN:2000\tM5:abc\n@PG\tID:x\n")
    out = str(tmp_path / "bed_header")
    core.write_bed_header(p, out)
    assert open(out).read() == "spA.p.c1\t1\t1000\n"      # the 4-field @SQ line is skipped exactly as the reference does


def test_malformed_inputs_fail_loudly(tmp_path):
    p = str(tmp_path / "bad.bam")
    with open(p, "wb") as f:
        f.write(b"not a bam")
    with pytest.raises(core._lib.MsnvError) as e:
        core.read_bam(p)
    assert e.value.code == core._lib.EFORMAT
    with pytest.raises(core._lib.MsnvError) as e:
        core.read_bam(str(tmp_path / "missing.bam"))
    assert e.value.code == core._lib.EIO


def test_synth_is_deterministic_and_sorted():
    p = core.synth_params(n_species=2, contig_len=3000, n_samples=2, mean_cov=5.0, seed=7)
    a, b = core.Synth(p), core.Synth(p)
    assert a.seqs == b.seqs and a.names == ["refGenome1clus", "refGenome2clus"]
    r0, r0b, r1 = a.sample_records(0), b.sample_records(0), a.sample_records(1)
    assert r0.tobytes() == r0b.tobytes() and r0.tobytes() != r1.tobytes()
    keys = [(r["tid"], r["pos"]) for r in bt.iter_records(r0)]
    assert keys == sorted(keys) and len(keys) > 50
    for r in bt.iter_records(r0):
        assert sum(n for n, op in r["cigar"] if op in (0, 1, 4, 7, 8)) == len(r["seq"])


def test_bgzf_crc_of_every_block_is_checked(tmp_path):
    """htslib verifies the CRC-32 of every BGZF block it inflates (what both reference tools read their BAMs through); so does the
    host decoder: a block whose trailer CRC does not match its inflated bytes fails the read, MSNV_INFLATE_CHECK=0 opts out."""
    import os
    import struct
    from bamtools import make_record, records
    from metasnv_amd._lib import MsnvError
    p = str(tmp_path / "a.bam")
    core.write_bam(p, ["c1"], [5000], records(*[make_record(0, 10 * i, "50M", "ACGT" * 12 + "AC") for i in range(400)]))
    assert core.read_bam(p)["records"].size > 0
    raw = bytearray(open(p, "rb").read())
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1           # first block: header ... payload, CRC32, ISIZE
    raw[bsize - 8] ^= 0x5a                                      # a bit of the stored CRC
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    try:
        core.read_bam(bad)
        raise AssertionError("a CRC mismatch must fail the read")
    except MsnvError as e:
        assert e.code == 3 and "CRC" in str(e)
    os.environ["MSNV_INFLATE_CHECK"] = "0"
    try:
        assert core.read_bam(bad)["records"].size > 0
    finally:
        del os.environ["MSNV_INFLATE_CHECK"]


def test_header_only_read_inflates_the_leading_blocks_only(tmp_path):
    """msnv_bam_read_header (what `samtools view -H` reads, metaSNV.py:88): names, lengths and text of msnv_bam_read without the records --
    also when the header spans several BGZF blocks, and when a LATER block of the file is corrupt (it is never looked at)."""
    names = ["contig_%05d_with_a_long_name_to_fill_blocks" % i for i in range(4000)]      # ~200 KB of contig table: four blocks
    lengths = [1000 + i for i in range(4000)]
    rec = bt.records(*[bt.make_record(7, 10 + i, "50M", "A" * 50, name="r%d" % i) for i in range(3000)])
    p = str(tmp_path / "big_header.bam")
    core.write_bam(p, names, lengths, rec)
    full, hdr = core.read_bam(p), core.read_bam(p, records=False)
    assert hdr["names"] == names and hdr["lengths"] == lengths and hdr["header_text"] == full["header_text"] and "records" not in hdr
    raw = bytearray(open(p, "rb").read())
    off, n = 0, 0
    while off < len(raw) - 28:                               # the last data block: flip payload bits
        last = off; off += (raw[off + 16] | raw[off + 17] << 8) + 1; n += 1
    assert n > 5
    raw[last + 30] ^= 0xff; raw[last + 31] ^= 0xff
    bad = str(tmp_path / "bad_tail.bam"); open(bad, "wb").write(raw)
    assert core.read_bam(bad, records=False)["names"] == names
    with pytest.raises(core._lib.MsnvError):
        core.read_bam(bad)


def test_nothing_is_added_behind_staged_streams(tmp_path):
    """msnv_dataset_stage_sample_bams reads now and packs in finalize -- LAST: every add_sample_* entry point refuses a dataset that holds
    staged streams (the sample order, and with it every per-sample output column, would come out wrong with no error)."""
    p = str(tmp_path / "x.bam")
    rec = _records()
    core.write_bam(p, ["c1", "c2"], [20, 30], rec)
    ds = core.Dataset(None, ["c1", "c2"], [20, 30], ["A" * 20, "C" * 30])           # no context yet (the runtime is still coming up)
    ds.stage_sample_bams([p])
    sp = core.synth_params(n_species=2, contig_len=20, n_samples=1, mean_cov=1.0, read_len=10, seed=1)
    for call in (lambda: ds.add_sample_records(rec), lambda: ds.add_sample_bams([p]), lambda: ds.add_synth_samples(sp, 0, 1, 1)):
        with pytest.raises(core._lib.MsnvError) as e:
            call()
        assert e.value.code == core._lib.EINVAL and "staged" in str(e.value)
    ds.stage_sample_bams([p])                                                      # (more staging is fine)
    ds.close()
