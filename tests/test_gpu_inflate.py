"""BGZF blocks inflated on the device (csrc/inflate_k.hip) against the host decoder: BAM files of the
synthetic workload written at several compression levels (stored, fixed and dynamic Huffman blocks), an empty file, and blocks the
device must hand back to the host (corrupted payloads: the host decoder then reports the error)."""
import os

import numpy as np
import pytest

from metasnv_amd import core, _lib

pytestmark = pytest.mark.gpu


def _bam(tmp_path, level, seed, **kw):
    sp = core.synth_params(n_species=2, contig_len=40000, n_samples=1, mean_cov=14.0, frac_paired=0.3, seed=seed, **kw)
    syn = core.Synth(sp)
    rec = syn.sample_records(0)
    p = str(tmp_path / ("l%d_%d.bam" % (level, seed)))
    core.write_bam(p, syn.names, syn.lengths, rec, level=level)
    return p


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_device_inflate_equals_host_inflate(level, tmp_path, monkeypatch):
    monkeypatch.setenv("MSNV_INFLATE_CHECK", "1")          # every block against the CRC-32 of its trailer (the default since round 3)
    ctx = core.Context(0)
    for seed in (1, 2):
        p = _bam(tmp_path, level, seed)
        host, _ = core.bgzf_inflate(p)
        dev, cnt = core.bgzf_inflate(p, ctx)
        assert host.size > 100000 and np.array_equal(host, dev)
        assert cnt["blocks"] > 5 and cnt["host_blocks"] == 0 and cnt["bytes"] == host.size
    ctx.close()


def test_blocks_the_device_refuses_go_to_the_host_decoder(tmp_path):
    ctx = core.Context(0)
    p = _bam(tmp_path, 6, 3)
    raw = bytearray(open(p, "rb").read())
    good, _ = core.bgzf_inflate(p)
    # flip bits inside the payload of the third block: the device refuses it, the host decoder refuses it too -> MSNV_EFORMAT
    off = 0
    for _ in range(2):
        off += (raw[off + 16] | raw[off + 17] << 8) + 1
    for k in (40, 41, 90):
        raw[off + 18 + k] ^= 0x5a
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(raw)
    for c in (None, ctx):
        with pytest.raises(_lib.MsnvError) as e:
            core.bgzf_inflate(bad, c)
        assert e.value.code == _lib.EFORMAT
    ctx.close()


@pytest.mark.parametrize("mode", ["device", "host", "device-without-staging"])
def test_file_entry_point_with_either_inflate(mode, tmp_path, monkeypatch):
    """msnv_dataset_add_sample_bams with the blocks inflated on the device (picked when it is the faster way for the call) and on the
    host: the same dataset -- sizes, called positions, bytes of called_SNPs.  "device-without-staging": the staging allocation fails
    (MSNV_ENOMEM: a multi-GB BAM on a full device) and the batch is inflated by the host decoder instead of failing the call."""
    monkeypatch.setenv("MSNV_INFLATE", mode.split("-")[0])
    if mode == "device-without-staging":
        monkeypatch.setenv("MSNV_TEST_NO_STAGING", "1")
    sp = core.synth_params(n_species=2, contig_len=30000, n_samples=6, mean_cov=12.0, frac_paired=0.4, snv_density=0.02, seed=11)
    syn = core.Synth(sp)
    fa = str(tmp_path / "ref.fa"); syn.write_fasta(fa)
    paths = []
    for i in range(sp.n_samples):
        p = str(tmp_path / ("s%d.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, syn.sample_records(i), level=[1, 6, 0, 9, 4, 2][i]); paths.append(p)
    ctx = core.Context(0)
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    ds.add_sample_bams(paths, 3)
    info = ds.finalize(); st = ds.run()
    ds.write_calls(str(tmp_path / "c"), str(tmp_path / "i"), None, None)
    got = open(tmp_path / "c").read()
    ds.close(); ctx.close()
    import orc
    from parity import run_oracle
    samples = [syn.sample_records(i) for i in range(sp.n_samples)]
    want = run_oracle(syn.names, syn.lengths, syn.seqs, samples)
    assert got == want[0] and got.count("\n") > 10 and info["n_pileup_bases"] == want[3]



def _bgzf(blocks_payload):
    """A BGZF file from (raw deflate payload, uncompressed bytes) pairs, plus the EOF block."""
    import struct, zlib
    out = bytearray()
    for comp, data in blocks_payload:
        bsize = 18 + len(comp) + 8
        out += bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def test_device_inflate_on_crafted_streams_of_every_kind(tmp_path):
    """Raw DEFLATE streams of every block type and strategy zlib can write (stored, fixed, dynamic; RLE, Huffman-only, filtered;
    levels 0-9; distance-1 runs, long matches, incompressible bytes, sizes 0 .. 65280) wrapped as BGZF: the device returns the host
    decoder's bytes.  Corrupted copies: whatever the host decoder says (error, or some bytes) the device path says too."""
    import random
    import zlib
    rnd = random.Random(5)

    def data_of(kind, n):
        if kind == 0: return bytes(rnd.getrandbits(8) for _ in range(n))
        if kind == 1: return bytes(rnd.choice(b"ACGT") for _ in range(n))
        if kind == 2:
            v = bytearray(rnd.getrandbits(8) for _ in range(min(n, 300)))
            while len(v) < n: v.append(v[len(v) - 1 - rnd.randrange(min(len(v), 300))])
            return bytes(v[:n])
        if kind == 3: return b"x" * n
        return bytes(30 + rnd.randrange(11) for _ in range(n))

    ctx = core.Context(0)
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]
    blocks = []
    for r in range(160):
        n = r if r < 6 else rnd.choice([17, 300, 4000, 30000, 65280])
        d = data_of(r % 5, n)
        c = zlib.compressobj(rnd.randrange(10), zlib.DEFLATED, -15, 8, strategies[(r // 5) % 5])
        blocks.append((c.compress(d) + c.flush(), d))
    # streams of SEVERAL deflate blocks: Z_SYNC_FLUSH (an empty stored block behind every piece), Z_FULL_FLUSH, stored data cut into
    # non-final stored blocks of odd sizes -- the stored-block path moves the input position back and forth across the decoder's window
    for r in range(24):
        n = rnd.choice([40, 700, 9000, 65280])
        d = data_of(r % 5, n)
        c = zlib.compressobj([0, 1, 6, 9][r % 4], zlib.DEFLATED, -15, 8, strategies[r % 5])
        out, o = b"", 0
        while o < n:
            step = rnd.choice([1, 3, 64, 255, 256, 257, 1000, 5000])
            out += c.compress(d[o:o + step]) + c.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
            o += step
            if len(out) > 52000:                                   # (a BGZF block payload must stay below 64 KiB)
                break
        d = d[:o]
        blocks.append((out + c.flush(), d))
    p = str(tmp_path / "crafted.gz")
    open(p, "wb").write(_bgzf(blocks))
    want = b"".join(d for _, d in blocks)
    host, _ = core.bgzf_inflate(p)
    dev, cnt = core.bgzf_inflate(p, ctx)
    assert host.tobytes() == want and dev.tobytes() == want and cnt["host_blocks"] == 0
    # corrupted payloads, one block per file
    n_err = n_same = 0
    for trial in range(60):
        comp, d = blocks[6 + rnd.randrange(len(blocks) - 6)]
        if len(comp) < 4:
            continue
        bad = bytearray(comp)
        for _ in range(rnd.randrange(1, 4)):
            bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
        q = str(tmp_path / ("bad%d.gz" % trial))
        open(q, "wb").write(_bgzf([(bytes(bad), d)]))
        res = []
        for c in (None, ctx):
            try:
                res.append(core.bgzf_inflate(q, c)[0].tobytes())
            except _lib.MsnvError as e:
                assert e.code == _lib.EFORMAT
                res.append(None)
        assert res[0] == res[1], trial
        n_err += res[0] is None
        n_same += res[0] is not None
    assert n_err > 10
    ctx.close()


def test_resident_device_inflate_checks_every_block_in_hbm(tmp_path, monkeypatch):
    """msnv_dataset_add_sample_bams with the device pack: the inflated bytes stay in HBM, every block's CRC-32 is checked there
    (inflate_k.hip: msnv_crc_blocks) -- no host inflate at all for good files; a wrong trailer CRC is MSNV_EFORMAT (as the host decoder and
    htslib have it), a corrupted payload too; MSNV_INFLATE_CHECK=0 lets the wrong trailer through."""
    monkeypatch.setenv("MSNV_INFLATE", "device"); monkeypatch.setenv("MSNV_PACK", "device")
    sp = core.synth_params(n_species=2, contig_len=60000, n_samples=4, mean_cov=16.0, frac_paired=0.3, snv_density=0.02, seed=13)
    syn = core.Synth(sp)
    fa = str(tmp_path / "ref.fa"); syn.write_fasta(fa)
    paths = []
    for i in range(sp.n_samples):
        p = str(tmp_path / ("s%d.bam" % i))
        core.write_bam(p, syn.names, syn.lengths, syn.sample_records(i), level=[6, 1, 0, 9][i]); paths.append(p)
    ctx = core.Context(0)
    t0 = core.host_timers()
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    ds.add_sample_bams(paths, 3)
    info = ds.finalize(); ds.run()
    ds.write_calls(str(tmp_path / "c"), str(tmp_path / "i"), None, None)
    t1 = core.host_timers()
    assert t1["inflate_host_s"] - t0["inflate_host_s"] == 0.0 and t1["inflate_device_wall_s"] > t0["inflate_device_wall_s"]
    assert ds.pack_stats()["records"] > 1000 and ds.pack_stats()["upload_wall_s"] < 0.5
    ds.close()
    from parity import run_oracle
    want = run_oracle(syn.names, syn.lengths, syn.seqs, [syn.sample_records(i) for i in range(sp.n_samples)])
    assert open(tmp_path / "c").read() == want[0] and info["n_pileup_bases"] == want[3]
    # several batches (the next one is read while the device works on the current one): the same dataset
    monkeypatch.setenv("MSNV_INFLATE_BATCH_MB", "1")
    assert sum(os.path.getsize(q) for q in paths) > (2 << 20)
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    ds.add_sample_bams(paths, 3)
    info2 = ds.finalize(); ds.run()
    ds.write_calls(str(tmp_path / "c2"), str(tmp_path / "i2"), None, None)
    assert open(tmp_path / "c2").read() == want[0] and info2["n_pileup_bases"] == want[3]
    ds.close()
    monkeypatch.delenv("MSNV_INFLATE_BATCH_MB")
    # block 3 of the second file: wrong CRC in the trailer / flipped payload bits
    raw = bytearray(open(paths[1], "rb").read())
    off = 0
    for _ in range(2):
        off += (raw[off + 16] | raw[off + 17] << 8) + 1
    bsize = (raw[off + 16] | raw[off + 17] << 8) + 1
    for kind in ("crc", "payload"):
        bad = bytearray(raw)
        if kind == "crc":
            bad[off + bsize - 8] ^= 0x01
        else:
            for k in (40, 41, 90):
                bad[off + 18 + k] ^= 0x5a
        badp = str(tmp_path / ("bad_%s.bam" % kind)); open(badp, "wb").write(bad)
        ds = core.Dataset.from_files(ctx, paths[0], fa)
        with pytest.raises(_lib.MsnvError) as e:
            ds.add_sample_bams([paths[0], badp, paths[2]], 3)
        assert e.value.code == _lib.EFORMAT
        ds.close()
    monkeypatch.setenv("MSNV_INFLATE_CHECK", "0")
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    ds.add_sample_bams([paths[0], str(tmp_path / "bad_crc.bam"), paths[2]], 3)
    assert ds.finalize()["n_pileup_bases"] > 0
    ds.close(); ctx.close()


def test_bam_files_inflated_and_dealt_on_the_device(tmp_path, monkeypatch):
    """msnv_dataset_deal_bams_device: the N-rank feed's decode + deal step with nothing of the inflated bytes on the host -- against the host route
    (msnv_bam_read + msnv_records_partition) file by file: part bytes, sizes, statistics; a BAM of another reference is refused like everywhere."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    sp = core.synth_params(n_species=4, contig_len=20000, n_samples=5, mean_cov=10.0, frac_paired=0.3, frac_absent=0.2, seed=17)
    syn = core.Synth(sp)
    fa = str(tmp_path / "ref.fa"); syn.write_fasta(fa)
    paths = []
    for i in range(sp.n_samples):
        p = str(tmp_path / ("s%d.bam" % i)); core.write_bam(p, syn.names, syn.lengths, syn.sample_records(i), level=[6, 0, 9, 1, 4][i]); paths.append(p)
    ctx = core.Context(0)
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    nc, n_parts, gap = len(syn.names), 3, 8 * len(paths)
    owner = np.array([c % n_parts if c != 2 else -1 for c in range(nc)], dtype=np.int32)
    want = [core.partition_records(core.read_bam(p)["records"], owner, n_parts) for p in paths]
    cap = sum(core.read_bam(p)["records"].size for p in paths) + n_parts * gap
    d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(cap + 64)) == 0
    t0 = core.host_timers()
    pb, stats, rbytes = ds.deal_bams_device(paths, owner, n_parts, d.value, cap, gap=gap, host_threads=3)
    t1 = core.host_timers()
    assert t1["inflate_host_s"] == t0["inflate_host_s"]                       # no block went through the host decoder
    got = np.zeros(cap, np.uint8); assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), d, C.c_size_t(cap), 2) == 0
    assert [int(x) for x in rbytes] == [core.read_bam(p)["records"].size for p in paths]
    assert np.array_equal(stats, np.stack([w[1] for w in want])) and np.array_equal(pb, np.array([[q.size for q in w[0]] for w in want], dtype=np.int64))
    o = 0
    for k in range(n_parts):
        o += gap
        for i in range(len(paths)):
            w = want[i][0][k]
            assert got[o:o + w.size].tobytes() == w.tobytes(), (k, i)
            o += w.size
    # a file whose header names other contigs
    other = str(tmp_path / "other.bam"); core.write_bam(other, ["x1", "x2"], [100, 200], np.zeros(0, np.uint8))
    with pytest.raises(_lib.MsnvError) as e:
        ds.deal_bams_device([paths[0], other], owner, n_parts, d.value, cap, gap=gap)
    assert e.value.code == _lib.EFORMAT
    # more than one batch of the device inflate: the caller is told to take the host route
    monkeypatch.setenv("MSNV_INFLATE_BATCH_MB", "1")
    if sum(os.path.getsize(p) for p in paths) > (1 << 20):
        with pytest.raises(_lib.MsnvError) as e:
            ds.deal_bams_device(paths, owner, n_parts, d.value, cap, gap=gap)
        assert e.value.code == _lib.EDOMAIN
    hip.hipFree(d); ds.close(); ctx.close()


def test_bam_files_written_by_another_deflate_writer(tmp_path, monkeypatch):
    """Whole BAM files whose BGZF members come from Python's zlib with every strategy / level / flush mode and odd member sizes
    (tests/bamtools.py: write_bam_py) -- not from the library's own writer -- through the device inflate + device pack, against the oracle
    on the plain records; the host inflate reads the same files to the same calls."""
    import random
    import bamtools as bt
    from parity import run_oracle
    sp = core.synth_params(n_species=2, contig_len=30000, n_samples=5, mean_cov=12.0, frac_paired=0.3, snv_density=0.02, seed=77)
    syn = core.Synth(sp)
    fa = str(tmp_path / "ref.fa"); syn.write_fasta(fa)
    rnd = random.Random(11)
    paths, recs = [], []
    for i in range(sp.n_samples):
        p = str(tmp_path / ("p%d.bam" % i))
        r = syn.sample_records(i); recs.append(r)
        bt.write_bam_py(p, syn.names, syn.lengths, r.tobytes(), rnd); paths.append(p)
    want = run_oracle(syn.names, syn.lengths, syn.seqs, recs)
    ctx = core.Context(0)
    for mode in ("device", "host"):
        monkeypatch.setenv("MSNV_INFLATE", mode)
        ds = core.Dataset.from_files(ctx, paths[0], fa)
        ds.add_sample_bams(paths, 3)
        info = ds.finalize(); ds.run()
        ds.write_calls(str(tmp_path / ("c_" + mode)), str(tmp_path / ("i_" + mode)), None, None)
        assert open(tmp_path / ("c_" + mode)).read() == want[0], mode
        assert open(tmp_path / ("i_" + mode)).read() == want[1], mode
        assert info["n_pileup_bases"] == want[3]
        ds.close()
    ctx.close()


def test_resident_inflate_that_fails_falls_back_to_the_host_decoder(tmp_path, monkeypatch):
    """The resident device inflate refusing a batch (no memory for its buffers, a HIP error) must hand the batch to the host decoder --
    round 4 wrote the host decoder's output through a NULL pointer there (MSNV_TEST_RESIDENT_FAIL makes the call fail)."""
    from parity import run_oracle
    monkeypatch.setenv("MSNV_INFLATE", "device"); monkeypatch.setenv("MSNV_PACK", "device")
    sp = core.synth_params(n_species=1, contig_len=20000, n_samples=3, mean_cov=10.0, snv_density=0.02, seed=5)
    syn = core.Synth(sp)
    fa = str(tmp_path / "ref.fa"); syn.write_fasta(fa)
    paths = []
    for i in range(sp.n_samples):
        p = str(tmp_path / ("s%d.bam" % i)); core.write_bam(p, syn.names, syn.lengths, syn.sample_records(i), level=6); paths.append(p)
    want = run_oracle(syn.names, syn.lengths, syn.seqs, [syn.sample_records(i) for i in range(sp.n_samples)])
    ctx = core.Context(0)
    monkeypatch.setenv("MSNV_TEST_RESIDENT_FAIL", "1")
    t0 = core.host_timers()
    ds = core.Dataset.from_files(ctx, paths[0], fa)
    ds.add_sample_bams(paths, 2)
    ds.finalize(); ds.run()
    ds.write_calls(str(tmp_path / "c"), str(tmp_path / "i"), None, None)
    assert core.host_timers()["inflate_host_s"] > t0["inflate_host_s"]
    assert open(tmp_path / "c").read() == want[0]
    ds.close(); ctx.close()
