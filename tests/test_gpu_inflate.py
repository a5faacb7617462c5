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
def test_device_inflate_equals_host_inflate(level, tmp_path):
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


@pytest.mark.parametrize("mode", ["device", "host"])
def test_file_entry_point_with_either_inflate(mode, tmp_path, monkeypatch):
    """msnv_dataset_add_sample_bams with the blocks inflated on the device (the default from 64 MB of BAM on) and on the host: the same
    dataset -- sizes, called positions, bytes of called_SNPs."""
    monkeypatch.setenv("MSNV_INFLATE", mode)
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

