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
