"""The library's own CRC-32 (csrc/crc32.cpp: carry-less-multiply folding, or sliced tables) against zlib's: BGZF files written in
Python -- payloads of every length 0 .. 299, 300 random lengths up to 60 000 and the lengths around the folding's block sizes, trailers
from zlib.crc32 -- are read with every block checked (a wrong CRC on either side is MSNV_EFORMAT); one flipped trailer bit is refused;
and the trailers the library writes are zlib's.  Both forms, each in a process of its own (the form is picked once per process)."""
import os
import subprocess
import sys

import pytest

WORKER = r'''
import os, sys, struct, zlib, random, tempfile
sys.path.insert(0, sys.argv[1])
from metasnv_amd import core, _lib
os.environ["MSNV_INFLATE_CHECK"] = "1"
rnd = random.Random(3)
def bgzf(blocks, corrupt=None):
    out = bytearray()
    for i, d in enumerate(blocks):
        c = zlib.compressobj(rnd.choice([0, 1, 6]), zlib.DEFLATED, -15)
        comp = c.compress(d) + c.flush()
        crc = zlib.crc32(d) & 0xffffffff
        if corrupt == i: crc ^= 1 << rnd.randrange(32)
        out += bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", 18 + len(comp) + 8 - 1) + comp + struct.pack("<II", crc, len(d))
    return bytes(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
lens = list(range(0, 300)) + [rnd.randrange(300, 60000) for _ in range(300)] + [64, 65, 79, 80, 127, 128, 4096, 65280, 65279, 16000, 16015]
blocks = [os.urandom(n) for n in lens]
with tempfile.TemporaryDirectory() as td:
    p = td + "/x.gz"
    open(p, "wb").write(bgzf(blocks))
    out, _ = core.bgzf_inflate(p)
    assert out.tobytes() == b"".join(blocks)
    for bad in (5, 100, 350, len(blocks) - 1):
        open(p, "wb").write(bgzf(blocks, corrupt=bad))
        try:
            core.bgzf_inflate(p)
            raise SystemExit("a wrong CRC was accepted in block %d" % bad)
        except _lib.MsnvError as e:
            assert e.code == _lib.EFORMAT
    syn = core.Synth(core.synth_params(n_species=1, contig_len=20000, n_samples=1, mean_cov=20.0, seed=2))
    q = td + "/w.bam"
    core.write_bam(q, syn.names, syn.lengths, syn.sample_records(0), level=1)
    raw, off, nb = open(q, "rb").read(), 0, 0
    while off < len(raw):
        bs = struct.unpack_from("<H", raw, off + 16)[0] + 1
        crc, isz = struct.unpack_from("<II", raw, off + bs - 8)
        data = zlib.decompress(raw[off + 18: off + bs - 8], -15)
        assert len(data) == isz and (zlib.crc32(data) & 0xffffffff) == crc
        off += bs; nb += 1
    assert nb > 5
print("ok")
'''


@pytest.mark.parametrize("form", ["auto", "table"])
def test_crc32_equals_zlib(form):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSNV_CRC=form)
    r = subprocess.run([sys.executable, "-c", WORKER, root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1000:] + r.stderr[-2000:]
