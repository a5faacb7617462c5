"""The product's own DEFLATE decoder (metasnv_amd/csrc/inflate.cpp) on the CPU:
  * built with -fsanitize=address,undefined together with tests/native/inflate_harness.cpp and run against zlib on streams of every
    block type (stored, fixed, dynamic; levels 0-9; five strategies; sizes 0 .. 64 KiB) and on corrupted / truncated streams,
    which may be refused or decoded to garbage but must never touch memory outside the buffers;
  * through the library: BAM files written at several compression levels come back byte-identical, with and without the
    decoder (MSNV_INFLATE=zlib), and no block needed the zlib fallback."""
import os
import subprocess
import sys

import numpy as np

from metasnv_amd import core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoder_against_zlib_under_sanitizers(tmp_path):
    exe = str(tmp_path / "inflate_harness")
    src = [os.path.join(ROOT, "tests", "native", "inflate_harness.cpp"), os.path.join(ROOT, "metasnv_amd", "csrc", "inflate.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"] + src + ["-lz", "-o", exe])
    r = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.startswith("ok 1500 streams") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_bam_files_of_every_compression_level_round_trip(tmp_path):
    sp = core.synth_params(n_species=2, contig_len=30000, n_samples=1, mean_cov=12.0, frac_paired=0.3, seed=9)
    syn = core.Synth(sp)
    rec = syn.sample_records(0)
    code = ("import sys; sys.path.insert(0, %r); from metasnv_amd import core, _lib; import hashlib, ctypes as C\n"
            "d = core.read_bam(sys.argv[1]); n = C.c_uint64(); _lib.lib.msnv_host_stats(C.byref(n))\n"
            "print(hashlib.md5(d['records'].tobytes()).hexdigest(), n.value)" % ROOT)
    import hashlib
    want = hashlib.md5(rec.tobytes()).hexdigest()
    for level in (0, 1, 6, 9):
        p = str(tmp_path / ("l%d.bam" % level))
        core.write_bam(p, syn.names, syn.lengths, rec, level=level)
        for env in ({}, {"MSNV_INFLATE": "zlib"}):
            r = subprocess.run([sys.executable, "-c", code, p], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            got, fallbacks = r.stdout.split()
            assert got == want and fallbacks == "0", (level, env, r.stdout)


def test_corrupt_bgzf_files_are_errors_not_crashes(tmp_path):
    sp = core.synth_params(n_species=1, contig_len=20000, n_samples=1, mean_cov=5.0, seed=3)
    syn = core.Synth(sp)
    good = str(tmp_path / "g.bam")
    core.write_bam(good, syn.names, syn.lengths, syn.sample_records(0))
    data = bytearray(open(good, "rb").read())
    rng = np.random.default_rng(5)
    outcomes = set()
    for k in range(40):
        bad = bytearray(data)
        if k % 4 == 0:
            bad = bad[:int(rng.integers(1, len(bad)))]                  # truncated
        elif k % 4 == 1:
            bad[10] = 0xff; bad[11] = 0xff                              # extra-field length beyond the file
        else:
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        p = str(tmp_path / "b.bam")
        open(p, "wb").write(bytes(bad))
        try:
            core.read_bam(p)
            outcomes.add("read")
        except core._lib.MsnvError as e:
            assert e.code in (core._lib.EFORMAT, core._lib.EIO, core._lib.ENOMEM)
            outcomes.add("refused")
    assert "refused" in outcomes
