"""ctypes wrapper of oracle/liborc.so -- the CPU checker.  Imported by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() only; never by the product package."""
import ctypes as C
import os
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_so = os.environ.get("ORC_LIBRARY") or os.path.join(ROOT, "oracle", "liborc.so")      # (ORC_LIBRARY: a sanitizer build outside the tree, tests/run_sanitized.sh)


class OrcRef(C.Structure):
    _fields_ = [("n_contigs", C.c_int), ("names", C.POINTER(C.c_char_p)), ("lengths", C.POINTER(C.c_int64)),
                ("seqs", C.POINTER(C.c_char_p)), ("seq_lens", C.POINTER(C.c_int64))]


class OrcSample(C.Structure):
    _fields_ = [("records", C.c_void_p), ("n_bytes", C.c_uint64)]


class MpOpts(C.Structure):
    _fields_ = [("min_baseq", C.c_int), ("flag_filter", C.c_int), ("count_orphans", C.c_int), ("max_depth", C.c_int),
                ("min_mapq", C.c_int), ("ignore_overlaps", C.c_int), ("n_bed", C.c_int), ("bed_tid", C.POINTER(C.c_int)),
                ("bed_beg", C.POINTER(C.c_int64)), ("bed_end", C.POINTER(C.c_int64))]


class ScOpts(C.Structure):
    _fields_ = [("min_coverage", C.c_int), ("calling_threshold", C.c_int), ("calling_min_fraction", C.c_double),
                ("fasta_path", C.c_char_p), ("genes_path", C.c_char_p), ("token_cap", C.c_int)]


class OrcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oracle error %d: %s" % (code, msg))
        self.code = code


ERR_DOMAIN = 3
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_so)
        L.orc_last_error.restype = C.c_char_p
        L.orc_call.argtypes = [C.POINTER(OrcRef), C.POINTER(OrcSample), C.c_int, C.POINTER(MpOpts), C.POINTER(ScOpts),
                               C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_mpileup_to_file.argtypes = [C.POINTER(OrcRef), C.POINTER(OrcSample), C.c_int, C.POINTER(MpOpts), C.c_char_p]
        L.orc_qacompute.argtypes = [C.POINTER(OrcRef), C.POINTER(OrcSample), C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise OrcError(rc, (lib().orc_last_error() or b"").decode())


class _Keep:
    """Holds the ctypes arrays that back an OrcRef / OrcSample[] alive."""


def make_ref(names, lengths, seqs=None):
    k = _Keep()
    n = len(names)
    k.names = (C.c_char_p * n)(*[x.encode() if isinstance(x, str) else x for x in names])
    k.lengths = (C.c_int64 * n)(*lengths)
    if seqs is None:
        k.ref = OrcRef(n, k.names, k.lengths, None, None)
    else:
        k.seq_bytes = [None if s is None else (s.encode() if isinstance(s, str) else s) for s in seqs]
        k.seqs = (C.c_char_p * n)(*k.seq_bytes)
        k.seq_lens = (C.c_int64 * n)(*[0 if s is None else len(s) for s in k.seq_bytes])
        k.ref = OrcRef(n, k.names, k.lengths, k.seqs, k.seq_lens)
    return k


def make_samples(record_arrays):
    k = _Keep()
    k.bufs = [np.ascontiguousarray(np.frombuffer(r, dtype=np.uint8) if isinstance(r, (bytes, bytearray)) else r, dtype=np.uint8)
              for r in record_arrays]
    k.arr = (OrcSample * len(k.bufs))()
    for i, b in enumerate(k.bufs):
        k.arr[i].records = b.ctypes.data if b.size else None
        k.arr[i].n_bytes = b.size
    return k


def mp_opts(bed=None, **kw):
    o = MpOpts()
    lib().orc_mpileup_default_opts(C.byref(o))
    for a, v in kw.items():
        setattr(o, a, v)
    keep = None
    if bed:
        n = len(bed)
        keep = ((C.c_int * n)(*[b[0] for b in bed]), (C.c_int64 * n)(*[b[1] for b in bed]), (C.c_int64 * n)(*[b[2] for b in bed]))
        o.n_bed, o.bed_tid, o.bed_beg, o.bed_end = n, keep[0], keep[1], keep[2]
    return o, keep


def sc_opts(fasta=None, genes=None, **kw):
    o = ScOpts()
    lib().orc_snpcall_default_opts(C.byref(o))
    for a, v in kw.items():
        setattr(o, a, v)
    if fasta:
        o.fasta_path = fasta.encode()
    if genes:
        o.genes_path = genes.encode()
    return o


def call(names, lengths, seqs, samples, bed=None, fasta=None, genes=None, mp=None, sc=None):
    """mpileup | snpCall restatement.  Returns (called_SNPs text, indiv_called text, n_lines, n_pileup_bases)."""
    L = lib()
    ref = make_ref(names, lengths, seqs)
    smp = make_samples(samples)
    mo, keep = mp_opts(bed, **(mp or {}))
    so = sc_opts(fasta, genes, **(sc or {}))
    with tempfile.TemporaryDirectory() as td:
        pp, ip = os.path.join(td, "pop"), os.path.join(td, "ind")
        nl, nb = C.c_uint64(), C.c_uint64()
        _check(L.orc_call(C.byref(ref.ref), smp.arr, len(smp.bufs), C.byref(mo), C.byref(so), pp.encode(), ip.encode(),
                          C.byref(nl), C.byref(nb)))
        return open(pp).read(), open(ip).read(), nl.value, nb.value


def mpileup_text(names, lengths, seqs, samples, bed=None, mp=None):
    L = lib()
    ref = make_ref(names, lengths, seqs)
    smp = make_samples(samples)
    mo, keep = mp_opts(bed, **(mp or {}))
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "mp")
        _check(L.orc_mpileup_to_file(C.byref(ref.ref), smp.arr, len(smp.bufs), C.byref(mo), p.encode()))
        return open(p).read()


def qacompute(names, lengths, sample, max_cov=10, min_mapq=1):
    """Returns (.cov text, .cov.detail text)."""
    L = lib()
    ref = make_ref(names, lengths)
    smp = make_samples([sample])
    with tempfile.TemporaryDirectory() as td:
        cp, dp = os.path.join(td, "x.cov"), os.path.join(td, "x.cov.detail")
        _check(L.orc_qacompute(C.byref(ref.ref), smp.arr, max_cov, min_mapq, cp.encode(), dp.encode()))
        return open(cp).read(), open(dp).read()


def snpcall_text(mpileup, fasta=None, genes=None, **kw):
    """snpCall restatement on mpileup text.  Returns (called_SNPs, indiv_called)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "orc_snpcall")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "orc_snpcall"], stdout=subprocess.DEVNULL)
    with tempfile.TemporaryDirectory() as td:
        ip = os.path.join(td, "ind")
        cmd = [exe, "-i", ip, "-c", str(kw.get("c", 4)), "-t", str(kw.get("t", 4))]
        if "p" in kw:
            cmd += ["-p", str(kw["p"])]
        if fasta:
            cmd += ["-f", fasta]
        if genes:
            cmd += ["-g", genes]
        r = subprocess.run(cmd, input=mpileup.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return r.returncode, r.stdout.decode(), open(ip).read() if os.path.exists(ip) else "", r.stderr.decode()
