"""Pure-Python BAM helpers for the tests: build raw alignment records from SAM-like fields and
parse BAM files with the standard library only (gzip reads BGZF because BGZF is a series of
gzip members) -- an independent check of the library's own BGZF/BAM reader and writer."""
import gzip
import re
import struct

import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def parse_cigar(cigar):
    if cigar in ("*", ""):
        return []
    return [(int(n), CIGAR_OPS.index(op)) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar)]


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def aux_fields(nm=None, md=None, score=None):
    """Auxiliary fields as bwa / ngless write them behind the qualities: NM:C, MD:Z, AS:i."""
    out = b""
    if nm is not None:
        out += b"NMC" + bytes([nm])
    if md is not None:
        out += b"MDZ" + md.encode() + b"\0"
    if score is not None:
        out += b"ASi" + struct.pack("<i", score)
    return out


def make_record(tid, pos, cigar, seq, qual=None, flag=0, mapq=60, name="r", mtid=-1, mpos=-1, tlen=0, aux=b"", cg_form=False):
    """One raw BAM alignment record (SAMv1 4.2).  pos is 0-based; qual is a list of ints or None; aux: bytes behind the qualities.
    cg_form: the CIGAR goes into a CG:B,I field behind the placeholder `<l_seq>S<ref_len>N`, as a writer does for more than 65535
    operations (SAMv1 4.2.2) -- htslib puts it back on reading, whatever its length."""
    ops = parse_cigar(cigar)
    if cg_form:
        real = ops
        l = 0 if seq in ("*", "") else len(seq)
        ref_len = sum(n for n, op in real if op in (0, 2, 3, 7, 8))
        ops = [(l, 4), (ref_len, 3)]
        aux = aux + b"CGBI" + struct.pack("<I", len(real)) + b"".join(struct.pack("<I", n << 4 | op) for n, op in real)
    l_seq = 0 if seq in ("*", "") else len(seq)
    if qual is None:
        qual = [40] * l_seq
    if isinstance(qual, str):
        qual = [ord(c) - 33 for c in qual]
    assert len(qual) == l_seq
    rlen = sum(n for n, op in (real if cg_form else ops) if op in (0, 2, 3, 7, 8))
    nm = name.encode() + b"\0"
    seqb = bytearray((l_seq + 1) // 2)
    for i in range(l_seq):
        code = NT16.index(seq[i].upper()) if seq[i].upper() in NT16 else 15
        seqb[i >> 1] |= code << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(nm), mapq, reg2bin(pos, pos + max(rlen, 1)), len(ops), flag,
                       l_seq, mtid, mpos, tlen)
    body += nm + b"".join(struct.pack("<I", n << 4 | op) for n, op in ops) + bytes(seqb) + bytes(qual) + aux
    return struct.pack("<i", len(body)) + body


def records(*recs):
    return np.frombuffer(b"".join(recs), dtype=np.uint8)


def read_bam_py(path):
    """Parse a BAM with gzip + struct only.  Returns (header_text, names, lengths, record_bytes)."""
    with gzip.open(path, "rb") as f:
        u = f.read()
    assert u[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", u, 4)[0]
    text = u[8:8 + l_text].rstrip(b"\0").decode()
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", u, off)[0]
    off += 4
    names, lengths = [], []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", u, off)[0]
        off += 4
        names.append(u[off:off + l_name - 1].decode())
        off += l_name
        lengths.append(struct.unpack_from("<i", u, off)[0])
        off += 4
    return text, names, lengths, u[off:]


def iter_records(buf):
    """Yield dicts of decoded fields from a raw record stream."""
    buf = bytes(buf)
    off = 0
    while off < len(buf):
        bs = struct.unpack_from("<i", buf, off)[0]
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", buf, off + 4)
        p = off + 36
        name = buf[p:p + l_name - 1].decode()
        p += l_name
        cig = [(c >> 4, c & 15) for c in struct.unpack_from("<%dI" % n_cig, buf, p)]
        p += 4 * n_cig
        sb = buf[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        seq = "".join(NT16[(sb[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = list(buf[p:p + l_seq])
        yield dict(tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=cig, seq=seq, qual=qual, name=name)
        off += bs + 4


def write_bam_py(path, names, lengths, record_bytes, rnd, header_text=None):
    """A BAM written by PYTHON's zlib, not by the library's writer: BGZF members of varying payload sizes (1 byte .. 65280), each deflated
    with another strategy / level / memory level (Z_FIXED, Z_RLE, Z_HUFFMAN_ONLY, Z_FILTERED, stored level 0, sync-flushed multi-block
    payloads), empty members in between, records and even the fixed 4-byte fields cut across members, the EOF marker at the end.  What
    other BGZF writers (htslib with libdeflate, bgzip -l, samtools -1/-9, Picard's Java Deflater) produce differs from zlib level 6 in
    exactly these ways: block split points, fixed vs dynamic codes, match-length / literal mixes."""
    import struct
    import zlib
    text = (header_text if header_text is not None else "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(names, lengths))).encode()
    body = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(names))
    for n, l in zip(names, lengths):
        nb = n.encode() + b"\0"
        body += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    body += bytes(record_bytes)
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]

    def member(payload):
        st = rnd.choice(strategies); lvl = rnd.choice([0, 1, 1, 4, 6, 6, 9]); mem = rnd.choice([1, 4, 8, 9])
        c = zlib.compressobj(lvl, zlib.DEFLATED, -15, mem, st)
        if len(payload) > 2000 and rnd.random() < 0.3:        # several deflate blocks in one member
            cut = rnd.randrange(1, len(payload))
            comp = c.compress(payload[:cut]) + c.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])) + c.compress(payload[cut:]) + c.flush()
        else:
            comp = c.compress(payload) + c.flush()
        if len(comp) + 26 > 65536:                            # incompressible under this strategy: stored
            c = zlib.compressobj(0, zlib.DEFLATED, -15)
            comp = c.compress(payload) + c.flush()
        bsize = len(comp) + 25
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))
    out = bytearray()
    pos = 0
    while pos < len(body):
        n = rnd.choice([1, 3, 17, 300, 4000, 30000, 65280, 65280, 65280])
        if len(body) - pos < n: n = len(body) - pos
        out += member(body[pos:pos + n]); pos += n
        if rnd.random() < 0.05: out += member(b"")              # an empty member in the middle of the file
    out += member(b"")                                         # the EOF marker (an empty member)
    open(path, "wb").write(out)
