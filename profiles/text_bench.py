"""profiles/text_bench.py [samples] [positions] -- msnv_call_from_mpileup (csrc/textcall.hip) on a synthetic pileup text of the
benchmark shape: `samples` columns, ~10x per sample, 0.7 % SNV positions, 0.1 % errors, ^ $ and indel markers.  Prints the text
size, the kernel time and the text bytes per second it parsed (the kernel's roofline is HBM: every byte of the text is read once,
the base strings a second time from cache)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from metasnv_amd import core

S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
P = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
rng = np.random.default_rng(7)


def make_text():
    out = []
    ref = rng.choice(list("ACGT"), size=P)
    snv = rng.random(P) < 0.007
    alt = rng.choice(list("ACGT"), size=P)
    for i in range(P):
        depth = rng.poisson(10.0, size=S)
        cols = ["ctg", str(i + 1), ref[i]]
        for s in range(S):
            d = int(depth[s])
            if d == 0:
                cols += ["0", "*", "*"]
                continue
            b = np.where(rng.random(d) < 0.5, ".", ",").astype(object)
            if snv[i] and alt[i] != ref[i] and (s % 3 == 0):
                b[:] = alt[i] if s % 2 else alt[i].lower()
            e = rng.random(d) < 0.001
            b[e] = "T"
            txt = "".join(b)
            if rng.random() < 0.1:
                txt = "^]" + txt + "$"
            if rng.random() < 0.02:
                txt += "+2AC"
            cols += [str(d), txt, "I" * d]
        out.append("\t".join(cols))
    return "\n".join(out) + "\n"


t0 = time.perf_counter()
text = make_text()
print("text: %d samples x %d positions = %.1f MB (%.1f s to generate)" % (S, P, len(text) / 1e6, time.perf_counter() - t0))
ctx = core.Context(0)
with tempfile.TemporaryDirectory() as td:
    for rep in range(3):
        t0 = time.perf_counter()
        st = core.call_from_mpileup(ctx, os.path.join(td, "c"), os.path.join(td, "i"), text=text)
        wall = time.perf_counter() - t0
        print("run %d: kernel %.3f ms = %.1f GB/s of text, %.2f G base characters/s; wall %.3f s; %d called lines" %
              (rep, st["kernel_ms"], st["text_bytes"] / st["kernel_ms"] / 1e6, st["base_chars"] / st["kernel_ms"] / 1e6, wall, st["called_lines"]))
ctx.close()
