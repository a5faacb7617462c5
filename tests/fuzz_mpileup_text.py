"""Randomised parity sweep of the text entry (msnv_call_from_mpileup vs the oracle's snpCall restatement) over malformed and
well-formed pileup text alike; run on the GPU box:  python3 tests/fuzz_mpileup_text.py [n_cases] [seed].
Both sides must agree on the outputs, or both must report a domain error (an input the reference crashes on)."""
import os, random, sys, tempfile
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
from metasnv_amd import core, _lib
import orc

GOOD = ".,.,.,.,..,,ACGTacgtNn*$"


def token(rnd, bad):
    n = rnd.choice([0, 0, 1, 3, 8, 20, 60])
    out = []
    for _ in range(n):
        u = rnd.random()
        if u < 0.80:
            out.append(rnd.choice(GOOD))
        elif u < 0.86:
            out.append("^" + rnd.choice("]I~~]]^+-. "))
        elif u < 0.93:
            k = rnd.choice([0, 1, 2, 3, 12, 150])
            out.append(rnd.choice("+-") + (str(k) if rnd.random() < 0.9 else "") + "".join(rnd.choice("ACGTNacgtn*") for _ in range(rnd.choice([k] * 30 + [max(0, k - 1), k + 1]))))
        elif bad and u < 0.95:
            out.append(rnd.choice("<>RYKMxX#@ 0123456789"))      # symbols the reference has no key for (blanks and stray digits among them)
    t = "".join(out)
    if rnd.random() < 0.1:
        t = " " * rnd.randint(1, 3) + t
    return t


def make_text(rnd):
    S = rnd.choice([0, 1, 2, 3, 5, 9, 70, 130])
    bad = rnd.random() < 0.15
    n_lines = rnd.choice([0, 1, 2, 5, 30, 200])
    lines = []
    for li in range(n_lines + 1):
        name = rnd.choice(["c1", "c1", "c2", "ctg.x", " c3"])
        pos = rnd.choice([str(li + 1), str(li + 1), " 7", "", "12x", "-3"])
        refc = rnd.choice(["A", "C", "G", "T", "N", "a", "c", "g", "t", "", "AC", " T"])
        cols = [name, pos, refc]
        s_here = max(0, S - rnd.choice([1, 2])) if rnd.random() < 0.05 else S + rnd.choice([1, 2]) if rnd.random() < 0.004 else S
        for _ in range(s_here):
            t = token(rnd, bad and li > 0)
            cols += [str(len(t)), t, "I" * len(t) if rnd.random() < 0.9 else ""]
        line = "\t".join(cols)
        u = rnd.random()
        if u < 0.05:
            line += "\t"
        elif u < 0.08:
            line = line[:rnd.randint(0, len(line))]
        elif u < 0.10 and li > 0:
            k = rnd.randint(0, len(line))
            line = line[:k] + "\0" + line[k:]
        lines.append(line)
    text = "\n".join(lines)
    if rnd.random() < 0.9:
        text += "\n"
    return text


def sweep(n_cases, seed):
    rnd = random.Random(seed)
    ctx = core.Context(0)
    bad = n_err = 0
    with tempfile.TemporaryDirectory() as td:
        pp, ip = os.path.join(td, "c"), os.path.join(td, "i")
        for case in range(n_cases):
            text = make_text(rnd)
            kw = dict(c=rnd.choice([1, 4, 4, 8]), t=rnd.choice([1, 2, 4, 4]), p=rnd.choice([0.01, 0.01, 0.3, 0.0]))
            os.environ["MSNV_TEXT_CHUNK"] = str(rnd.choice([1, 300, 1 << 28]))
            raw = text.encode("latin-1")
            with tempfile.NamedTemporaryFile(dir=td, delete=False) as f:
                f.write(raw)
            import subprocess
            exe = os.path.join(_ROOT, "oracle", "orc_snpcall")
            oi = os.path.join(td, "oi")
            if os.path.exists(oi):
                os.remove(oi)
            r = subprocess.run([exe, "-i", oi, "-c", str(kw["c"]), "-t", str(kw["t"]), "-p", str(kw["p"])], input=raw, capture_output=True)
            os.remove(f.name)
            o_pop, o_ind = r.stdout, (open(oi, "rb").read() if os.path.exists(oi) else b"")
            try:
                core.call_from_mpileup(ctx, pp, ip, text=raw, params=core.default_params(min_coverage=kw["c"], calling_threshold=kw["t"], min_fraction=kw["p"]))
                got = (0, open(pp, "rb").read(), open(ip, "rb").read())
            except _lib.MsnvError as e:
                got = (e.code, b"", b"")
            if r.returncode == orc.ERR_DOMAIN or got[0] == _lib.EDOMAIN:
                n_err += 1
                ok = r.returncode == orc.ERR_DOMAIN and got[0] == _lib.EDOMAIN
            else:
                ok = r.returncode == 0 and got == (0, o_pop, o_ind)
            if not ok:
                bad += 1
                dump = os.path.join(_ROOT, "gpurun_out", "fuzz_text_case_%d_%d.txt" % (seed, case))
                os.makedirs(os.path.dirname(dump), exist_ok=True)
                open(dump, "wb").write(raw)
                print("MISMATCH case %d (seed %d): oracle rc %d, product rc %d, opts %r -> %s" % (case, seed, r.returncode, got[0], kw, dump))
                print("  oracle:", r.stderr[-200:], o_pop[:200]); print("  product:", got[1][:200])
    ctx.close()
    print("%d cases, %d mismatches, %d domain errors on both sides" % (n_cases, bad, n_err - bad if n_err >= bad else n_err))
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if sweep(n, seed) else 0)
