#!/usr/bin/env python3
"""profiles/perturb_build.py OUT_DIR -- the EXPERIMENT build behind profiles/r04e_perturb.txt: a copy of metasnv_amd/csrc with a run-time word
(MSNV_PERTURB) that switches parts of msnv_pileup_tiles_narrow32 off, built into OUT_DIR/libmsnv.so (select it with MSNV_LIBRARY=...).
Results of such a run are wrong by construction; only times and counters are read.  Bits: 1 no mismatch walk, 2 no low-quality atomics,
4 no per-pair pass, 8 no classify at all, 16 no coverage difference array, 32 no column loads.
    python3 profiles/perturb_build.py /tmp/msnv_perturb && MSNV_LIBRARY=/tmp/msnv_perturb/libmsnv.so bash profiles/perturb.sh 0 1 2 4 8 12 28 60
"""
import os, shutil, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/msnv_perturb"
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out)
src = os.path.join(out, "metasnv_amd", "csrc")
shutil.copytree(os.path.join(root, "metasnv_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o", "*.so"))
shutil.copytree(os.path.join(root, "include"), os.path.join(out, "include"))
p = os.path.join(src, "kernels.hip")
s = open(p).read()
EDITS = [
    ("    uint32_t       *stage_ovf;    // record-list indices", "    uint32_t        perturb;\n    uint32_t       *stage_ovf;    // record-list indices"),
    ("a.counters = counters; a.min_baseq = (uint32_t)std::max(0, p.min_baseq);",
     "a.counters = counters; a.min_baseq = (uint32_t)std::max(0, p.min_baseq); a.perturb = getenv(\"MSNV_PERTURB\") ? (uint32_t)atoi(getenv(\"MSNV_PERTURB\")) : 0u;"),
    ("__device__ __forceinline__ void narrow_classify32(NarrowLds &L, const uint32_t lq_all, const uint4 sq, const uint32_t P0, const int vhi) {",
     "__device__ __forceinline__ void narrow_classify32(NarrowLds &L, const uint32_t lq_all, const uint4 sq, const uint32_t P0, const int vhi, const uint32_t pt = 0u) {"),
    ("        if (byte) atomicAdd(&L.exc[wi + w], spread_bits(byte));      // (one test per lane for the first four words",
     "        if (byte && !(pt & 2u)) atomicAdd(&L.exc[wi + w], spread_bits(byte));      // (one test per lane for the first four words"),
    ("    if constexpr (SEQ_ALIGN_LOG2 < 3) E &= L.emask[min(max(vhi, 0), 32)];\n    while (E) {",
     "    if constexpr (SEQ_ALIGN_LOG2 < 3) E &= L.emask[min(max(vhi, 0), 32)];\n    if (pt & 1u) E = 0u;\n    while (E) {"),
    ("            if (__any(vh[i] > 0)) narrow_classify32(L, lowq_bits(ql[i], qsh[i]), sq[i], P0[i], vh[i]);\n\n        if (tid < N_HCAP) put_hdr((c + 1u) & 1u, hreg);",
     "            if (__any(vh[i] > 0) && !(a.perturb & 8u)) narrow_classify32(L, lowq_bits(ql[i], qsh[i]), sq[i], P0[i], vh[i], a.perturb);\n\n        if (tid < N_HCAP) put_hdr((c + 1u) & 1u, hreg);"),
    ("        if (c + 1u < nch) issue_loads(c + 1u);                       // in flight under the per-sample pass\n        if (last_chunk) narrow_pass<NarrowLds, 0, MERGED, fused, DA>",
     "        if (c + 1u < nch && !(a.perturb & 32u)) issue_loads(c + 1u);                       // in flight under the per-sample pass\n        if (last_chunk && !(a.perturb & 4u)) narrow_pass<NarrowLds, 0, MERGED, fused, DA>"),
    ("        if (tid < N_HCAP) {\n            const uint32_t hx = get_hdr(c & 1u, (uint32_t)tid).x;\n            const uint32_t s = hx & (TILE - 1u), sb = s + ((hx >> 11) & 0xffu);\n            if (sb != s) {                                           // coverage difference array: +1 at the start, -1 behind the end",
     "        if (tid < N_HCAP && !(a.perturb & 16u)) {\n            const uint32_t hx = get_hdr(c & 1u, (uint32_t)tid).x;\n            const uint32_t s = hx & (TILE - 1u), sb = s + ((hx >> 11) & 0xffu);\n            if (sb != s) {                                           // coverage difference array: +1 at the start, -1 behind the end"),
]
for a, b in EDITS:
    if s.count(a) < 1:
        sys.exit("perturb_build.py: kernels.hip no longer holds the text of an edit:\n" + a[:120])
    s = s.replace(a, b, 1)
open(p, "w").write(s)
subprocess.check_call(["make", "-C", src, "libmsnv.so"], stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(src, "libmsnv.so"), os.path.join(out, "libmsnv.so"))
print(os.path.join(out, "libmsnv.so"))
