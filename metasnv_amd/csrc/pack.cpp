// metasnv_amd/csrc/pack.cpp -- host side of the boundary: BAM records -> packed read columns
// (dataset.h), tile index, upload.  This is the "decoded on the host ... streamed to the device
// as packed per-read columns" stage of the north star; no pileup arithmetic happens here, only
// the read-level filters that `samtools mpileup` applies before its pileup engine
// (bam_plcmd.c mplp_func, restated in SURVEY.md Appendix C) and qaCompute's read filter
// (qaCompute.cpp:461-526), both reduced to two bits per read.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <queue>
#include <thread>
#include <unordered_map>

#include "device.h"
#include "devpack.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace msnv {

// Piece alignment in the seq column (bytes; the qual column is twice that): dataset.h SEQ_ALIGN, 2 bytes since round 3 (round 1,
// 16 bases per lane: 16 -> 0.725 ms, 8 -> 0.665, 4 -> 0.664, 2 -> 0.651; round 3, 32 bases per lane: 8 -> 0.573, 4 -> 0.548, 2 -> 0.547).
// The compact headers store seq offsets in SEQ_ALIGN units.
constexpr uint32_t seq_align = SEQ_ALIGN;

// The device's quality column (kernels.hip: lowq_fetch): bit i = quality byte i of the host staging is below the cutoff.  The staged
// bytes are clamped to <= 127 (pack_sample), padding bytes are 0 (flagged unless the cutoff is 0; the kernels mask what lies beyond a piece).
static void pack_lowq(const uint8_t *q, size_t n, int cutoff, std::vector<uint8_t> &out) {
    out.assign((n + 7) / 8, 0);
    // staged bytes are 0 .. 127, or 0x80 for a base behind snpCall's token limit: as SIGNED bytes that one is below every cutoff,
    // the cutoff 0 (mpileup -Q 0: no quality is too low) included
    const int c_eff = std::min(std::max(cutoff, -127), 127);
    const bool all = cutoff > 127;
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i c = _mm_set1_epi8((char)c_eff);
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(q + i));
        const uint32_t m = all ? 0xffffu : (uint32_t)_mm_movemask_epi8(_mm_cmpgt_epi8(c, v));
        out[i >> 3] = (uint8_t)m; out[(i >> 3) + 1] = (uint8_t)(m >> 8);
    }
#endif
    for (; i < n; ++i) if (all || (int)(int8_t)q[i] < c_eff) out[i >> 3] |= (uint8_t)(1u << (i & 7));
}


// Layout of the narrow path: padded per-piece columns (msnv_pileup_tiles_narrow32) or the dense block stream
// (msnv_pileup_tiles_dense: no alignment padding, every lane owns 32 real bases, but up to two segments per block).
// Dense wins for short pieces (50-base reads: -18 % kernel time), loses from ~100 bases on (kernels.hip), so the
// dataset picks it when the mean piece is shorter than 62 bases.  MSNV_LAYOUT=pieces|dense overrides.
static bool layout_dense(uint64_t n_pieces, uint64_t n_bases) {
    const char *e = getenv("MSNV_LAYOUT");
    if (e && e[0] == 'p') return false;
    if (e && e[0] == 'd') return true;
    return n_pieces && n_bases / n_pieces < 62;      // (round 3, one bit of quality per base, eight workgroups per CU: 50-base reads 0.589 vs 0.535 ms, 75-base reads 0.476 vs 0.535 -- break-even near 62; it was 72: profiles/r03zap_layout_by_read_length.txt)
}

// A (contig, tile) run of one sample whose depth can reach NARROW_MAX_DEPTH does not fit the byte bins of the narrow
// kernels, and the kernels slow down well before that (8 samples at ~100x: 0.531 ms unsplit, 0.336 ms when runs deeper than
// 192 are dealt into groups of <= 128; at ~400x 2.68 ms with groups of 240, 1.66 ms with groups of 128).  Instead of a second kernel with wider bins, the run is dealt into groups of pieces that each stay below the
// limit (round robin in start order, verified with an exact sweep, more groups if needed); every group becomes its own
// (sample, tile) pair and the per-sample results are summed on the device (gather / scatter accumulate).  The pieces of a
// group are made contiguous -- headers AND bases / qualities (MSNV_DEEP_RELOCATE=0 leaves the columns in read order: every group's
// workgroups then fetch almost every cache line of the run).  MSNV_DEEP=wide keeps such runs whole for msnv_pileup_tiles_wide instead.
static bool deep_runs_split() {
    const char *e = getenv("MSNV_DEEP");                     // read per dataset (tests switch it)
    return !(e && e[0] == 'w');
}
static uint32_t deep_split_at() {
    static const uint32_t v = [] { const char *e = getenv("MSNV_SPLIT_AT"); const int x = e ? atoi(e) : 192; return (uint32_t)std::min<int>((int)NARROW_MAX_DEPTH, std::max(32, x)); }();
    return v;
}
static uint32_t deep_group_depth() {
    static const uint32_t v = [] { const char *e = getenv("MSNV_GROUP_DEPTH"); const int x = e ? atoi(e) : 128; return (uint32_t)std::min(250, std::max(16, x)); }();
    return v;
}
static int split_deep_runs(SampleCols &sc, int device) {
    const size_t n = sc.hdr.size();
    sc.grp.assign(n, 0);
    if (!deep_runs_split()) return MSNV_OK;
    struct Ev { uint32_t pos; int32_t delta; uint32_t g; };
    std::vector<Ev> ev;
    std::vector<uint32_t> cur, mx, order;
    bool any_split = false;
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        while (j < n && sc.tid[j] == sc.tid[i] && sc.hdr[j].gpos / TILE == sc.hdr[i].gpos / TILE) ++j;
        uint32_t bound = 0;
        for (size_t k = i; k < j; ++k) bound = std::max<uint32_t>(bound, sc.depth[k]);
        const uint32_t split_at = deep_split_at();
        if (bound >= split_at) {
            const size_t m = j - i;
            auto sweep = [&](uint32_t G) -> uint32_t {            // largest per-position depth of any group
                ev.clear();
                for (size_t k = 0; k < m; ++k) {
                    const ReadHdr &h = sc.hdr[i + k];
                    ev.push_back(Ev{h.gpos, +1, (uint32_t)(k % G)});
                    ev.push_back(Ev{h.gpos + h.cig, -1, (uint32_t)(k % G)});
                }
                std::sort(ev.begin(), ev.end(), [](const Ev &a, const Ev &b) { return a.pos != b.pos ? a.pos < b.pos : a.delta < b.delta; });
                cur.assign(G, 0); mx.assign(G, 0);
                for (const Ev &e : ev) { cur[e.g] += (uint32_t)e.delta; mx[e.g] = std::max(mx[e.g], cur[e.g]); }
                return *std::max_element(mx.begin(), mx.end());
            };
            const uint32_t exact = sweep(1);
            if (exact < split_at) {
                for (size_t k = i; k < j; ++k) sc.depth[k] = (uint16_t)exact;        // the start-time bound was pessimistic
            } else {
                const uint32_t group_depth = deep_group_depth();
                uint32_t G = exact / group_depth + 1;
                while (sweep(G) >= NARROW_MAX_DEPTH) ++G;
                std::vector<uint32_t> gmax = mx;
                order.resize(m);
                for (size_t k = 0; k < m; ++k) order[k] = (uint32_t)k;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return a % G < b % G; });
                std::vector<ReadHdr> h2(m); std::vector<int32_t> t2(m), e2(m);
                for (size_t k = 0; k < m; ++k) { h2[k] = sc.hdr[i + order[k]]; t2[k] = sc.tid[i + order[k]]; e2[k] = sc.end[i + order[k]]; }
                for (size_t k = 0; k < m; ++k) {
                    sc.hdr[i + k] = h2[k]; sc.tid[i + k] = t2[k]; sc.end[i + k] = e2[k];
                    sc.grp[i + k] = (uint16_t)(1 + order[k] % G);
                    sc.depth[i + k] = (uint16_t)gmax[order[k] % G];
                }
                any_split = true;
            }
        }
        i = j;
    }
    // The bases and qualities follow their headers: a group's pieces were every G-th piece of the run in memory, so the workgroups
    // of the G groups -- at G different times -- each fetched (almost) every cache line of the run (one sample at 1600x: 6.2 GB of
    // HBM reads per pass for 2.2 GB of columns).  Same pieces (bases + twice as many quality bytes, alignment padding included), new order.
    if (any_split && !(getenv("MSNV_DEEP_RELOCATE") && getenv("MSNV_DEEP_RELOCATE")[0] == '0')) {
        if (sc.on_device) {                              // (the relocation still runs on host staging: a device-packed sample comes back for it)
            if (int rc = dev_set_device(device)) return rc;
            if (int rc = devpack_sample_to_host(sc)) return rc;
        }
        std::vector<uint8_t> nseq(sc.seq.size(), 0xff), nqual(sc.qual.size(), 0);
        size_t so = 0;
        auto stored = [](uint32_t bases) { return (size_t)(((bases + 1u) / 2u + seq_align - 1u) & ~(seq_align - 1u)); };      // seq bytes of a piece with its alignment padding (pack_sample)
        for (size_t k = 0; k < n; ++k) {
            const size_t blk = stored(sc.hdr[k].cig);
            if (so + blk > nseq.size() || 2 * (so + blk) > nqual.size()) return MSNV_OK;       // (cannot happen: the blocks are a permutation; keep the old layout rather than write past the end)
            memcpy(nseq.data() + so, sc.seq.data() + sc.hdr[k].seqoff, blk);
            memcpy(nqual.data() + 2 * so, sc.qual.data() + 2 * (size_t)sc.hdr[k].seqoff, 2 * blk);
            so += blk;
        }
        so = 0;
        for (size_t k = 0; k < n; ++k) { sc.hdr[k].seqoff = (uint32_t)so; so += stored(sc.hdr[k].cig); }
        sc.seq.swap(nseq); sc.qual.swap(nqual);
    }
    return MSNV_OK;
}

// Rewrites seq / qual of one sample into the dense block streams (dataset.h) and fills blk / run_*.
// hdr must already be grouped by (contig, tile); hdr[i].gpos is still contig-relative (contigs start on tile boundaries).
static void relayout_dense(SampleCols &sc) {
    const size_t n = sc.hdr.size();
    std::vector<uint8_t> nseq, nqual;
    nseq.reserve(sc.seq.size()); nqual.reserve(sc.qual.size());
    sc.blk.clear(); sc.run_blk_lo.clear(); sc.run_nblk.clear(); sc.run_seq0.clear();
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        while (j < n && sc.tid[j] == sc.tid[i] && sc.hdr[j].gpos / TILE == sc.hdr[i].gpos / TILE && sc.grp[j] == sc.grp[i]) ++j;
        const uint32_t blk0 = (uint32_t)sc.blk.size();
        const size_t seq0 = nseq.size();                       // multiple of 16 bytes
        uint64_t cursor = 0;                                   // bases from the start of this run's stream
        auto block = [&](uint64_t b) -> uint32_t & {
            while (sc.blk.size() <= blk0 + b) sc.blk.push_back(BLK_EMPTY);
            return sc.blk[blk0 + b];
        };
        for (size_t k = i; k < j; ++k) {
            ReadHdr &h = sc.hdr[k];
            const uint32_t len = h.cig, P = h.gpos % TILE;
            cursor = (cursor + 1) & ~1ull;
            uint64_t b = cursor / 32; uint32_t o = (uint32_t)(cursor % 32);
            if (o != 0) {
                const uint32_t cur = block(b);
                const bool has_b = ((cur >> 17) & 0xfffu) != BLK_NO_B;
                if (has_b || len < 32 - o) { cursor = (b + 1) * 32; ++b; o = 0; }      // a second head, or one that ends inside the block: fresh block
            }
            // copy the piece: seq nibbles are byte aligned at both ends (even cursor), qualities 1:1
            const size_t so = seq0 + cursor / 2, qo = 2 * seq0 + cursor;
            if (nseq.size() < so + (len + 1) / 2) nseq.resize(so + (len + 1) / 2, 0xff);
            if (nqual.size() < qo + len) nqual.resize(qo + len, 0);
            memcpy(nseq.data() + so, sc.seq.data() + h.seqoff, (len + 1) / 2);
            if (len & 1u) nseq[so + len / 2] |= 0xf0;           // the pad nibble of an odd piece reads as N
            memcpy(nqual.data() + qo, sc.qual.data() + 2 * (size_t)h.seqoff, len);
            h.seqoff = (uint32_t)so;                           // (sample-relative; the wide kernel reads pieces through it)
            // descriptors
            uint32_t done = 0;
            if (o != 0) {                                      // head of the piece = segment B of block b
                uint32_t &w = block(b);
                w = (w & ~(0xfffu << 17)) | ((P - o + 32u) << 17);
                done = 32 - o;
                if (done == len) w |= BLK_END_B;
                ++b;
            }
            bool first = (o == 0);
            while (done < len) {
                const uint32_t na = std::min<uint32_t>(32, len - done);
                uint32_t &w = block(b);
                w = (w & (0xfffu << 17)) | (P + done) | na << 11 | (first ? BLK_START_A : 0u) | (done + na == len ? BLK_END_A : 0u);
                first = false; done += na; ++b;
            }
            cursor += len;
        }
        const uint32_t nb = (uint32_t)sc.blk.size() - blk0;
        nseq.resize(seq0 + (size_t)nb * 16, 0xff);
        nqual.resize(2 * seq0 + (size_t)nb * 32, 0);
        sc.run_blk_lo.push_back(blk0); sc.run_nblk.push_back(nb); sc.run_seq0.push_back((uint32_t)seq0);
        i = j;
    }
    sc.seq.swap(nseq); sc.qual.swap(nqual);
}

std::vector<std::string> synth_contigs(const msnv_synth_params &p);
void synth_sample_records(const msnv_synth_params &p, int sample, const std::vector<std::string> &contigs, std::vector<uint8_t> &out);

static inline bool consumes_ref(uint32_t t) { return t == C_M || t == C_D || t == C_N || t == C_EQ || t == C_X; }
static inline bool consumes_query(uint32_t t) { return t == C_M || t == C_I || t == C_S || t == C_EQ || t == C_X; }

// qaCompute's per-read bookkeeping (qaCompute.cpp:461-473 unmapped, :518-526 MAPQ / duplicate filter): the "Other" block of
// OUT counts every record of the BAM, whatever shard it belongs to.  Returns false for reads qaCompute skips as unmapped;
// cov_ok = the read enters the coverage difference array.
static inline bool read_stats(const RecView &r, int cov_min_mapq, msnv_sample_stats &st, bool &cov_ok) {
    cov_ok = false;
    st.total_reads++;
    if ((r.flag & BAM_FUNMAP) || r.tid < 0) { st.unmapped++; return false; }
    st.any_mapped = 1;
    if (r.mapq >= cov_min_mapq) {
        if (r.flag & BAM_FPROPER_PAIR) st.proper_pairs++;
        if (r.flag & BAM_FDUP) st.duplicates++; else cov_ok = true;
    } else st.zero_quality++;
    return true;
}

// Multi-GPU decode sharding (msnv.h: msnv_records_partition): one walk over a sample's record stream that deals the
// mapped records to the rank owning their contig (order preserved inside a part) and counts qaCompute's statistics.
int records_partition(const uint8_t *rec, uint64_t n_bytes, const int32_t *owner, int n_contigs, int n_parts, int cov_min_mapq,
                      uint8_t *out, uint64_t *part_bytes, msnv_sample_stats &st) {
    std::vector<uint64_t> size((size_t)n_parts, 0);
    st = msnv_sample_stats{};
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<uint64_t> cur((size_t)n_parts, 0);
        if (pass == 1) { uint64_t o = 0; for (int k = 0; k < n_parts; ++k) { cur[(size_t)k] = o; o += size[(size_t)k]; } }
        uint64_t off = 0;
        while (off < n_bytes) {
            RecView r;
            if (!rec_parse(rec + off, n_bytes - off, r)) return fail(MSNV_EFORMAT, "malformed BAM record at byte %llu", (unsigned long long)off);
            const uint8_t *src = rec + off;
            off += r.size;
            if (pass == 0) {
                bool cov_ok;
                if (!read_stats(r, cov_min_mapq, st, cov_ok)) continue;
                if (r.tid >= n_contigs) return fail(MSNV_EFORMAT, "record refers to contig %d but the header has %d", r.tid, n_contigs);
            } else if ((r.flag & BAM_FUNMAP) || r.tid < 0) continue;
            const int32_t k = owner[r.tid];
            if (k < 0) continue;                                  // contig outside every shard (BED split)
            if (k >= n_parts) return fail(MSNV_EINVAL, "contig %d is owned by part %d of %d", r.tid, k, n_parts);
            if (pass == 0) size[(size_t)k] += r.size;
            else { memcpy(out + cur[(size_t)k], src, r.size); cur[(size_t)k] += r.size; }
        }
    }
    for (int k = 0; k < n_parts; ++k) part_bytes[k] = size[(size_t)k];
    return MSNV_OK;
}

// ---------------------------------------------------------------------------------- overlapping mates
// `samtools mpileup` without -x (metaSNV.py:160-165) lets htslib's pileup engine edit the base qualities of proper-pair
// mates that overlap on the reference before the -Q cutoff sees them (sam.c overlap_push / tweak_overlap_quality, htslib
// >= 1.10): where both mates have an aligned base at a reference position, agreeing bases give min(200, qa + qb) to the
// mate that was pushed first and 0 to the other; disagreeing bases give 0.8 * q (truncated) to the one of higher quality
// (ties: the first) and 0 to the other.  So each template counts once per position, and snpCall's counts depend on it.
//
// The engine walks the two CIGARs with a shared reference cursor t: x = first match position of a at or after t,
// y = first match position of b at or after x, the pair is edited iff x == y, then t = y + 1.  Consequence kept here:
// behind a stretch where a has bases and b has none (deletion / skip in b), b's next base is never edited.
struct MatchCursor {           // the M/=/X bases of one alignment in reference order
    const RecView &r; int k = 0; int64_t op_ref, op_q; int64_t ref = -1, q = -1;
    explicit MatchCursor(const RecView &rv) : r(rv), op_ref(rv.pos), op_q(0) {}
    // first match base at a reference position >= target; false when the alignment has none left
    bool seek(const int64_t target) {
        for (; k < r.n_cigar; ++k) {
            const uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u; const int64_t l = c >> 4;
            if (t == C_M || t == C_EQ || t == C_X) {
                if (target < op_ref + l) { ref = std::max(target, op_ref); q = op_q + (ref - op_ref); return true; }
                op_ref += l; op_q += l;
            } else {
                if (t == C_D || t == C_N) op_ref += l;
                if (t == C_I || t == C_S) op_q += l;
            }
        }
        return false;
    }
};

static void tweak_overlapping_mates(const RecView &a, uint8_t *qa, const RecView &b, uint8_t *qb) {
    if (a.l_seq == 0 || b.l_seq == 0) return;
    MatchCursor ca(a), cb(b);
    int64_t t = b.pos;
    while (ca.seek(t) && cb.seek(ca.ref)) {
        t = cb.ref + 1;
        if (ca.ref != cb.ref) continue;
        if (ca.q >= a.l_seq || cb.q >= b.l_seq) return;
        const uint32_t ba = (a.seq[ca.q >> 1] >> ((~ca.q & 1) << 2)) & 0xfu, bb = (b.seq[cb.q >> 1] >> ((~cb.q & 1) << 2)) & 0xfu;
        uint8_t &x = qa[ca.q], &y = qb[cb.q];
        if (ba == bb) { const int sum = (int)x + (int)y; x = (uint8_t)(sum > 200 ? 200 : sum); y = 0; }
        else if (x >= y) { x = (uint8_t)(0.8 * x); y = 0; }
        else { y = (uint8_t)(0.8 * y); x = 0; }
    }
}

// One filtered read of pass 1, kept for pass 2 (packing).
struct KeptRead { uint64_t off; int64_t endpos; uint16_t depth_here; bool pile_ok, cov_ok; uint32_t idx; };      // idx: ordinal of the record in the stream

// ---------------------------------------------------------------------------------- snpCall's token limit
// snpCall copies every tab-separated field of a pileup line through a 10000-character token (call_vC.cpp:92-111,481-483):
// a sample's base string is CUT there and the bases behind the cut are never counted (SURVEY.md Appendix A "D1").  With
// mpileup -d 8000 that takes a stack of reads starting at one position (`^]` costs two characters per read start) or
// thousands of indel suffixes, but the reference's counts are what they are.  The string itself is never built here: where
// a sample's string can reach the limit at all (upper bound kept by the caller's sweep), the characters of every element
// are counted in pileup order -- [^ mapq] base|*|<> [+n ins | -n del] [$], elements below -Q are not printed at all
// (bam_plcmd.c pileup_seq and the text loop) -- and a base whose character lies at or behind the limit gets quality 0, so
// the device leaves it out exactly like a base below the cutoff.  Nothing else of a cut element can be counted: what
// follows a base inside its element (indel suffix, $) is skipped by snpCall's parser anyway (call_vC.cpp:506-523).
static uint32_t decimal_digits(uint32_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

static uint32_t max_element_chars(const RecView &r) {
    uint64_t ins = 0, del = 0;
    for (int k = 0; k < r.n_cigar; ++k) {
        const uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u;
        if (t == C_I) ins += c >> 4;                       // consecutive insertions around a P op print as one suffix
        else if (t == C_D) del = std::max<uint64_t>(del, c >> 4);
    }
    // `^` mapq, the base, `$`; behind a base that ends an aligned block: sign, up to ten digits, the inserted / deleted bases (round 6: the 11 only
    // for a read that has an indel at all -- a sample's string is "in reach" of the limit at 2 500 plain reads, not at 667)
    return (uint32_t)std::min<uint64_t>(0x7fffffffu, 4 + ((ins | del) ? 11 + std::max(ins, del) : 0));
}

// The mark of a base whose character is cut off.  (It used to be quality 0 -- "the device leaves it out like a base below the cutoff" --
// which is not below a cutoff of 0: with -Q 0 the cut bases were counted.  metaSNV never runs -Q 0; the fuzz sweep does.)  Qualities of
// 128 and more pass every cutoff just like 127 (pack_sample clamps them), so the reads the walk below can touch are clamped first and
// the mark cannot be mistaken for a stored quality.
constexpr uint8_t QUAL_CUT = 0xfe;

template <typename MutableQual>
static void apply_token_limit(const msnv_params &P, const uint8_t *rec, uint64_t n_bytes, const std::vector<KeptRead> &kept, MutableQual &mutable_qual) {
    for (const KeptRead &kr : kept) {
        if (!kr.pile_ok) continue;
        RecView r;
        rec_parse(rec + kr.off, n_bytes - kr.off, r);
        if (r.l_seq <= 0) continue;
        uint8_t *q = mutable_qual(r);
        for (int32_t j = 0; j < r.l_seq; ++j) if (q[j] > 127) q[j] = 127;
    }
    struct Act { RecView r; int64_t end; uint32_t maxc; int k; int64_t x, y; };
    std::vector<Act> act;                                   // pileup reads alive at the current position, in push (= file) order
    uint64_t bound = 0;
    const uint64_t limit = (uint64_t)P.token_limit;
    size_t next = 0;
    while (next < kept.size() && !kept[next].pile_ok) ++next;
    int32_t tid = -1; int64_t pos = 0;
    auto peek = [&](RecView &r) { rec_parse(rec + kept[next].off, n_bytes - kept[next].off, r); };
    while (next < kept.size() || !act.empty()) {
        RecView nr{};
        if (next < kept.size()) peek(nr);
        if (act.empty()) {
            if (next >= kept.size()) break;
            tid = nr.tid; pos = nr.pos;
        }
        // reads that start here
        while (next < kept.size() && nr.tid == tid && nr.pos == pos) {
            Act a{nr, kept[next].endpos, max_element_chars(nr), 0, nr.pos, 0};
            for (; a.k < nr.n_cigar; ++a.k) {               // cursor on the first reference-consuming op (sam.c resolve_cigar2)
                const uint32_t c = ld_u32(nr.cigar + 4 * a.k), t = c & 15u;
                if (consumes_ref(t)) break;
                if (t == C_I || t == C_S) a.y += c >> 4;
            }
            act.push_back(a); bound += a.maxc;
            do ++next; while (next < kept.size() && !kept[next].pile_ok);
            if (next < kept.size()) peek(nr);
        }
        if (bound >= limit) {
            uint64_t off = 0;                               // characters of this sample's base string so far
            for (Act &a : act) {
                const RecView &r = a.r;
                bool found = false;
                while (a.k < r.n_cigar) {                   // advance to the op that holds `pos`
                    const uint32_t c = ld_u32(r.cigar + 4 * a.k), t = c & 15u; const int64_t l = c >> 4;
                    if (consumes_ref(t) && pos < a.x + l) { found = true; break; }
                    if (consumes_ref(t)) a.x += l;
                    if (consumes_query(t)) a.y += l;
                    ++a.k;
                }
                if (!found) continue;
                const uint32_t c = ld_u32(r.cigar + 4 * a.k), t = c & 15u; const int64_t l = c >> 4;
                const bool is_del = !(t == C_M || t == C_EQ || t == C_X);
                const int64_t qpos = is_del ? a.y : a.y + (pos - a.x);
                int64_t indel = 0;
                if (!is_del && a.x + l - 1 == pos && a.k + 1 < r.n_cigar) {
                    const uint32_t c2 = ld_u32(r.cigar + 4 * (a.k + 1)), t2 = c2 & 15u;
                    if (t2 == C_D) indel = -(int64_t)(c2 >> 4);
                    else if (t2 == C_I) indel = c2 >> 4;
                    else if (t2 == C_P && a.k + 2 < r.n_cigar) {
                        for (int k = a.k + 2; k < r.n_cigar; ++k) {
                            const uint32_t c3 = ld_u32(r.cigar + 4 * k), t3 = c3 & 15u;
                            if (t3 == C_I) indel += c3 >> 4;
                            else if (t3 == C_D || t3 == C_M || t3 == C_N || t3 == C_EQ || t3 == C_X) break;
                        }
                    }
                }
                uint8_t *q = (r.l_seq > 0) ? mutable_qual(r) : nullptr;
                const int qv = (q && qpos < r.l_seq) ? q[qpos] : 0;
                if (qv < P.min_baseq) continue;             // not printed at all
                const bool head = pos == r.pos, tail = pos == a.end - 1;
                if (!is_del && off + (head ? 2u : 0u) >= limit && q) q[qpos] = QUAL_CUT;      // the base's own character is cut off
                const uint64_t n_indel = (uint64_t)(indel < 0 ? -indel : indel);
                off += (head ? 2u : 0u) + 1u + (indel ? 1u + decimal_digits((uint32_t)n_indel) + n_indel : 0u) + (tail ? 1u : 0u);
            }
        }
        // next position: the following one while reads are alive, else the next read's start
        ++pos;
        size_t w = 0;
        for (size_t i = 0; i < act.size(); ++i) { if (act[i].end > pos) act[w++] = act[i]; else bound -= act[i].maxc; }
        act.resize(w);
        if (bound < limit) {                                // nothing to count before the next read arrives
            RecView r2{};
            if (next < kept.size()) peek(r2);
            if (next >= kept.size() || r2.tid != tid) { act.clear(); bound = 0; }
            else if (r2.pos > pos) {
                pos = r2.pos;
                w = 0;
                for (size_t i = 0; i < act.size(); ++i) { if (act[i].end > pos) act[w++] = act[i]; else bound -= act[i].maxc; }
                act.resize(w);
            }
        }
    }
}

struct NameKey {
    const uint8_t *p; uint32_t n;
    bool operator==(const NameKey &o) const { return n == o.n && memcmp(p, o.p, n) == 0; }
};
struct NameHash {
    size_t operator()(const NameKey &k) const { uint64_t h = 1469598103934665603ull; for (uint32_t i = 0; i < k.n; ++i) { h ^= k.p[i]; h *= 1099511628211ull; } return (size_t)h; }
};

// Pass 1 of packing a sample: the read-level filters of both tools and everything that edits base qualities BEFORE the
// pileup is counted (the overlapping-mate tweak, the token limit), on a private copy of the record stream (`patched`,
// left empty when nothing was edited).  kept: the reads pass 2 packs, in file order.
static int filter_and_edit(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, msnv_sample_stats &st,
                           std::vector<KeptRead> &kept, std::vector<uint8_t> &patched, bool *cut_marks = nullptr, uint32_t *n_records = nullptr) {
    const msnv_params &P = ds.params;
    if (cut_marks) *cut_marks = false;
    const int n_contigs = (int)ds.names.size();
    uint64_t off = 0;
    int32_t last_tid = -1, last_pos = -1;
    // depth cap (mpileup -d): live pileup reads of this sample, by reference end
    typedef std::pair<int64_t, uint32_t> LiveRead;                  // {reference end, upper bound of its pileup element's characters}
    std::priority_queue<LiveRead, std::vector<LiveRead>, std::greater<LiveRead>> live;
    uint64_t live_chars = 0;                                        // upper bound of this sample's base-string length at the current position
    bool token_limit_in_reach = false;
    int32_t cap_tid = -1, cap_pos = -1; int nth_at_pos = 0; bool first_push_done = false;
    auto mutable_qual = [&](const RecView &r) -> uint8_t * {
        if (patched.empty()) patched.assign(rec, rec + n_bytes);
        return patched.data() + (r.qual - rec);
    };
    // qname -> the mate that was pushed first and still waits for its partner (htslib's overlap hash).  An entry is
    // visible while its read is alive in the pileup (end > start of the read being pushed); htslib drops it a few pushes
    // later, which can only matter for templates with three or more alignments in the file.
    struct Waiting { uint64_t off; int32_t tid; int64_t end; };
    std::unordered_map<NameKey, Waiting, NameHash> waiting;
    size_t waiting_sweep_at = 8192;

    uint32_t rec_idx = 0xffffffffu;
    while (off < n_bytes) {
        RecView r;
        if (!rec_parse(rec + off, n_bytes - off, r)) return fail(MSNV_EFORMAT, "malformed BAM record at byte %llu", (unsigned long long)off);
        const uint64_t rec_off = off;
        off += r.size;
        ++rec_idx;
        bool cov_ok = false;
        if (!read_stats(r, P.cov_min_mapq, st, cov_ok)) continue;                // unmapped (qaCompute.cpp:461-473)
        if (r.tid >= n_contigs) return fail(MSNV_EFORMAT, "record refers to contig %d but the header has %d", r.tid, n_contigs);
        if (r.tid < last_tid || (r.tid == last_tid && r.pos < last_pos)) return fail(MSNV_EFORMAT, "BAM is not coordinate sorted");
        last_tid = r.tid; last_pos = r.pos;
        if (!ds.sel[(size_t)r.tid]) continue;     // not this shard's contig

        // ---- CIGAR geometry
        int64_t rlen = 0, qlen = 0;
        bool has_ref_op = false;
        for (int k = 0; k < r.n_cigar; ++k) {
            uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u, l = c >> 4;
            if (consumes_ref(t)) { rlen += l; has_ref_op = true; }
            if (consumes_query(t)) qlen += l;
        }
        const int64_t endpos = r.pos + (rlen ? rlen : 1);                        // bam_endpos

        // ---- mpileup read-level filters (bam_plcmd.c mplp_func order)
        bool pile_ok = !(r.flag & P.flag_filter);
        if (pile_ok && ds.has_bed) pile_ok = ds.bed_beg[(size_t)r.tid] < endpos && r.pos < ds.bed_end[(size_t)r.tid];
        if (pile_ok && ds.has_seq[(size_t)r.tid] && (int64_t)ds.seqs[(size_t)r.tid].size() <= r.pos) pile_ok = false;
        if (pile_ok && r.mapq < P.min_mapq) pile_ok = false;
        if (pile_ok && !P.count_orphans && (r.flag & BAM_FPAIRED) && !(r.flag & BAM_FPROPER_PAIR)) pile_ok = false;
        if (pile_ok && !has_ref_op) pile_ok = false;
        if (pile_ok && r.l_seq > 0 && qlen != r.l_seq)
            return fail(MSNV_EFORMAT, "CIGAR consumes %lld query bases but the read has %d", (long long)qlen, r.l_seq);
        if (pile_ok) {
            // depth cap, sam.c bam_plp_push (sample-local restatement)
            while (!live.empty() && live.top().first <= r.pos) { live_chars -= live.top().second; live.pop(); }
            if (cap_tid != r.tid) { while (!live.empty()) live.pop(); live_chars = 0; }
            if (cap_tid != r.tid || cap_pos != r.pos) { cap_tid = r.tid; cap_pos = r.pos; nth_at_pos = 0; }
            bool capped = (nth_at_pos > 0 || !first_push_done) && P.max_depth > 0 && (int64_t)live.size() > (int64_t)P.max_depth;
            first_push_done = true; ++nth_at_pos;
            if (capped) {
                pile_ok = false;
                if (!P.ignore_overlaps) waiting.erase(NameKey{r.qname, r.l_name});     // overlap_remove of a capped read
            } else {
                const uint32_t mc = max_element_chars(r);
                live.push(LiveRead{endpos, mc});
                live_chars += mc;
                if (P.token_limit > 0 && live_chars >= (uint64_t)P.token_limit) token_limit_in_reach = true;
            }
        }
        const uint16_t depth_here = (uint16_t)std::min<size_t>(live.size(), 0xffff);
        if (pile_ok && !P.ignore_overlaps && !(r.flag & BAM_FMUNMAP) && (r.flag & BAM_FPROPER_PAIR) &&
            !((r.mtid >= 0 && r.tid != r.mtid) || (std::llabs((long long)r.tlen) >= 2ll * r.l_seq && r.mpos >= endpos))) {
            // sam.c overlap_push: the mate waiting under this name gets its overlap with this read edited; otherwise this
            // read waits for its mate when that one is still to come (mate position at or after this one, or unknown)
            const NameKey key{r.qname, r.l_name};
            auto it = waiting.find(key);
            if (it != waiting.end() && (it->second.tid != r.tid || it->second.end <= r.pos)) { waiting.erase(it); it = waiting.end(); }
            if (it != waiting.end()) {
                RecView a;
                rec_parse(rec + it->second.off, n_bytes - it->second.off, a);
                uint8_t *qa = mutable_qual(a), *qb = mutable_qual(r);
                tweak_overlapping_mates(a, qa, r, qb);
                waiting.erase(it);
            } else if (r.mpos >= r.pos || ((r.flag & BAM_FPAIRED) && r.mpos == -1)) {
                waiting.emplace(key, Waiting{rec_off, r.tid, endpos});
                if (waiting.size() >= waiting_sweep_at) {                      // drop the entries no later read can see
                    for (auto w = waiting.begin(); w != waiting.end();) w = (w->second.tid != r.tid || w->second.end <= r.pos) ? waiting.erase(w) : std::next(w);
                    waiting_sweep_at = std::max<size_t>(8192, 2 * waiting.size());
                }
            }
        }
        if (!pile_ok && !cov_ok) continue;
        kept.push_back(KeptRead{rec_off, endpos, depth_here, pile_ok, cov_ok, rec_idx});
    }
    if (n_records) *n_records = rec_idx + 1u;
    if (token_limit_in_reach) { apply_token_limit(P, rec, n_bytes, kept, mutable_qual); if (cut_marks) *cut_marks = true; }
    return MSNV_OK;
}

// The record stream with the base qualities as the pileup engine sees them (msnv.h: msnv_dataset_pileup_qualities).
int pileup_qualities(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, uint8_t *out) {
    msnv_sample_stats st{};
    std::vector<KeptRead> kept;
    std::vector<uint8_t> patched;
    bool cut_marks = false;
    if (int rc = filter_and_edit(ds, rec, n_bytes, st, kept, patched, &cut_marks)) return rc;
    if (n_bytes) memcpy(out, patched.empty() ? rec : patched.data(), n_bytes);
    if (cut_marks) {                              // the documented form of a cut base is quality 0 (msnv.h)
        for (const KeptRead &kr : kept) {
            RecView r;
            rec_parse(rec + kr.off, n_bytes - kr.off, r);
            uint8_t *q = out + (r.qual - rec);
            for (int32_t j = 0; j < r.l_seq; ++j) if (q[j] == QUAL_CUT) q[j] = 0;
        }
    }
    return MSNV_OK;
}

// The sequential edits for the device pack (devpack.hip): verdict per record + edited qualities.
int host_prepass(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, std::vector<uint32_t> &ovr, std::vector<uint8_t> &patched, bool &cut_marks) {
    HostTimerScope ts(HT_PACK);
    msnv_sample_stats st{};
    std::vector<KeptRead> kept;
    uint32_t n_records = 0;
    cut_marks = false;
    if (int rc = filter_and_edit(ds, rec, n_bytes, st, kept, patched, &cut_marks, &n_records)) return rc;
    ovr.assign(n_records, 1u);
    for (const KeptRead &kr : kept) ovr[kr.idx] = 1u | (kr.pile_ok ? 2u : 0u) | (kr.cov_ok ? 4u : 0u) | (uint32_t)kr.depth_here << 16;
    return MSNV_OK;
}

// Packs one sample.  `ds` supplies contig selection, BED and parameters.
int pack_sample(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, SampleCols &sc) {
    HostTimerScope ts(HT_PACK);
    const int n_contigs = (int)ds.names.size();
    std::vector<KeptRead> kept;
    std::vector<uint8_t> patched;                 // copy of the record stream with edited qualities (empty: nothing was edited)
    bool cut_marks = false;                       // the stream carries QUAL_CUT marks (snpCall's token limit was in reach)
    if (int rc = filter_and_edit(ds, rec, n_bytes, sc.st, kept, patched, &cut_marks)) return rc;
    const uint8_t *qual_base = patched.empty() ? rec : patched.data();          // qualities as the pileup sees them

    for (const KeptRead &kr : kept) {
        RecView r;
        rec_parse(rec + kr.off, n_bytes - kr.off, r);
        const bool pile_ok = kr.pile_ok, cov_ok = kr.cov_ok;
        const int64_t endpos = kr.endpos;
        const uint16_t depth_here = kr.depth_here;
        const uint8_t *r_qual = qual_base + (r.qual - rec);
        int64_t m_bases = 0;
        for (int k = 0; k < r.n_cigar; ++k) {
            const uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u;
            if (t == C_M || t == C_EQ || t == C_X) m_bases += c >> 4;
        }

        if (cov_ok) {
            // qaCompute's M intervals in its own index space (qaCompute.cpp:530-552): index = pos + 1, a leading
            // S/H op is skipped without advancing, every other non-M op advances the cursor
            int64_t pp = (int64_t)r.pos + 1;
            int k = 0;
            if (r.n_cigar > 0) { const uint32_t t = ld_u32(r.cigar) & 15u; if (t == C_S || t == C_H) k = 1; }
            for (; k < r.n_cigar; ++k) {
                const uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u, l = c >> 4;
                if (t == C_M && ds.sel[(size_t)r.tid]) {
                    // qaCompute.cpp:542-549: `++entireChr[pp]`, then `--entireChr[min(pp + l, chrSize - 1)]`.  With pp >= chrSize
                    // (a read whose cursor -- pos + 1, advanced by EVERY earlier op -- reaches the contig end; pp == chrSize is still
                    // inside the chrSize + 1 slots, beyond that the reference writes out of bounds) only the decrement at
                    // chrSize - 1 lands in the scanned range, and that position can never be covered (every read that reaches it is
                    // clamped onto it), so its coverage goes to -1 and the reference increments coverageHist[-1]: undefined there.
                    // One such read must not take a whole metaSNV run down: the library warns once per sample and computes what
                    // the reference's arithmetic says short of the out-of-bounds write -- covSum takes the -1, the position lands in
                    // no histogram bin.  Stored as the interval {L, L - 1} (begin > end) = "-1 at L - 1".
                    const int64_t L = ds.lengths[(size_t)r.tid];
                    if (pp >= L) {
                        if (!sc.warned_beyond_end) {
                            sc.warned_beyond_end = true;
                            fprintf(stderr, "msnv: warning: read at %s:%d reaches the contig end in qaCompute's index space (undefined behaviour in the reference: coverageHist[-1]); "
                                            "the last position of the contig is left out of the coverage histogram\n", ds.names[(size_t)r.tid].c_str(), r.pos + 1);
                        }
                        if (L >= 1) { sc.cov_tid.push_back(r.tid); sc.cov_beg.push_back((int32_t)L); sc.cov_end.push_back((int32_t)(L - 1)); }
                    } else { sc.cov_tid.push_back(r.tid); sc.cov_beg.push_back((int32_t)pp); sc.cov_end.push_back((int32_t)(pp + l)); }
                }
                pp += l;
            }
        }
        if (!pile_ok) continue;

        // ---- one 16-byte header per M/=/X segment piece of at most SEG_MAX bases; only aligned bases are shipped
        sc.n_pileup_bases += (uint64_t)m_bases;
        sc.n_pileup_reads++;
        // algorithmic bytes of a read = 16 B header + 4 B per CIGAR op + 4-bit bases + 1 B qualities
        // (only the aligned bases are counted; clipped / inserted bases are not shipped)
        sc.alg_8d_bytes += 16u + 4u * r.n_cigar + ((uint64_t)m_bases + 1) / 2 + (uint64_t)m_bases;
        sc.alg_cigar_bytes += 4u * r.n_cigar;
        if (sc.first_tid < 0) {
            // first pileup line of this sample (call_vC.cpp:423 drops the first line of the run)
            int64_t b = r.pos, e = endpos;
            if (ds.has_bed) { b = std::max(b, ds.bed_beg[(size_t)r.tid]); e = std::min(e, ds.bed_end[(size_t)r.tid]); }
            if (b < e) { sc.first_tid = r.tid; sc.first_beg = (int32_t)b; sc.first_end = (int32_t)e; }
        }
        // per contig: the first pileup line of an invocation that starts at this contig -- without -l, and with metaSNV's
        // split BED `name 1 LEN` (0-based [1, LEN): position 0 is excluded, metaSNV.py:92).  The driver derives the dropped
        // first line of every best_split_K file from these when all splits are written from one resident dataset.
        if (sc.first_any.empty()) { sc.first_any.assign((size_t)n_contigs, -1); sc.first_from1.assign((size_t)n_contigs, -1); }
        if (sc.first_any[(size_t)r.tid] < 0) sc.first_any[(size_t)r.tid] = r.pos;
        if (sc.first_from1[(size_t)r.tid] < 0 && endpos > 1) sc.first_from1[(size_t)r.tid] = std::max<int32_t>(r.pos, 1);
        // SEQ '*': every base prints as 'N' with quality 0 (bam_plcmd.c pileup_seq [EXT]) -- below every cutoff but -Q 0, and with -Q 0 an N
        // over a reference N is a match ('.' / ','): only then the read's pieces are shipped, as N bases of quality 0
        const bool noseq = r.l_seq == 0;
        if (noseq && ds.params.min_baseq > 0) continue;
        const std::string &refseq = ds.seqs[(size_t)r.tid];
        const bool has_ref = ds.has_seq[(size_t)r.tid];
        int64_t rp = r.pos; int64_t q = 0;
        for (int k = 0; k < r.n_cigar; ++k) {
            const uint32_t c = ld_u32(r.cigar + 4 * k), t = c & 15u, l = c >> 4;
            if (t == C_M || t == C_EQ || t == C_X) {
                for (uint32_t off = 0, n = 0; off < l; off += n) {
                    // a piece never crosses a tile boundary (contigs start on tile boundaries), so the device
                    // needs no clipping and every piece belongs to exactly one (tile, sample) pair
                    n = std::min<uint32_t>(std::min<uint32_t>(SEG_MAX, l - off), TILE - (uint32_t)((rp + off) % TILE));
                    if (sc.seq.size() > 0xffffff00ull) return fail(MSNV_EDOMAIN, "one sample holds more than 8.5 G aligned bases in this shard: shard the contigs further");
                    ReadHdr h;
                    h.gpos = (uint32_t)(rp + off);        // contig-relative until finalize
                    h.seqoff = (uint32_t)sc.seq.size();
                    h.cig = n;                            // segment length in bases
                    h.meta = META_PILEUP_OK | (uint32_t)r.mapq << 16;
                    sc.seq.resize(sc.seq.size() + (n + 1) / 2, 0xff);        // pad nibble = N
                    sc.alg_seq_bytes += (n + 1) / 2; sc.alg_qual_bytes += n;    // algorithmic bytes exclude the alignment padding
                    uint8_t *dst = sc.seq.data() + h.seqoff;
                    // BAM packs base 2i in the HIGH nibble; the device wants base j of the piece in nibble j, low first.
                    // Whole bytes at a time; a '=' base (code 0, rare) sends the piece through the per-base path below.
                    const int64_t q0 = q + off;
                    const uint8_t *src = r.seq + (q0 >> 1);
                    bool has_eq = false;
                    if (noseq) {
                        // (the bytes are N already)
                    } else if (!(q0 & 1)) {
                        for (uint32_t i = 0; i < n / 2; ++i) { const uint8_t b = src[i]; has_eq |= !(b & 0xf0u) || !(b & 0x0fu); dst[i] = (uint8_t)(b >> 4 | b << 4); }
                        if (n & 1u) { const uint8_t b = (uint8_t)(src[n / 2] >> 4); has_eq |= !b; dst[n / 2] = (uint8_t)(0xf0u | b); }
                    } else {
                        for (uint32_t i = 0; i < n / 2; ++i) { const uint8_t lo = src[i] & 0x0fu, hi = src[i + 1] & 0xf0u; has_eq |= !lo || !hi; dst[i] = (uint8_t)(lo | hi); }
                        if (n & 1u) { const uint8_t lo = src[n / 2] & 0x0fu; has_eq |= !lo; dst[n / 2] = (uint8_t)(0xf0u | lo); }
                    }
                    if (has_eq) {
                        for (uint32_t j = 0; j < n; ++j) {
                            const int64_t qq = q0 + j;
                            uint32_t code = (r.seq[qq >> 1] >> ((~qq & 1) << 2)) & 0xfu;
                            if (code == 0) {      // '=' always counts as a match (pileup_seq): ship the reference code instead
                                const int64_t g = rp + off + j;
                                code = (has_ref && g >= 0 && (size_t)g < refseq.size()) ? nt16_of_char((unsigned char)refseq[(size_t)g]) : 15u;
                                if (code == 0) code = 15u;
                            }
                            const int sh = (int)(j & 1u) * 4;
                            dst[j >> 1] = (uint8_t)((dst[j >> 1] & ~(0xf << sh)) | code << sh);  // low nibble first
                        }
                    }
                    if (has_ref && (sc.hdr.size() & 15u) == 0u) {      // one piece in 16: how noisy are these reads? (finalize_dataset: dense allele planes)
                        uint32_t mm = 0;
                        for (uint32_t j = 0; j < n; ++j) {
                            const int64_t g = rp + off + j;
                            if (g < 0 || (size_t)g >= refseq.size()) break;
                            mm += ((dst[j >> 1] >> (4u * (j & 1u))) & 0xfu) != nt16_of_char((unsigned char)refseq[(size_t)g]);
                        }
                        sc.mm_sampled_bases += n; sc.mm_sampled += mm;
                    }
                    {   // qualities above 127 (0xff = "not stored") pass every cutoff; clamping keeps the comparison
                        const size_t qs = sc.qual.size();
                        sc.qual.resize(qs + n);
                        uint8_t *qd = sc.qual.data() + qs;
                        const uint8_t *qsrc = r_qual + q0;
                        // (0x80: a base behind the token limit -- below every cutoff, pack_lowq)
                        if (noseq) for (uint32_t j = 0; j < n; ++j) qd[j] = 0;
                        else for (uint32_t j = 0; j < n; ++j) qd[j] = (cut_marks && qsrc[j] == QUAL_CUT) ? 0x80 : qsrc[j] > 127 ? 127 : qsrc[j];
                    }
                    // every piece starts on an 8-byte (seq) / 16-byte (qual) boundary
                    while (sc.seq.size() & (seq_align - 1u)) sc.seq.push_back(0xff);
                    while (sc.qual.size() < 2 * sc.seq.size()) sc.qual.push_back(0);
                    sc.hdr.push_back(h);
                    sc.tid.push_back(r.tid);
                    sc.depth.push_back(depth_here);
                    sc.end.push_back((int32_t)(rp + off + n));
                }
                rp += l; q += l;
            } else {
                if (consumes_ref(t)) rp += l;
                if (consumes_query(t)) q += l;
            }
        }
    }
    // ---- group the pieces by tile (stable: read order inside a tile); reads are sorted by start, so this
    // only moves the few pieces of reads that run into the next tile
    {
        const size_t n = sc.hdr.size();
        bool sorted = true;
        for (size_t i = 1; i < n && sorted; ++i)
            sorted = sc.tid[i - 1] < sc.tid[i] || (sc.tid[i - 1] == sc.tid[i] && sc.hdr[i - 1].gpos / TILE <= sc.hdr[i].gpos / TILE);
        if (!sorted) {
            std::vector<uint32_t> idx(n);
            for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
                if (sc.tid[x] != sc.tid[y]) return sc.tid[x] < sc.tid[y];
                return sc.hdr[x].gpos / TILE < sc.hdr[y].gpos / TILE;
            });
            std::vector<ReadHdr> h2(n); std::vector<int32_t> t2(n), e2(n); std::vector<uint16_t> d2(n);
            for (size_t i = 0; i < n; ++i) { h2[i] = sc.hdr[idx[i]]; t2[i] = sc.tid[idx[i]]; e2[i] = sc.end[idx[i]]; d2[i] = sc.depth[idx[i]]; }
            sc.hdr.swap(h2); sc.tid.swap(t2); sc.end.swap(e2); sc.depth.swap(d2);
        }
    }
    // tail padding: kernels read 16 B (qual) / 8 B (seq) chunks and may run past the last read
    for (int i = 0; i < 32; ++i) sc.seq.push_back(0xff);
    while (sc.qual.size() < 2 * sc.seq.size()) sc.qual.push_back(0);
    return MSNV_OK;
}

// Index loops of finalize that run per sample (16 M pieces on the benchmark shape, 6e8 at BASELINE configs[2] scale): dealt to host threads.
template <typename F>
static void parallel_for(size_t n, F fn) {
    const unsigned hw = msnv_default_threads();
    const size_t nt = std::min<size_t>(std::min<size_t>(n, hw), 64);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t) th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; fn(i); } });
    for (auto &t : th) t.join();
}

// The index tables of a dataset go up TOGETHER: finalize_dataset builds some thirty small tables (pairs, work items, tile tables, gate
// descriptors ...); an allocation and a synchronous copy each was a millisecond of finalize.  add() copies a table into one staging block,
// commit() makes one allocation and one copy of everything added so far and sets the tables' device pointers (so nothing on the device may
// use a table between its add() and the next commit()).  The block is owned by DeviceCols::blocks.  With guarded allocations
// (MSNV_GUARD_ALLOC=1) every table keeps an allocation of its own, so that a read past its end still faults.
struct TableArena {
    struct Ent { void **dst; size_t off, bytes; };
    std::vector<Ent> ents;
    std::vector<uint8_t> stage;
    template <typename T>
    int add(T **dst, const std::vector<T> &v, uint64_t *, size_t pad_elems = 0) {
        const size_t off = (stage.size() + 255) & ~(size_t)255, bytes = (v.size() + pad_elems) * sizeof(T);
        stage.resize(off + std::max<size_t>(bytes, 16), 0);
        if (!v.empty()) memcpy(stage.data() + off, v.data(), v.size() * sizeof(T));
        ents.push_back(Ent{reinterpret_cast<void **>(dst), off, std::max<size_t>(bytes, 16)});
        return MSNV_OK;
    }
    int commit(DeviceCols &d) {
        if (ents.empty()) return MSNV_OK;
        static const bool guard = [] { const char *e = getenv("MSNV_GUARD_ALLOC"); return e && e[0] == '1'; }();
        if (guard) {
            for (const Ent &e : ents) {
                if (int rc = dev_alloc(e.dst, e.bytes, &d.device_bytes)) return rc;
                if (int rc = dev_upload(*e.dst, stage.data() + e.off, e.bytes)) return rc;
            }
        } else {
            void *block = nullptr;
            if (int rc = dev_alloc(&block, stage.size() + 256, &d.device_bytes)) return rc;
            d.blocks.emplace_back(block, stage.size() + 256);
            if (int rc = dev_upload(block, stage.data(), stage.size())) return rc;
            for (const Ent &e : ents) *e.dst = static_cast<uint8_t *>(block) + e.off;
        }
        ents.clear(); stage.clear();
        return MSNV_OK;
    }
};

int finalize_dataset(msnv_dataset &ds) {
    const size_t S = ds.samples.size();
    const size_t NC = ds.names.size();
    if (S == 0) return fail(MSNV_EINVAL, "dataset has no samples");
    if (S >= 16384) return fail(MSNV_EDOMAIN, "more than 16383 samples per dataset are not supported");
    // MSNV_FINALIZE_TRACE=1: wall seconds of every stage to stderr (where a big dataset's finalize goes)
    fin_trace_reset();
    auto lap = [&](const char *what) { fin_trace(what); };
    // Device-packed samples (devpack.hip) keep their piece headers and intervals in HBM.  When every sample is one, the per-piece and
    // per-interval loops below run there as kernels (`fast`; devfin_* in devpack.hip) and the host works on (sample, tile) pairs only.  The
    // dense piece re-layout (short reads) still runs on host staging, and so do mixed datasets: the headers come down first
    // (MSNV_FINALIZE=host forces that; tests compare the two).
    bool fast = false;
    {
        bool any_dev = false, all_dev = true;
        for (const SampleCols &sc : ds.samples) { any_dev |= sc.dev_index; all_dev &= sc.dev_index; }
        fast = any_dev && all_dev && HDR4 && deep_runs_split();
        if (const char *e = getenv("MSNV_FINALIZE")) if (e[0] == 'h') fast = false;
        if (fast) {
            uint64_t np = 0, nb = 0;
            for (const SampleCols &sc : ds.samples) { np += sc.n_dev_pieces; nb += sc.n_pileup_bases; }
            if (layout_dense(np, nb)) if (const char *e = getenv("MSNV_DENSE_RELAYOUT")) if (e[0] == 'h') fast = false;      // (the dense re-layout on host staging: tests compare the two)
            // deep (sample, tile) runs: their exact depth, and -- where a run really is that deep -- its pieces dealt into groups and the sample's
            // columns re-laid, by kernels (devpack.hip: devfin_deep_runs; MSNV_DEEP_RELOCATE=0, a host-only experiment, takes the host loops)
            const uint32_t split_at = deep_split_at();
            bool any_deep = false;
            for (const SampleCols &sc : ds.samples) for (const DevPair &p : sc.dev_pairs) if (p.maxd >= split_at) { any_deep = true; break; }
            if (fast && any_deep) {
                if (getenv("MSNV_DEEP_RELOCATE") && getenv("MSNV_DEEP_RELOCATE")[0] == '0') fast = false;
                else {
                    bool fallback = false;
                    if (int rc = devfin_deep_runs(ds, split_at, deep_group_depth(), &fallback)) return rc;
                    if (fallback) fast = false;
                }
            }
        }
        if (any_dev && !fast) { if (int rc = devpack_sync_pending(ds)) return rc; if (int rc = devpack_download_pieces(ds)) return rc; }
    }

    lap("decide / download");
    // ---- tile layout: every selected contig owns ceil(max(L, furthest read end) / TILE) tiles
    std::vector<int64_t> maxend(NC, 0);
    for (size_t c = 0; c < NC; ++c) maxend[c] = ds.sel[c] ? ds.lengths[c] : 0;
    if (!fast) {       // (fast: the pieces are in HBM -- devfin_overhang below; qaCompute's intervals never reach beyond a contig's length)
        // per sample: the contigs it touches and how far (reads are sorted by contig: runs), merged under a lock
        std::mutex mu;
        parallel_for(S, [&](size_t s) {
            const SampleCols &sc = ds.samples[s];
            std::vector<std::pair<int32_t, int64_t>> mine;
            for (size_t i = 0; i < sc.hdr.size(); ++i) {
                if (mine.empty() || mine.back().first != sc.tid[i]) mine.emplace_back(sc.tid[i], 0);
                mine.back().second = std::max<int64_t>(mine.back().second, sc.end[i]);
            }
            for (size_t i = 0; i < sc.cov_tid.size(); ++i) {
                if (!ds.sel[(size_t)sc.cov_tid[i]]) continue;
                if (mine.empty() || mine.back().first != sc.cov_tid[i]) mine.emplace_back(sc.cov_tid[i], 0);
                mine.back().second = std::max<int64_t>(mine.back().second, std::min<int64_t>((int64_t)sc.cov_beg[i] + 1, ds.lengths[(size_t)sc.cov_tid[i]]));
            }
            std::lock_guard<std::mutex> lk(mu);
            for (const auto &m : mine) maxend[(size_t)m.first] = std::max(maxend[(size_t)m.first], m.second);
        });
    }
    lap("  maxend loops");
    if (fast) if (int rc = devfin_overhang(ds, maxend)) return rc;       // (reads that run past their contig: from the device's per-contig maxima)
    lap("  devfin_overhang");
    ds.tile_base.assign(NC, UINT32_MAX);
    ds.tile_contig.clear();
    uint64_t nt = 0;
    for (size_t c = 0; c < NC; ++c) {
        if (!ds.sel[c]) continue;
        ds.tile_base[c] = (uint32_t)nt;
        uint64_t n = ((uint64_t)maxend[c] + TILE - 1) / TILE;
        for (uint64_t k = 0; k < n; ++k) ds.tile_contig.push_back((uint32_t)c);
        nt += n;
        if (nt * TILE >= 0xffffffffull) return fail(MSNV_EDOMAIN, "shard spans more than 2^32 positions: shard the contigs further");
    }
    ds.n_tiles = (uint32_t)nt;
    const uint64_t npos = nt * TILE;

    lap("  tile tables");
    DeviceCols *d = new DeviceCols();
    ds.dev = d;
    TableArena arena;
    d->n_tiles = ds.n_tiles; d->n_samples = (uint32_t)S;

    if (fast) if (int rc = devfin_coverage_launch(ds, *d)) return rc;     // (the coverage index's kernels go first: they run while the host builds the pair tables)
    // ---- genome coverage index: intervals of qaCompute's difference array, grouped by tile.  The host's share -- pair tables by tile, accumulator
    // rows, work items (0.5 ms of loops on the benchmark shape) -- needs the index's kernels' results and nothing else of finalize.  On a thread
    // of its own beside the tile index (MSNV_COV_THREAD=1) it made finalize SLOWER (median of twelve builds 2.9-3.3 ms against 2.3: the thread's
    // start and wake-up cost more than the loops; profiles/r06_variants.txt): it runs behind the wait for those kernels, on this thread.
    uint64_t cov_bytes = 0;
    auto cov_index = [&](TableArena &A, bool in_thread) -> int {
        std::vector<Pair32> iv;
        std::vector<uint64_t> cvbase(S + 1, 0);
        struct CP { uint32_t tile, sample, lo, hi; };
        std::vector<CP> flat;                                        // all samples' (tile, sample) runs, sample after sample, tiles ascending inside a sample
        bool tables_on_device = false;                               // (round 6: devfin_coverage has cut the pair tables, the rows and the work items in HBM)
        if (fast) {
            // the intervals are in HBM: filtered, made linear and grouped by (sample, tile) there (devpack.hip: devfin_coverage); the runs come
            // back sample-major with the tiles ascending, which is the order the loops below want
            std::vector<DevCovPair> cp;
            if (int rc = devfin_coverage(ds, *d, cvbase, cp)) return rc;
            if (!in_thread) lap("  devfin_coverage");
            tables_on_device = ds.dp.cov_tables_done;
            flat.resize(cp.size());
            bool ordered = true;
            for (size_t i = 0; i < cp.size(); ++i) {
                flat[i] = CP{cp[i].tile, cp[i].sample, cp[i].lo, cp[i].hi};
                if (i && (cp[i].sample < cp[i - 1].sample || (cp[i].sample == cp[i - 1].sample && cp[i].tile < cp[i - 1].tile))) ordered = false;
            }
            if (!ordered) std::stable_sort(flat.begin(), flat.end(), [](const CP &a, const CP &b) { return a.sample != b.sample ? a.sample < b.sample : a.tile < b.tile; });
        }
        else {
        std::vector<std::vector<CP>> per(S);
        for (size_t s = 0; s < S; ++s) {
            const SampleCols &sc = ds.samples[s];
            uint32_t n_here = 0;
            for (size_t i = 0; i < sc.cov_tid.size(); ++i) {
                const size_t c = (size_t)sc.cov_tid[i];
                const int64_t L = ds.lengths[c];
                const int64_t b = sc.cov_beg[i];
                const bool minus_one = b > sc.cov_end[i];                          // {L, L - 1}: "-1 at L - 1" (pack_sample)
                const int64_t e = minus_one ? sc.cov_end[i] : (sc.cov_end[i] >= L ? L - 1 : sc.cov_end[i]);       // qaCompute.cpp:544-549
                if (!minus_one && b >= e) continue;
                const uint64_t g0 = (uint64_t)ds.tile_base[c] * TILE;
                const uint32_t idx = n_here++;
                iv.push_back(Pair32{(uint32_t)(g0 + (uint64_t)b), (uint32_t)(g0 + (uint64_t)e)});
                std::vector<CP> &pv = per[s];
                const uint32_t t_first = (uint32_t)((g0 + (uint64_t)(minus_one ? e : b)) / TILE), t_last = minus_one ? t_first : (uint32_t)((g0 + e - 1) / TILE);
                for (uint32_t t = t_first; t <= t_last; ++t) {
                    size_t k = pv.size();
                    while (k > 0 && pv[k - 1].tile > t) --k;
                    if (k > 0 && pv[k - 1].tile == t) pv[k - 1].hi = idx + 1;
                    else pv.insert(pv.begin() + (ptrdiff_t)k, CP{t, (uint32_t)s, idx, idx + 1});
                }
            }
            cvbase[s + 1] = cvbase[s] + n_here;
        }
        for (size_t s = 0; s < S; ++s) flat.insert(flat.end(), per[s].begin(), per[s].end());
        }
        std::vector<TilePair> cpairs;
        std::vector<WorkItem> cwork;
        if (!tables_on_device) {
        // one pass for the tile counts, one for the pairs (sample order inside a tile comes with the order of `flat`) and the accumulator rows
        std::vector<uint32_t> cps(nt + 1, 0);
        for (const CP &p : flat) cps[p.tile + 1]++;
        for (uint64_t t = 0; t < nt; ++t) cps[t + 1] += cps[t];
        cpairs.resize(cps[nt]);
        {
            std::vector<uint32_t> fill(cps.begin(), cps.end() - 1);
            // blk_lo / nblk: absolute index of the sample's first interval (the kernel needs no per-sample base lookup)
            // max_depth: the accumulator row of the pair's (sample, contig) -- rows exist for the combinations that have intervals only
            ds.cov_row_sample.clear(); ds.cov_row_contig.clear(); ds.cov_row_start.assign(S + 1, 0);
            size_t i = 0;
            for (size_t s = 0; s < S; ++s) {
                ds.cov_row_start[s] = ds.cov_row_sample.size();
                uint32_t last_contig = UINT32_MAX;
                for (; i < flat.size() && flat[i].sample == s; ++i) {                     // (tile order = contig order)
                    const CP &p = flat[i];
                    const uint32_t c = ds.tile_contig[p.tile];
                    if (c != last_contig) { ds.cov_row_sample.push_back((uint32_t)s); ds.cov_row_contig.push_back(c); last_contig = c; }
                    if (ds.cov_row_sample.size() > 0xffffffffull) return fail(MSNV_EDOMAIN, "more than 2^32 (sample, contig) pairs with coverage in one shard");
                    cpairs[fill[p.tile]++] = TilePair{p.sample, p.lo, p.hi, (uint32_t)(ds.cov_row_sample.size() - 1), (uint32_t)cvbase[s], (uint32_t)(cvbase[s] >> 32), 0, 0};
                }
            }
            ds.cov_row_start[S] = ds.cov_row_sample.size();
        }
        if (!in_thread) lap("    cov tables: pairs by tile, rows");
        cwork.reserve(cpairs.size() / 2 + nt + 16);
        // a coverage work item = COV_ITEM_PAIRS consecutive pairs of a tile = one wavefront of msnv_coverage_tiles, which loads their
        // descriptors up front and a pair's intervals while it works on the pair before (fewer when one pair alone is deep:
        // MSNV_COV_ITEM intervals)
        const uint64_t cov_item_intervals = [] { const char *e = getenv("MSNV_COV_ITEM"); const long long v = e ? atoll(e) : 16384; return (uint64_t)(v > 0 ? v : 16384); }();
        for (uint64_t t = 0; t < nt; ++t) {
            uint32_t lo = cps[t]; uint64_t acc = 0;
            for (uint32_t k = cps[t]; k < cps[t + 1]; ++k) {
                acc += cpairs[k].read_hi - cpairs[k].read_lo;
                if (acc >= cov_item_intervals || k + 1 - lo >= COV_ITEM_PAIRS || k + 1 == cps[t + 1]) { cwork.push_back(WorkItem{(uint32_t)t, lo, k + 1, 0, 0, 0, 0, 0, ChunkDesc{}}); lo = k + 1; acc = 0; }
            }
        }
        // the items with a pair of more than 32 767 intervals go last: msnv_coverage_tiles<true> (one word per position) runs them,
        // the 16-bit difference array of the usual variant holds +-32 767 per position and per 16 positions of one parity
        // (MSNV_COV_NARROW_MAX is read per dataset: tests lower it to run the other variant)
        const uint32_t cov_narrow_max = [] { const char *e = getenv("MSNV_COV_NARROW_MAX"); const long long v = e ? atoll(e) : 32767; return (uint32_t)std::min<long long>(32767, std::max<long long>(1, v)); }();
        auto cov_wide = [&](const WorkItem &w) { for (uint32_t k = w.pair_lo; k < w.pair_hi; ++k) if (cpairs[k].read_hi - cpairs[k].read_lo > cov_narrow_max) return true; return false; };
        bool any_wide = false;
        for (const TilePair &q : cpairs) if (q.read_hi - q.read_lo > cov_narrow_max) { any_wide = true; break; }
        d->n_cov_work_wide = 0;
        if (any_wide) {
            const auto first_wide = std::stable_partition(cwork.begin(), cwork.end(), [&](const WorkItem &w) { return !cov_wide(w); });
            d->n_cov_work_wide = (uint32_t)(cwork.end() - first_wide);
        }
        if (!in_thread) lap("    cov tables: work items");
        d->n_cov_pairs = (uint32_t)cpairs.size(); d->n_cov_work = (uint32_t)cwork.size();
        }
        std::vector<uint32_t> tlen(nt + 1, 0), tcont(nt + 1, 0);
        for (uint64_t t = 0; t < nt; ++t) {
            const size_t c = ds.tile_contig[t];
            const int64_t t0 = (int64_t)(t - ds.tile_base[c]) * TILE;
            tlen[t] = (uint32_t)std::min<int64_t>(std::max<int64_t>(ds.lengths[c] - t0, 0), TILE);
            tcont[t] = (uint32_t)c;
        }
        if (!fast) d->n_cov_iv = iv.size();
        for (int k = 0; k < 4; ++k) iv.push_back(Pair32{0u, 0u});     // behind the last interval: what the idle lanes of msnv_coverage_tiles load, four at a time (they touch nothing)
        d->n_contigs = (uint32_t)NC;
        if (!fast) if (int rc = A.add(&d->cov_iv, iv, &cov_bytes, 1)) return rc;
        if (int rc = A.add(&d->s_cov_base, cvbase, &cov_bytes)) return rc;
        if (!tables_on_device) {
            if (int rc = A.add(&d->cov_pairs, cpairs, &cov_bytes, 1)) return rc;
            if (int rc = A.add(&d->cov_work, cwork, &cov_bytes, 1)) return rc;
        }
        if (int rc = A.add(&d->tile_len, tlen, &cov_bytes)) return rc;
        if (int rc = A.add(&d->tile_contig_dev, tcont, &cov_bytes)) return rc;
        return MSNV_OK;
    };
    TableArena cov_arena;
    struct CovJob {
        std::thread th; int rc = MSNV_OK; std::string msg;
        int join() { if (th.joinable()) th.join(); if (rc) return fail(rc, "%s", msg.c_str()); return MSNV_OK; }
        ~CovJob() { if (th.joinable()) th.join(); }
    } cov_job;
    if (fast && ds.dp.cov_launched && getenv("MSNV_COV_THREAD")) {
        const int device = ds.ctx ? ds.ctx->device : 0;
        cov_job.th = std::thread([&cov_job, &cov_index, &cov_arena, device]() {
            (void)dev_set_device(device);
            try { cov_job.rc = cov_index(cov_arena, true); } catch (const std::exception &e) { cov_job.rc = fail_quiet(MSNV_ENOMEM, "coverage index: %s", e.what()); }
            if (cov_job.rc) cov_job.msg = msnv_last_error();
        });
    }
    lap("tile layout");
    // ---- reference: nt16 codes (N beyond the contig end, as mpileup prints) + lower-case bits
    std::vector<uint32_t> ref4(npos / 8 + 1, 0xffffffffu);
    {
        std::vector<uint32_t> lc(npos / 32 + 1, 0u);
        // (contigs start on tile boundaries: their words of both tables are disjoint, so the contigs are dealt to the host threads --
        // a database shard is gigabases of FASTA)
        std::vector<size_t> with_seq;
        for (size_t c = 0; c < NC; ++c) if (ds.sel[c] && ds.has_seq[c]) with_seq.push_back(c);
        // the device pack has converted the selected contigs once already (devpack.hip: build_tables -- codes 8 per word, lower-case bits 32 per
        // word, per contig): a contig starts on a tile, so its words go into place as they are
        const bool converted = ds.dp.ready && !ds.dp.h_codes.empty();
        std::vector<size_t> by_char;
        for (size_t c : with_seq) {
            const uint64_t n = std::min<uint64_t>(ds.seqs[c].size(), (uint64_t)maxend[c]);
            if (!converted || n != ds.seqs[c].size() || ds.dp.h_code_off[c] == ~0ull) { by_char.push_back(c); continue; }      // (a FASTA record longer than its contig's tiles: character by character)
            const uint64_t g0 = (uint64_t)ds.tile_base[c] * TILE;
            memcpy(ref4.data() + g0 / 8, ds.dp.h_codes.data() + ds.dp.h_code_off[c], ((n + 7) / 8) * 4);
            memcpy(lc.data() + g0 / 32, ds.dp.h_lc.data() + ds.dp.h_lc_off[c], ((n + 31) / 32) * 4);
        }
        parallel_for(by_char.size(), [&](size_t k) {
            const size_t c = by_char[k];
            const std::string &s = ds.seqs[c];
            const uint64_t g0 = (uint64_t)ds.tile_base[c] * TILE;
            const uint64_t lim = std::min<uint64_t>(s.size(), (uint64_t)maxend[c]);
            for (uint64_t i = 0; i < lim; ++i) {
                const uint64_t g = g0 + i;
                const uint32_t code = nt16_of_char((unsigned char)s[i]);
                ref4[g >> 3] = (ref4[g >> 3] & ~(0xfu << (4 * (g & 7)))) | code << (4 * (g & 7));
                const char ch = s[i];
                if (ch == 'a' || ch == 'c' || ch == 'g' || ch == 't') lc[g >> 5] |= 1u << (g & 31);
            }
        });
        if (int rc = arena.add(&d->ref4, ref4, &d->device_bytes)) return rc;
        if (int rc = arena.add(&d->ref_lc, lc, &d->device_bytes)) return rc;
        ds.info.bytes_ref = npos / 2;
    }
    lap("reference");
    // ---- callable range per tile (BED -l regions; without BED every covered position)
    std::vector<uint32_t> vb_host, ve_host;
    {
        std::vector<uint32_t> vb(nt + 1, 0), ve(nt + 1, 0);
        for (uint64_t t = 0; t < nt; ++t) {
            const size_t c = ds.tile_contig[t];
            const int64_t t0 = (int64_t)(t - ds.tile_base[c]) * TILE;
            int64_t b = ds.has_bed ? ds.bed_beg[c] : 0, e = ds.has_bed ? ds.bed_end[c] : INT64_MAX;
            b = std::min<int64_t>(std::max<int64_t>(b - t0, 0), TILE);
            e = std::min<int64_t>(std::max<int64_t>(e - t0, 0), TILE);
            vb[t] = (uint32_t)b; ve[t] = (uint32_t)e;
        }
        if (int rc = arena.add(&d->tile_vbeg, vb, &d->device_bytes)) return rc;
        if (int rc = arena.add(&d->tile_vend, ve, &d->device_bytes)) return rc;
        vb_host = vb; ve_host = ve;
    }

    lap("callable ranges");
    // ---- per sample: gpos, tile overlap index; concatenate columns
    const int device_id = ds.ctx ? ds.ctx->device : 0;
    if (!fast) {
        std::atomic<int> split_err{0}; std::mutex split_mu; std::string split_msg;
        parallel_for(S, [&](size_t s) {
            if (int rc = split_deep_runs(ds.samples[s], device_id)) { std::lock_guard<std::mutex> lk(split_mu); if (!split_err.load()) { split_msg = msnv_last_error(); split_err.store(rc); } }
        });
        if (split_err.load()) return fail(split_err.load(), "%s", split_msg.c_str());
    }
    uint64_t all_pieces = 0, all_bases = 0;
    for (const SampleCols &sc : ds.samples) { all_pieces += fast ? (size_t)sc.n_dev_pieces : sc.hdr.size(); all_bases += sc.n_pileup_bases; }
    const bool dense = layout_dense(all_pieces, all_bases);
    d->dense = dense;
    if (dense && fast) {
        if (int rc = devfin_dense(ds)) return rc;                    // devpack.hip: block streams, descriptors and run tables of every sample, in HBM
    } else if (dense) {
        for (size_t s = 0; s < S; ++s) if (ds.samples[s].on_device) { if (int rc = dev_set_device(device_id)) return rc; if (int rc = devpack_sample_to_host(ds.samples[s])) return rc; }
        parallel_for(S, [&](size_t s) {
            SampleCols &sc = ds.samples[s];
            relayout_dense(sc);
            for (int i = 0; i < 32; ++i) sc.seq.push_back(0xff);     // tail padding as in pack_sample
            while (sc.qual.size() < 2 * sc.seq.size()) sc.qual.push_back(0);
        });
    }
    std::vector<uint64_t> rbase(S + 1, 0), sbase(S + 1, 0), bbase(S + 1, 0);
    for (size_t s = 0; s < S; ++s) {
        bbase[s + 1] = bbase[s] + (fast ? (size_t)ds.samples[s].n_dev_blk : ds.samples[s].blk.size());
        rbase[s + 1] = rbase[s] + (fast ? (size_t)ds.samples[s].n_dev_pieces : ds.samples[s].hdr.size());
        const size_t seq_bytes = ds.samples[s].on_device ? (size_t)ds.samples[s].d_seq_bytes : ds.samples[s].seq.size();
        sbase[s + 1] = sbase[s] + ((seq_bytes + 15) & ~(size_t)15);
    }
    struct PairTmp { uint32_t tile, sample, lo, hi, maxd, run, grp; };
    std::vector<std::vector<PairTmp>> per_sample(S);
    ds.first_tid = -1; ds.first_pos = -1;
    uint64_t tot_reads = 0, tot_pile_reads = 0, tot_bases = 0;
    if (fast) for (size_t s = 0; s < S; ++s) {                           // the runs of pieces as the device found them (devpack.hip: msnv_pair_starts / msnv_pair_maxd)
        const SampleCols &sc = ds.samples[s];
        std::vector<PairTmp> &pv = per_sample[s];
        pv.reserve(sc.dev_pairs.size());
        for (const DevPair &p : sc.dev_pairs)
            pv.push_back(PairTmp{ds.tile_base[(size_t)p.tid] + p.tile, (uint32_t)s, p.lo, p.hi, p.maxd, (uint32_t)pv.size(), p.grp});
    }
    else parallel_for(S, [&](size_t s) {
        SampleCols &sc = ds.samples[s];
        std::vector<PairTmp> &pv = per_sample[s];
        // pieces are grouped by tile: one pair per run
        for (size_t i = 0; i < sc.hdr.size(); ++i) {
            const size_t c = (size_t)sc.tid[i];
            const uint64_t gs = (uint64_t)ds.tile_base[c] * TILE + sc.hdr[i].gpos;
            sc.hdr[i].gpos = (uint32_t)gs;
            const uint32_t t = (uint32_t)(gs / TILE);
            if (!dense && !sc.on_device) {                       // (device-packed samples: devpack.hip msnv_fill_padding, once the columns are in place)
                // The alignment padding behind a piece (up to the next 16 bases) reads as the reference the kernel compares
                // it with (N beyond the tile): msnv_pileup_tiles_narrow32 then masks mismatch flags per 16 bases, not per base.
                const uint32_t len = sc.hdr[i].cig, stop = (len + 2u * seq_align - 1u) & ~(2u * seq_align - 1u);
                uint8_t *sp = sc.seq.data() + sc.hdr[i].seqoff;
                for (uint32_t j = len; j < stop; ++j) {
                    const uint64_t g = gs + j;
                    const uint32_t code = (gs % TILE + j < TILE) ? (ref4[g >> 3] >> (4 * (g & 7))) & 0xfu : 0xfu;
                    const int sh = (int)(j & 1u) * 4;
                    sp[j >> 1] = (uint8_t)((sp[j >> 1] & ~(0xf << sh)) | code << sh);
                }
            }
            if (!pv.empty() && pv.back().tile == t && pv.back().grp == sc.grp[i]) pv.back().hi = (uint32_t)i + 1;
            else pv.push_back(PairTmp{t, (uint32_t)s, (uint32_t)i, (uint32_t)i + 1, 0, (uint32_t)pv.size(), sc.grp[i]});
        }
        for (PairTmp &p : pv) {       // depth bound: every read alive inside the tile was alive when one of [lo,hi) started
            uint32_t m = 0;
            for (uint32_t i = p.lo; i < p.hi; ++i) m = std::max<uint32_t>(m, sc.depth[i]);
            p.maxd = m;
        }
    });
    for (size_t s = 0; s < S; ++s) {
        const SampleCols &sc = ds.samples[s];
        tot_reads += fast ? (size_t)sc.n_dev_pieces : sc.hdr.size(); tot_pile_reads += sc.n_pileup_reads; tot_bases += sc.n_pileup_bases;
        if (sc.first_tid >= 0 && (ds.first_tid < 0 || sc.first_tid < ds.first_tid || (sc.first_tid == ds.first_tid && sc.first_beg < ds.first_pos))) {
            ds.first_tid = sc.first_tid; ds.first_pos = sc.first_beg;
        }
    }
    lap("pairs per sample");
    // ---- CSR of pairs by tile (sample order inside a tile)
    std::vector<uint8_t> fuse_tile(nt, 0);                          // tiles piled up by ONE whole-tile work item (see below)
    std::vector<uint32_t> tps(nt + 1, 0);
    for (size_t s = 0; s < S; ++s) for (const PairTmp &p : per_sample[s]) tps[p.tile + 1]++;
    for (uint64_t t = 0; t < nt; ++t) tps[t + 1] += tps[t];
    std::vector<TilePair> pairs(tps[nt]);
    {
        std::vector<uint32_t> fill(tps.begin(), tps.end() - 1);
        for (size_t s = 0; s < S; ++s)
            for (const PairTmp &p : per_sample[s]) {
                const SampleCols &sc = ds.samples[s];
                TilePair tp{p.sample, p.lo, p.hi, p.maxd, 0, 0, 0, p.grp ? 1u : 0u};      // pad = 1: one of several pairs of this sample in the tile
                if (dense) {
                    if (p.run >= sc.run_nblk.size()) return fail(MSNV_EINVAL, "internal: dense runs and tile pairs disagree");
                    tp.blk_lo = sc.run_blk_lo[p.run]; tp.nblk = sc.run_nblk[p.run]; tp.seq0 = sc.run_seq0[p.run];
                }
                pairs[fill[p.tile]++] = tp;
            }
        // inside a tile: narrow pairs (every per-position count fits a byte) first, then wide ones, then the SHALLOW pairs that
        // are merged into groups (msnv_pileup_tiles_merged): pad = 2.  A pair of ~20 pieces costs the narrow kernel a chunk
        // iteration and a pass over all 2048 positions whatever it holds, so cohorts of many shallow samples (1600 x 1x ran at
        // 24 % of the roofline) and contigs much shorter than a tile put several pairs into the same bins.
        // MSNV_SHALLOW_PIECES: number of pieces up to which a pair counts as shallow (default 48 = 3/8 of a chunk; 0 = never
        // merge); its depth bound must leave room for at least three pairs in a group, and a tile needs two such pairs.
        const uint32_t shallow_pieces = [] { const char *e = getenv("MSNV_SHALLOW_PIECES"); const int v = e ? atoi(e) : 48; return (uint32_t)std::max(0, v); }();   // read per dataset (tests switch it)
        const bool can_merge = !dense && shallow_pieces > 0 && sbase[S] < ((uint64_t)SEQ_ALIGN << 37);   // merged headers hold absolute seq offsets / SEQ_ALIGN in 37 bits
        auto is_shallow = [&](const TilePair &p) { return p.read_hi - p.read_lo <= shallow_pieces && p.max_depth <= MERGE_MAX_DEPTH / 3 && !p.pad; };
        // ... and only when the shallow pairs are a real share of the dataset (>= 3 % of its pieces): a few of them -- the
        // partial last tile of every contig of the benchmark shape -- are not worth the second code path in the tail
        std::vector<uint8_t> merge_tile(nt, 0);
        // Whole-tile work items: in a SPARSE cohort (a pair or two per tile: BASELINE configs[3]) a tile whose pairs fit ONE merged group
        // is piled up by one workgroup, which then holds the tile's totals and applies the gates itself (kernels.hip: fused_tile_gate).
        // MSNV_FUSE=0 switches it off, MSNV_FUSE_PIECES sets the pieces a pair may hold (default 256), MSNV_FUSE=1 forces it on
        // whatever the cohort looks like (tests).
        {
            uint64_t tiles_with_pairs = 0;
            for (uint64_t t = 0; t < nt; ++t) tiles_with_pairs += tps[t + 1] > tps[t];
            const char *fe = getenv("MSNV_FUSE");
            const bool sparse = tiles_with_pairs && pairs.size() < 4 * tiles_with_pairs;
            const bool fuse_on = can_merge && !(fe && fe[0] == '0') && (sparse || (fe && fe[0] == '1'));
            const uint32_t fuse_pieces = [] { const char *e = getenv("MSNV_FUSE_PIECES"); return (uint32_t)std::max(1, e ? atoi(e) : 256); }();
            if (fuse_on) for (uint64_t t = 0; t < nt; ++t) {
                const uint32_t n = tps[t + 1] - tps[t];
                if (n == 0 || n > MERGE_MAX_PAIRS) continue;
                uint64_t depth = 0; bool ok = true;
                for (uint32_t k = tps[t]; k < tps[t + 1] && ok; ++k) { depth += pairs[k].max_depth; ok = !pairs[k].pad && pairs[k].read_hi - pairs[k].read_lo <= fuse_pieces; }
                if (ok && depth <= MERGE_MAX_DEPTH) fuse_tile[t] = 1;
            }
        }
        if (can_merge) {
            uint64_t shallow_pieces_total = 0, all_pieces = 0;
            for (uint64_t t = 0; t < nt; ++t) {
                uint32_t n_shallow = 0; uint64_t np = 0;
                for (uint32_t k = tps[t]; k < tps[t + 1]; ++k) { all_pieces += pairs[k].read_hi - pairs[k].read_lo; if (is_shallow(pairs[k])) { ++n_shallow; np += pairs[k].read_hi - pairs[k].read_lo; } }
                if (n_shallow >= 2) { merge_tile[t] = 1; shallow_pieces_total += np; }
            }
            const bool force = [] { const char *e = getenv("MSNV_MERGE_ALWAYS"); return e && e[0] == '1'; }();   // (tests)
            if (!force && shallow_pieces_total * 100 < all_pieces * 3) std::fill(merge_tile.begin(), merge_tile.end(), 0);
        }
        for (uint64_t t = 0; t < nt; ++t) {
            auto b = pairs.begin() + tps[t], e = pairs.begin() + tps[t + 1];
            if (fuse_tile[t]) for (auto it = b; it != e; ++it) it->pad = 2;
            else if (merge_tile[t]) for (auto it = b; it != e; ++it) if (is_shallow(*it)) it->pad = 2;
            auto cls = [](const TilePair &p) { return p.pad == 2 ? 2 : p.max_depth < NARROW_MAX_DEPTH ? 0 : 1; };
            bool in_class_order = true;                              // (nearly every tile's pairs are of one class: the sort's temporary buffer per tile was a third of this stage)
            for (auto it = b; it != e && it + 1 != e; ++it) if (cls(*(it + 1)) < cls(*it)) { in_class_order = false; break; }
            if (!in_class_order) std::stable_sort(b, e, [&](const TilePair &x, const TilePair &y) { return cls(x) < cls(y); });
        }
    }
    std::vector<uint32_t> tpm(nt + 1, 0);                           // per tile: first merged pair
    for (uint64_t t = 0; t < nt; ++t) { uint32_t k = tps[t]; while (k < tps[t + 1] && pairs[k].pad != 2) ++k; tpm[t] = k; }
    lap("pair CSR, merge / fuse classes");
    // ---- slots: the samples that have reads in a tile are numbered 0 .. n - 1 (the pairs of a split sample, consecutive, share
    // one); the per-sample cells of the tile's called positions are stored per slot (kernels.hip: CellMap) and the host expands
    // to all samples when it fetches.  From here on TilePair::pad = kind (0 narrow or wide, 1 split, 2 merged) | slot << 8.
    ds.tile_slot_base.assign(nt + 1, 0);
    ds.slot_sample.clear();
    {
        std::vector<uint32_t> nslots(nt + 1, 0);
        for (uint64_t t = 0; t < nt; ++t) {
            ds.tile_slot_base[t] = (uint64_t)ds.slot_sample.size();
            uint32_t n = 0;
            for (uint32_t k = tps[t]; k < tps[t + 1]; ++k) {
                const bool same = k > tps[t] && pairs[k].pad == 1 && (pairs[k - 1].pad & 0xffu) == 1 && pairs[k].sample == pairs[k - 1].sample;
                if (!same) { ds.slot_sample.push_back(pairs[k].sample); ++n; }
                pairs[k].pad |= (n - 1u) << 8;
            }
            // the device's row stride: tiles of >= 16 slots get rows that are multiples of 16 bytes (8 coverage cells), so that the
            // many-site gather can write them with 16-byte stores (kernels.hip: gather_cov_wide); the padding cells stay zero
            nslots[t] = n >= 16u ? (n + 7u) & ~7u : n;
        }
        ds.tile_slot_stride = nslots;
        ds.tile_slot_base[nt] = (uint64_t)ds.slot_sample.size();
        if (int rc = arena.add(&d->tile_nslots, nslots, &d->device_bytes)) return rc;
        if (int rc = dev_alloc((void **)&d->tile_cell_base, (nt + 1) * sizeof(unsigned long long), &d->device_bytes)) return rc;
        if (int rc = dev_memset_async(d->tile_cell_base, 0, (nt + 1) * sizeof(unsigned long long), ds.ctx ? ds.ctx->stream : nullptr)) return rc;
    }
    lap("slots");
    // ---- work list: split each tile's pairs so that work items carry similar read counts
    std::vector<WorkItem> work;
    struct MergedGroup { uint32_t pair_lo, pair_hi, tile; };
    std::vector<MergedGroup> groups;                                // in work-item order
    {
        uint64_t total_reads_in_pairs = 0;
        for (const TilePair &p : pairs) total_reads_in_pairs += p.read_hi - p.read_lo;
        // work item size: pieces per workgroup.  MSNV_ITEM_PIECES overrides (tuning experiments).
        // ~1000 pieces (a few (tile, sample) pairs) per workgroup: measured on the benchmark shape (16 M pieces, one box,
        // pileup kernel): 400 -> 0.616 ms, 700 -> 0.614, 1000 -> 0.599-0.615, 1500 -> 0.608-0.633, 2628 -> 0.636-0.644.
        // Small items keep the last wave of workgroups short (an item of 2600 pieces runs ~190 us of a 640 us kernel);
        // below ~700 the per-item costs (LDS init, partial row, gate summing more rows) take over.
        // (at 4x the benchmark size 1000 still beats 2000: 55.5 vs 54.5 % of the roofline, so the size is a constant)
        // Items stay in tile order: dispatching the longest items first (shorter last wave of workgroups) measured 3 % SLOWER
        // (0.595 -> 0.612 ms) -- neighbouring items of a tile share the reference and the allele-total lines in L2.
        // (round 3, after the kernel had lost 9 % of its time: 1000 -> 0.5544 ms kernel / 0.6269 ms pass, 1400 -> 0.5520 / 0.6232,
        // 2000 -> 0.5510 / 0.6197, 2800 -> 0.5536 / 0.6235 over four alternations, profiles/r03k_ab_items.txt: flat in the kernel, the gate
        // kernel sums fewer partial rows)
        uint64_t target = 2000;
        if (const char *e = getenv("MSNV_ITEM_PIECES")) target = std::max<uint64_t>(64, (uint64_t)atoll(e));
        std::vector<WorkItem> wide, merged;
        auto chunks_of = [&](const TilePair &q) -> uint64_t {
            const bool nar = q.max_depth < NARROW_MAX_DEPTH;
            if (dense && nar) return (q.nblk + DENSE_CHUNK_BLOCKS - 1) / DENSE_CHUNK_BLOCKS;
            return (q.read_hi - q.read_lo + CHUNK_READS - 1) / CHUNK_READS;
        };
        // Taper: the items of the last tiles are cut smaller, so that the last wave of workgroups (dispatch is in index order)
        // ends on short items.  The thresholds are in units of one full wave of workgroups (resident workgroups x pieces per
        // item), not fractions of the dataset: on the benchmark shape (16 M pieces) they are the last 20 / 8 / 3 %, at
        // BASELINE configs[2] scale (514 M pieces) the last 0.6 % -- tapering a fixed fraction there cost 5 %.
        // MSNV_ITEM_TAPER=0 switches it off; MSNV_TAPER_AT=u1,u2,u3 moves the thresholds.
        const bool taper = [] { const char *e = getenv("MSNV_ITEM_TAPER"); return !(e && e[0] == '0'); }();   // read per dataset (tests switch it)
        double u1 = 1.8, u2 = 0.73, u3 = 0.27;
        if (const char *e = getenv("MSNV_TAPER_AT")) sscanf(e, "%lf,%lf,%lf", &u1, &u2, &u3);
        const uint64_t base_target = target;
        const double wave_pieces = (double)dev_resident_workgroups(7) * (double)base_target;   // 7 workgroups of 256 threads per CU (kernels.hip)
        uint64_t seen = 0;
        for (uint64_t t = 0; t < nt; ++t) {
            uint32_t lo = tps[t];
            uint64_t acc = 0, nch = 0;
            if (taper && total_reads_in_pairs) {
                const double left = (double)(total_reads_in_pairs - seen) / wave_pieces;
                target = left < u3 ? std::max<uint64_t>(64, base_target / 8) : left < u2 ? std::max<uint64_t>(64, base_target / 4) : left < u1 ? std::max<uint64_t>(64, base_target / 2) : base_target;
            }
            // merged groups: consecutive shallow pairs whose depth bounds add up to <= MERGE_MAX_DEPTH; an item = whole groups
            {
                uint32_t g_lo = tpm[t], i_lo = tpm[t];
                uint64_t g_depth = 0, i_pieces = 0, i_chunks = 0, g_pieces = 0;
                for (uint32_t k = tpm[t]; k < tps[t + 1]; ++k) {
                    const uint32_t nr = pairs[k].read_hi - pairs[k].read_lo;
                    seen += nr;
                    if (k > g_lo && (g_depth + pairs[k].max_depth > MERGE_MAX_DEPTH || k - g_lo >= MERGE_MAX_PAIRS)) {   // close the group
                        groups.push_back(MergedGroup{g_lo, k, (uint32_t)t});
                        i_pieces += g_pieces; i_chunks += (g_pieces + CHUNK_READS - 1) / CHUNK_READS;
                        g_lo = k; g_depth = 0; g_pieces = 0;
                        if (i_pieces >= target || i_chunks >= MAX_CHUNKS_PER_ITEM / 2) { merged.push_back(WorkItem{(uint32_t)t, i_lo, k, 0, 0, 0, 0, 0, ChunkDesc{}}); i_lo = k; i_pieces = 0; i_chunks = 0; }
                    }
                    g_depth += pairs[k].max_depth; g_pieces += nr;
                }
                if (tps[t + 1] > g_lo) { groups.push_back(MergedGroup{g_lo, tps[t + 1], (uint32_t)t}); merged.push_back(WorkItem{(uint32_t)t, i_lo, tps[t + 1], 0, 0, 0, 0, 0, ChunkDesc{}}); }
            }
            for (uint32_t k = tps[t]; k < tpm[t]; ++k) {
                const uint32_t nr = pairs[k].read_hi - pairs[k].read_lo;
                seen += nr;
                acc += nr; nch += chunks_of(pairs[k]);
                const bool narrow = pairs[k].max_depth < NARROW_MAX_DEPTH;
                const bool boundary = k + 1 == tpm[t] || (narrow != (pairs[k + 1].max_depth < NARROW_MAX_DEPTH));
                const uint64_t next_ch = boundary ? 0 : chunks_of(pairs[k + 1]);
                if (acc >= target || boundary || nch + next_ch > MAX_CHUNKS_PER_ITEM) {
                    (narrow ? work : wide).push_back(WorkItem{(uint32_t)t, lo, k + 1, 0, 0, 0, 0, 0, ChunkDesc{}}); lo = k + 1; acc = 0; nch = 0;
                }
            }
        }
        // whole-tile items behind the other merged items: the pileup kernel picks its code by the range a work item is in
        // (and among them the tiles with ONE pair last: their per-sample cells are the tile's totals, written by the gate kernel -- the
        // merged gather skips their groups, the last n_groups_solo of its list)
        for (uint64_t t = 0; t < nt; ++t) if (fuse_tile[t] && tps[t + 1] - tps[t] == 1) fuse_tile[t] = 2;
        std::stable_sort(merged.begin(), merged.end(), [&](const WorkItem &x, const WorkItem &y) { return fuse_tile[x.tile] < fuse_tile[y.tile]; });
        std::stable_sort(groups.begin(), groups.end(), [&](const MergedGroup &x, const MergedGroup &y) { return fuse_tile[x.tile] < fuse_tile[y.tile]; });
        d->n_groups_solo = 0;
        for (const MergedGroup &g : groups) d->n_groups_solo += fuse_tile[g.tile] == 2;
        d->n_work_narrow = (uint32_t)work.size();
        d->n_work_merged = (uint32_t)merged.size();
        d->n_work_fused = 0;
        for (const WorkItem &w : merged) d->n_work_fused += fuse_tile[w.tile] != 0;
        work.insert(work.end(), merged.begin(), merged.end());
        work.insert(work.end(), wide.begin(), wide.end());
    }
    lap("work list");
    // ---- coverage partials: one row of TILE counters per work item, rows of a tile contiguous (the gate kernel sums them):
    // u8 rows first (narrow items whose pairs' depth bounds add up to < 256: most of them at ~10x), then u16 rows (the other
    // narrow items: <= 32 pairs x 254), then the u32 rows of wide items.  Bit 0 of part_lo marks a u8 row, bits 1-2 hold the tile's allele-total mode.
    {
        std::vector<uint32_t> tss(nt + 1, 0), t16(nt + 1, 0), twide(nt + 1, 0);
        for (const WorkItem &w : work) ++tss[w.tile + 1];
        for (uint64_t t = 0; t < nt; ++t) tss[t + 1] += tss[t];
        std::vector<uint8_t> cls(work.size(), 0);
        // Allele totals (kernels.hip: tot_add / msnv_gate_sites): a tile's 4 x TILE words hold its mismatch totals position by
        // position, as narrow as the tile's summed depth bound allows -- 4 bytes (A, C, G, T) in ONE word per position when the
        // bound is below 256 (a sparse cohort: the gate kernel then reads 4 B per position instead of 16), 2 x u16 in two words
        // below 65536 (the benchmark shape), else four words.  MSNV_TOT_MODE=0..2 sets the narrowest mode allowed (tests).
        std::vector<uint64_t> tile_bound(nt, 0);
        for (size_t i = 0; i < work.size(); ++i) {
            uint64_t bound = 0;
            for (uint32_t k = work[i].pair_lo; k < work[i].pair_hi; ++k) bound += pairs[k].max_depth;
            tile_bound[work[i].tile] += bound;
            cls[i] = i >= d->n_work_narrow + d->n_work_merged ? 2 : bound < 256 ? 0 : 1;
        }
        const uint32_t min_tot_mode = [] { const char *e = getenv("MSNV_TOT_MODE"); return e ? (uint32_t)std::min(2, std::max(0, atoi(e))) : 0u; }();
        auto tot_mode = [&](uint64_t t) { return std::max<uint32_t>(min_tot_mode, tile_bound[t] < 256 ? 0u : tile_bound[t] < 65536 ? 1u : 2u); };
        std::vector<uint32_t> fill(tss.begin(), tss.end() - 1);
        std::vector<uint8_t> slot_cls(work.size(), 0);
        for (int c = 0; c < 3; ++c)
            for (size_t i = 0; i < work.size(); ++i)
                if (cls[i] == c) { work[i].slot = fill[work[i].tile]++; slot_cls[work[i].slot] = (uint8_t)c; }
        std::vector<uint64_t> off(work.size() + 1, 0);
        for (size_t s = 0; s < work.size(); ++s) off[s + 1] = off[s] + ((uint64_t)TILE << slot_cls[s]);
        for (size_t i = 0; i < work.size(); ++i) {
            WorkItem &w = work[i];
            w.part_lo = (uint32_t)off[w.slot] | (cls[i] == 0 ? 1u : 0u) | tot_mode(w.tile) << 1 | (fuse_tile[w.tile] ? WORK_FUSED : 0u); w.part_hi = (uint32_t)(off[w.slot] >> 32);
        }
        for (uint64_t t = 0; t < nt; ++t) {                    // first u16 row and first u32 row of every tile
            uint32_t s = tss[t];
            while (s < tss[t + 1] && slot_cls[s] < 1) ++s;
            t16[t] = s;
            while (s < tss[t + 1] && slot_cls[s] < 2) ++s;
            twide[t] = s;
        }
        std::vector<uint32_t> active;
        for (uint64_t t = 0; t < nt; ++t) if (tss[t + 1] > tss[t]) active.push_back((uint32_t)t);
        d->n_active_tiles = (uint32_t)active.size();
        if (int rc = arena.add(&d->active_tiles, active, &d->device_bytes, 1)) return rc;
        // the spill gather only has something to do in tiles that hold pairs outside merged groups (the others -- every tile of a sparse
        // cohort -- would each cost a workgroup that looks its tile up and leaves)
        {
            std::vector<uint32_t> gather_tiles;
            for (uint32_t t : active) if (tpm[t] > tps[t]) gather_tiles.push_back(t);
            d->n_gather_tiles = (uint32_t)gather_tiles.size();
            if (int rc = arena.add(&d->gather_tiles, gather_tiles, &d->device_bytes, 1)) return rc;
        }
        d->part_bytes = std::max<uint64_t>(16, off[work.size()]);
        if (int rc = arena.add(&d->tile_slot_start, tss, &d->device_bytes)) return rc;
        if (int rc = arena.add(&d->tile_slot_u16, t16, &d->device_bytes)) return rc;
        if (int rc = arena.add(&d->tile_slot_wide, twide, &d->device_bytes)) return rc;
        if (int rc = arena.add(&d->slot_off, off, &d->device_bytes)) return rc;
        if (int rc = dev_alloc((void **)&d->part, d->part_bytes, &d->device_bytes)) return rc;
        // one descriptor per active tile for the gate kernel: everything it looks up about its tile in one load
        std::vector<uint32_t> nslots_host(nt + 1, 0);
        for (uint64_t t = 0; t < nt; ++t) nslots_host[t] = ds.tile_slot_stride[t];
        std::vector<DeviceCols::GateTileH> gts;
        gts.reserve(active.size());
        for (uint32_t t : active) gts.push_back(DeviceCols::GateTileH{t, tss[t], t16[t], twide[t], tss[t + 1], vb_host[t], ve_host[t], nslots_host[t], off[tss[t]], tot_mode(t), (uint32_t)fuse_tile[t],
                                                                      tps[t], tpm[t] - tps[t], 0, 0});
        if (int rc = arena.add(&d->gate_tiles, gts, &d->device_bytes, 1)) return rc;
        d->gather_split = (uint32_t)std::min<uint64_t>(4, std::max<uint64_t>(1, (active.empty() ? 0 : pairs.size() / active.size()) / 32));
        if (const char *e = getenv("MSNV_GATHER_SPLIT")) d->gather_split = (uint32_t)std::max(1, atoi(e));      // (tuning experiments)
        {   // whole-tile work items write their candidate records per active tile
            std::vector<uint32_t> stage_idx(nt + 1, 0xffffffffu);
            uint32_t n_fused = 0;
            for (size_t i = 0; i < active.size(); ++i) { stage_idx[active[i]] = (uint32_t)i; n_fused += fuse_tile[active[i]] != 0; }
            d->n_fused_tiles = n_fused;
            if (n_fused) {
                std::vector<DeviceCols::GateTileH> dense_l, staged_l;
                for (size_t i = 0; i < gts.size(); ++i) {
                    if (!gts[i].staged) { dense_l.push_back(gts[i]); continue; }
                    DeviceCols::GateTileH g = gts[i];
                    g.row0 = (uint64_t)i;                              // (a whole-tile item writes no partial row: the field carries the index of its record list)
                    staged_l.push_back(g);
                }
                if (int rc = arena.add(&d->gate_tiles_dense, dense_l, &d->device_bytes, 1)) return rc;
                if (int rc = arena.add(&d->gate_tiles_staged, staged_l, &d->device_bytes, 1)) return rc;
                if (int rc = arena.add(&d->tile_stage_idx, stage_idx, &d->device_bytes)) return rc;
                // (+ one index per whole-tile item behind the lists: the tiles whose candidates do not fit a list -- kernels.hip: stage_ovf_list)
                if (int rc = dev_alloc((void **)&d->tile_stage, (uint64_t)active.size() * sizeof(TileStage) + (uint64_t)n_fused * sizeof(uint32_t), &d->device_bytes)) return rc;
                if (int rc = dev_memset_async(d->tile_stage, 0, (uint64_t)active.size() * sizeof(TileStage), ds.ctx ? ds.ctx->stream : nullptr)) return rc;
            }
        }
        d->wide_tot = false;
        for (uint32_t t : active) if (tot_mode(t) == 2u) d->wide_tot = true;
        d->use_dirty = !active.empty() && work.size() < 4 * active.size();       // a sparse cohort: fewer than four work items per tile
    }
    lap("partial rows, gate tiles");
    // ---- chunk descriptors.  Layout of d->chunks (round 6): the merged groups' chunks FIRST -- their number is the host's --, the narrow work
    // items' behind them: on the fast path those are cut in HBM and nobody waits for their count (devfin_chunks_launch).
    std::vector<ChunkDesc> mchunks, nchunks;                        // merged groups' / narrow items' (host loops only)
    std::vector<PieceHdr> hm;
    std::vector<DevMergedSrc> hm_src;                               // fast: the pairs whose headers a kernel writes
    uint64_t hm_count = 0;
    std::vector<MergedGroupDev> mgroups;
    {   // merged groups: their piece headers, group by group, and chunks that run across the group's pairs
        size_t gi = 0;
        for (uint32_t wi = d->n_work_narrow; wi < d->n_work_narrow + d->n_work_merged; ++wi) {
            WorkItem &w = work[wi];
            w.chunk_lo = (uint32_t)mchunks.size();
            for (; gi < groups.size() && groups[gi].pair_lo >= w.pair_lo && groups[gi].pair_hi <= w.pair_hi; ++gi) {
                const MergedGroup &g = groups[gi];
                const uint64_t h0 = fast ? hm_count : hm.size();
                for (uint32_t k = g.pair_lo; k < g.pair_hi; ++k) {
                    const TilePair &p = pairs[k];
                    if (fast) { hm_src.push_back(DevMergedSrc{k, k - g.pair_lo, hm_count}); hm_count += p.read_hi - p.read_lo; continue; }
                    const SampleCols &sc = ds.samples[p.sample];
                    for (uint32_t r = p.read_lo; r < p.read_hi; ++r) {
                        const uint64_t so = (sbase[p.sample] + sc.hdr[r].seqoff) >> SEQ_ALIGN_LOG2;          // 37 bits: bits 32-36 ride in bits 27-31 of the first word
                        hm.push_back(PieceHdr{(sc.hdr[r].gpos % TILE) | sc.hdr[r].cig << 11 | (k - g.pair_lo) << 19 | (uint32_t)(so >> 32) << 27, (uint32_t)so});
                    }
                }
                const uint64_t n_h = (fast ? hm_count : hm.size()) - h0;
                mgroups.push_back(MergedGroupDev{h0, w.tile, g.pair_lo, g.pair_hi - g.pair_lo, (uint32_t)n_h});
                for (uint64_t r = 0; r < n_h; r += CHUNK_READS) {
                    const uint32_t n = (uint32_t)std::min<uint64_t>(CHUNK_READS, n_h - r);
                    mchunks.push_back(ChunkDesc{h0 + r, 0, pairs[g.pair_lo].sample, g.pair_lo, n | (r + n >= n_h ? 1u << 16 : 0u), g.pair_hi - g.pair_lo});
                }
            }
            w.chunk_hi = (uint32_t)mchunks.size();
        }
        if (gi != groups.size()) return fail(MSNV_EINVAL, "internal: merged groups and work items disagree");
    }
    const uint64_t M = mchunks.size();
    if (M > 0x7ffffff0ull) return fail(MSNV_EDOMAIN, "more than 2^31 chunks of merged groups in one shard");
    std::vector<std::vector<uint32_t>> hdr4_of(HDR4 && !dense && !fast ? S : 0);      // 4-byte piece headers, per sample (chunk-relative offsets: filled with the chunks)
    for (size_t s = 0; s < hdr4_of.size(); ++s) hdr4_of[s].assign(ds.samples[s].hdr.size(), 0u);
    std::vector<uint32_t> narrow_pairs, item_first;                 // fast, piece layout: the narrow items' pairs in item order; every item's first entry
    uint64_t chunk_cap = 0;                                         // ... and the room their chunks get
    const bool chunks_on_device = fast && !dense;
    if (chunks_on_device) {
        if (rbase[S] > 0xfffffff0ull) return fail(MSNV_EDOMAIN, "more than 2^32 pieces in one shard: shard the contigs further");      // (chunks <= pieces: the 32-bit scan of their counts cannot wrap)
        item_first.reserve((size_t)d->n_work_narrow + 1);
        for (uint32_t wi = 0; wi < d->n_work_narrow; ++wi) {
            item_first.push_back((uint32_t)narrow_pairs.size());
            for (uint32_t k = work[wi].pair_lo; k < work[wi].pair_hi; ++k) { narrow_pairs.push_back(k); chunk_cap += (pairs[k].read_hi - pairs[k].read_lo + CHUNK_READS - 1) / CHUNK_READS + 2; }
        }
        item_first.push_back((uint32_t)narrow_pairs.size());
        if (const char *e = getenv("MSNV_CHUNK_CAP")) chunk_cap = (uint64_t)std::max<long long>(0, atoll(e));      // (tests: a table too small on purpose -> the exact, waiting form)
        if (M + chunk_cap > 0xfffffff0ull) return fail(MSNV_EDOMAIN, "more than 2^32 chunks in one shard");
    }
    else for (uint32_t wi = 0; wi < d->n_work_narrow; ++wi) {
        WorkItem &w = work[wi];
        w.chunk_lo = (uint32_t)(M + nchunks.size());
        for (uint32_t k = w.pair_lo; k < w.pair_hi; ++k) {
            const TilePair &p = pairs[k];
            if (dense) {        // chunk = up to DENSE_CHUNK_BLOCKS blocks of the pair's stream: {first block, seq byte offset of that block}
                for (uint32_t b = 0; b < p.nblk; b += DENSE_CHUNK_BLOCKS) {
                    const uint32_t n = std::min<uint32_t>(DENSE_CHUNK_BLOCKS, p.nblk - b);
                    nchunks.push_back(ChunkDesc{bbase[p.sample] + p.blk_lo + b, sbase[p.sample] + p.seq0 + 16ull * b, p.pad >> 8, k,
                                                n | (b + n >= p.nblk ? 1u << 16 : 0u), p.pad & 0xffu});      // "sample" = the sample's slot in the tile
                }
                continue;
            }
            if (HDR4) {
                // a chunk = up to CHUNK_READS consecutive pieces whose seq bytes lie within 2^HDR4_OFF_BITS alignment units of the chunk's
                // lowest offset (always true for pieces in storage order; the pieces a read leaves in the NEXT tile sit a little before
                // that tile's other pieces); seq_base = the absolute offset of that lowest byte, the headers hold the distance to it
                const SampleCols &sc = ds.samples[p.sample];
                std::vector<uint32_t> &h4 = hdr4_of[p.sample];
                constexpr uint64_t span_max = (uint64_t)SEQ_ALIGN << HDR4_OFF_BITS;
                uint32_t r = p.read_lo;
                while (r < p.read_hi) {
                    uint64_t lo = sc.hdr[r].seqoff, hi = lo;
                    uint32_t e = r;
                    while (e < p.read_hi && e - r < CHUNK_READS) {
                        const uint64_t o = sc.hdr[e].seqoff, nlo = std::min(lo, o), nhi = std::max(hi, o);
                        if (nhi - nlo >= span_max) break;
                        lo = nlo; hi = nhi; ++e;
                    }
                    for (uint32_t i = r; i < e; ++i)
                        h4[i] = (sc.hdr[i].gpos % TILE) | sc.hdr[i].cig << 11 | (uint32_t)((sc.hdr[i].seqoff - lo) >> SEQ_ALIGN_LOG2) << 19;
                    nchunks.push_back(ChunkDesc{rbase[p.sample] + r, sbase[p.sample] + lo, p.pad >> 8, k, (e - r) | (e >= p.read_hi ? 1u << 16 : 0u), p.pad & 0xffu});
                    r = e;
                }
                continue;
            }
            for (uint32_t r = p.read_lo; r < p.read_hi; r += CHUNK_READS) {
                const uint32_t n = std::min<uint32_t>(CHUNK_READS, p.read_hi - r);
                nchunks.push_back(ChunkDesc{rbase[p.sample] + r, sbase[p.sample], p.pad >> 8, k, n | (r + n >= p.read_hi ? 1u << 16 : 0u), p.pad & 0xffu});
            }
        }
        w.chunk_hi = (uint32_t)(M + nchunks.size());
    }
    lap("chunks (host part)");
    if (!chunks_on_device) for (size_t wi = 0; wi < (size_t)d->n_work_narrow + d->n_work_merged; ++wi)
        if (work[wi].chunk_hi > work[wi].chunk_lo) work[wi].first = work[wi].chunk_lo < M ? mchunks[work[wi].chunk_lo] : nchunks[work[wi].chunk_lo - M];
    // ---- the index tables up, in one block; then the kernels that finish them are queued (nothing here waits for them)
    d->n_pairs = (uint32_t)pairs.size(); d->n_work = (uint32_t)work.size();
    d->n_merged_groups = (uint32_t)mgroups.size();
    d->max_group_pairs = 0;
    for (const MergedGroupDev &g : mgroups) d->max_group_pairs = std::max(d->max_group_pairs, g.n_pairs);
    if (int rc = arena.add(&d->pairs, pairs, &d->device_bytes)) return rc;
    if (int rc = arena.add(&d->s_read_base, rbase, &d->device_bytes)) return rc;
    if (int rc = arena.add(&d->s_seq_base, sbase, &d->device_bytes)) return rc;
    if (int rc = arena.add(&d->merged_groups, mgroups, &d->device_bytes, 1)) return rc;
    if (int rc = arena.add(&d->tile_pair_merged, tpm, &d->device_bytes)) return rc;
    if (int rc = arena.add(&d->tile_pair_start, tps, &d->device_bytes)) return rc;
    if (int rc = arena.add(&d->work, work, &d->device_bytes)) return rc;
    if (fast) {
        ds.info.bytes_headers += hm_count * sizeof(PieceHdr);
        d->n_hdr8m = hm_count;
    } else {
        ds.info.bytes_headers += hm.size() * sizeof(PieceHdr);
        d->n_hdr8m = hm.size();
        if (int rc = arena.add(&d->hdr8m, hm, &d->device_bytes, 1)) return rc;
    }
    if (int rc = arena.commit(*d)) return rc;
    const uint64_t chunk_room = M + (chunks_on_device ? chunk_cap : nchunks.size());
    d->n_chunks = M + nchunks.size();                               // (fast, piece layout: the narrow chunks are counted behind finalize's last wait)
    if (int rc = dev_alloc((void **)&d->chunks, (chunk_room + 1) * sizeof(ChunkDesc), &d->device_bytes)) return rc;
    if (int rc = dev_upload(d->chunks, mchunks.data(), M * sizeof(ChunkDesc))) return rc;
    if (int rc = dev_upload(d->chunks + M, nchunks.data(), nchunks.size() * sizeof(ChunkDesc))) return rc;
    lap("  index tables up");
    if (fast) {
        if (int rc = dev_alloc((void **)&d->hdr, (rbase[S] + 1) * sizeof(ReadHdr), &d->device_bytes)) return rc;
        if (int rc = dev_alloc((void **)&d->hdr8m, (hm_count + 1) * sizeof(PieceHdr), &d->device_bytes)) return rc;
    }
    if (chunks_on_device) {
        // the pieces are in HBM: the chunks of every narrow pair are cut there by a kernel that runs the same greedy rule as the loop above,
        // the work items learn their ranges there too
        if (int rc = dev_alloc((void **)&d->hdr4, (rbase[S] + 4) * sizeof(uint32_t), &d->device_bytes)) return rc;
        if (int rc = dev_memset_async(d->hdr4, 0, (rbase[S] + 4) * sizeof(uint32_t), ds.ctx->stream)) return rc;
        if (int rc = devfin_chunks_launch(ds, *d, narrow_pairs, item_first, (uint32_t)M, chunk_cap)) return rc;
    }
    if (fast) {
        if (int rc = devfin_merged_headers(ds, *d, hm_src)) return rc;
        if (int rc = devfin_work_first(ds, *d, d->n_work_narrow + d->n_work_merged)) return rc;
        if (int rc = devfin_headers(ds, *d, rbase)) return rc;     // (the 16-byte headers with linear positions: wide kernel, padding, host mapping -- behind everything the first pass waits for)
    }

    lap("merged groups");
    // ---- columns
    d->n_reads = rbase[S]; d->n_seq_bytes = sbase[S]; d->n_blk = bbase[S];
    if (!fast) if (int rc = dev_alloc((void **)&d->hdr, (rbase[S] + 1) * sizeof(ReadHdr), &d->device_bytes)) return rc;
    if (!dense && !HDR4) if (int rc = dev_alloc((void **)&d->hdr8, (rbase[S] + 1) * sizeof(PieceHdr), &d->device_bytes)) return rc;
    if (!dense && HDR4 && !fast) if (int rc = dev_alloc((void **)&d->hdr4, (rbase[S] + 4) * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (dense) if (int rc = dev_alloc((void **)&d->blk, (bbase[S] + 1) * sizeof(uint32_t), &d->device_bytes)) return rc;
    const bool cols_on_device = fast && !dense;                    // every sample's columns are in the rounds' buffers, already in this layout (devpack_place_columns)
    if (cols_on_device) { if (int rc = devpack_place_columns(ds, *d, sbase)) return rc; }
    else {
    if (int rc = dev_alloc((void **)&d->seq, sbase[S] + 256, &d->device_bytes)) return rc;     // lanes past the end of the last piece read on
    if (int rc = dev_memset(d->seq, 0xff, sbase[S] + 256)) return rc;                          // (the < 16 bytes between two samples' columns: defined, so that two builds of a dataset can be compared)
    // the quality column of the device is ONE BIT per base -- "below the -Q cutoff" -- at the index of the base's nibble in the seq column: the
    // cutoff is a parameter of the dataset (mpileup -Q, metaSNV.py:160-165 never changes it) and nothing else of a quality is ever looked at
    // behind the overlap tweak and the token limit, which ran on the host (pass 1 above).  1 B -> 1/8 B per base of HBM and of upload.
    if (int rc = dev_alloc((void **)&d->qual, sbase[S] / 4 + 64, &d->device_bytes)) return rc;
    if (int rc = dev_memset(d->qual, 0, sbase[S] / 4 + 64)) return rc;
    }
    d->qlow_cutoff = ds.params.min_baseq;
    uint64_t alg = 0;
    lap("  column allocs + memsets");
    {
        // the samples' columns go up from a few host threads at a time (a pageable copy is staged by the runtime: several in flight keep
        // the link busy while the compact headers of the next samples are built)
        std::atomic<int> up_err{0};
        std::mutex up_mu; std::string up_msg;
        const int device = ds.ctx ? ds.ctx->device : 0;
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            (void)dev_set_device(device);
            std::vector<uint8_t> qbits;
            for (;;) {
                const size_t s = next.fetch_add(1);
                if (s >= S || up_err.load()) break;
                SampleCols &sc = ds.samples[s];
                int rc = fast ? MSNV_OK : dev_upload(d->hdr + rbase[s], sc.hdr.data(), sc.hdr.size() * sizeof(ReadHdr));
                if (fast) {
                    // (headers and 4-byte headers were built in HBM; so were the block descriptors of the dense layout)
                    if (dense) rc = devpack_copy_blocks(sc, d->blk + bbase[s], ds.ctx ? ds.ctx->stream : nullptr);
                } else if (!rc && dense) {
                    rc = dev_upload(d->blk + bbase[s], sc.blk.data(), sc.blk.size() * sizeof(uint32_t));
                } else if (!rc && HDR4) {
                    rc = dev_upload(d->hdr4 + rbase[s], hdr4_of[s].data(), hdr4_of[s].size() * sizeof(uint32_t));
                    std::vector<uint32_t>().swap(hdr4_of[s]);
                } else if (!rc) {   // compact tile-local headers of the narrow kernel: {start in tile | length << 11, seq offset / SEQ_ALIGN}
                    std::vector<PieceHdr> h8(sc.hdr.size());
                    for (size_t i = 0; i < sc.hdr.size(); ++i) h8[i] = PieceHdr{(sc.hdr[i].gpos % TILE) | sc.hdr[i].cig << 11, sc.hdr[i].seqoff >> SEQ_ALIGN_LOG2};
                    rc = dev_upload(d->hdr8 + rbase[s], h8.data(), h8.size() * sizeof(PieceHdr));
                }
                if (cols_on_device) {}                                     // (placed above, round by round)
                else if (!rc && sc.on_device) rc = devpack_copy_columns(sc, d->seq + sbase[s], d->qual + sbase[s] / 4, ds.ctx ? ds.ctx->stream : nullptr);      // packed on the device: HBM to HBM
                else {
                    if (!rc) rc = dev_upload(d->seq + sbase[s], sc.seq.data(), sc.seq.size());
                    if (!rc) {
                        pack_lowq(sc.qual.data(), sc.qual.size(), ds.params.min_baseq, qbits);
                        rc = dev_upload(d->qual + sbase[s] / 4, qbits.data(), qbits.size());      // (sbase: multiples of 16 bytes of seq = 32 flags)
                    }
                }
                if (rc) { std::lock_guard<std::mutex> lk(up_mu); if (!up_err.load()) { up_msg = msnv_last_error(); up_err.store(rc); } continue; }
                // release host staging of the bulky columns; headers stay (coverage pass, results mapping)
                std::vector<uint8_t>().swap(sc.seq);
                std::vector<uint8_t>().swap(sc.qual);
                if (dense) std::vector<uint32_t>().swap(sc.blk);
            }
        };
        const size_t n_up = std::min<size_t>(S, std::min<size_t>(8, msnv_default_threads()));
        std::vector<std::thread> th;
        if (cols_on_device) worker();                                  // (nothing to upload: no threads)
        else for (size_t t = 0; t < n_up; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
        if (up_err.load()) return fail(up_err.load(), "%s", up_msg.c_str());
        lap("  column copies");
        {   // device-packed samples: the alignment padding behind their pieces, then their round buffers and the pack tables go back
            std::vector<uint8_t> on_dev(S, 0);
            for (size_t s = 0; s < S; ++s) on_dev[s] = ds.samples[s].on_device ? 1 : 0;
            // (round 6: the device pack's emit kernels leave the padding in place -- the pass over every piece is only taken when a FASTA record
            // is longer than its contig, where "the reference" behind the contig's last tile position differs between the two)
            bool fasta_longer = false;
            for (size_t c = 0; c < NC; ++c) if (ds.sel[c] && ds.has_seq[c] && (int64_t)ds.seqs[c].size() > maxend[c]) fasta_longer = true;
            if (!dense && (!ds.dp.pad_in_emit || fasta_longer || getenv("MSNV_FILL_PADDING"))) if (int rc = devpack_fill_padding(*d, on_dev, ds.ctx ? ds.ctx->stream : nullptr)) return rc;
        }
        for (size_t s = 0; s < S; ++s) {
            const SampleCols &sc = ds.samples[s];
            ds.info.bytes_headers += dense ? (bbase[s + 1] - bbase[s]) * sizeof(uint32_t) : (rbase[s + 1] - rbase[s]) * (HDR4 ? sizeof(uint32_t) : sizeof(PieceHdr));
            ds.info.bytes_cigar += sc.alg_cigar_bytes;
            alg += sc.alg_8d_bytes;
            ds.info.bytes_seq += sc.alg_seq_bytes;
            ds.info.bytes_qual += sc.alg_qual_bytes;
        }
    }
    d->algorithmic_bytes = alg;                  // SURVEY.md section 8d figure; the shipped bytes are bytes_headers + bytes_seq + bytes_qual

    lap("columns");
    // ---- intermediates (before the coverage index: their allocations and fills -- queued on the context's stream, in front of the first
    // pass -- run while the device still works on the index; the coverage index below holds finalize's last wait)
    void *const fin_stream = ds.ctx ? ds.ctx->stream : nullptr;
    if (int rc = dev_alloc((void **)&d->tot, std::max<uint64_t>(1, 4 * npos) * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_memset_async(d->tot, 0, std::max<uint64_t>(1, 4 * npos) * sizeof(uint32_t), fin_stream)) return rc;   // the gate kernel keeps it zero between passes
    if (int rc = dev_alloc((void **)&d->spill, std::max<uint64_t>(1, (uint64_t)pairs.size()) * TILE, &d->device_bytes)) return rc;
    double est_events = 0.0;                                        // mismatching bases the sampled rate predicts: what the event list is sized by
    {
        // Allele bookkeeping of the narrow work items.  Clean reads (the benchmark's 0.1 % errors): a mismatching base is an EVENT -- one
        // memory-side atomic on the position's totals + 8 bytes in the event list, scattered into the called sites' cells afterwards.
        // At a few per cent of mismatches (real metagenomic reads against a species representative) every position carries some and the
        // events are the pass: 37.7 M of them at 3 % = 1.2 ms of atomics in the pileup kernel + 1.5 ms of scatter (profiles/r03e).  Then
        // the alleles go the way the coverage goes: every (sample, tile) pair writes four byte PLANES (A, C, G, T counts per position,
        // 8 KB a pair, plain 16-byte stores), the gate kernel sums the planes, the gather transposes them into the cells.  Picked when the
        // sampled mismatch rate reaches 1.0 % (with byte qualities the break-even was 1.5 %; the kernel that reads one bit of quality per base
        // no longer hides the events behind HBM time: at 0.6 % of errors -- 0.85 % sampled with the cohort's SNVs -- events 0.603 vs planes
        // 0.643 ms a pass, at 1 % -- 1.25 % sampled -- 0.773 vs 0.601 ms, profiles/r03zam_crossover.txt; at 3 % 3.55 vs 2.08 ms, at 10 % 8.5 vs
        // 2.1 ms, profiles/r03g_*; a cost model in events per pair put the sigma = 2 cohort -- 0.4 % of mismatches, deep
        // pairs -- on planes, where the five-plane gather over its 526 k sites cost more than the events: 4.95 vs 4.08 ms);
        // MSNV_ALLELES=planes | events overrides.
        // Needs byte counts everywhere: not with wide work items (MSNV_DEEP=wide) and not in the dense piece layout's kernel.
        if (int rc = devpack_sync_pending(ds)) return rc;               // (the last round's mismatch sample: its kernels were left running behind the pack)
        uint64_t sb = 0, sm = 0;
        for (const SampleCols &sc : ds.samples) { sb += sc.mm_sampled_bases; sm += sc.mm_sampled; }
        const double rate = sb ? (double)sm / (double)sb : 0.0;
        est_events = rate * (double)tot_bases;
        bool planes = rate >= 0.010;
        if (const char *e = getenv("MSNV_ALLELES")) planes = e[0] == 'p';
        if (dense || d->n_work > d->n_work_narrow + d->n_work_merged || pairs.empty()) planes = false;
        d->allele_planes = planes;
        if (planes) {
            const uint64_t bytes = (uint64_t)pairs.size() * 4 * TILE;
            if (int rc = dev_alloc((void **)&d->aspill, bytes, &d->device_bytes)) return rc;
            if (int rc = dev_memset_async(d->aspill, 0, bytes, fin_stream)) return rc;      // (rows of merged pairs are never written and never read)
        }
        ds.info.allele_planes = planes ? 1 : 0;
        ds.info.sampled_mismatch_ppm = (uint64_t)(rate * 1e6);
    }
    // sparse buffers: generous first guess, grown on MSNV_ECAPACITY by the caller
    // (the event list: four times the mismatching bases the sample of every 16th piece predicts, at most one per 16 bases -- the flat
    // "one per 16 bases" of the earlier rounds was 29 GB of HBM, and 1 s of hipMalloc, for BASELINE configs[2]'s 5.8e10 bases at 0.4 % of mismatches)
    d->cap_events = (uint32_t)std::min<uint64_t>(0x7fffffffull, std::max<uint64_t>(1u << 20, std::min<uint64_t>(tot_bases / 16, (uint64_t)(4.0 * est_events) + (1u << 20))));
    if (const char *e = getenv("MSNV_CAP_EVENTS")) d->cap_events = (uint32_t)std::max<long long>(EV_LISTS, atoll(e));   // tests: force the grow-and-rerun path
    d->cap_overflow = (uint32_t)std::min<uint64_t>(0x7fffffffull, std::max<uint64_t>(1u << 16, npos / 8));
    d->cap_sites = (uint32_t)std::min<uint64_t>(0x7fffffffull, std::max<uint64_t>(1u << 16, npos / 4));
    if (int rc = dev_alloc((void **)&d->events, (uint64_t)d->cap_events * sizeof(Pair32), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->overflow, (uint64_t)d->cap_overflow * sizeof(Pair32), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->sites, (uint64_t)d->cap_sites * sizeof(SiteRec), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->unc_sites, (uint64_t)d->cap_sites * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->site_row, (npos / 64 + 1) * sizeof(unsigned long long), &d->device_bytes)) return rc;   // per 64 positions (kernels.hip: CellMap::block_row)
    if (int rc = dev_alloc((void **)&d->tile_dirty, ((uint64_t)work.size() + 1) * sizeof(uint32_t), &d->device_bytes)) return rc;    // one word per work item (by slot)
    if (int rc = dev_memset_async(d->tile_dirty, 0, ((uint64_t)work.size() + 1) * sizeof(uint32_t), fin_stream)) return rc;
    // no memset per pass: the counter blocks alternate (the gate kernel zeroes the next one) and the gate kernel leaves the
    // individual-rule bits it consumes zero, like the allele totals
    if (int rc = dev_alloc((void **)&d->counters, 2 * CNT_WORDS * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_memset_async(d->counters, 0, 2 * CNT_WORDS * sizeof(uint32_t), fin_stream)) return rc;
    if (int rc = dev_alloc((void **)&d->ind4, (npos / 8 + npos / 32 + 2) * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_memset_async(d->ind4, 0, (npos / 8 + npos / 32 + 2) * sizeof(uint32_t), fin_stream)) return rc;
    d->unc_bits = d->ind4 + npos / 8 + 1;
    for (const TilePair &tp : pairs) if ((tp.pad & 0xffu) == 1) { d->any_split = true; break; }
    if (int rc = dev_alloc((void **)&d->site_bits, (npos / 64 + 1) * sizeof(unsigned long long), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->site_rank, (npos / 64 + 1) * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->tile_site_base, (nt + 1) * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d->tile_site_cnt, (nt + 1) * sizeof(uint32_t), &d->device_bytes)) return rc;
    if (int rc = dev_memset_async(d->tile_site_cnt, 0, (nt + 1) * sizeof(uint32_t), fin_stream)) return rc;
    if (int rc = dev_memset_async(d->tile_site_base, 0, (nt + 1) * sizeof(uint32_t), fin_stream)) return rc;

    ds.info.n_samples = S; ds.info.n_contigs = 0; ds.info.n_positions = 0;
    for (size_t c = 0; c < NC; ++c) if (ds.sel[c]) { ds.info.n_contigs++; ds.info.n_positions += (uint64_t)ds.lengths[c]; }
    ds.info.n_reads = tot_reads; ds.info.n_reads_pileup = tot_pile_reads; ds.info.n_pileup_bases = tot_bases;
    ds.info.n_tiles = nt; ds.info.n_pairs = pairs.size(); ds.info.n_work = work.size();
    lap("intermediates");

    // ---- the coverage index's tables: joined (device-packed datasets: built beside everything above) or built here
    if (cov_job.th.joinable()) { if (int rc = cov_job.join()) return rc; }
    else if (int rc = cov_index(cov_arena, false)) return rc;
    lap("    cov tables: staged");
    d->device_bytes += cov_bytes;
    if (int rc = cov_arena.commit(*d)) return rc;
    lap("    cov tables: up");
    {   // accumulator copies: as many as fit 64 MB, at most 8 (many contigs = few tiles per contig = little contention anyway)
        d->n_cov_rows = ds.cov_row_sample.size();
        const uint64_t acc_bytes = std::max<uint64_t>(1, d->n_cov_rows) * (1 + COV_BINS) * sizeof(unsigned long long);
        d->cov_copies = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8, (64ull << 20) / std::max<uint64_t>(1, acc_bytes)));
        if (int rc = dev_alloc((void **)&d->cov_acc, d->cov_copies * acc_bytes, &d->device_bytes)) return rc;
    }
    lap("  coverage tables");
    // ---- the last wait: how many chunks the narrow items got (cut in HBM, counted there; an event behind the cut's kernels)
    if (chunks_on_device) {
        uint64_t n_narrow = 0; bool overflow = false;
        if (int rc = devfin_chunks_result(ds, &n_narrow, &overflow)) return rc;
        if (overflow) {
            // (more chunks than the bound gave room for: once more with the exact number, which is known now)
            dev_free(d->chunks); d->chunks = nullptr;
            if (M + n_narrow > 0xfffffff0ull) return fail(MSNV_EDOMAIN, "more than 2^32 chunks in one shard");
            if (int rc = dev_alloc((void **)&d->chunks, (M + n_narrow + 1) * sizeof(ChunkDesc), &d->device_bytes)) return rc;
            if (int rc = dev_upload(d->chunks, mchunks.data(), M * sizeof(ChunkDesc))) return rc;
            if (int rc = devfin_chunks_launch(ds, *d, narrow_pairs, item_first, (uint32_t)M, n_narrow)) return rc;
            if (int rc = devfin_work_first(ds, *d, d->n_work_narrow + d->n_work_merged)) return rc;
            if (int rc = dev_stream_wait(fin_stream)) return rc;
            if (int rc = devfin_chunks_result(ds, &n_narrow, &overflow)) return rc;
            if (overflow) return fail(MSNV_EINVAL, "internal: the chunk count changed between two cuts of the same pairs");
        }
        d->n_chunks = M + n_narrow;
    }
    if (int rc = devpack_finish(ds)) return rc;                // (device-packed samples: the rounds' buffers and the pack tables go back)
    lap("pack tables released");
    ds.info.bytes_index = pairs.size() * sizeof(TilePair) + work.size() * sizeof(WorkItem) + d->n_chunks * sizeof(ChunkDesc) + (nt + 1) * 4;
    ds.info.device_bytes = d->device_bytes;
    ds.finalized = true;
    return MSNV_OK;
}

}  // namespace msnv
