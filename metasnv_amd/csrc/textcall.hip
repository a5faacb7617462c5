// metasnv_amd/csrc/textcall.hip -- snpCall on its OWN input: mpileup text (msnv_call_from_mpileup).
//
//   main loop of snpCall       call_vC.cpp:466-668   one mpileup line = one reference position
//   first line                 call_vC.cpp:423-431   sample count from its tabs; the line itself is never processed
//   toksplit                   call_vC.cpp:92-111    tab tokeniser: skips leading blanks, keeps 10 000 characters of a token
//   base-string parse          call_vC.cpp:503-535   ^x, +n..., -n..., * $ N n ignored, the ten counted symbols
//   gates / allele loop        call_vC.cpp:545-552, 577-601
//
// This is the reference's literal boundary -- `snpCall ... < mpileup.txt` -- next to msnv_call (which replaces the whole
// `samtools mpileup | snpCall` pipe and never sees text).  The text is HBM-bound byte work: ~4 KB per position at 160 samples.
// One wavefront per line: it finds the tabs with ballots (the field number of every byte = tabs before it), notes where
// the base strings start, then every lane parses the base strings of its samples (8 bytes per load) and the wavefront
// sums them up, applies the gates and the calling rule, and appends a record for a called position: the site header and
// the per-sample counts the formatter of the BAM path takes (format.cpp).  The host reads the text, cuts it into lines,
// parses the three leading fields of the called lines (contig, position, reference character) and writes the files; gene /
// codon annotation (-g) runs through msnv_annotate_sites like the BAM path's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "device.h"
#include "msnv_internal.h"

namespace msnv {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(MSNV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int write_calls_text(msnv_dataset &ds, const char *called_path, const char *indiv_path, const msnv_site_ann *ann, const std::vector<std::string> *gene_names);

constexpr uint32_t TOK_CAP = 10000;            // call_vC.cpp:482: characters of a token that toksplit keeps
constexpr int TC_NT = 256;                     // four wavefronts = four lines per workgroup

struct TextRec { uint32_t line, cov, n[4]; uint32_t masks; uint32_t pad; };   // 32 B: header of one called line (masks: pop | ind << 4)
struct TextArgs {
    const uint8_t *text; const uint64_t *line_off; uint32_t n_lines, n_samples;     // line i = text[line_off[i], line_off[i + 1])
    int min_cov, min_snvs; double min_frac;
    uint32_t *fstart;                          // [waves][n_samples]: where each sample's base string starts (line-relative)
    msnv_site_sample *scratch;                 // [waves][n_samples]: the line's per-sample counts
    TextRec *rec; msnv_site_sample *rec_samples; uint32_t cap_rec;
    uint32_t *counters;                        // [0] records, [1] first line with a domain error (min), [2] its kind | byte << 8
    uint32_t *next_line;                       // work counter: the wavefronts take lines in order
    uint64_t *bases_parsed;
};

__device__ __forceinline__ uint32_t ld_byte(const uint8_t *p) { return *p; }

// index of the counted symbol or -1 / -2:  . , -> 0 (match)   a A -> 1   c C -> 2   g G -> 3   t T -> 4;   * $ N n -> -1 (ignored);  anything else -> -2
__device__ __forceinline__ int sym_class(const uint32_t c) {
    switch (c) {
        case '.': case ',': return 0;
        case 'a': case 'A': return 1;
        case 'c': case 'C': return 2;
        case 'g': case 'G': return 3;
        case 't': case 'T': return 4;
        case '*': case '$': case 'N': case 'n': return -1;
        default: return -2;
    }
}

__global__ __launch_bounds__(TC_NT) void msnv_parse_pileup_lines(const TextArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (TC_NT / 64) + (threadIdx.x >> 6);
    uint32_t *const fstart = a.fstart + (uint64_t)wave * a.n_samples;
    msnv_site_sample *const scratch = a.scratch + (uint64_t)wave * a.n_samples;
    const uint32_t S = a.n_samples;
    uint64_t my_bases = 0;
    for (;;) {
        uint32_t li = 0;
        if (lane == 0) li = atomicAdd(a.next_line, 1u);
        li = (uint32_t)__shfl((int)li, 0);
        if (li >= a.n_lines) break;
        const uint8_t *const L = a.text + a.line_off[li];
        uint32_t len = (uint32_t)(a.line_off[li + 1] - a.line_off[li]);
        // ---- phase A: one pass over the line.  strlen (a NUL ends the line, fgets + strlen: call_vC.cpp:473), the tabs, and
        // the start of every base string: the token behind the k-th tab is field k; fields 4, 7, 10 ... are the samples' base
        // strings (call_vC.cpp:503: pos > 3 && pos % 3 == 1, sample pos / 3).
        uint32_t tabs = 0;                      // tabs before the current window (uniform)
        uint32_t smax = 0;                      // samples whose base string starts inside the line
        bool too_many = false;
        for (uint32_t w0 = 0; w0 < len; w0 += 64u) {
            const uint32_t p = w0 + (uint32_t)lane;
            const uint32_t c = p < len ? ld_byte(L + p) : 1u;
            const unsigned long long nul = __ballot(c == 0u);
            if (nul) len = min(len, w0 + (uint32_t)__builtin_ctzll(nul));            // the bytes behind a NUL do not exist
            const unsigned long long tb = __ballot(c == '\t' && p < len);
            if (c == '\t' && p < len) {
                const uint32_t k = tabs + (uint32_t)__popcll(tb & ((1ull << lane) - 1ull)) + 1u;     // field that starts at p + 1
                if (k > 3u && k % 3u == 1u) {
                    const uint32_t s = k / 3u;                                      // 1-based sample
                    if (s <= S) fstart[s - 1u] = p + 1u; else too_many = true;
                }
            }
            tabs += (uint32_t)__popcll(tb);
        }
        if (len > 0u) --len;                                                        // line[--lLen] = 0 (call_vC.cpp:475)
        too_many = __any(too_many);                                                 // (found by the lane that looked at the tab)
        __threadfence();                                                            // fstart[] is written by one lane and read by another: no stale L1 line of the previous position
        // fields counted from the tabs inside the stripped line; a field is only processed if something follows the tab that ends
        // it (while (*rest), call_vC.cpp:490) -- checked per field below
        {
            const uint32_t k_last = tabs;                                           // the line holds fields 0 .. k_last (before stripping)
            smax = k_last >= 4u ? min(S, (k_last - 1u) / 3u) : 0u;                  // fields 4, 7, ...: sample s starts at field 3 s + 1
        }
        // ---- phase B: every lane parses the base strings of its samples
        uint32_t t_cov = 0, t_n[4] = {0u, 0u, 0u, 0u}, ind = 0, err = 0;              // my samples' sums; alleles some sample of mine holds >= t reads of; error kind | byte << 8
        bool extra = false;                                                         // a processed base string of a sample beyond the first line's count
        for (uint32_t s = (uint32_t)lane; s < S; s += 64u) {
            uint32_t cnt[5] = {0u, 0u, 0u, 0u, 0u};
            if (s < smax) {
                uint32_t b = fstart[s];
                if (b <= len) {
                    // the token: leading blanks skipped, up to the next tab or the end of the line (toksplit)
                    while (b < len && ld_byte(L + b) == ' ') ++b;
                    uint32_t e = b;
                    while (e < len && ld_byte(L + e) != '\t') ++e;
                    const bool processed = e < len && e + 1u < len;                 // a tab ends it and the line goes on behind the tab
                    if (processed) {
                        const uint32_t n = min(e - b, TOK_CAP);
                        uint32_t i = 0;
                        while (i < n) {
                            const uint32_t c = ld_byte(L + b + i);
                            if (c == '^') { ++i; }                                  // call_vC.cpp:511-514: the mapping quality character
                            else if (c == '+' || c == '-') {                        // :515-522: skip the inserted / deleted bases
                                uint32_t skip = 0;
                                for (;;) {
                                    ++i;
                                    const uint32_t d = i < n ? ld_byte(L + b + i) : 0u;          // the token is NUL-terminated in the reference
                                    if (d < '0' || d > '9') break;
                                    skip = skip * 10u + (d - '0');
                                    if (skip > 0x0fffffffu) skip = 0x0fffffffu;
                                }
                                i += skip - 1u;                                     // (wraps by one for skip == 0: undone by the ++i below)
                            } else {
                                const int k = sym_class(c);
                                if (k >= 0) ++cnt[k];
                                else if (k == -2 && !err) err = 1u | c << 8;        // the reference writes through an empty vector (SIGSEGV)
                            }
                            ++i;
                        }
                        my_bases += n;
                    }
                }
            }
            const uint32_t cov = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4];
            msnv_site_sample r;
            r.cov = (uint16_t)cov; r.n[0] = (uint16_t)cnt[1]; r.n[1] = (uint16_t)cnt[2]; r.n[2] = (uint16_t)cnt[3]; r.n[3] = (uint16_t)cnt[4];
            scratch[s] = r;
            t_cov += cov;
#pragma unroll
            for (int x = 0; x < 4; ++x) { t_n[x] += cnt[1 + x]; if ((int)cnt[1 + x] >= a.min_snvs) ind |= 1u << x; }
        }
        // base strings of samples beyond the first line's count: the reference writes out of bounds as soon as one is processed
        if (too_many) {
            // find out whether such a field is processed: field k = 3 (S + 1) + 1 starts behind tab number k; it is processed when a tab
            // ends it and the stripped line goes on behind that tab.  The wavefront redoes the tab count for that one field.
            uint32_t seen = 0; bool hit = false;
            const uint32_t want = 3u * (S + 1u) + 1u;
            for (uint32_t w0 = 0; w0 < len && !hit; w0 += 64u) {
                const uint32_t p = w0 + (uint32_t)lane;
                const bool t = p < len && ld_byte(L + p) == '\t';
                const unsigned long long tb = __ballot(t);
                const uint32_t n_here = (uint32_t)__popcll(tb);
                if (seen + n_here >= want + 1u) {                                    // the tab that ENDS field `want` is tab number want + 1
                    unsigned long long m = tb; uint32_t k = seen;
                    uint32_t q = 0;
                    while (m) { const uint32_t bpos = (uint32_t)__builtin_ctzll(m); m &= m - 1ull; if (++k == want + 1u) { q = w0 + bpos; break; } }
                    hit = q + 1u < len;
                    break;
                }
                seen += n_here;
            }
            extra = hit;
        }
        // ---- the line's totals
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            t_cov += (uint32_t)__shfl_xor((int)t_cov, o);
#pragma unroll
            for (int x = 0; x < 4; ++x) t_n[x] += (uint32_t)__shfl_xor((int)t_n[x], o);
            ind |= (uint32_t)__shfl_xor((int)ind, o);
        }
        const unsigned long long errs = __ballot(err != 0u);
        if (errs || extra) {                                                        // (uniform) a domain error: the earliest line wins
            uint32_t code = extra ? 2u : 0u;
            if (errs) code = (uint32_t)__shfl((int)err, (int)__builtin_ctzll(errs));
            if (lane == 0) {
                const uint32_t old = atomicMin(&a.counters[1], li);
                if (li < old) a.counters[2] = code;                                 // (racy between lines; the host re-reads the winner's line to word the message)
            }
            continue;
        }
        // ---- gates and the calling rule (call_vC.cpp:545-552, 577-601)
        if ((int)t_cov < a.min_cov) continue;
        if ((int)(t_n[0] + t_n[1] + t_n[2] + t_n[3]) < a.min_snvs) continue;
        // reference character: field 2, tok[0] (call_vC.cpp:502) -- the allele that equals it AS A CHARACTER is skipped (:580)
        uint32_t refc = 0;
        {
            // fields 0 and 1 end at the first two tabs; lane 0 walks them (a few bytes)
            uint32_t p = 0, t = 0;
            while (p < len && t < 2u) { if (ld_byte(L + p) == '\t') ++t; ++p; }
            while (p < len && ld_byte(L + p) == ' ') ++p;                           // toksplit skips leading blanks
            if (t == 2u && p < len && ld_byte(L + p) != '\t') refc = ld_byte(L + p);
        }
        uint32_t pop = 0, indm = 0;
        const double lim = (double)(int)t_cov * a.min_frac;
        const char lower[4] = {'a', 'c', 'g', 't'};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (refc == (uint32_t)lower[x]) continue;
            if ((int)t_n[x] >= a.min_snvs && (double)t_n[x] >= lim) pop |= 1u << x;
            else if ((ind >> x) & 1u) indm |= 1u << x;
        }
        if (!(pop | indm)) continue;
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(&a.counters[0], 1u);
        slot = (uint32_t)__shfl((int)slot, 0);
        if (slot >= a.cap_rec) continue;
        if (lane == 0) {
            TextRec r;
            r.line = li; r.cov = t_cov; r.n[0] = t_n[0]; r.n[1] = t_n[1]; r.n[2] = t_n[2]; r.n[3] = t_n[3]; r.masks = pop | indm << 4; r.pad = 0;
            a.rec[slot] = r;
        }
        // (the lanes that wrote scratch[s] read it back: same lane, same addresses)
        for (uint32_t s = (uint32_t)lane; s < S; s += 64u) a.rec_samples[(uint64_t)slot * S + s] = scratch[s];
    }
    for (int o = 32; o >= 1; o >>= 1) my_bases += (uint64_t)__shfl_xor((long long)my_bases, o);
    if (lane == 0 && my_bases) atomicAdd(reinterpret_cast<unsigned long long *>(a.bases_parsed), (unsigned long long)my_bases);
}

namespace {

// toksplit on the host (call_vC.cpp:92-111) for the three leading fields of a called line
const char *toksplit_host(const char *s, const char *end, std::string &tok) {
    tok.clear();
    while (s < end && *s == ' ') ++s;
    while (s < end && *s != '\t') { if (tok.size() < TOK_CAP) tok.push_back(*s); ++s; }
    if (s < end && *s == '\t') ++s;
    return s;
}

struct DevBuf { void *p = nullptr; ~DevBuf() { if (p) dev_free(p); } };

}  // namespace

// One chunk of whole lines through the device; appends the called lines' records (line numbers relative to the chunk) to `recs` /
// `samples` in line order.
static int run_chunk(msnv_ctx *ctx, const char *text, const std::vector<uint64_t> &off, uint32_t S, const msnv_params &p,
                     std::vector<TextRec> &recs, std::vector<msnv_site_sample> &samples, double *ms_kernel, uint64_t *bases,
                     uint32_t *err_line, uint32_t *err_code) {
    hipStream_t st = (hipStream_t)ctx->stream;
    const uint32_t n_lines = (uint32_t)(off.size() - 1);
    const uint64_t n_bytes = off.back();
    *err_line = UINT32_MAX; *err_code = 0;
    if (n_lines == 0) return MSNV_OK;
    uint64_t acct = 0;
    DevBuf d_text, d_off, d_fs, d_scr, d_rec, d_rs, d_cnt;
    // a wavefront per line in flight; fewer when the per-wavefront rows (fstart + scratch) of a many-sample cohort get large
    const uint64_t row_bytes = (uint64_t)std::max<uint32_t>(S, 1) * (sizeof(uint32_t) + sizeof(msnv_site_sample));
    uint32_t waves = dev_resident_workgroups(8) * (TC_NT / 64);
    waves = (uint32_t)std::min<uint64_t>(waves, std::max<uint64_t>(256, (512ull << 20) / row_bytes));
    waves = std::max<uint32_t>(TC_NT / 64, std::min<uint32_t>(waves, (n_lines + 3u) & ~3u));
    waves = (waves + 3u) & ~3u;
    const uint64_t cap_rec = n_lines;                                               // every line may be called
    if (int rc = dev_alloc(&d_text.p, n_bytes + 64, &acct)) return rc;
    if (int rc = dev_alloc(&d_off.p, off.size() * sizeof(uint64_t), &acct)) return rc;
    if (int rc = dev_alloc(&d_fs.p, (uint64_t)waves * std::max<uint32_t>(S, 1) * sizeof(uint32_t), &acct)) return rc;
    if (int rc = dev_alloc(&d_scr.p, (uint64_t)waves * std::max<uint32_t>(S, 1) * sizeof(msnv_site_sample), &acct)) return rc;
    if (int rc = dev_alloc(&d_rec.p, cap_rec * sizeof(TextRec), &acct)) return rc;
    if (int rc = dev_alloc(&d_rs.p, std::max<uint64_t>(16, cap_rec * S * sizeof(msnv_site_sample)), &acct)) return rc;
    if (int rc = dev_alloc(&d_cnt.p, 64, &acct)) return rc;
    HIP_TRY(hipMemcpyAsync(d_text.p, text, n_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    uint32_t init[16] = {0};
    init[1] = UINT32_MAX;
    HIP_TRY(hipMemcpyAsync(d_cnt.p, init, sizeof init, hipMemcpyHostToDevice, st));
    TextArgs a;
    a.text = (const uint8_t *)d_text.p; a.line_off = (const uint64_t *)d_off.p; a.n_lines = n_lines; a.n_samples = S;
    a.min_cov = p.min_coverage; a.min_snvs = p.calling_threshold; a.min_frac = p.min_fraction;
    a.fstart = (uint32_t *)d_fs.p; a.scratch = (msnv_site_sample *)d_scr.p;
    a.rec = (TextRec *)d_rec.p; a.rec_samples = (msnv_site_sample *)d_rs.p; a.cap_rec = (uint32_t)cap_rec;
    a.counters = (uint32_t *)d_cnt.p; a.next_line = a.counters + 4; a.bases_parsed = reinterpret_cast<uint64_t *>(a.counters + 8);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipError_t he = hipEventRecord(e0, st);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(msnv_parse_pileup_lines, dim3(waves / (TC_NT / 64)), dim3(TC_NT), 0, st, a);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    uint32_t cnt[16];
    if (he == hipSuccess) he = hipMemcpyAsync(cnt, d_cnt.p, sizeof cnt, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    float t = 0;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(MSNV_EHIP, "mpileup text kernel: %s", hipGetErrorString(he));
    if (ms_kernel) *ms_kernel += t;
    if (bases) { uint64_t b; memcpy(&b, cnt + 8, 8); *bases += b; }
    if (cnt[1] != UINT32_MAX) { *err_line = cnt[1]; *err_code = cnt[2]; return MSNV_OK; }
    const uint32_t n = cnt[0];
    if (n > cap_rec) return fail(MSNV_EINVAL, "internal: %u records from %u lines", n, n_lines);
    std::vector<TextRec> r(n);
    std::vector<msnv_site_sample> rs((size_t)n * S);
    if (n) {
        if (int rc = dev_download(r.data(), d_rec.p, (uint64_t)n * sizeof(TextRec))) return rc;
        if (S) if (int rc = dev_download(rs.data(), d_rs.p, (uint64_t)n * S * sizeof(msnv_site_sample))) return rc;
    }
    // the wavefronts finish in any order: back to line order
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return r[x].line < r[y].line; });
    for (uint32_t i : order) {
        recs.push_back(r[i]);
        samples.insert(samples.end(), rs.begin() + (size_t)i * S, rs.begin() + (size_t)(i + 1) * S);
    }
    return MSNV_OK;
}

// snpCall over mpileup text that sits in memory.  stats (optional): [0] lines read (with the first), [1] samples, [2] called_SNPs lines,
// [3] indiv_called lines, [4] kernel microseconds, [5] text bytes through the kernel, [6] base-string characters parsed.
int text_call(msnv_ctx *ctx, const char *text, uint64_t n_text, const msnv_params &p, const char *ref_fasta, const char *ann_path,
              const char *called_path, const char *indiv_path, uint64_t stats[8]) {
    if (int rc = dev_set_device(ctx->device)) return rc;
    if (stats) for (int i = 0; i < 8; ++i) stats[i] = 0;
    // ---- lines (fgets: a line ends behind its '\n'; the last one may have none)
    std::vector<uint64_t> lines;                                                     // start of every line, then the end of the text
    for (uint64_t o = 0; o < n_text;) {
        lines.push_back(o);
        const void *nl = memchr(text + o, '\n', n_text - o);
        o = nl ? (uint64_t)((const char *)nl - text) + 1 : n_text;
    }
    lines.push_back(n_text);
    const uint64_t n_lines = lines.size() - 1;
    msnv_dataset tmp;                                                                // formatter / annotation state: names, records
    uint32_t S = 0;
    if (n_lines > 0) {                                                               // call_vC.cpp:423-431: the first line counts the samples and is dropped
        const uint64_t b = lines[0], e = std::min<uint64_t>(lines[1], b + strnlen(text + b, lines[1] - b));
        uint32_t tabs = 0;
        for (uint64_t i = b; i < e; ++i) tabs += text[i] == '\t';
        const int ns = (int)(tabs + 1u - 3u) / 3;
        S = ns > 0 ? (uint32_t)ns : 0u;
    }
    if (S >= 16384) return fail(MSNV_EDOMAIN, "more than 16383 samples are not supported");
    tmp.samples.resize(S);
    std::vector<TextRec> recs;
    std::vector<msnv_site_sample> samples;
    std::vector<uint64_t> rec_line;                                                  // absolute line of every record
    double ms = 0; uint64_t bases = 0;
    // ---- chunks of whole lines (<= ~256 MB of text each, MSNV_TEXT_CHUNK overrides: tests)
    uint64_t chunk_bytes = 256ull << 20;
    if (const char *e = getenv("MSNV_TEXT_CHUNK")) chunk_bytes = std::max<uint64_t>(1, (uint64_t)atoll(e));
    for (uint64_t l0 = 1; l0 < n_lines;) {
        uint64_t l1 = l0 + 1;
        while (l1 < n_lines && lines[l1 + 1] - lines[l0] <= chunk_bytes && l1 - l0 < (1u << 30)) ++l1;
        if (lines[l1] - lines[l0] >= (1ull << 32)) return fail(MSNV_EDOMAIN, "an mpileup line of 4 GB or more");
        std::vector<uint64_t> off(l1 - l0 + 1);
        for (uint64_t i = l0; i <= l1; ++i) off[i - l0] = lines[i] - lines[l0];
        const size_t first = recs.size();
        uint32_t err_line, err_code;
        if (int rc = run_chunk(ctx, text + lines[l0], off, S, p, recs, samples, &ms, &bases, &err_line, &err_code)) return rc;
        if (err_line != UINT32_MAX) {
            const uint64_t ln = l0 + err_line;
            if ((err_code & 0xffu) == 2u)
                return fail(MSNV_EDOMAIN, "mpileup line %llu holds more samples than the first line (%u; reference: out-of-bounds write)", (unsigned long long)(ln + 1), S);
            const unsigned c = (err_code >> 8) & 0xffu;
            return fail(MSNV_EDOMAIN, "mpileup line %llu: pileup symbol '%c' (0x%02x): the reference dereferences an empty vector (SIGSEGV)",
                        (unsigned long long)(ln + 1), c >= 32 && c < 127 ? (char)c : '?', c);
        }
        for (size_t i = first; i < recs.size(); ++i) rec_line.push_back(l0 + recs[i].line);
        l0 = l1;
    }
    // ---- the three leading fields of the called lines; contigs are numbered in order of appearance
    std::unordered_map<std::string, int32_t> contig_id;
    std::vector<int64_t> max_pos;
    std::string tok, name;
    tmp.sites.resize(recs.size());
    tmp.site_samples = std::move(samples);
    for (size_t i = 0; i < recs.size(); ++i) {
        const char *s = text + lines[rec_line[i]], *e = text + lines[rec_line[i] + 1];
        e = s + strnlen(s, (size_t)(e - s));
        if (e > s) --e;                                                              // call_vC.cpp:475
        s = toksplit_host(s, e, name);
        s = toksplit_host(s, e, tok);
        const long lp = atol(tok.c_str()) - 1;                                       // :499
        s = toksplit_host(s, e, tok);
        auto it = contig_id.find(name);
        if (it == contig_id.end()) { it = contig_id.emplace(name, (int32_t)tmp.names.size()).first; tmp.names.push_back(name); max_pos.push_back(0); }
        msnv_site &o = tmp.sites[i];
        o.tid = it->second; o.pos = (int32_t)(int)lp; o.cov = recs[i].cov;
        for (int x = 0; x < 4; ++x) o.n[x] = recs[i].n[x];
        o.pop_mask = (uint8_t)(recs[i].masks & 15u); o.ind_mask = (uint8_t)(recs[i].masks >> 4);
        o.refchar = tok.empty() ? 0 : (uint8_t)tok[0];                                // :502
        o.dropped = 0;
        max_pos[(size_t)o.tid] = std::max<int64_t>(max_pos[(size_t)o.tid], (int64_t)o.pos);
    }
    uint64_t n_pop = 0, n_ind = 0;
    for (const msnv_site &o : tmp.sites) { n_pop += o.pop_mask != 0; n_ind += o.ind_mask != 0; }
    if (stats) { stats[0] = n_lines; stats[1] = S; stats[2] = n_pop; stats[3] = n_ind; stats[4] = (uint64_t)(ms * 1000.0); stats[5] = n_lines > 1 ? n_text - lines[1] : 0; stats[6] = bases; }
    if (!(ann_path && ref_fasta)) return write_calls_text(tmp, called_path, indiv_path, nullptr, nullptr);     // call_vC.cpp:448
    // ---- gene / codon annotation on the device: the called positions in a linear position space of their own
    Annotation an;
    if (int rc = load_annotation(ann_path, ref_fasta, an)) return rc;
    tmp.lengths.resize(tmp.names.size());
    tmp.tile_base.resize(tmp.names.size());
    uint64_t nt = 0;
    for (size_t c = 0; c < tmp.names.size(); ++c) {
        int64_t len = max_pos[c] + 1;
        auto g = an.genome.find(tmp.names[c]);
        if (g != an.genome.end()) len = std::max<int64_t>(len, (int64_t)g->second.size());
        tmp.lengths[c] = len;
        tmp.tile_base[c] = (uint32_t)nt;
        nt += ((uint64_t)len + TILE - 1) / TILE;
        if (nt * TILE >= 0xffffffffull) return fail(MSNV_EDOMAIN, "the called contigs span more than 2^32 positions: split the input by contig");
    }
    for (const msnv_site &o : tmp.sites)
        if (o.pos < 0) return fail(MSNV_EDOMAIN, "position %d on %s (reference: undefined behaviour)", o.pos + 1, tmp.names[(size_t)o.tid].c_str());
    DeviceCols d;
    struct Guard { DeviceCols &d; ~Guard() { dev_free_all(d); } } guard{d};
    AnnHost h;
    if (int rc = ann_build(tmp, an, h)) return rc;
    if (int rc = dev_ann_upload(d, h)) return rc;
    std::vector<std::string> gene_names;
    ann_gene_names(an, tmp.names, gene_names);
    const uint32_t n = (uint32_t)tmp.sites.size();
    std::vector<SiteRec> sr(std::max<uint32_t>(n, 1));
    std::vector<uint8_t> fl(std::max<uint32_t>(n, 1));
    for (uint32_t i = 0; i < n; ++i) {
        const msnv_site &o = tmp.sites[i];
        sr[i].gpos = tmp.tile_base[(size_t)o.tid] * TILE + (uint32_t)o.pos; sr[i].cov = o.cov;
        for (int x = 0; x < 4; ++x) sr[i].n[x] = o.n[x];
        fl[i] = (uint8_t)(o.pop_mask | o.ind_mask << 4);
    }
    if (int rc = dev_alloc((void **)&d.sites, sr.size() * sizeof(SiteRec), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&d.site_flags, fl.size(), &d.device_bytes)) return rc;
    if (int rc = dev_upload(d.sites, sr.data(), sr.size() * sizeof(SiteRec))) return rc;
    if (int rc = dev_upload(d.site_flags, fl.data(), fl.size())) return rc;
    uint32_t err[2] = {UINT32_MAX, UINT32_MAX};
    if (int rc = dev_annotate(d, n, UINT32_MAX, ctx->stream, nullptr, err)) return rc;
    for (int k = 0; k < 2; ++k) {
        if (err[k] == UINT32_MAX) continue;
        size_t c = tmp.names.size() - 1;
        while (c > 0 && (uint64_t)tmp.tile_base[c] * TILE > err[k]) --c;
        return fail(MSNV_EDOMAIN, k == 0 ? "contig %s has genes but no FASTA record (position %u; reference: undefined behaviour)"
                                         : "codon at %s:%u runs past the contig end (reference: undefined behaviour)",
                    tmp.names[c].c_str(), err[k] - tmp.tile_base[c] * TILE + 1u);
    }
    std::vector<msnv_site_ann> ann(std::max<uint32_t>(n, 1));
    if (n) if (int rc = dev_download(ann.data(), d.ann.out, (uint64_t)n * sizeof(msnv_site_ann))) return rc;
    return write_calls_text(tmp, called_path, indiv_path, ann.data(), &gene_names);
}

}  // namespace msnv
