// metasnv_amd/csrc/hostio.cpp -- host-side file formats of the pileup path.
//
// The reference decodes BAM through htslib (qaCompute.cpp:26-27,276,367,441) and lets
// samtools read BAM/FASTA/BED (metaSNV.py:160-165).  htslib is not available in the build
// image, so this is a minimal BGZF/BAM reader+writer on zlib
// written from the SAM/BAM specification (SAMv1 section 4), plus FASTA and 3-column BED.
#include "msnv_internal.h"

#include <zlib.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

#include <sys/mman.h>

namespace msnv {

// Host threads a call may use when the caller names none: the hardware's, capped by the CPU time the container is given (cgroup v2 `cpu.max`,
// v1 `cpu.cfs_quota_us`): on a 256-thread node that grants 16 cores, 32 threads read + inflate 160 BAMs in 0.29-0.34 s, 128 in 0.37-0.49 s
// (profiles/stage_threads.py).  Twice the quota: a thread that waits for a page fault or a read leaves its share to another.
static double g_quota_cores = 0;       // CPU time the container is given, in cores (0: no quota)
static std::chrono::steady_clock::time_point g_fin_trace_t;
static bool fin_trace_on() { static const bool on = [] { const char *e = getenv("MSNV_FINALIZE_TRACE"); return e && e[0] == '1'; }(); return on; }
void fin_trace_reset() { if (fin_trace_on()) g_fin_trace_t = std::chrono::steady_clock::now(); }
void fin_trace(const char *what) {
    if (!fin_trace_on()) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[finalize] %-44s %9.3f ms\n", what, 1e3 * std::chrono::duration<double>(now - g_fin_trace_t).count());
    g_fin_trace_t = now;
}

unsigned msnv_default_threads() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        double cores = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64]; long long per = 0;
            if (fscanf(f, "%63s %lld", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0) cores = (double)atoll(q) / (double)per;
            fclose(f);
        } else {
            long long quota = -1, per = 0;
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (quota > 0 && per > 0) cores = (double)quota / (double)per;
        }
        g_quota_cores = cores;
        if (cores >= 1.0) hw = std::min<unsigned>(hw, (unsigned)(2.0 * cores + 0.5));
        return std::max(1u, hw);
    }();
    return n;
}

void msnv_drop_pages(void *p, size_t bytes) {
#if defined(MADV_DONTNEED)
    const uintptr_t a = ((uintptr_t)p + 4095u) & ~(uintptr_t)4095u, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095u;
    if (e > a) (void)madvise((void *)a, e - a, MADV_DONTNEED);
#else
    (void)p; (void)bytes;
#endif
}

void msnv_advise_huge(void *p, size_t bytes) {
#if defined(MADV_HUGEPAGE)
    static const int mode = [] { const char *e = getenv("MSNV_HUGE"); return e ? atoi(e) : 1; }();      // 0: small pages; 2: huge pages, populated now (experiments: profiles/stage_threads.py)
    if (mode == 0) return;
    (void)madvise(p, bytes, MADV_HUGEPAGE);
#if defined(MADV_POPULATE_WRITE)
    if (mode == 2) (void)madvise(p, bytes, MADV_POPULATE_WRITE);
#endif
#else
    (void)p; (void)bytes;
#endif
}


// ------------------------------------------------------------------------------ error state
static thread_local char t_err[1024];

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
    fprintf(stderr, "libmsnv: %s\n", t_err);
    return code;
}
int fail_quiet(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { t_err[0] = 0; }

}  // namespace msnv

extern "C" const char *msnv_last_error(void) { return msnv::t_err; }
namespace msnv { uint64_t inflate_zlib_fallbacks(); }
extern "C" int msnv_host_stats(uint64_t *zlib_fallbacks) { if (zlib_fallbacks) *zlib_fallbacks = msnv::inflate_zlib_fallbacks(); return MSNV_OK; }

namespace msnv {
static std::atomic<uint64_t> g_timer_us[HT_N];
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void host_timer_add(int which, double seconds) { if (which >= 0 && which < HT_N && seconds > 0) g_timer_us[which].fetch_add((uint64_t)(seconds * 1e6)); }
HostTimerScope::HostTimerScope(int w) : which(w), t0(now_s()) {}
HostTimerScope::~HostTimerScope() { host_timer_add(which, now_s() - t0); }
}  // namespace msnv
extern "C" int msnv_host_timers(double *seconds, int32_t n, int32_t reset) {
    for (int i = 0; i < n && seconds; ++i) seconds[i] = i < msnv::HT_N ? (double)msnv::g_timer_us[i].load() * 1e-6 : 0.0;
    if (reset) for (int i = 0; i < msnv::HT_N; ++i) msnv::g_timer_us[i].store(0);
    return MSNV_OK;
}

namespace msnv {

static int read_file(const char *path, ByteBuf &buf, size_t &n_out) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MSNV_EIO, "cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return fail(MSNV_EIO, "cannot stat %s", path); }
    if (!buf.alloc((size_t)n + 16)) { fclose(f); return fail(MSNV_ENOMEM, "out of memory reading %s (%ld bytes)", path, n); }
    if (n && fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); return fail(MSNV_EIO, "short read on %s", path); }
    fclose(f);
    memset(buf.data() + n, 0, 16);       // the inflate fast path loads 8 bytes at a time
    n_out = (size_t)n;
    return MSNV_OK;
}
// ... into a buffer that lives with the calling thread and only grows: a host thread that decodes one BAM after the other (160 of 8.5 MB
// each on the benchmark shape) maps, faults in and unmaps its input once instead of once per file -- every unmap takes the lock of the
// address space away from the page faults of all other decode threads.
static int read_file_reuse(const char *path, const uint8_t *&data, size_t &n_out) {
    static thread_local ByteBuf tl;
    static thread_local size_t tl_cap = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MSNV_EIO, "cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return fail(MSNV_EIO, "cannot stat %s", path); }
    if ((size_t)n + 16 > tl_cap) {
        const size_t want = (size_t)n + (size_t)n / 4 + 16;
        tl_cap = 0;
        if (!tl.alloc(want)) { fclose(f); return fail(MSNV_ENOMEM, "out of memory reading %s (%ld bytes)", path, n); }
        tl_cap = want;
    }
    if (n && fread(tl.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); return fail(MSNV_EIO, "short read on %s", path); }
    fclose(f);
    memset(tl.data() + n, 0, 16);
    data = tl.data(); n_out = (size_t)n;
    return MSNV_OK;
}
static int read_file(const char *path, std::vector<uint8_t> &buf) {      // small files (FASTA / BED / annotation readers)
    ByteBuf b; size_t n = 0;
    if (int rc = read_file(path, b, n)) return rc;
    try { buf.assign(b.data(), b.data() + n); } catch (const std::exception &) { return fail(MSNV_ENOMEM, "out of memory reading %s", path); }
    return MSNV_OK;
}

// ------------------------------------------------------------------------------ BGZF
// A BGZF block is a gzip member with an extra field "BC" holding BSIZE-1 (SAMv1 4.1).
using BlockRef = BgzfBlock;                  // msnv_internal.h: {in_off, in_size, out_size, out_off}

struct ConstBytes { const uint8_t *p; size_t n; const uint8_t *data() const { return p; } size_t size() const { return n; } };
static int bgzf_index(const ConstBytes in, const char *path, std::vector<BlockRef> &blocks, uint64_t &total_out) {
    uint64_t off = 0;
    total_out = 0;
    while (off < in.size()) {
        if (in.size() - off < 18) return fail(MSNV_EFORMAT, "%s: truncated BGZF block header", path);
        const uint8_t *p = in.data() + off;
        if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return fail(MSNV_EFORMAT, "%s: not a BGZF file", path);
        uint32_t xlen = p[10] | p[11] << 8;
        if (12ull + xlen > in.size() - off) return fail(MSNV_EFORMAT, "%s: truncated BGZF extra field", path);      // nothing of the file is trusted
        uint32_t bsize = 0;
        bool found = false;
        uint32_t x = 0;
        while (x + 4 <= xlen) {
            const uint8_t *e = p + 12 + x;
            uint32_t slen = e[2] | e[3] << 8;
            if (x + 4 + slen > xlen) break;
            if (e[0] == 'B' && e[1] == 'C' && slen == 2) { bsize = (e[4] | e[5] << 8) + 1u; found = true; }
            x += 4 + slen;
        }
        if (!found || off + bsize > in.size() || bsize < 12 + xlen + 8) return fail(MSNV_EFORMAT, "%s: bad BGZF block", path);
        uint32_t isize = ld_u32(p + bsize - 4);
        if (isize > 65536u) return fail(MSNV_EFORMAT, "%s: BGZF block claims %u uncompressed bytes (the format allows 65536)", path, isize);
        BlockRef b{off + 12 + xlen, bsize - 12 - xlen - 8, isize, total_out};
        blocks.push_back(b);
        total_out += isize;
        off += bsize;
    }
    return MSNV_OK;
}

bool inflate_raw(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out);          // inflate.cpp
bool inflate_raw_bmi2(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out);     // the same, compiled with -mbmi2
static std::atomic<uint64_t> g_zlib_fallbacks{0};
uint64_t inflate_zlib_fallbacks() { return g_zlib_fallbacks.load(); }

static bool inflate_block(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out) {
    if (n_out == 0) return true;
    static const bool use_zlib = [] { const char *e = getenv("MSNV_INFLATE"); return e && e[0] == 'z'; }();     // MSNV_INFLATE=zlib: A/B of the two decoders
    static const bool bmi2 = __builtin_cpu_supports("bmi2") != 0;
    if (!use_zlib && (bmi2 ? inflate_raw_bmi2(src, n_in, dst, n_out) : inflate_raw(src, n_in, dst, n_out))) return true;
    // zlib decides about anything the fast decoder refuses (a malformed block fails here too); counted: a well-formed file never gets here
    if (!use_zlib) g_zlib_fallbacks.fetch_add(1);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = n_in;
    zs.next_out = dst; zs.avail_out = n_out;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}

// The two halves of bgzf_read_all for callers that inflate elsewhere (the device: api.cpp / inflate_k.hip)
int bgzf_load(const char *path, ByteBuf &comp, size_t &n_in, std::vector<BgzfBlock> &blocks, uint64_t &total_out) {
    if (int rc = read_file(path, comp, n_in)) return rc;
    blocks.clear();
    return bgzf_index(ConstBytes{comp.data(), n_in}, path, blocks, total_out);
}
bool bgzf_inflate_block_host(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out) { return inflate_block(src, n_in, dst, n_out); }
int bgzf_index_bytes(const uint8_t *data, size_t n, const char *path, std::vector<BgzfBlock> &blocks, uint64_t &total_out) {
    blocks.clear();
    return bgzf_index(ConstBytes{data, n}, path, blocks, total_out);
}

int bgzf_read_all(const char *path, ByteBuf &out, int threads) {
    ByteBuf inb; size_t n_in = 0;
    const uint8_t *in_p = nullptr;
    {
        HostTimerScope ts(HT_READ);
        if (threads <= 1) { if (int rc = read_file_reuse(path, in_p, n_in)) return rc; }      // (one of many files of this thread: its input buffer is reused)
        else { if (int rc = read_file(path, inb, n_in)) return rc; in_p = inb.data(); }
    }
    const ConstBytes in{in_p, n_in};
    std::vector<BlockRef> blocks;
    uint64_t total = 0;
    if (int rc = bgzf_index(in, path, blocks, total)) return rc;
    if (!out.alloc(total)) return fail(MSNV_ENOMEM, "%s: out of memory for %llu inflated bytes", path, (unsigned long long)total);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    // every block's output is checked against the CRC-32 of its BGZF trailer, as htslib does (MSNV_INFLATE_CHECK=0 skips it: benchmarks)
    const uint32_t check_every = inflate_check_every();
    auto worker = [&]() {
        HostTimerScope ts(HT_INFLATE_HOST);
        for (;;) {
            size_t i = next.fetch_add(16);
            if (i >= blocks.size()) break;
            size_t e = std::min(blocks.size(), i + 16);
            for (; i < e; ++i) {
                const BlockRef &b = blocks[i];
                if (!inflate_block(in.data() + b.in_off, b.in_size, out.data() + b.out_off, b.out_size)) { bad = true; continue; }
                if (check_every && i % check_every == 0 && b.out_size) {
                    const uint8_t *t = in.data() + b.in_off + b.in_size;
                    if (bgzf_crc32(out.data() + b.out_off, b.out_size) != ld_u32(t)) bad = true;
                }
            }
        }
    };
    if (threads <= 1 || blocks.size() < 64) worker();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    if (bad) return fail(MSNV_EFORMAT, "%s: BGZF inflate failed (malformed DEFLATE stream or CRC-32 mismatch)", path);
    return MSNV_OK;
}

int bgzf_write_all(const char *path, const uint8_t *data, uint64_t n, int level) {
    FILE *f = fopen(path, "wb");
    if (!f) return fail(MSNV_EIO, "cannot create %s", path);
    const uint32_t CHUNK = 0xff00;
    std::vector<uint8_t> comp(CHUNK + 1024);
    uint64_t off = 0;
    bool wrote_eof = false;
    while (!wrote_eof) {
        uint32_t m = (uint32_t)std::min<uint64_t>(CHUNK, n - off);
        if (m == 0) wrote_eof = true;      // the empty block is the BGZF EOF marker
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { fclose(f); return fail(MSNV_ENOMEM, "deflateInit2 failed"); }
        zs.next_in = const_cast<uint8_t *>(data + off); zs.avail_in = m;
        zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        int rc = deflate(&zs, Z_FINISH);
        uint32_t clen = (uint32_t)(comp.size() - zs.avail_out);
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) { fclose(f); return fail(MSNV_EIO, "deflate failed"); }
        uint32_t bsize = 18 + clen + 8;
        uint8_t h[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
        uint32_t crc = bgzf_crc32(data + off, (uint32_t)m);
        uint8_t t[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), (uint8_t)m, (uint8_t)(m >> 8), (uint8_t)(m >> 16), (uint8_t)(m >> 24)};
        if (fwrite(h, 1, 18, f) != 18 || fwrite(comp.data(), 1, clen, f) != clen || fwrite(t, 1, 8, f) != 8) { fclose(f); return fail(MSNV_EIO, "write failed on %s", path); }
        off += m;
    }
    fclose(f);
    return MSNV_OK;
}

// ------------------------------------------------------------------------------ BAM
template <typename Bytes>
static int bam_parse_header(const Bytes &u, const char *path, BamHeader &hdr, uint64_t &rec_off) {
    if (u.size() < 12 || memcmp(u.data(), "BAM\1", 4) != 0) return fail(MSNV_EFORMAT, "%s: not a BAM file", path);
    uint32_t l_text = ld_u32(u.data() + 4);
    if (8ull + l_text + 4 > u.size()) return fail(MSNV_EFORMAT, "%s: truncated BAM header", path);
    hdr.text.assign((const char *)u.data() + 8, l_text);
    while (!hdr.text.empty() && hdr.text.back() == '\0') hdr.text.pop_back();
    uint64_t off = 8ull + l_text;
    uint32_t n_ref = ld_u32(u.data() + off);
    off += 4;
    hdr.names.clear(); hdr.lengths.clear();
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (off + 4 > u.size()) return fail(MSNV_EFORMAT, "%s: truncated BAM contig table", path);
        uint32_t l_name = ld_u32(u.data() + off);
        off += 4;
        if (off + l_name + 4 > u.size() || l_name == 0) return fail(MSNV_EFORMAT, "%s: truncated BAM contig table", path);
        hdr.names.emplace_back((const char *)u.data() + off, l_name - 1);
        off += l_name;
        hdr.lengths.push_back((int64_t)ld_u32(u.data() + off));
        off += 4;
    }
    rec_off = off;
    return MSNV_OK;
}

int bam_parse_header_bytes(const uint8_t *data, size_t n, const char *path, BamHeader &hdr, uint64_t &rec_off) {
    return bam_parse_header(ConstBytes{data, n}, path, hdr, rec_off);
}

int bam_read(const char *path, BamHeader &hdr, ByteBuf &buf, uint64_t &rec_off, int threads) {
    if (int rc = bgzf_read_all(path, buf, threads)) return rc;
    return bam_parse_header(buf, path, hdr, rec_off);
}

// The BAM header of a file whose compressed bytes are in memory (blocks: bgzf_index_bytes): headers are small, the leading blocks are
// inflated until the contig table is complete.  rec_off = offset of the first alignment record in the inflated stream.
int bam_header_from_blocks(const uint8_t *file_bytes, const std::vector<BgzfBlock> &blocks, const char *path, BamHeader &hdr, uint64_t &rec_off) {
    std::vector<uint8_t> u;
    for (const BlockRef &b : blocks) {
        size_t old = u.size();
        u.resize(old + b.out_size);
        if (!inflate_block(file_bytes + b.in_off, b.in_size, u.data() + old, b.out_size)) return fail(MSNV_EFORMAT, "%s: BGZF inflate failed", path);
        // try to parse
        if (u.size() >= 12 && memcmp(u.data(), "BAM\1", 4) == 0) {
            uint32_t l_text = ld_u32(u.data() + 4);
            uint64_t off = 8ull + l_text;
            if (off + 4 <= u.size()) {
                uint32_t n_ref = ld_u32(u.data() + off);
                off += 4;
                bool ok = true;
                for (uint32_t i = 0; i < n_ref && ok; ++i) {
                    if (off + 4 > u.size()) { ok = false; break; }
                    uint32_t l_name = ld_u32(u.data() + off);
                    off += 4ull + l_name + 4;
                    if (off > u.size()) ok = false;
                }
                if (ok) return bam_parse_header(u, path, hdr, rec_off);
            }
        }
    }
    return bam_parse_header(u, path, hdr, rec_off);
}

int bam_read_header(const char *path, BamHeader &hdr) {
    ByteBuf inb; size_t n_in = 0;
    if (int rc = read_file(path, inb, n_in)) return rc;
    const ConstBytes in{inb.data(), n_in};
    std::vector<BlockRef> blocks;
    uint64_t total = 0;
    if (int rc = bgzf_index(in, path, blocks, total)) return rc;
    uint64_t ro;
    return bam_header_from_blocks(inb.data(), blocks, path, hdr, ro);
}

int bam_write(const char *path, const BamHeader &hdr, const uint8_t *records, uint64_t n, int level) {
    std::vector<uint8_t> u;
    auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) u.push_back((uint8_t)(v >> (8 * i))); };
    u.insert(u.end(), {'B', 'A', 'M', 1});
    put32((uint32_t)hdr.text.size());
    u.insert(u.end(), hdr.text.begin(), hdr.text.end());
    put32((uint32_t)hdr.names.size());
    for (size_t i = 0; i < hdr.names.size(); ++i) {
        put32((uint32_t)hdr.names[i].size() + 1);
        u.insert(u.end(), hdr.names[i].begin(), hdr.names[i].end());
        u.push_back(0);
        put32((uint32_t)hdr.lengths[i]);
    }
    u.insert(u.end(), records, records + n);
    return bgzf_write_all(path, u.data(), u.size(), level);
}

bool rec_parse(const uint8_t *p, uint64_t avail, RecView &r) {
    if (avail < 36) return false;
    int32_t bs = (int32_t)ld_u32(p);
    if (bs < 32 || (uint64_t)bs + 4 > avail) return false;
    r.tid = (int32_t)ld_u32(p + 4);
    r.pos = (int32_t)ld_u32(p + 8);
    uint32_t l_name = p[12];
    r.mapq = p[13];
    r.n_cigar = (int32_t)(p[16] | p[17] << 8);
    r.flag = (uint16_t)(p[18] | p[19] << 8);
    r.l_seq = (int32_t)ld_u32(p + 20);
    if (r.l_seq < 0) return false;
    uint64_t need = 36ull + l_name + 4ull * r.n_cigar + ((uint64_t)r.l_seq + 1) / 2 + (uint64_t)r.l_seq;
    if (need > (uint64_t)bs + 4) return false;
    r.mtid = (int32_t)ld_u32(p + 24); r.mpos = (int32_t)ld_u32(p + 28); r.tlen = (int32_t)ld_u32(p + 32);
    r.qname = p + 36; r.l_name = l_name;
    r.cigar = p + 36 + l_name;
    r.seq = r.cigar + 4ull * r.n_cigar;
    r.qual = r.seq + ((uint64_t)r.l_seq + 1) / 2;
    r.size = (uint32_t)bs + 4;
    // A CIGAR of more than 65535 operations travels in the auxiliary field CG:B,I behind a placeholder `<l_seq>S<ref_len>N`; htslib puts it
    // back when it reads the record (sam.c bam_tag2cigar, called by bam_read1 [EXT]), so samtools mpileup and qaCompute (sam_read1,
    // qaCompute.cpp:441) both walk the real one.  Same conditions here: mapped, first operation a soft clip of exactly l_seq bases, a CG
    // field of type B,I / B,i with at least as many operations as the placeholder (devpack.hip: rec_load does the same on the device).
    if (r.n_cigar > 0 && r.tid >= 0 && r.pos >= 0) {
        const uint32_t c0 = ld_u32(r.cigar);
        if ((c0 & 15u) == C_S && (int32_t)(c0 >> 4) == r.l_seq) {
            const uint8_t *aux = r.qual + r.l_seq, *end = p + r.size;
            while (aux + 3 <= end) {
                const uint8_t t = aux[2], *v = aux + 3;
                uint64_t sz;
                if (t == 'A' || t == 'c' || t == 'C') sz = 1;
                else if (t == 's' || t == 'S') sz = 2;
                else if (t == 'i' || t == 'I' || t == 'f') sz = 4;
                else if (t == 'd') sz = 8;
                else if (t == 'Z' || t == 'H') { const uint8_t *q = v; while (q < end && *q) ++q; sz = (uint64_t)(q - v) + 1; }
                else if (t == 'B') {
                    if (v + 5 > end) break;
                    const uint64_t es = (v[0] == 'c' || v[0] == 'C') ? 1 : (v[0] == 's' || v[0] == 'S') ? 2 : 4, n = ld_u32(v + 1);
                    if (aux[0] == 'C' && aux[1] == 'G') {
                        if ((v[0] == 'I' || v[0] == 'i') && n >= (uint64_t)r.n_cigar && n < (1ull << 29) && v + 5 + 4 * n <= end) { r.cigar = v + 5; r.n_cigar = (int32_t)n; }
                        break;
                    }
                    sz = 5 + es * n;
                } else break;
                if (aux[0] == 'C' && aux[1] == 'G') break;
                if (v + sz > end) break;
                aux = v + sz;
            }
        }
    }
    return true;
}

uint8_t nt16_of_char(unsigned char c) {
    switch (c) {
        case '=': return 0;
        case '0': return 1; case '1': return 2; case '2': return 4; case '3': return 8;
        case 'A': case 'a': return 1;  case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;  case 'T': case 't': return 8;
        case 'M': case 'm': return 3;  case 'R': case 'r': return 5;
        case 'S': case 's': return 6;  case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;  case 'Y': case 'y': return 10;
        case 'H': case 'h': return 11; case 'K': case 'k': return 12;
        case 'D': case 'd': return 13; case 'B': case 'b': return 14;
        default: return 15;
    }
}

// ------------------------------------------------------------------------------ FASTA / BED
int fasta_read(const char *path, std::vector<FastaSeq> &out) {
    std::vector<uint8_t> buf;
    if (int rc = read_file(path, buf)) return rc;
    out.clear();
    size_t i = 0, n = buf.size();
    while (i < n) {
        const void *nl = memchr(buf.data() + i, '\n', n - i);
        const size_t e = nl ? (size_t)((const uint8_t *)nl - buf.data()) : n;
        size_t le = e;
        if (le > i && buf[le - 1] == '\r') --le;
        if (buf[i] == '>') {
            size_t k = i + 1;
            while (k < le && !isspace(buf[k])) ++k;
            out.push_back(FastaSeq{std::string((const char *)buf.data() + i + 1, k - i - 1), std::string()});
        } else if (!out.empty()) {
            // faidx keeps every isgraph() character of a sequence line
            std::string &s = out.back().seq;
            bool plain = true;                                   // (nearly every line: appended in one piece)
            for (size_t k = i; k < le; ++k) plain &= (buf[k] > 0x20 && buf[k] < 0x7f);
            if (plain) s.append((const char *)buf.data() + i, le - i);
            else for (size_t k = i; k < le; ++k) if (isgraph(buf[k])) s.push_back((char)buf[k]);
        }
        i = e + 1;
    }
    return MSNV_OK;
}

int bed_read(const char *path, std::vector<BedRegion> &out) {
    FILE *f = fopen(path, "r");
    if (!f) return fail(MSNV_EIO, "cannot open %s", path);
    char line[4096];
    out.clear();
    while (fgets(line, sizeof line, f)) {
        char name[2048];
        long long b, e;
        if (line[0] == '#' || line[0] == '\n') continue;
        if (sscanf(line, "%2047s %lld %lld", name, &b, &e) != 3) { fclose(f); return fail(MSNV_EFORMAT, "%s: expected `name beg end` lines", path); }
        out.push_back(BedRegion{name, (int64_t)b, (int64_t)e});
    }
    fclose(f);
    return MSNV_OK;
}

}  // namespace msnv

// ------------------------------------------------------------------------------ C ABI (I/O helpers)
using namespace msnv;

extern "C" int32_t msnv_host_cores(void) {
    (void)msnv_default_threads();
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int32_t)(g_quota_cores >= 1.0 ? std::min<double>(hw, g_quota_cores + 0.5) : hw);
}

extern "C" int msnv_bam_write_bed_header(const char *bam_path, const char *out_path) {
    clear_error();
    BamHeader h;
    if (int rc = bam_read_header(bam_path, h)) return rc;
    FILE *f = fopen(out_path, "w");
    if (!f) return fail(MSNV_EIO, "cannot create %s", out_path);
    // metaSNV.py:85-93 keeps only @SQ lines that have exactly three tab-separated fields and
    // strips "SN:" / "LN:"; it skips the first line of `samtools view -H` unconditionally.
    size_t i = 0;
    int lineno = 0;
    const std::string &t = h.text;
    while (i < t.size()) {
        size_t e = t.find('\n', i);
        if (e == std::string::npos) e = t.size();
        std::string line = t.substr(i, e - i);
        i = e + 1;
        if (lineno++ == 0) continue;
        while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back();
        std::vector<std::string> fld;
        size_t s = 0;
        for (;;) {
            size_t tb = line.find('\t', s);
            fld.push_back(line.substr(s, tb == std::string::npos ? std::string::npos : tb - s));
            if (tb == std::string::npos) break;
            s = tb + 1;
        }
        if (fld.size() != 3 || fld[0] != "@SQ") continue;
        auto strip = [](std::string v, const char *tag) { size_t p = v.find(tag); if (p != std::string::npos) v.erase(p, strlen(tag)); return v; };
        fprintf(f, "%s\t1\t%s\n", strip(fld[1], "SN:").c_str(), strip(fld[2], "LN:").c_str());
    }
    fclose(f);
    return MSNV_OK;
}

extern "C" int msnv_bam_read(const char *bam_path, msnv_bam_data *out) {
    clear_error();
    if (!bam_path || !out) return fail(MSNV_EINVAL, "msnv_bam_read: NULL argument");
    memset(out, 0, sizeof *out);
    BamHeader h;
    ByteBuf buf; uint64_t rec_off = 0;
    if (int rc = bam_read(bam_path, h, buf, rec_off, 1)) return rc;
    struct { const uint8_t *p; size_t n; const uint8_t *data() const { return p; } size_t size() const { return n; } } rec{buf.data() + rec_off, buf.size() - (size_t)rec_off};
    out->n_contigs = (int32_t)h.names.size();
    out->names = (char **)calloc(h.names.size() + 1, sizeof(char *));
    out->lengths = (int64_t *)calloc(h.names.size() + 1, sizeof(int64_t));
    for (size_t i = 0; i < h.names.size(); ++i) { out->names[i] = strdup(h.names[i].c_str()); out->lengths[i] = h.lengths[i]; }
    out->records = (uint8_t *)malloc(rec.size() + 1);
    memcpy(out->records, rec.data(), rec.size());
    out->n_record_bytes = rec.size();
    out->header_text = strdup(h.text.c_str());
    return MSNV_OK;
}

extern "C" int msnv_bam_read_header(const char *bam_path, msnv_bam_data *out) {
    clear_error();
    if (!bam_path || !out) return fail(MSNV_EINVAL, "msnv_bam_read_header: NULL argument");
    memset(out, 0, sizeof *out);
    BamHeader h;
    if (int rc = bam_read_header(bam_path, h)) return rc;        // (inflates the leading blocks only)
    out->n_contigs = (int32_t)h.names.size();
    out->names = (char **)calloc(h.names.size() + 1, sizeof(char *));
    out->lengths = (int64_t *)calloc(h.names.size() + 1, sizeof(int64_t));
    for (size_t i = 0; i < h.names.size(); ++i) { out->names[i] = strdup(h.names[i].c_str()); out->lengths[i] = h.lengths[i]; }
    out->header_text = strdup(h.text.c_str());
    return MSNV_OK;
}

extern "C" void msnv_bam_data_free(msnv_bam_data *d) {
    if (!d) return;
    for (int i = 0; i < d->n_contigs; ++i) free(d->names[i]);
    free(d->names); free(d->lengths); free(d->records); free(d->header_text);
    memset(d, 0, sizeof *d);
}

extern "C" int msnv_bam_write(const char *bam_path, const char *header_text, int32_t n_contigs,
                              const char *const *names, const int64_t *lengths,
                              const uint8_t *records, uint64_t n_record_bytes, int32_t compress_level) {
    clear_error();
    BamHeader h;
    if (header_text) h.text = header_text;
    else {
        h.text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int i = 0; i < n_contigs; ++i) h.text += "@SQ\tSN:" + std::string(names[i]) + "\tLN:" + std::to_string(lengths[i]) + "\n";
    }
    for (int i = 0; i < n_contigs; ++i) { h.names.push_back(names[i]); h.lengths.push_back(lengths[i]); }
    return bam_write(bam_path, h, records, n_record_bytes, compress_level < 0 ? 1 : compress_level);
}

extern "C" void msnv_free(void *p) { free(p); }
