// metasnv_amd/csrc/msnv_internal.h -- shared declarations of libmsnv.so (not installed).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/msnv.h"

namespace msnv {

// ---------------------------------------------------------------------------------- errors
int  fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int  fail_quiet(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));   // sets the message, prints nothing
void clear_error();

// ---------------------------------------------------------------------------------- host-stage timers (msnv.h: msnv_host_timers)
// Cumulative microseconds since the library was loaded (or the last reset), summed over the host threads that did the work:
// what the wall time of BAM files -> resident dataset is made of.
enum HostTimer { HT_READ = 0, HT_INFLATE_HOST, HT_INFLATE_DEVICE_WALL, HT_PACK, HT_UPLOAD_WALL, HT_FORMAT_WALL, HT_ADD_WALL, HT_PACK_DEVICE_WALL, HT_SYNTH_WALL, HT_N };
void host_timer_add(int which, double seconds);
struct HostTimerScope {
    int which; double t0;
    explicit HostTimerScope(int w);
    ~HostTimerScope();
};

// MSNV_FINALIZE_TRACE=1: wall seconds since the previous mark, to stderr (where a dataset's finalize goes; hostio.cpp)
void fin_trace(const char *what);
void fin_trace_reset();

// ---------------------------------------------------------------------------------- host IO
struct BamHeader {
    std::string              text;
    std::vector<std::string> names;
    std::vector<int64_t>     lengths;
};

void msnv_advise_huge(void *p, size_t bytes);   // madvise(MADV_HUGEPAGE) where the platform has it (hostio.cpp)

// Uninitialised byte buffer (a std::vector would zero hundreds of megabytes that the inflate overwrites right away).
unsigned msnv_default_threads();                       // hostio.cpp: hardware threads, capped by the container's CPU quota
void msnv_drop_pages(void *p, size_t bytes);       // hostio.cpp: the pages of a buffer back to the system ahead of its free (MADV_DONTNEED: shared lock)
struct ByteBuf {
    uint8_t *p = nullptr; size_t n = 0;
    ByteBuf() = default;
    ByteBuf(const ByteBuf &) = delete;
    ByteBuf &operator=(const ByteBuf &) = delete;
    ByteBuf(ByteBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ByteBuf &operator=(ByteBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~ByteBuf() { free(p); }
    // Large buffers (the inflated bytes of a BAM) ask for transparent huge pages where the system grants them on request (THP "madvise"):
    // a 22 MB stream is 11 page faults instead of 5 600 when the inflate threads first touch it, and unmapping 3.5 GB of them at the end of
    // a 160-BAM job takes milliseconds instead of 0.3 s under the address-space lock (where it stalled the device allocations of finalize).
    bool alloc(size_t m) {
        free(p); p = nullptr;
        constexpr size_t huge = 2u << 20;
        if (m >= 4 * huge) {
            void *q = nullptr;
            if (posix_memalign(&q, huge, (m + huge - 1) / huge * huge) == 0) { p = (uint8_t *)q; msnv_advise_huge(p, (m + huge - 1) / huge * huge); }
        }
        if (!p) p = (uint8_t *)malloc(m ? m : 1);
        n = p ? m : 0;
        return p != nullptr;
    }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

// MSNV_INFLATE_CHECK=n: block i of a file is checked against the CRC-32 of its BGZF trailer when i % n == 0 (default 1: every block, as
// htslib does; 0: none -- benchmarks).  ONE reading for the host decoder and the device inflate (read per call: tests switch it).
inline uint32_t inflate_check_every() { const char *e = getenv("MSNV_INFLATE_CHECK"); const int v = e ? atoi(e) : 1; return (uint32_t)(v < 0 ? 0 : v); }

// Whole-file BGZF inflate (blocks are independent; `threads` > 1 inflates them in parallel).
int bgzf_read_all(const char *path, ByteBuf &out, int threads);
// ... in two steps, for the device inflate (inflate_k.hip): the file's bytes + its blocks (offsets of the raw DEFLATE payloads), and
// the host decoder for one block (own decoder, zlib behind it)
struct BgzfBlock { uint64_t in_off; uint32_t in_size; uint32_t out_size; uint64_t out_off; };
int bgzf_load(const char *path, ByteBuf &comp, size_t &n_in, std::vector<BgzfBlock> &blocks, uint64_t &total_out);
bool bgzf_inflate_block_host(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out);
uint32_t bgzf_crc32(const uint8_t *data, uint32_t n);           // CRC-32 as in the BGZF trailer
int bgzf_index_bytes(const uint8_t *data, size_t n, const char *path, std::vector<BgzfBlock> &blocks, uint64_t &total_out);   // `data` needs 16 readable bytes behind n
int bam_parse_header_bytes(const uint8_t *data, size_t n, const char *path, BamHeader &hdr, uint64_t &rec_off);
int bam_header_from_blocks(const uint8_t *file_bytes, const std::vector<BgzfBlock> &blocks, const char *path, BamHeader &hdr, uint64_t &rec_off);   // leading blocks inflated on the host
int bgzf_write_all(const char *path, const uint8_t *data, uint64_t n, int level);

// BAM = BGZF(magic, header text, contig table, records...)
// The records start at buf.data() + rec_off (the header sits in front of them in the same buffer: no second copy).
int bam_read(const char *path, BamHeader &hdr, ByteBuf &buf, uint64_t &rec_off, int threads);
int bam_read_header(const char *path, BamHeader &hdr);
int bam_write(const char *path, const BamHeader &hdr, const uint8_t *records, uint64_t n, int level);

// FASTA: name = header line up to the first whitespace (faidx semantics, what `samtools
// mpileup -f` sees); sequence characters kept verbatim (case preserved).
struct FastaSeq { std::string name; std::string seq; };
int fasta_read(const char *path, std::vector<FastaSeq> &out);

// 3-column BED as written by metaSNV.py:92 (`name\t1\tLEN`): 0-based half-open regions.
struct BedRegion { std::string name; int64_t beg, end; };
int bed_read(const char *path, std::vector<BedRegion> &out);

// ---------------------------------------------------------------------------------- gene annotation (parsing only)
// --db_ann rows as snpCall keeps them (call_vC.cpp:116-199,205-284): per contig in file order, start<=end only,
// and the genome characters as gene.h stores them (anything but A/T/C/G/N is 'A').
struct GeneRow { long start, end; std::string name; char strand; };
struct Annotation {
    bool active = false;
    std::map<std::string, std::vector<GeneRow>> genes;
    std::map<std::string, std::string> genome;
};
int load_annotation(const char *ann_path, const char *fasta_path, Annotation &an);

// ---------------------------------------------------------------------------------- BAM record view
constexpr int BAM_FPAIRED = 1, BAM_FPROPER_PAIR = 2, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16,
              BAM_FSECONDARY = 256, BAM_FQCFAIL = 512, BAM_FDUP = 1024;
enum CigarOp : uint32_t { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8 };

struct RecView {
    int32_t  tid, pos, l_seq;
    uint16_t flag; int32_t n_cigar;      // n_cigar / cigar: the record's, or the CG:B,I field's when the CIGAR lives there (rec_parse)
    uint8_t  mapq;
    const uint8_t *cigar, *seq, *qual;   // unaligned little-endian
    const uint8_t *qname; uint32_t l_name;   // NUL-terminated read name, l_name includes the NUL
    int32_t  mtid, mpos, tlen;            // mate contig / position, template length
    uint32_t size;                        // bytes consumed including block_size
};
// returns false on a malformed record
bool rec_parse(const uint8_t *p, uint64_t avail, RecView &r);
inline uint32_t ld_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// htslib's seq_nt16_table: FASTA/IUPAC character -> 4-bit code (unknown -> 15)
uint8_t nt16_of_char(unsigned char c);

}  // namespace msnv
