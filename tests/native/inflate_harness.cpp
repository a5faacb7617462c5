// tests/native/inflate_harness.cpp -- built and run by tests/test_inflate.py (CPU, with -fsanitize=address,undefined):
// metasnv_amd/csrc/inflate.cpp against zlib on streams of every block type, and on corrupted streams (which must be refused or
// decoded to something, but never read or write out of bounds).  argv[1] = rounds, argv[2] = "bench" for a timing line.
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace msnv { bool inflate_raw(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out); }

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static std::vector<uint8_t> make_data(int kind, size_t n) {
    std::vector<uint8_t> v(n);
    switch (kind) {
        case 0: for (auto &b : v) b = (uint8_t)rnd(); break;                                         // incompressible
        case 1: for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)("ACGT"[rnd() & 3]); break;           // 2 bits of entropy per byte
        case 2: for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)(i < 300 ? rnd() : v[i - 1 - rnd() % 300]); break;   // long matches, all distances
        case 3: memset(v.data(), 'x', n); break;                                                      // distance-1 runs
        case 4: for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)(30 + rnd() % 11); break;             // quality-like
        default: {                                                                                   // BAM-like records: ids, packed bases, qualities
            size_t i = 0;
            while (i < n) {
                for (int k = 0; k < 36 && i < n; ++k) v[i++] = (uint8_t)(k < 8 ? rnd() : k);
                for (int k = 0; k < 12 && i < n; ++k) v[i++] = (uint8_t)("s12r0456789\0"[k]);
                for (int k = 0; k < 50 && i < n; ++k) v[i++] = (uint8_t)(0x11 << (rnd() & 3));
                for (int k = 0; k < 100 && i < n; ++k) v[i++] = (uint8_t)(rnd() % 10 ? 30 + rnd() % 11 : 2 + rnd() % 11);
            }
        }
    }
    return v;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &in, int level, int strategy) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(compressBound((uLong)in.size()) + 64);
    zs.next_in = const_cast<uint8_t *>(in.data()); zs.avail_in = (uInt)in.size();
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    const bool bench = argc > 2 && !strcmp(argv[2], "bench");
    int n_ok = 0, n_refused = 0;
    for (int r = 0; r < rounds; ++r) {
        const int kind = (int)(rnd() % 6);
        const size_t n = r < 8 ? (size_t)r : (size_t)(rnd() % 65536 + 1);
        const std::vector<uint8_t> data = make_data(kind, n);
        const int level = (int)(rnd() % 10);
        const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
        std::vector<uint8_t> comp = deflate_raw(data, level, strategies[rnd() % 5]);
        // exact-size buffers so that the sanitizer sees every overrun; 8 readable bytes behind the input as the callers guarantee
        std::vector<uint8_t> cin(comp.size() + 8, 0);
        memcpy(cin.data(), comp.data(), comp.size());
        std::vector<uint8_t> out(n);
        if (!msnv::inflate_raw(cin.data(), (uint32_t)comp.size(), out.data(), (uint32_t)n) || (n && memcmp(out.data(), data.data(), n) != 0)) {
            fprintf(stderr, "MISMATCH round %d kind %d n %zu level %d\n", r, kind, n, level);
            return 1;
        }
        ++n_ok;
        // wrong expected size must be refused
        if (n > 1) { std::vector<uint8_t> o2(n - 1); if (msnv::inflate_raw(cin.data(), (uint32_t)comp.size(), o2.data(), (uint32_t)(n - 1))) { fprintf(stderr, "short output accepted\n"); return 1; } }
        // corrupted streams: anything goes except touching memory outside the buffers
        for (int m = 0; m < 6 && !comp.empty(); ++m) {
            std::vector<uint8_t> bad = cin;
            const int flips = 1 + (int)(rnd() % 4);
            for (int f = 0; f < flips; ++f) bad[rnd() % comp.size()] ^= (uint8_t)(1u << (rnd() & 7));
            uint32_t cut = (uint32_t)comp.size();
            if (rnd() % 3 == 0) cut = (uint32_t)(rnd() % (comp.size() + 1));                           // truncated
            std::vector<uint8_t> bin(cut + 8, 0);
            memcpy(bin.data(), bad.data(), cut);
            std::vector<uint8_t> o3(n);
            if (!msnv::inflate_raw(bin.data(), cut, o3.data(), (uint32_t)n)) ++n_refused;
        }
    }
    printf("ok %d streams, %d corrupted streams refused\n", n_ok, n_refused);
    if (bench) {
        std::vector<uint8_t> data = make_data(5, 65280);
        std::vector<uint8_t> comp = deflate_raw(data, 6, Z_DEFAULT_STRATEGY);
        comp.resize(comp.size() + 8);
        std::vector<uint8_t> out(data.size());
        const int reps = 4000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) msnv::inflate_raw(comp.data(), (uint32_t)comp.size() - 8, out.data(), (uint32_t)out.size());
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) {
            z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
            zs.next_in = comp.data(); zs.avail_in = (uInt)comp.size() - 8; zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
            inflate(&zs, Z_FINISH); inflateEnd(&zs);
        }
        auto t2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double>(t1 - t0).count(), b = std::chrono::duration<double>(t2 - t1).count();
        printf("bench: %zu -> %zu bytes; msnv %.0f MB/s out (%.0f MB/s in), zlib %.0f MB/s out (%.0f MB/s in), ratio %.2f\n", comp.size() - 8, data.size(),
               reps * data.size() / a / 1e6, reps * (comp.size() - 8) / a / 1e6, reps * data.size() / b / 1e6, reps * (comp.size() - 8) / b / 1e6, b / a);
    }
    return 0;
}
