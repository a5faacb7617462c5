// metasnv_amd/csrc/inflate_k.hip -- BGZF blocks inflated on the device (host BAM decode).
//
// The reference reads its BAMs through htslib + zlib, one thread per file (qaCompute.cpp:441, `samtools mpileup`); here the
// host stage is inflate (2/3 of its time) + record parse + pack (pack.cpp).  A BGZF block is an independent raw DEFLATE
// stream (RFC 1951) of at most 64 KiB: a BAM of the benchmark shape is ~500 of them, a job tens of thousands, so the
// device takes ONE WAVEFRONT per block and thousands of blocks at a time.  DEFLATE is sequential inside a block, so the
// wavefront's lanes all walk the same symbol stream (uniform control flow: table lookups in LDS broadcast, the 64-bit bit
// buffer refilled from memory ahead of its use) and split the work that has width: a match copy is 64 bytes per step
// (overlapping matches repeat their pattern by index arithmetic), literals are collected lane by lane and stored 64 at a
// time, the decode tables are filled a symbol per lane.  Malformed input is a status word, never an out-of-bounds access:
// the host inflates such a block with its own decoder (csrc/inflate.cpp), which words the error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "device.h"
#include "msnv_internal.h"

namespace msnv {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(MSNV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct InfBlock { unsigned long long in_off, out_off; uint32_t in_size, out_size; };

namespace {

#if !defined(MSNV_INFLATE_WAVES)
#define MSNV_INFLATE_WAVES 5                                   // wavefronts per SIMD the register allocation aims at (86 registers; 6 = 80 registers, two of them spilled: the same 130.6 ms)
#endif
constexpr int I_LL_BITS = 9, I_D_BITS = 6;                     // root table bits (zlib's choice: 852 / 592 entries bound the tables)
constexpr int I_LL_CAP = 852, I_D_CAP = 592;                    // zlib's ENOUGH_LENS / ENOUGH_DISTS for these root bits: the most entries a valid code needs (round 5: 1024 / 640 -- the
                                                                // kernel's rate is wavefronts in flight / latency per symbol, and LDS is what bounds the wavefronts: 18 -> 22 per CU)
// Table entries are 16 bits (round 6; 32 until then): the kernel's rate is wavefronts in flight / latency per symbol, and LDS bounded the
// wavefronts -- 7.4 KB a wavefront = 20 per CU; with 16-bit entries and the base / extra tables beside them 4.7 KB = 32 per CU, the most a CU runs.
//   literal / length table:  bits 0-2 kind (LL_LIT, LL_LEN, LL_EOB, LL_SUB, LL_BAD)
//                            LIT / LEN / EOB: bits 3-6 code bits to drop; LIT: bits 8-15 the byte; LEN: bits 8-12 symbol - 257
//                            SUB: bits 3-5 subtable bits - 1, bits 6-15 first entry of the subtable
//   distance / code-length table: bits 0-1 kind (D_SYM, D_SUB, D_BAD); SYM: bits 2-5 code bits, bits 8-12 symbol; SUB: bits 2-5 subtable bits - 1, bits 6-15 first entry
enum : uint32_t { LL_LIT = 0, LL_LEN = 1, LL_EOB = 2, LL_SUB = 3, LL_BAD = 4, D_SYM = 0, D_SUB = 2, D_BAD = 3 };

__constant__ uint16_t k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t k_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct InfLds {
    uint16_t ll[I_LL_CAP];
    uint16_t d[I_D_CAP];                       // (its first 128 entries are the code-length code's table while a dynamic block's header is read: d is built behind that)
    uint32_t len_be[32], dist_be[32];          // base << 8 | extra bits of every length / distance symbol (from the constant tables, once per wavefront)
    uint8_t  lens[32 + 320 + 8];               // [0, 19): code-length code; [32, ...): literal/length lengths, then the distance lengths
    uint16_t code[320];                        // bit-reversed canonical code of every symbol
    uint8_t  sub_bits[1 << I_LL_BITS];
    uint32_t count[16];
    uint32_t next_free;
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// entry of symbol s with `nb` code bits (to drop at the table level it sits in)
__device__ __forceinline__ uint32_t ll_entry(const int s, const uint32_t nb) {
    if (s < 256) return LL_LIT | nb << 3 | (uint32_t)s << 8;
    if (s == 256) return LL_EOB | nb << 3;
    if (s < 286) return LL_LEN | nb << 3 | (uint32_t)(s - 257) << 8;
    return LL_BAD;
}
__device__ __forceinline__ uint32_t d_entry(const int s, const uint32_t nb) { return s < 30 ? D_SYM | nb << 2 | (uint32_t)s << 8 : D_BAD; }
__device__ __forceinline__ uint32_t cl_entry(const int s, const uint32_t nb) { return D_SYM | nb << 2 | (uint32_t)s << 8; }

// Canonical Huffman decode table from code lengths (the algorithm of csrc/inflate.cpp build_table, a symbol per lane where the work
// has width).  kind: 0 literal/length, 1 distance, 2 code lengths.  Returns false for an over-subscribed code or a table that
// does not fit.  Called by the whole wavefront; barriers inside.
__device__ bool build_table(InfLds &L, uint16_t *tab, const int main_bits, const int cap, const uint8_t *lens, const int n_sym, const int kind, const int lane) {
    const uint32_t bad = kind == 0 ? LL_BAD : D_BAD;
    if (lane < 16) L.count[lane] = 0;
    __syncthreads();
    for (int s = lane; s < n_sym; s += 64) if (lens[s]) atomicAdd(&L.count[lens[s]], 1u);
    __syncthreads();
    // lengths -> first code of every length; over-subscription check (uniform: every lane computes it)
    uint32_t left = 1, code = 0, nc[16];
    bool ok = true;
    nc[0] = 0;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        const uint32_t c = L.count[l];
        if (c > left) ok = false;
        left -= min(c, left);
        code = (code + (l > 1 ? L.count[l - 1] : 0u)) << 1;
        nc[l] = code;
    }
    if (!ok) return false;
    const int main_size = 1 << main_bits;
    // canonical codes: symbol s of length l gets next_code[l] + (number of symbols < s with the same length).  Serial over the
    // symbols (a few hundred steps), done by every lane identically; only lane 0 stores.
    if (lane == 0) {
        uint32_t nxt[16];
        for (int l = 0; l < 16; ++l) nxt[l] = nc[l];
        for (int s = 0; s < n_sym; ++s) { const int l = lens[s]; if (l) L.code[s] = (uint16_t)rev_bits(nxt[l]++, l); }
    }
    for (int i = lane; i < main_size; i += 64) { tab[i] = (uint16_t)bad; L.sub_bits[i] = 0; }      // an incomplete code leaves holes
    __syncthreads();
    // subtables: per root prefix the longest code that starts with it (serial: a few hundred steps)
    if (lane == 0) {
        for (int s = 0; s < n_sym; ++s) {
            const int l = lens[s];
            if (l > main_bits) { uint8_t &b = L.sub_bits[L.code[s] & (uint32_t)(main_size - 1)]; if (l - main_bits > b) b = (uint8_t)(l - main_bits); }
        }
        int next_free = main_size;
        bool fits = true;
        for (int i = 0; i < main_size && fits; ++i) {
            if (!L.sub_bits[i]) continue;
            const int sz = 1 << L.sub_bits[i];
            if (next_free + sz > cap) { fits = false; break; }
            tab[i] = (uint16_t)(kind == 0 ? LL_SUB | (uint32_t)(L.sub_bits[i] - 1) << 3 | (uint32_t)next_free << 6 : D_SUB | (uint32_t)(L.sub_bits[i] - 1) << 2 | (uint32_t)next_free << 6);
            for (int k = 0; k < sz; ++k) tab[next_free + k] = (uint16_t)bad;
            next_free += sz;
        }
        L.next_free = fits ? (uint32_t)next_free : 0u;
    }
    __syncthreads();
    if (L.next_free == 0u) return false;
    for (int s = lane; s < n_sym; s += 64) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = L.code[s];
        if (l <= main_bits) {
            const uint32_t e = kind == 0 ? ll_entry(s, (uint32_t)l) : kind == 1 ? d_entry(s, (uint32_t)l) : cl_entry(s, (uint32_t)l);
            for (uint32_t k = c; k < (uint32_t)main_size; k += 1u << l) tab[k] = (uint16_t)e;
        } else {
            const uint32_t link = tab[c & (uint32_t)(main_size - 1)];
            const int sb = (int)(kind == 0 ? ((link >> 3) & 7u) : ((link >> 2) & 15u)) + 1, start = (int)(link >> 6);
            const uint32_t nb = (uint32_t)(l - main_bits);
            const uint32_t e = kind == 0 ? ll_entry(s, nb) : kind == 1 ? d_entry(s, nb) : cl_entry(s, nb);
            for (uint32_t k = c >> main_bits; k < (1u << sb); k += 1u << (l - main_bits)) tab[start + (int)k] = (uint16_t)e;
        }
    }
    __syncthreads();
    return true;
}

}  // namespace

// The symbol stream is the same in every lane; values that come back from LDS or from another lane are pinned to scalar registers so
// that the bit-buffer arithmetic runs on the scalar unit (64-bit shifts in one instruction, no vector latency between dependent
// steps) -- the decoder is a single dependency chain and its speed is instructions x latency.
__device__ __forceinline__ uint32_t uni(const uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// status: 0 = inflated to exactly out_size bytes; 1 = malformed / unsupported (the host decides)
__global__ __launch_bounds__(64, MSNV_INFLATE_WAVES) void msnv_inflate_blocks(const uint8_t *__restrict__ comp, const InfBlock *__restrict__ blocks, const uint32_t n_blocks,
                                                          uint8_t *__restrict__ out, uint32_t *__restrict__ status) {
    __shared__ InfLds L;
    const int lane = threadIdx.x;
    const uint32_t bi = blockIdx.x;
    if (bi >= n_blocks) return;
    if (lane < 32) {
        L.len_be[lane] = lane < 29 ? (uint32_t)k_len_base[lane] << 8 | k_len_extra[lane] : 0u;
        L.dist_be[lane] = lane < 30 ? (uint32_t)k_dist_base[lane] << 8 | k_dist_extra[lane] : 0u;
    }
    __syncthreads();
    InfBlock B = blocks[bi];
    B.in_size = uni(B.in_size); B.out_size = uni(B.out_size);
    const uint8_t *const src = comp + B.in_off;
    uint8_t *const dst = out + B.out_off;
    const uint32_t n_in = B.in_size, n_out = B.out_size;
    // ---- bit reader: 64-bit buffer, LSB first, refilled a byte at a time from 8-byte aligned words (every lane the same: broadcast loads)
    // The input is looked at through a 256-byte window held in a register (lane i = dword i of the window, ONE coalesced load;
    // the next window is requested when this one is entered): a refill is a cross-lane read, not a trip to memory -- with a
    // dependent load per refill a symbol cost ~260 ns, all of it latency.
    unsigned long long bb = 0; int bc = 0; uint32_t ip = 0;     // ip: next input byte
    const uint32_t a0 = uni((uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u));           // the windows start at the aligned address in front of src
    const uint32_t *const wsrc = reinterpret_cast<const uint32_t *>(src - a0);
    const uint32_t n_words = (a0 + n_in + 3u) / 4u;
    uint32_t win_at = 0;                                        // first dword of the current window
    uint32_t win = (uint32_t)lane < n_words ? wsrc[lane] : 0u;
    uint32_t win_next = 64u + (uint32_t)lane < n_words ? wsrc[64 + lane] : 0u;
    auto refill = [&](const int need) -> bool {                 // make `need` (<= 48) bits available; false: the input ran out
        while (bc < need) {
            if (ip >= n_in) return false;
            const uint32_t q = a0 + ip, wi = q >> 2, al = q & 3u;                   // byte q of the window space
            if (wi >= win_at + 64u) {                            // (uniform) next window; the one behind it is requested now
                win = win_next; win_at += 64u;
                win_next = win_at + 64u + (uint32_t)lane < n_words ? wsrc[win_at + 64u + lane] : 0u;
            }
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(wi - win_at)) >> (8u * al);
            const uint32_t take = min(min(4u - al, n_in - ip), (uint32_t)(64 - bc) >> 3);
            if (take == 0u) break;
            bb |= (unsigned long long)(w & (take == 4u ? 0xffffffffu : (1u << (8u * take)) - 1u)) << bc;
            bc += 8 * (int)take; ip += take;
        }
        return bc >= need;
    };
    // a stored block copies from src and moves ip: the window follows through the same test (wi >= win_at + 64); moving by more than
    // one window at a time is handled by reloading
    auto resync = [&]() {
        const uint32_t wi = (a0 + ip) >> 2;
        // (also backwards: the stored-block path hands unread bytes of the bit buffer back; with today's refill sizes ip never falls
        // behind the window, but readlane(win, wi - win_at) with a wrapped index would read the wrong dword without any error)
        if (wi < win_at || wi >= win_at + 64u) {
            win_at = wi & ~63u;
            win = win_at + (uint32_t)lane < n_words ? wsrc[win_at + lane] : 0u;
            win_next = win_at + 64u + (uint32_t)lane < n_words ? wsrc[win_at + 64u + lane] : 0u;
        }
    };
    uint32_t op = 0;                                            // output position
    bool fail_ = false;
    // literals wait in a register, lane i holding literal i of the batch, and are stored 64 at a time (or in front of a match)
    uint32_t lit_reg = 0; uint32_t n_lit = 0;
    auto flush_lits = [&]() {
        if ((uint32_t)lane < n_lit) dst[op + (uint32_t)lane] = (uint8_t)lit_reg;
        op += n_lit; n_lit = 0;
    };
    for (;;) {
        if (!refill(3)) { fail_ = true; break; }
        const uint32_t final_ = (uint32_t)bb & 1u, type = ((uint32_t)bb >> 1) & 3u;
        bb >>= 3; bc -= 3;
        if (type == 0u) {                                        // stored block
            flush_lits();
            const int drop = bc & 7;
            bb >>= drop; bc -= drop;
            if (!refill(32)) { fail_ = true; break; }
            const uint32_t len = (uint32_t)bb & 0xffffu, nlen = (uint32_t)(bb >> 16) & 0xffffu;
            bb >>= 32; bc -= 32;
            if ((len ^ nlen) != 0xffffu) { fail_ = true; break; }
            ip -= (uint32_t)bc >> 3; bb = 0; bc = 0;            // the unread whole bytes of the buffer go back
            if (n_in - ip < len || n_out - op < len) { fail_ = true; break; }
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) dst[op + i] = src[ip + i];
            ip += len; op += len;
            resync();
        } else if (type == 3u) { fail_ = true; break; }
        else {
            int hlit = 288, hdist = 32;
            if (type == 2u) {                                    // dynamic Huffman: read the code lengths
                if (!refill(14)) { fail_ = true; break; }
                hlit = (int)(bb & 31u) + 257; hdist = (int)((bb >> 5) & 31u) + 1;
                const int hclen = (int)((bb >> 10) & 15u) + 4;
                bb >>= 14; bc -= 14;
                if (hlit > 286 || hdist > 30) { fail_ = true; break; }
                if (lane < 19) L.lens[lane] = 0;
                __syncthreads();
                bool ok = true;
                for (int i = 0; i < hclen; ++i) {
                    if (!refill(3)) { ok = false; break; }
                    if (lane == 0) L.lens[k_cl_order[i]] = (uint8_t)(bb & 7u);
                    bb >>= 3; bc -= 3;
                }
                __syncthreads();
                if (!ok || !build_table(L, L.d, 7, 128, L.lens, 19, 2, lane)) { fail_ = true; break; }
                // the code lengths themselves: a serial stream again (every lane decodes it, lane 0 stores)
                int n = 0; uint32_t prev = 0;
                while (n < hlit + hdist) {
                    refill(14);                                  // (the tail of the stream may hold fewer bits: checked through l > bc below)
                    const uint32_t e = uni((uint32_t)L.d[bb & 127u]);
                    if ((e & 3u) != D_SYM) { ok = false; break; }
                    const int l = (int)((e >> 2) & 15u), sym = (int)(e >> 8);
                    if (l > bc) { ok = false; break; }
                    bb >>= l; bc -= l;
                    if (sym < 16) { if (lane == 0) L.lens[32 + n] = (uint8_t)sym; prev = (uint32_t)sym; ++n; }
                    else {
                        int rep, xb; uint32_t v = 0;
                        if (sym == 16) { if (!n) { ok = false; break; } v = prev; xb = 2; rep = 3; }
                        else if (sym == 17) { xb = 3; rep = 3; }
                        else { xb = 7; rep = 11; }
                        if (xb > bc) { ok = false; break; }
                        rep += (int)(bb & ((1u << xb) - 1u));
                        bb >>= xb; bc -= xb;
                        if (n + rep > hlit + hdist) { ok = false; break; }
                        if (lane == 0) for (int k = 0; k < rep; ++k) L.lens[32 + n + k] = (uint8_t)v;
                        n += rep; prev = v;
                    }
                }
                __syncthreads();
                if (!ok || L.lens[32 + 256] == 0) { fail_ = true; break; }      // (no end-of-block code)
            } else {                                             // fixed Huffman codes
                for (int s = lane; s < 288; s += 64) L.lens[32 + s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                if (lane < 32) L.lens[32 + 288 + lane] = 5;
                __syncthreads();
            }
            // (the distance lengths are copied in front of the literal/length table build: it reuses L.code / L.sub_bits)
            if (!build_table(L, L.ll, I_LL_BITS, I_LL_CAP, L.lens + 32, hlit, 0, lane)) { fail_ = true; break; }
            if (!build_table(L, L.d, I_D_BITS, I_D_CAP, L.lens + 32 + hlit, hdist, 1, lane)) { fail_ = true; break; }
            // ---------------------------------------------------------------- symbols
            bool done = false;
            while (!done) {
                refill(48);                                      // a literal/length code (<= 15 + 5 bits) and a distance code (<= 15 + 13): 48 bits
                uint32_t e = uni((uint32_t)L.ll[bb & ((1u << I_LL_BITS) - 1u)]);
                int used = 0;
                if ((e & 7u) == LL_SUB) { used = I_LL_BITS; e = uni((uint32_t)L.ll[(e >> 6) + (uint32_t)((bb >> I_LL_BITS) & ((2u << ((e >> 3) & 7u)) - 1u))]); }
                const uint32_t kind = e & 7u;
                if (kind >= LL_SUB) { fail_ = true; break; }          // (a hole of an incomplete code; a link inside a subtable cannot be)
                used += (int)((e >> 3) & 15u);
                if (used > bc) { fail_ = true; break; }
                bb >>= used; bc -= used;
                if (kind == LL_LIT) {
                    if (op + n_lit >= n_out) { fail_ = true; break; }
                    if ((uint32_t)lane == n_lit) lit_reg = e >> 8;
                    if (++n_lit == 64u) flush_lits();
                    continue;
                }
                if (kind == LL_EOB) { done = true; break; }
                const uint32_t lbe = uni(L.len_be[(e >> 8) & 31u]);
                const int xl = (int)(lbe & 0xffu);
                if (xl > bc) { fail_ = true; break; }
                const uint32_t len = (lbe >> 8) + (uint32_t)(bb & ((1u << xl) - 1u));
                bb >>= xl; bc -= xl;
                uint32_t d = uni((uint32_t)L.d[bb & ((1u << I_D_BITS) - 1u)]);
                used = 0;
                if ((d & 3u) == D_SUB) { used = I_D_BITS; d = uni((uint32_t)L.d[(d >> 6) + (uint32_t)((bb >> I_D_BITS) & ((2u << ((d >> 2) & 15u)) - 1u))]); }
                if ((d & 3u) != D_SYM) { fail_ = true; break; }
                used += (int)((d >> 2) & 15u);
                const uint32_t dbe = uni(L.dist_be[(d >> 8) & 31u]);
                const int xd = (int)(dbe & 0xffu);
                if (used + xd > bc) { fail_ = true; break; }
                bb >>= used; bc -= used;
                const uint32_t dist = (dbe >> 8) + (uint32_t)(bb & ((1u << xd) - 1u));
                bb >>= xd; bc -= xd;
                flush_lits();
                if (dist > op || len > n_out - op) { fail_ = true; break; }
                // the bytes behind the match source were stored by this wavefront (other lanes, earlier instructions): make them visible
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // an overlapping match (dist < len) repeats its pattern: byte i comes from position i mod dist of the pattern
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) dst[op + i] = dst[op - dist + (dist >= len ? i : i % dist)];
                op += len;
            }
            if (fail_) break;
        }
        if (final_) break;
    }
    if (!fail_) {
        flush_lits();
        if (op != n_out) fail_ = true;                          // exactly n_out bytes (the reader never looks behind the input)
    }
    if (lane == 0) status[bi] = fail_ ? 1u : 0u;
}

// Pinned staging of a context, grown on demand: *in holds `in_bytes` compressed bytes (the caller fills it), *out receives the output.
int dev_inflate_staging(msnv_ctx *ctx, uint64_t in_bytes, uint64_t out_bytes, uint8_t **in, uint8_t **out) {
    if (const char *e = getenv("MSNV_TEST_NO_STAGING")) if (e[0] == '1') return fail_quiet(MSNV_ENOMEM, "staging refused (MSNV_TEST_NO_STAGING=1)");   // tests: the host takes the batch
    auto grow = [](void **p, uint64_t *cap, uint64_t need, bool host) -> int {
        if (need <= *cap) return MSNV_OK;
        if (*p) { if (host) (void)hipHostFree(*p); else dev_free(*p); *p = nullptr; *cap = 0; }
        const uint64_t want = need + need / 4 + (1u << 20);
        if (!host) {
            if (dev_alloc(p, want, nullptr) != MSNV_OK) { *p = nullptr; return fail(MSNV_ENOMEM, "device staging of %llu bytes for the device inflate", (unsigned long long)want); }
            *cap = want;
            return MSNV_OK;
        }
        const hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
        if (e != hipSuccess) { *p = nullptr; return fail(MSNV_ENOMEM, "%s staging of %llu bytes for the device inflate: %s", host ? "pinned host" : "device", (unsigned long long)want, hipGetErrorString(e)); }
        *cap = want;
        return MSNV_OK;
    };
    if (int rc = grow(&ctx->pin_in, &ctx->pin_in_cap, in_bytes + 64, true)) return rc;
    if (int rc = grow(&ctx->pin_out, &ctx->pin_out_cap, out_bytes + 64, true)) return rc;
    if (int rc = grow(&ctx->dev_in, &ctx->dev_in_cap, in_bytes + 64, false)) return rc;
    if (int rc = grow(&ctx->dev_out, &ctx->dev_out_cap, out_bytes + 512, false)) return rc;      // (+ read-ahead room of the kernels that take the records where they lie: devpack.hip)
    *in = (uint8_t *)ctx->pin_in; *out = (uint8_t *)ctx->pin_out;
    return MSNV_OK;
}
// The DEVICE half of the staging only (what a finished msnv_dataset_add_sample_bams / msnv_bam_records_many call gives back before the
// dataset is uploaded: compressed + inflated bytes of a batch would otherwise sit in HBM next to the columns); the pinned half stays.
void dev_inflate_release_device(msnv_ctx *ctx) {
    if (ctx->dev_in) dev_free(ctx->dev_in);
    if (ctx->dev_out) dev_free(ctx->dev_out);
    ctx->dev_in = ctx->dev_out = nullptr;
    ctx->dev_in_cap = ctx->dev_out_cap = 0;
}
void dev_inflate_release(msnv_ctx *ctx) {
    if (ctx->pin_in) (void)hipHostFree(ctx->pin_in);
    if (ctx->pin_out) (void)hipHostFree(ctx->pin_out);
    if (ctx->dev_in) dev_free(ctx->dev_in);
    if (ctx->dev_out) dev_free(ctx->dev_out);
    ctx->pin_in = ctx->pin_out = ctx->dev_in = ctx->dev_out = nullptr;
    ctx->pin_in_cap = ctx->pin_out_cap = ctx->dev_in_cap = ctx->dev_out_cap = 0;
}

// ------------------------------------------------------------------------------------------ CRC-32 of the inflated blocks, on the device
// htslib checks every BGZF block against the CRC-32 of its trailer (bgzf.c `[EXT]`; the reference's tools read through it); csrc/crc32.cpp
// does that on host threads over a pinned copy of the output.  When the inflated bytes stay in HBM (the device pack takes them from there)
// the check runs here: one wavefront per block, lane l runs the table-driven register (four bytes per step, four 256-entry tables in LDS
// shared by the workgroup's four blocks) over bytes [1024 l, 1024 l + 1024) -- lane 0 from the all-ones register, the others from zero --
// and moves its register behind the bytes that follow (the register update is linear: a register r followed by n zero bytes is
// r * x^(8 n) mod P; zlib's crc32_combine is the same identity), the 64 registers are XOR-ed.  ~3 400 vector instructions per block.
namespace {
constexpr uint32_t CRC_POLY = 0xEDB88320u;      // reflected CRC-32 (ISO-HDLC), as in the BGZF / gzip trailer
constexpr uint32_t CRC_SPAN = 1024;             // bytes per lane: 64 lanes cover a block of up to 64 KiB
// a * b mod P, polynomials in reflected bit order (x^0 = bit 31)
__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) { p ^= (a & 0x80000000u) ? b : 0u; a <<= 1; b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u); }
    return p;
}
__global__ __launch_bounds__(256) void msnv_crc_blocks(const uint8_t *out, const uint8_t *in, const InfBlock *blocks, const uint32_t *blk_in_file, uint32_t n_blocks, uint32_t check_every,
                                                       const uint32_t *xs /* x^(8 * 1024 k), k < 64 */, const uint32_t *xr /* x^(8 r), r <= 1024 */, uint32_t *status) {
    __shared__ uint32_t T[4][256];
    const int tid = threadIdx.x, lane = tid & 63;
    {   // slicing tables: T[0][i] = register after byte i from zero; T[k][i] = T[k-1][i] moved one byte on
        uint32_t c = (uint32_t)tid;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        T[0][tid] = c;
        __syncthreads();
        for (int k = 1; k < 4; ++k) { c = T[0][c & 0xffu] ^ (c >> 8); T[k][tid] = c; }
        __syncthreads();
    }
    const uint32_t bi = blockIdx.x * 4u + (uint32_t)(tid >> 6);
    if (bi >= n_blocks) return;
    if (status[bi] != 0u || (check_every > 1u && blk_in_file[bi] % check_every)) return;      // refused by the inflate kernel (the host decoder takes it), or not checked
    const InfBlock b = blocks[bi];
    const uint32_t n = b.out_size, L = (n + CRC_SPAN - 1u) / CRC_SPAN;
    uint32_t r = 0;
    if ((uint32_t)lane < L) {
        const uint8_t *p = out + b.out_off + (unsigned long long)CRC_SPAN * (uint32_t)lane;
        const uint32_t len = min(CRC_SPAN, n - CRC_SPAN * (uint32_t)lane);
        r = lane == 0 ? 0xffffffffu : 0u;
        uint32_t i = 0;
        for (; i + 4u <= len; i += 4u) {
            uint32_t w; __builtin_memcpy(&w, p + i, 4);
            r ^= w;
            r = T[3][r & 0xffu] ^ T[2][(r >> 8) & 0xffu] ^ T[1][(r >> 16) & 0xffu] ^ T[0][r >> 24];
        }
        for (; i < len; ++i) r = T[0][(r ^ p[i]) & 0xffu] ^ (r >> 8);
        if ((uint32_t)lane + 1u < L) {               // bytes behind mine: (L - 2 - lane) whole spans and the last lane's n - 1024 (L - 1)
            r = crc_mulmod(r, xs[L - 2u - (uint32_t)lane]);
            r = crc_mulmod(r, xr[n - CRC_SPAN * (L - 1u)]);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r ^= (uint32_t)__shfl_xor((int)r, o);
    if (lane == 0) {
        const uint8_t *t = in + b.in_off + b.in_size;
        const uint32_t want = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        if ((r ^ 0xffffffffu) != want) status[bi] = 2u;
    }
}
}  // namespace

// The device halves of the staging only (compressed batch in, inflated batch out): the resident path below needs no pinned memory.
int dev_inflate_device_buffers(msnv_ctx *ctx, uint64_t in_bytes, uint64_t out_bytes) {
    if (const char *e = getenv("MSNV_TEST_NO_STAGING")) if (e[0] == '1') return fail_quiet(MSNV_ENOMEM, "staging refused (MSNV_TEST_NO_STAGING=1)");
    // (through dev_alloc: with MSNV_GUARD_ALLOC=1 the buffers are exact and end at the end of their mapping -- tests/test_gpu_guard.py)
    const bool exact = [] { const char *e = getenv("MSNV_GUARD_ALLOC"); return e && e[0] == '1'; }();
    auto grow = [&](void **p, uint64_t *cap, uint64_t need) -> int {
        if (need <= *cap && !exact) return MSNV_OK;
        if (*p) { dev_free(*p); *p = nullptr; *cap = 0; }
        const uint64_t want = exact ? need : need + need / 4 + (1u << 20);
        if (int rc = dev_alloc(p, want, nullptr)) { *p = nullptr; return rc == MSNV_ENOMEM ? rc : fail(MSNV_ENOMEM, "device staging of %llu bytes for the device inflate", (unsigned long long)want); }
        *cap = want;
        return MSNV_OK;
    };
    if (int rc = grow(&ctx->dev_in, &ctx->dev_in_cap, in_bytes + 8)) return rc;       // (+8: the trailer of the last block is read as four bytes behind its payload; files carry 16 bytes of slack anyway)
    return grow(&ctx->dev_out, &ctx->dev_out_cap, out_bytes + 512);      // (+ read-ahead room of the kernels that take the records where they lie: devpack.hip)
}

// A batch inflated AND checked on the device, the output left in ctx->dev_out: `host_in` (pageable) goes up, the blocks are inflated, every
// checked block's CRC-32 is compared with its trailer (status 2: mismatch); only the status words come back.  blk_in_file: index of every
// block inside its file (MSNV_INFLATE_CHECK counts per file).
int dev_inflate_resident(msnv_ctx *ctx, const uint8_t *host_in, uint64_t comp_bytes, const std::vector<InfBlock> &blocks, const std::vector<uint32_t> &blk_in_file,
                         uint32_t check_every, std::vector<uint32_t> &status, double *ms_kernel) {
    hipStream_t st = (hipStream_t)ctx->stream;
    status.assign(blocks.size(), 0u);
    if (blocks.empty()) return MSNV_OK;
    struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } d_blk, d_st, d_bif, d_x;
    std::vector<uint32_t> xpow(64 + CRC_SPAN + 1);
    {   // x^(8 r) for r = 0 .. 1024, then x^(8 * 1024 k) for k = 0 .. 63
        uint32_t v = 0x80000000u;
        for (uint32_t r = 0; r <= CRC_SPAN; ++r) { xpow[64 + r] = v; for (int k = 0; k < 8; ++k) v = (v >> 1) ^ ((v & 1u) ? CRC_POLY : 0u); }
        const uint32_t span = xpow[64 + CRC_SPAN];
        xpow[0] = 0x80000000u;
        for (uint32_t k = 1; k < 64; ++k) xpow[k] = crc_mulmod(xpow[k - 1], span);
    }
    HIP_TRY(hipMalloc(&d_blk.p, blocks.size() * sizeof(InfBlock)));
    HIP_TRY(hipMalloc(&d_st.p, blocks.size() * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_bif.p, blocks.size() * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_x.p, xpow.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(ctx->dev_in, host_in, comp_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_blk.p, blocks.data(), blocks.size() * sizeof(InfBlock), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_bif.p, blk_in_file.data(), blocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_x.p, xpow.data(), xpow.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipError_t he = hipEventRecord(e0, st);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(msnv_inflate_blocks, dim3((unsigned)blocks.size()), dim3(64), 0, st, (const uint8_t *)ctx->dev_in, (const InfBlock *)d_blk.p,
                           (uint32_t)blocks.size(), (uint8_t *)ctx->dev_out, (uint32_t *)d_st.p);
        he = hipGetLastError();
    }
    if (he == hipSuccess && check_every) {
        hipLaunchKernelGGL(msnv_crc_blocks, dim3((unsigned)((blocks.size() + 3) / 4)), dim3(256), 0, st, (const uint8_t *)ctx->dev_out, (const uint8_t *)ctx->dev_in, (const InfBlock *)d_blk.p,
                           (const uint32_t *)d_bif.p, (uint32_t)blocks.size(), check_every, (const uint32_t *)d_x.p, (const uint32_t *)d_x.p + 64, (uint32_t *)d_st.p);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess) he = hipMemcpyAsync(status.data(), d_st.p, blocks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    float t = 0;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(MSNV_EHIP, "device inflate: %s", hipGetErrorString(he));
    if (ms_kernel) *ms_kernel += t;
    return MSNV_OK;
}
// one block the host decoder produced, into the batch's output in HBM (resident path: a block the device refused or that did not check)
int dev_inflate_patch(msnv_ctx *ctx, uint64_t out_off, const uint8_t *data, uint32_t n) {
    if (n) HIP_TRY(hipMemcpy(static_cast<uint8_t *>(ctx->dev_out) + out_off, data, n, hipMemcpyHostToDevice));
    return MSNV_OK;
}

// Inflates `blocks` (offsets into the context's staging buffers, dev_inflate_staging) on the device; status[i] != 0: block i was refused.
int dev_inflate(msnv_ctx *ctx, uint64_t comp_bytes, const std::vector<InfBlock> &blocks, uint64_t out_bytes, std::vector<uint32_t> &status, double *ms_kernel) {
    hipStream_t st = (hipStream_t)ctx->stream;
    status.assign(blocks.size(), 0u);
    if (blocks.empty()) return MSNV_OK;
    struct Buf { void *p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } d_blk, d_st;
    HIP_TRY(hipMalloc(&d_blk.p, blocks.size() * sizeof(InfBlock)));
    HIP_TRY(hipMalloc(&d_st.p, blocks.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(ctx->dev_in, ctx->pin_in, comp_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_blk.p, blocks.data(), blocks.size() * sizeof(InfBlock), hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipError_t he = hipEventRecord(e0, st);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(msnv_inflate_blocks, dim3((unsigned)blocks.size()), dim3(64), 0, st, (const uint8_t *)ctx->dev_in, (const InfBlock *)d_blk.p,
                           (uint32_t)blocks.size(), (uint8_t *)ctx->dev_out, (uint32_t *)d_st.p);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess && out_bytes) he = hipMemcpyAsync(ctx->pin_out, ctx->dev_out, out_bytes, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipMemcpyAsync(status.data(), d_st.p, blocks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    float t = 0;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(MSNV_EHIP, "device inflate: %s", hipGetErrorString(he));
    if (ms_kernel) *ms_kernel += t;
    return MSNV_OK;
}

}  // namespace msnv
