// metasnv_amd/csrc/inflate.cpp -- raw DEFLATE (RFC 1951) decoder for BGZF blocks.
//
// Host decode is row f2 of SURVEY.md section 8: once the kernels run at terabases per second, inflating the BAMs is what an
// end-to-end user waits for (the reference reads them through htslib + zlib, qaCompute.cpp:441; zlib's inflate runs at
// ~200 MB/s of compressed BAM per core).  A BGZF block is a complete, independent DEFLATE stream of at most 64 KiB that sits
// whole in memory together with its output buffer, which allows what a streaming inflate cannot do:
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load, no per-byte input checks inside the fast loop;
//   * one table lookup per symbol: 11-bit main table for literal/length codes (8-bit for distances) whose entries carry the
//     symbol's value, its extra-bit count and its code length; longer codes go through per-prefix subtables;
//   * up to three literals per refill; matches copied eight bytes at a time (the loop keeps 258 + 8 bytes of output slack and
//     never writes beyond the block's own output range -- neighbouring blocks are inflated by other threads);
//   * a careful byte-exact loop for the last few hundred bytes of a block.
// Written from the RFC; table layout in the spirit of the well-known fast decoders (libdeflate, zlib-ng's inffast).
// Malformed input is an error return, never an out-of-bounds access: the callers guarantee 8 readable bytes behind the
// input (the BGZF trailer / padding) and the decoder never writes past `dst + n_out`.
#include <cstdint>
#include <cstring>

namespace msnv {
namespace {

constexpr int LL_BITS = 11, D_BITS = 8;
constexpr uint32_t F_LIT = 1u << 12, F_EOB = 1u << 13, F_SUB = 1u << 14, F_BAD = 1u << 15;
// entry: bits 0-7 code length (bits to drop), 8-11 extra bits, 12-15 flags, 16-31 literal / base value / subtable start

struct Tables {
    uint32_t ll[(1 << LL_BITS) + 1200];
    uint32_t d[(1 << D_BITS) + 600];
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

uint8_t g_rev8[256];
inline uint32_t rev_bits(uint32_t v, int n) {                // n <= 15
    return ((uint32_t)g_rev8[v & 0xffu] << 8 | g_rev8[(v >> 8) & 0xffu]) >> (16 - n);
}

// Builds a decode table from canonical code lengths.  sym_entry[s] = value << 16 | extra << 8 | flags (no length yet).
// Returns false for an over-subscribed code.  Unused slots are F_BAD.
bool build_table(uint32_t *tab, int main_bits, int cap, const uint8_t *lens, int n_sym, const uint32_t *sym_entry) {
    int count[16] = {0};
    for (int s = 0; s < n_sym; ++s) count[lens[s]]++;
    count[0] = 0;
    uint32_t left = 1;
    for (int l = 1; l <= 15; ++l) { left <<= 1; if ((uint32_t)count[l] > left) return false; left -= (uint32_t)count[l]; }
    uint32_t next_code[16], code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
    const int main_size = 1 << main_bits;
    if (left) for (int i = 0; i < main_size; ++i) tab[i] = F_BAD | 1u;      // an incomplete code leaves holes (a complete one fills every slot)
    // subtables: per main-table prefix the longest code that starts with it
    uint8_t sub_bits[1 << LL_BITS];
    memset(sub_bits, 0, (size_t)main_size);
    uint32_t codes[288 + 32];
    for (int s = 0; s < n_sym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = rev_bits(next_code[l]++, l);
        codes[s] = c;
        if (l > main_bits) { uint8_t &b = sub_bits[c & (uint32_t)(main_size - 1)]; if (l - main_bits > b) b = (uint8_t)(l - main_bits); }
    }
    int next_free = main_size;
    for (int i = 0; i < main_size; ++i) {
        if (!sub_bits[i]) continue;
        const int sz = 1 << sub_bits[i];
        if (next_free + sz > cap) return false;
        tab[i] = (uint32_t)next_free << 16 | (uint32_t)sub_bits[i] << 8 | F_SUB | (uint32_t)main_bits;
        for (int k = 0; k < sz; ++k) tab[next_free + k] = F_BAD | 1u;
        next_free += sz;
    }
    for (int s = 0; s < n_sym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = codes[s];
        if (l <= main_bits) {
            const uint32_t e = sym_entry[s] | (uint32_t)l;
            for (uint32_t k = c; k < (uint32_t)main_size; k += 1u << l) tab[k] = e;
        } else {
            const uint32_t link = tab[c & (uint32_t)(main_size - 1)];
            const int sb = (int)((link >> 8) & 0xfu), start = (int)(link >> 16);
            const uint32_t e = sym_entry[s] | (uint32_t)(l - main_bits);
            for (uint32_t k = c >> main_bits; k < (1u << sb); k += 1u << (l - main_bits)) tab[start + (int)k] = e;
        }
    }
    return true;
}

// Two literals per lookup: where a main-table slot holds a literal of l1 bits and the bits behind it decode -- within the 11
// bits of the index -- to a second literal, the slot becomes {lit1, lit2, l1 + l2} (extra-bits field = 1 marks the pair).
// BAM is literal-heavy (qualities, packed bases): this is where most of the decoder's time goes.
void pair_literals(uint32_t *tab) {
    uint32_t single[1 << LL_BITS];
    memcpy(single, tab, sizeof single);
    for (uint32_t i = 0; i < (1u << LL_BITS); ++i) {
        const uint32_t e1 = single[i];
        if (!(e1 & F_LIT)) continue;
        const uint32_t l1 = e1 & 0xffu;
        const uint32_t e2 = single[i >> l1];               // the unknown high bits are zero: valid only if e2's code fits the known ones
        if (!(e2 & F_LIT)) continue;
        const uint32_t l2 = e2 & 0xffu;
        if (l1 + l2 > (uint32_t)LL_BITS) continue;
        tab[i] = ((e1 >> 16) & 0xffu) << 16 | ((e2 >> 16) & 0xffu) << 24 | 1u << 8 | F_LIT | (l1 + l2);
    }
}

uint32_t g_ll_entry[288], g_d_entry[32];
Tables g_fixed;
bool g_init = false;

void init_static() {
    for (int v = 0; v < 256; ++v) { uint32_t r = 0; for (int i = 0; i < 8; ++i) r |= ((uint32_t)(v >> i) & 1u) << (7 - i); g_rev8[v] = (uint8_t)r; }
    for (int s = 0; s < 256; ++s) g_ll_entry[s] = (uint32_t)s << 16 | F_LIT;
    g_ll_entry[256] = F_EOB;
    for (int s = 257; s < 286; ++s) g_ll_entry[s] = (uint32_t)kLenBase[s - 257] << 16 | (uint32_t)kLenExtra[s - 257] << 8;
    g_ll_entry[286] = g_ll_entry[287] = F_BAD;
    for (int s = 0; s < 30; ++s) g_d_entry[s] = (uint32_t)kDistBase[s] << 16 | (uint32_t)kDistExtra[s] << 8;
    g_d_entry[30] = g_d_entry[31] = F_BAD;
    uint8_t lens[288];
    for (int s = 0; s < 144; ++s) lens[s] = 8;
    for (int s = 144; s < 256; ++s) lens[s] = 9;
    for (int s = 256; s < 280; ++s) lens[s] = 7;
    for (int s = 280; s < 288; ++s) lens[s] = 8;
    build_table(g_fixed.ll, LL_BITS, (int)(sizeof g_fixed.ll / 4), lens, 288, g_ll_entry);
    pair_literals(g_fixed.ll);
    uint8_t dl[32];
    for (int s = 0; s < 32; ++s) dl[s] = 5;
    build_table(g_fixed.d, D_BITS, (int)(sizeof g_fixed.d / 4), dl, 32, g_d_entry);
    g_init = true;
}
struct StaticInit { StaticInit() { init_static(); } } g_static_init;

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

}  // namespace

// Inflates one raw DEFLATE stream of exactly n_out bytes.  `src` must have 8 readable bytes behind src + n_in.
// Compiled twice: as is, and with -mbmi2 as inflate_raw_bmi2 (inflate_bmi2.cpp; shrx / bzhi take the variable shifts off the
// flags and the CL register: +25 % on the build host); hostio.cpp picks one at run time.
#ifndef MSNV_INFLATE_NAME
#define MSNV_INFLATE_NAME inflate_raw
#endif
bool MSNV_INFLATE_NAME(const uint8_t *src, uint32_t n_in, uint8_t *dst, uint32_t n_out) {
    if (!g_init) init_static();
    const uint8_t *in = src, *const in_end = src + n_in;
    uint8_t *out = dst, *const out_end = dst + n_out;
    uint64_t bb = 0;       // bit buffer, LSB first
    int bc = 0;            // valid bits in bb
    Tables dyn;
    // careful refill: never reads behind in_end (zeros enter instead; running out of real bits is caught by the in > in_end check)
    auto need = [&](int n) -> bool {
        while (bc < n) {
            if (in >= in_end) return false;
            bb |= (uint64_t)*in++ << bc;
            bc += 8;
        }
        return true;
    };
    for (;;) {
        if (!need(3)) return false;
        const uint32_t final = (uint32_t)bb & 1u, type = ((uint32_t)bb >> 1) & 3u;
        bb >>= 3; bc -= 3;
        if (type == 0) {                                     // stored
            // drop to a byte boundary: the bit buffer holds whole bytes behind it
            const int drop = bc & 7;
            bb >>= drop; bc -= drop;
            if (!need(32)) return false;
            const uint32_t len = (uint32_t)bb & 0xffffu, nlen = ((uint32_t)(bb >> 16)) & 0xffffu;
            bb >>= 32; bc -= 32;
            if ((len ^ nlen) != 0xffffu) return false;
            // give the unread whole bytes of the bit buffer back to the input
            in -= bc >> 3; bb = 0; bc = 0;
            if ((uint32_t)(in_end - in) < len || (uint32_t)(out_end - out) < len) return false;
            memcpy(out, in, len);
            in += len; out += len;
        } else if (type == 3) {
            return false;
        } else {
            const Tables *T = &g_fixed;
            if (type == 2) {                                 // dynamic Huffman: read the code lengths
                if (!need(14)) return false;
                const int hlit = (int)(bb & 31u) + 257, hdist = (int)((bb >> 5) & 31u) + 1, hclen = (int)((bb >> 10) & 15u) + 4;
                bb >>= 14; bc -= 14;
                if (hlit > 286 || hdist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; ++i) { if (!need(3)) return false; cl[order[i]] = (uint8_t)(bb & 7u); bb >>= 3; bc -= 3; }
                uint32_t cl_entry[19];
                for (int s = 0; s < 19; ++s) cl_entry[s] = (uint32_t)s << 16;
                uint32_t cltab[1 << 7];
                if (!build_table(cltab, 7, 1 << 7, cl, 19, cl_entry)) return false;
                uint8_t lens[286 + 30 + 140];
                int n = 0;
                while (n < hlit + hdist) {
                    if (!need(7 + 7)) { if (in < in_end) return false; }          // (the tail of the stream may hold fewer bits; zeros are harmless: checked below)
                    const uint32_t e = cltab[bb & 127u];
                    if (e & F_BAD) return false;
                    const int l = (int)(e & 0xffu), sym = (int)(e >> 16);
                    if (l > bc) return false;
                    bb >>= l; bc -= l;
                    if (sym < 16) lens[n++] = (uint8_t)sym;
                    else {
                        int rep, xb; uint8_t v = 0;
                        if (sym == 16) { if (!n) return false; v = lens[n - 1]; xb = 2; rep = 3; }
                        else if (sym == 17) { xb = 3; rep = 3; }
                        else { xb = 7; rep = 11; }
                        if (xb > bc) return false;
                        rep += (int)(bb & ((1u << xb) - 1u));
                        bb >>= xb; bc -= xb;
                        if (n + rep > hlit + hdist) return false;
                        while (rep--) lens[n++] = v;
                    }
                }
                if (lens[256] == 0) return false;            // no end-of-block code
                if (!build_table(dyn.ll, LL_BITS, (int)(sizeof dyn.ll / 4), lens, hlit, g_ll_entry)) return false;
                pair_literals(dyn.ll);
                if (!build_table(dyn.d, D_BITS, (int)(sizeof dyn.d / 4), lens + hlit, hdist, g_d_entry)) return false;
                T = &dyn;
            }
            const uint32_t *const ll = T->ll, *const dt = T->d;
            // ---------------------------------------------------------------- fast loop
            // needs: 8 readable bytes at `in` for every refill (callers pad), >= 266 writable bytes at `out`
            bool done = false;
            while (in_end - in >= 16 && out_end - out >= 274) {
                bb |= load64(in) << bc;                      // bits above 63 fall off: they are re-read by the next refill
                in += (63 - bc) >> 3;
                bc |= 56;
                uint32_t e = ll[bb & ((1u << LL_BITS) - 1u)];
                if (e & F_LIT) {                             // up to three lookups = up to six literals out of one refill (3 x <= 11 bits < 56)
                    // a slot holds one literal or two (pair_literals): both bytes are stored, the pointer moves by one or two
                    bb >>= e & 0xffu; bc -= (int)(e & 0xffu);
                    out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24); out += 1u + ((e >> 8) & 1u);
                    e = ll[bb & ((1u << LL_BITS) - 1u)];
                    if (e & F_LIT) {
                        bb >>= e & 0xffu; bc -= (int)(e & 0xffu);
                        out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24); out += 1u + ((e >> 8) & 1u);
                        e = ll[bb & ((1u << LL_BITS) - 1u)];
                        if (e & F_LIT) {
                            bb >>= e & 0xffu; bc -= (int)(e & 0xffu);
                            out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24); out += 1u + ((e >> 8) & 1u);
                            continue;
                        }
                    }
                }
                if (e & F_SUB) {
                    bb >>= e & 0xffu; bc -= (int)(e & 0xffu);
                    e = ll[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 0xfu)) - 1u))];
                }
                if (e & (F_LIT | F_EOB | F_BAD)) {
                    if (e & F_BAD) return false;
                    bb >>= e & 0xffu; bc -= (int)(e & 0xffu);
                    if (e & F_EOB) { done = true; break; }
                    *out++ = (uint8_t)(e >> 16);
                    continue;
                }
                // length (<= 15 + 5 bits so far at most 45 used of 56: the distance needs <= 15 + 13 more -> refill first if short)
                bb >>= e & 0xffu; bc -= (int)(e & 0xffu);
                const uint32_t xl = (e >> 8) & 0xfu;
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1u));
                bb >>= xl; bc -= (int)xl;
                if (bc < 28) { bb |= load64(in) << bc; in += (63 - bc) >> 3; bc |= 56; }
                uint32_t d = dt[bb & ((1u << D_BITS) - 1u)];
                if (d & F_SUB) {
                    bb >>= d & 0xffu; bc -= (int)(d & 0xffu);
                    d = dt[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 0xfu)) - 1u))];
                }
                if (d & F_BAD) return false;
                bb >>= d & 0xffu; bc -= (int)(d & 0xffu);
                const uint32_t xd = (d >> 8) & 0xfu;
                const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1u));
                bb >>= xd; bc -= (int)xd;
                if (dist > (uint32_t)(out - dst)) return false;
                const uint8_t *from = out - dist;
                uint8_t *const stop = out + len;
                if (dist >= 8) {
                    do { memcpy(out, from, 8); out += 8; from += 8; } while (out < stop);       // may run <= 7 bytes past stop: slack guaranteed
                } else if (dist == 1) {
                    memset(out, *from, len);
                } else {
                    do { *out++ = *from++; } while (out < stop);
                }
                out = stop;
            }
            if (in > in_end + 8) return false;
            // ---------------------------------------------------------------- careful loop (block tails)
            while (!done) {
                // make sure a whole code (<= 15 bits) is decodable; zeros behind the input are fine as long as we do not consume them
                while (bc < 15 && in < in_end) { bb |= (uint64_t)*in++ << bc; bc += 8; }
                uint32_t e = ll[bb & ((1u << LL_BITS) - 1u)];
                int used = 0;
                if (e & F_SUB) { used = (int)(e & 0xffu); e = ll[(e >> 16) + (uint32_t)((bb >> used) & ((1u << ((e >> 8) & 0xfu)) - 1u))]; }
                if (e & F_BAD) return false;
                used += (int)(e & 0xffu);
                if (used > bc) return false;
                bb >>= used; bc -= used;
                if (e & F_LIT) {
                    const uint32_t n_lit = 1u + ((e >> 8) & 1u);
                    if ((uint32_t)(out_end - out) < n_lit) return false;
                    *out++ = (uint8_t)(e >> 16);
                    if (n_lit == 2u) *out++ = (uint8_t)(e >> 24);
                    continue;
                }
                if (e & F_EOB) break;
                const int xl = (int)((e >> 8) & 0xfu);
                while (bc < xl + 15 && in < in_end) { bb |= (uint64_t)*in++ << bc; bc += 8; }
                if (xl > bc) return false;
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1u));
                bb >>= xl; bc -= xl;
                uint32_t d = dt[bb & ((1u << D_BITS) - 1u)];
                used = 0;
                if (d & F_SUB) { used = (int)(d & 0xffu); d = dt[(d >> 16) + (uint32_t)((bb >> used) & ((1u << ((d >> 8) & 0xfu)) - 1u))]; }
                if (d & F_BAD) return false;
                used += (int)(d & 0xffu);
                if (used > bc) return false;
                bb >>= used; bc -= used;
                const int xd = (int)((d >> 8) & 0xfu);
                while (bc < xd && in < in_end) { bb |= (uint64_t)*in++ << bc; bc += 8; }
                if (xd > bc) return false;
                const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1u));
                bb >>= xd; bc -= xd;
                if (dist > (uint32_t)(out - dst) || len > (uint32_t)(out_end - out)) return false;
                const uint8_t *from = out - dist;
                for (uint32_t i = 0; i < len; ++i) out[i] = from[i];
                out += len;
            }
        }
        if (final) break;
    }
    // the fast loop reads ahead of what it has consumed: what counts is that no bit behind the input was needed
    if ((int64_t)(in - in_end) * 8 > (int64_t)bc) return false;
    return out == out_end;
}

}  // namespace msnv
