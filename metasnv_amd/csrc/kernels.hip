// metasnv_amd/csrc/kernels.hip -- CDNA4 (gfx950) kernels of the pileup SNV-calling path.
//
// What they replace (reference, CPU, text based):
//   samtools mpileup's per-position counting of the aligned bases that pass -Q  [EXT] -> msnv_pileup_tiles_* (the CIGAR walk to aligned
//     pieces, the -Q test of every base and the read filters run BEFORE these kernels: devpack.hip on the device, pack.cpp on host threads)
//   snpCall base-string parse + bpCounts      call_vC.cpp:503-535 -> msnv_pileup_tiles_*
//   snpCall gates                             call_vC.cpp:545-552 -> msnv_gate_sites
//   snpCall per-sample strings (counts only)  call_vC.cpp:316-325 -> msnv_gather_scatter
//   snpCall population / individual rule      call_vC.cpp:577-601 -> msnv_decide_sites
//
// Everything is integer counting: the bound is HBM bandwidth, there is no MFMA-shaped work.
// One workgroup owns one tile of TILE reference positions and keeps that tile's bins in LDS;
// reads are streamed from HBM with 16-byte loads per lane; only compact results are written.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device.h"

namespace msnv {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(MSNV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int LANES_PER_READ = 8;              // 8 lanes x 16 bases = one segment piece (<= SEG_MAX bases)
static_assert(LANES_PER_READ * 16 == SEG_MAX, "segment pieces are sized for 8 lanes of 16 bases");

struct PileupArgs {
    const ReadHdr  *hdr;          // one 16-byte header per M/=/X segment piece: {gpos, seqoff, length, meta}
    const uint32_t *blk;          // dense layout: one descriptor per 32-base block
    const PieceHdr *hdr8;         // the same pieces, tile-local 8-byte form: {start | length << 11, seqoff / SEQ_ALIGN} (MSNV_HDR4=0 builds)
    const uint32_t *hdr4;         // ... 4-byte form with chunk-relative offsets (dataset.h: HDR4), what the ordinary narrow work items read
    const PieceHdr *hdr8m;        // headers of the merged groups (pair index in bits 19+, absolute seq offset / 8)
    uint32_t        n_narrow;     // narrow32 launch: work items [0, n_narrow) are ordinary, the rest hold merged groups
    const uint8_t  *seq;
    const uint8_t  *qual;         // ONE BIT per base, in the order of the seq column's nibbles: the base's quality is below the -Q cutoff (pack.cpp packs them at upload: the cutoff is a parameter of the dataset)
    const uint64_t *s_read_base, *s_seq_base;
    const uint32_t *ref4;
    const TilePair *pairs;
    const WorkItem *work;
    const ChunkDesc *chunks;
    uint32_t       *tot;          // 4 words per position, laid out per tile (tot_add): A, C, G, T mismatch totals over all samples (atomics, sparse)
    uint8_t        *part;         // coverage summed over the item's samples, one row per work item (plain stores)
    uint64_t        npos;
    uint8_t        *spill;
    uint8_t        *aspill;       // allele rows (noisy reads): [pair][TILE] words A | C << 8 | G << 16 | T << 24 = the pair's mismatch counts per position, instead of events and total atomics
    Pair32         *events;   uint32_t cap_events;   // this work item's sub-list (ev_list) and the capacity of ONE sub-list
    Pair32         *overflow; uint32_t cap_overflow;
    uint32_t       *counters;
    uint32_t       *ev_count;     // set by ev_list() inside the kernel
    uint32_t        min_baseq;
    uint32_t       *ind4;         // 4 bits per position (A, C, G, T): some sample holds >= min_snvs reads of that mismatching allele
    uint32_t       *unc_bits;     // 1 bit per position: a sample that was split into several pairs holds the allele (it may reach the threshold only in sum)
    uint32_t       *slot_dirty;   // per work item (by the slot of its coverage row), 1 bit per 64 positions: the item added to the allele totals there
    uint32_t        min_snvs;
    // whole-tile work items (dataset.h: WORK_FUSED): the gates and the calling rule are applied by the workgroup that piled the tile up
    uint32_t        n_fused_lo;   // work items from here on are whole-tile items AND the pass uses their record lists (else = the item count)
    int             min_cov; double min_frac;
    const uint32_t *ref_lc, *tile_vbeg, *tile_vend, *tile_stage_idx;
    TileStage      *tile_stage;
    uint32_t       *stage_ovf;    // record-list indices of the tiles with more candidates than a list holds (counters[CNT_STAGE] of them): fused_tile_gate
};

// allele index (A,C,G,T -> 0..3) of a one-hot nt16 code, 4 for anything else
__device__ __forceinline__ uint32_t allele_index(uint32_t code) {
    return (code == 1u) ? 0u : (code == 2u) ? 1u : (code == 4u) ? 2u : (code == 8u) ? 3u : 4u;
}
__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) {         // bit 8j+7 set iff byte j != 0
    return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
__device__ __forceinline__ uint32_t count_nz_bytes(const uint4 a, const uint4 b) {
    return (uint32_t)__popc(nz_bytes(a.x) | nz_bytes(a.y) >> 1 | nz_bytes(a.z) >> 2 | nz_bytes(a.w) >> 3 |
                            nz_bytes(b.x) >> 4 | nz_bytes(b.y) >> 5 | nz_bytes(b.z) >> 6 | nz_bytes(b.w) >> 7);
}
__device__ __forceinline__ uint32_t nz_nibbles(uint32_t x) {       // bit 4j+3 set iff nibble j != 0
    return (((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u;
}
__device__ __forceinline__ uint32_t nibflags_to_bits(uint32_t f) {  // bit 4j+3 -> bit j (8 bits)
    const uint32_t e = f & 0x08080808u, o = (f >> 4) & 0x08080808u;
    return (__builtin_amdgcn_udot4(o, 0x80200802u, __builtin_amdgcn_udot4(e, 0x40100401u, 0u, false), false)) >> 3;
}
// A register-resident uint4 whose value does not matter (an empty asm "defines" it): lets predicated loads skip the zero fill.
__device__ __forceinline__ uint4 any_uint4() {
    uint4 v;
    asm volatile("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w));
    return v;
}
// Low-quality flags of up to 32 consecutive bases: `bit` = index of the first base's flag in the column (= its nibble index in the seq
// column).  Pieces start on SEQ_ALIGN bytes of seq = a multiple of 4 bases, so the flags start anywhere in a byte: eight bytes, shifted.
__device__ __forceinline__ uint2 lowq_fetch(const uint8_t *qlow, const uint64_t bit) {
    uint2 v;
    __builtin_memcpy(&v, qlow + (bit >> 3), 8);
    return v;
}
__device__ __forceinline__ uint32_t lowq_bits(const uint2 v, const uint32_t shift) {
    return (uint32_t)(((unsigned long long)v.y << 32 | v.x) >> shift);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int x) {
    return x + __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int x) {      // LLVM's DPP scan sequence (gfx9)
    x = dpp_add<0x111, 0xf>(x);   // row_shr:1
    x = dpp_add<0x112, 0xf>(x);   // row_shr:2
    x = dpp_add<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add<0x118, 0xf>(x);   // row_shr:8
    x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 -> rows 1 and 3
    x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 -> rows 2 and 3
    return x;
}

// Allele totals of a tile: 4 x TILE words, position-major, as narrow as the tile's summed depth bound allows (pack.cpp sets the
// mode per tile; bits 1-2 of WorkItem::part_lo, GateTile::tot_mode): 0 = the four counts in the bytes of one word per position
// (ONE atomic adds a pair's byte bin as it is), 1 = A | C << 16 and G | T << 16 in two words, 2 = four words.  No field can carry
// into its neighbour: every total is at most the bound.
__device__ __forceinline__ uint32_t tot_mode_of(const WorkItem &w) { return (w.part_lo >> 1) & 3u; }
__device__ __forceinline__ void tot_add(uint32_t *tot, const uint32_t mode, const uint32_t t0, const uint32_t p, const uint32_t n0, const uint32_t n1,
                                        const uint32_t n2, const uint32_t n3) {
    uint32_t *base = tot + 4ull * t0;
    if (mode == 0u) atomicAdd(base + p, n0 | n1 << 8 | n2 << 16 | n3 << 24);
    else if (mode == 1u) {
        if (n0 | n1) atomicAdd(base + 2u * p, n0 | n1 << 16);
        if (n2 | n3) atomicAdd(base + 2u * p + 1u, n2 | n3 << 16);
    } else {
        if (n0) atomicAdd(base + 4u * p, n0);
        if (n1) atomicAdd(base + 4u * p + 1u, n1);
        if (n2) atomicAdd(base + 4u * p + 2u, n2);
        if (n3) atomicAdd(base + 4u * p + 3u, n3);
    }
}
// one allele of one position (the narrow kernels: mismatches are rare and mostly one allele per position); the mode is uniform, so
// the index and shift arithmetic is scalar
__device__ __forceinline__ void tot_add_one(uint32_t *tile_tot, const uint32_t mode, const uint32_t p, const uint32_t x, const uint32_t n) {
    const uint32_t shpack = mode == 0u ? 0x18100800u : mode == 1u ? 0x10001000u : 0u;     // shifts of x = 0..3, one byte each
    // (uniform base + 32-bit byte offset: the address needs no 64-bit register pair -- the kernel sits at a register step: 64 for eight workgroups per CU)
    const uint32_t byte_off = ((p << mode) + (x >> (2u - mode))) * 4u;
    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tile_tot) + byte_off), n << ((shpack >> (8u * x)) & 0xffu));
}

// ------------------------------------------------------------------------------------------
// msnv_pileup_tiles_wide: one work item = (tile, range of (tile,sample) pairs) of ANY depth.
//   for every sample: the sample's segment pieces are classified 16 bases per lane with SWAR
//   arithmetic; coverage is a difference array (+1/-1 per piece, prefix-summed once per sample);
//   only EXCEPTIONS touch the per-position bins: bases below the BQ cutoff / N (subtracted from
//   the span coverage) and mismatching A/C/G/T bases (allele counts).  16-bit bins.
//   Then one pass over the tile adds the sample to the running totals (registers), spills the
//   per-sample coverage byte (>= 255 goes to an overflow list) and emits the sparse allele events.
// Algorithmic HBM bytes: 16 B per segment piece + 0.5 B/base seq + 1 B/base qual (resident: one bit of quality per base).
// ------------------------------------------------------------------------------------------
constexpr int W_NT = 512;
constexpr int W_PPT = TILE / W_NT;             // 4
constexpr int W_HCAP = 256;
constexpr int W_EVCAP = 1024;
static_assert(W_PPT == 4, "wide per-sample pass is written for 4 positions per thread");

struct WideLds {
    int32_t  span[TILE + 4];
    uint32_t exc[TILE / 2];                    // two u16 per word
    unsigned long long al[TILE];               // four u16 (A,C,G,T)
    uint32_t ref[TILE / 8 + 4];
    uint4    hdr[2][W_HCAP];
    Pair32   ev[W_EVCAP];
    int32_t  wsum[W_NT / 64];
    uint32_t evn, ev_base;
};

// Work item i appends to sub-list i % EV_LISTS (device.h): base pointer and fill counter of that sub-list.
__device__ __forceinline__ void ev_list(PileupArgs &a) {
    const uint32_t k = blockIdx.x % EV_LISTS;
    a.events += (uint64_t)k * a.cap_events;
    a.ev_count = a.counters + 16u + k * EV_CNT_STRIDE;
}

template <typename LDS, int NT, int CAP>
__device__ __forceinline__ void flush_events(LDS &L, const PileupArgs &a, int tid) {
    const uint32_t n = min(L.evn, (uint32_t)CAP);              // called by all threads between barriers
    if (tid == 0) L.ev_base = n ? atomicAdd(a.ev_count, n) : 0u;
    __syncthreads();
    const uint32_t base = L.ev_base;
    for (uint32_t i = tid; i < n; i += NT)
        if (base + i < a.cap_events) a.events[base + i] = L.ev[i];
    __syncthreads();
    if (tid == 0) L.evn = 0;
}

template <typename LDS, int CAP>
__device__ __forceinline__ void stage_allele_event(LDS &L, const PileupArgs &a, Pair32 e) {
    const uint32_t i = atomicAdd(&L.evn, 1u);
    if (i < (uint32_t)CAP) { L.ev[i] = e; return; }
    const uint32_t g = atomicAdd(a.ev_count, 1u);               // staging full: slow path
    if (g < a.cap_events) a.events[g] = e;
}

__device__ __forceinline__ void wide_classify(WideLds &L, const uint32_t lq, const uint32_t s0, const uint32_t s1,
                                              const uint32_t P0, const uint32_t vmask) {
    const uint32_t pr = P0;
    const uint32_t wi = pr >> 3, sh = (pr & 7u) * 4u;
    const uint32_t w0 = L.ref[wi], w1 = L.ref[wi + 1], w2 = L.ref[wi + 2];
    const uint32_t r0 = __builtin_amdgcn_alignbit(w1, w0, sh), r1 = __builtin_amdgcn_alignbit(w2, w1, sh);
    const uint32_t nm = nibflags_to_bits(nz_nibbles(s0 ^ r0)) | nibflags_to_bits(nz_nibbles(s1 ^ r1)) << 8;
    uint32_t e = (lq | nm) & vmask;
    while (e) {
        const uint32_t j = (uint32_t)__builtin_ctz(e);
        e &= e - 1u;
        const uint32_t p = P0 + j;
        uint32_t ai = 4u;
        if (!((lq >> j) & 1u)) ai = allele_index(((j < 8u ? s0 : s1) >> (4u * (j & 7u))) & 0xfu);
        if (ai < 4u) atomicAdd(&L.al[p], 1ull << (16u * ai));
        else atomicAdd(&L.exc[p >> 1], 1u << (16u * (p & 1u)));     // not counted: low BQ, N, other IUPAC
    }
}

__global__ __launch_bounds__(W_NT) void msnv_pileup_tiles_wide(PileupArgs a) {
    ev_list(a);
    __shared__ WideLds L;
    const WorkItem w = a.work[blockIdx.x];
    const uint32_t t0 = w.tile * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane8 = tid & (LANES_PER_READ - 1), grp = tid / LANES_PER_READ;
    const int b0 = 16 * lane8;

    for (int i = tid; i < (int)(TILE / 8 + 4); i += W_NT)
        L.ref[i] = (i < (int)(TILE / 8)) ? a.ref4[(t0 >> 3) + i] : 0xffffffffu;
    for (int i = tid; i < (int)(TILE + 4); i += W_NT) L.span[i] = 0;
    for (int i = tid; i < (int)(TILE / 2); i += W_NT) L.exc[i] = 0;
    for (int i = tid; i < (int)TILE; i += W_NT) L.al[i] = 0;
    if (tid == 0) L.evn = 0;

    uint32_t tc[W_PPT], tn[W_PPT][4];
#pragma unroll
    for (int j = 0; j < W_PPT; ++j) { tc[j] = 0; tn[j][0] = tn[j][1] = tn[j][2] = tn[j][3] = 0; }

    uint32_t k = w.pair_lo;
    TilePair pr = (k < w.pair_hi) ? a.pairs[k] : TilePair{0, 0, 0, 0};
    uint32_t rbeg = pr.read_lo;
    uint4 hreg = make_uint4(0, 0, 0, 0);
    if (k < w.pair_hi && tid < W_HCAP && rbeg + (uint32_t)tid < pr.read_hi)
        hreg = *reinterpret_cast<const uint4 *>(a.hdr + a.s_read_base[pr.sample] + rbeg + tid);
    int buf = 0;

    while (k < w.pair_hi) {
        const uint32_t nrd = min((uint32_t)W_HCAP, pr.read_hi - rbeg);
        const bool last_chunk = rbeg + nrd >= pr.read_hi;
        const uint32_t sample = pr.pad >> 8;                       // the sample's SLOT in the tile (what events and overflow entries carry)
        if (tid < W_HCAP) L.hdr[buf][tid] = hreg;                   // slots beyond nrd hold meta = 0
        uint32_t nk = k, nrbeg = rbeg + nrd;
        TilePair npr = pr;
        if (last_chunk) { nk = k + 1; if (nk < w.pair_hi) { npr = a.pairs[nk]; nrbeg = npr.read_lo; } }
        hreg = make_uint4(0, 0, 0, 0);
        if (nk < w.pair_hi && tid < W_HCAP && nrbeg + (uint32_t)tid < npr.read_hi)
            hreg = *reinterpret_cast<const uint4 *>(a.hdr + a.s_read_base[npr.sample] + nrbeg + tid);
        __syncthreads();                                            // (A) headers visible, bins clean

        // (pr.sample, not the slot: a tile that some samples have no reads in numbers its slots without them)
        const uint8_t *seq = a.seq + a.s_seq_base[pr.sample];
        const uint64_t qbit0 = 2ull * a.s_seq_base[pr.sample];   // the sample's first flag in the low-quality column
        for (uint32_t r = (uint32_t)grp; r < nrd; r += W_NT / LANES_PER_READ) {
            const uint4 h = L.hdr[buf][r];
            if (!(h.w & META_PILEUP_OK)) continue;
            const uint32_t s = h.x - t0, len = h.z;                   // a piece never leaves its tile
            if (lane8 == 0) { atomicAdd(&L.span[s], 1); atomicAdd(&L.span[s + len], -1); }
            const int vhi = min(max((int)len - b0, 0), 16);
            if (vhi <= 0) continue;
            uint2 sv;
            const uint64_t qbit = qbit0 + 2ull * h.y + (uint32_t)b0;
            const uint2 qv = lowq_fetch(a.qual, qbit);
            __builtin_memcpy(&sv, seq + (uint64_t)h.y + (uint32_t)(b0 >> 1), 8);
            wide_classify(L, lowq_bits(qv, (uint32_t)qbit & 7u), sv.x, sv.y, s + (uint32_t)b0, (1u << vhi) - 1u);
        }

        if (last_chunk) {
            __syncthreads();                                        // (B) all exceptions of this sample are in LDS
            int4 sp = *reinterpret_cast<int4 *>(&L.span[W_PPT * tid]);
            const uint2 ex = *reinterpret_cast<uint2 *>(&L.exc[2 * tid]);   // two u16 per word: words 2 tid, 2 tid + 1 hold my 4 positions
            unsigned long long al[W_PPT];
#pragma unroll
            for (int j = 0; j < W_PPT; ++j) al[j] = L.al[W_PPT * tid + j];
            *reinterpret_cast<int4 *>(&L.span[W_PPT * tid]) = make_int4(0, 0, 0, 0);
            if (tid == 0) L.span[TILE] = 0;
            *reinterpret_cast<uint2 *>(&L.exc[2 * tid]) = make_uint2(0u, 0u);
#pragma unroll
            for (int j = 0; j < W_PPT; ++j) L.al[W_PPT * tid + j] = 0ull;
            sp.y += sp.x; sp.z += sp.y; sp.w += sp.z;
            const int incl = wave_inclusive_scan(sp.w);
            if (lane == 63) L.wsum[wave] = incl;
            __syncthreads();                                        // (C)
            int off = incl - sp.w;
            for (int wv = 0; wv < wave; ++wv) off += L.wsum[wv];
            const int depth[W_PPT] = {off + sp.x, off + sp.y, off + sp.z, off + sp.w};
            const uint32_t exc[W_PPT] = {ex.x & 0xffffu, ex.x >> 16, ex.y & 0xffffu, ex.y >> 16};
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < W_PPT; ++j) {
                const uint32_t cov = (uint32_t)depth[j] - exc[j];
                tc[j] += cov;
                packed |= (cov < 255u ? cov : 255u) << (8 * j);
                const uint32_t gpos = t0 + W_PPT * tid + j;
                if (cov >= 255u) {
                    const uint32_t g = atomicAdd(&a.counters[1], 1u);
                    if (g < a.cap_overflow) a.overflow[g] = Pair32{gpos, sample << 16 | (cov & 0xffffu)};
                }
                if (al[j] != 0ull) {
                    const uint32_t nn[4] = {(uint32_t)(al[j] & 0xffffu), (uint32_t)((al[j] >> 16) & 0xffffu),
                                            (uint32_t)((al[j] >> 32) & 0xffffu), (uint32_t)(al[j] >> 48)};
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (nn[x]) {
                            tn[j][x] += nn[x];
                            if (nn[x] >= a.min_snvs) atomicOr(&a.ind4[gpos >> 3], 1u << (4u * (gpos & 7u) + (uint32_t)x));
                            stage_allele_event<WideLds, W_EVCAP>(L, a, Pair32{gpos, sample << 18 | (uint32_t)x << 16 | nn[x]});
                        }
                }
            }
            *reinterpret_cast<uint32_t *>(a.spill + (uint64_t)k * TILE + W_PPT * tid) = packed;
        }
        k = nk; pr = npr; rbeg = nrbeg; buf ^= 1;
        if (last_chunk && k < w.pair_hi) {
            __syncthreads();                                        // staged events are complete
            if (L.evn >= (uint32_t)(W_EVCAP / 2)) flush_events<WideLds, W_NT, W_EVCAP>(L, a, tid);
        }
    }
    __syncthreads();
    flush_events<WideLds, W_NT, W_EVCAP>(L, a, tid);

    *reinterpret_cast<uint4 *>(a.part + ((uint64_t)w.part_hi << 32 | (w.part_lo & ~15u)) + 4u * W_PPT * tid) = make_uint4(tc[0], tc[1], tc[2], tc[3]);   // u32 row
    bool any_allele = false;
#pragma unroll
    for (int j = 0; j < W_PPT; ++j) {
        if (tn[j][0] | tn[j][1] | tn[j][2] | tn[j][3]) {
            tot_add(a.tot, tot_mode_of(w), t0, W_PPT * tid + j, tn[j][0], tn[j][1], tn[j][2], tn[j][3]);
            any_allele = true;
        }
    }
    if (any_allele && tot_mode_of(w) != 0u) atomicOr(&a.slot_dirty[w.slot], 1u << ((W_PPT * (uint32_t)tid) >> 6));      // (gate kernel: blocks whose allele totals are not all zero; one word per item)
}

// ------------------------------------------------------------------------------------------
// Narrow path = the dominant kernel (msnv_pileup_tiles_narrow32 below): the same algorithm for
// (tile, sample) pairs whose per-position depth is known (host bound) to stay below 255, which lets
// every LDS bin be ONE BYTE:
//   start[p] / end[p]  segment pieces that begin at / end before p  (coverage = running sum)
//   exc[p]             bases not counted (BQ below cutoff, N, other IUPAC)
//   al[p]              4 bytes: mismatching A, C, G, T
// 20.4 KB of LDS, 63 registers and 256 threads per workgroup -> EIGHT resident workgroups per CU (round 3; six in round 1, seven in round 2).
// The hot loop is written to minimise instructions and vector-memory operations per base:
//   * every header is one segment piece of <= 128 aligned bases inside one tile (the host resolves
//     the CIGAR and splits at tile boundaries): one code path, no clipping, no branches on data;
//   * chunk descriptors are staged in LDS, headers are prefetched one chunk ahead, and all data
//     loads of a 128-piece chunk are issued before the first one is consumed;
//   * BQ cutoff: resolved by the host into ONE BIT per base (pack.cpp: pack_lowq; the cutoff is a parameter of the dataset) -- a
//     lane's 32 flags are one 8-byte load and a shift; match = nibble equality against the LDS-staged reference (host rewrites
//     '=' codes); mismatch flags stay in the nibble domain (they are rare);
//   * low-quality bases reach the byte bins 8 positions at a time: 8 flag bits become 8 bytes (two 24-bit multiplies: spread_bits)
//     for one 64-bit LDS atomic, skipped when zero;
//   * the per-sample prefix sum uses DPP row shifts; allele totals go straight to global memory and
//     allele events are staged in LDS (one returning global atomic per flush).
// ------------------------------------------------------------------------------------------
// bit k of a byte of flags -> byte k of a 64-bit word (the increment of eight byte bins): two 24-bit multiplies.  (Until the quality column
// was one bit per base this came from a 256-entry LDS table -- the same speed while the kernel waited for HBM, 2.5 % slower since:
// the atomic had to wait for the table read, five times per lane and round; profiles/r03zs_ab_spread.txt)
__device__ __forceinline__ unsigned long long spread_bits(const uint32_t byte) {
    return (unsigned long long)(__umul24(byte >> 4, 0x00204081u) & 0x01010101u) << 32 | (__umul24(byte & 0xfu, 0x00204081u) & 0x01010101u);
}

constexpr int N_NT = 256;
constexpr int N_PPT = TILE / N_NT;             // 8 positions per thread in the per-sample pass
constexpr int N_HCAP = CHUNK_READS;
// EIGHT workgroups per CU since the low-quality spread table left the LDS: 20 464 bytes (224 staged events instead of 256 made the last
// 256 bytes) and <= 64 registers (the compiler is told: 8 waves per SIMD; it lands on 63 without scratch).  Seven: 0.405 ms, eight:
// 0.392 ms on the benchmark shape, 0.512 -> 0.480 ms on the sparse shard (profiles/r03zt_ab_occ8.txt)
#if !defined(MSNV_N_EVCAP)
#define MSNV_N_EVCAP 224
#endif
#if !defined(MSNV_N32_WAVES)
#define MSNV_N32_WAVES 8
#endif
constexpr int N_EVCAP = MSNV_N_EVCAP;

static_assert(N_PPT == 8, "narrow per-sample pass is written for 8 positions per thread");

struct alignas(16) NarrowLds {
    uint32_t start[TILE / 4 + 4];
    uint32_t end[TILE / 4 + 4];
    unsigned long long exc[TILE / 8 + 4];      // byte bins, updated 8 positions at a time with 64-bit LDS atomics
    uint32_t al[TILE];
    uint32_t ref[TILE / 8 + 4];
    uint2    hdr[2][N_HCAP];
    Pair32   ev[N_EVCAP];
    ChunkDesc desc[MAX_CHUNKS_PER_ITEM];
    uint32_t carry[N_NT / 64 + 1];             // depth at the left edge of each wavefront's quarter of the tile ([4]: scratch for pieces that end at the tile end)
    uint32_t evn, ev_base;
    uint32_t emask[SEQ_ALIGN_LOG2 < 3 ? 33 : 1];   // pieces closer than 16 bases apart: mismatch flags of a lane's first n bases (narrow_classify32's flag order)
};

// Per-sample pass of the narrow kernel: prefix sum of start/end -> depth, minus the not-counted bases;
// adds the sample to the running totals, spills the per-sample coverage bytes, emits allele events,
// and leaves every bin zero for the next sample.  Called by all threads right behind barrier (B); no barrier inside:
// wavefront w owns positions [512 w, 512 w + 512) and knows the depth at its left edge from L.carry[w] (the pieces that
// cover position 512 w - 1, counted while they were classified), so the prefix sum never leaves the wavefront; allele
// events are placed per wavefront too (one LDS reservation in the staging buffer, or -- staging full, noisy reads --
// one reservation in the event list; the lanes write at their prefix-sum offsets).
// MERGED: the bins hold a GROUP of shallow (sample, tile) pairs (pack.cpp: merged groups).  Their pieces went into the same
// bins and nobody needs their per-sample bytes or allele events: the gather recomputes the few per-sample cells at called
// positions -- coverage and allele counts -- from the pieces themselves (gather_merged_block), so the pass only adds the group to
// the running totals and the allele totals, marks positions where one sample MIGHT hold >= min_snvs reads of an allele (the
// calling rule then reads the summed per-sample records), and leaves every bin zero.
// DA ("dense alleles", pack.cpp: allele planes): the pair's mismatch counts leave as four byte planes -- plain stores next to the coverage
// bytes -- instead of one total atomic + one event per (position, allele): what a pass costs no longer depends on how noisy the reads are.
template <typename LDS, int EXC_PAD, bool MERGED = false, bool FUSED = false, bool DA = false>
__device__ __forceinline__ void narrow_pass(LDS &L, const PileupArgs &a, uint32_t (&tc)[N_PPT / 2], bool &dirty, const uint32_t t0,
                                            const int tid, const int lane, const int wave, const uint32_t sample, const uint32_t k,
                                            const uint32_t split, const uint32_t tmode) {
    const uint2 st = *reinterpret_cast<uint2 *>(&L.start[2 * tid]);
    const uint2 en = *reinterpret_cast<uint2 *>(&L.end[2 * tid]);
    const uint2 ex = *reinterpret_cast<uint2 *>(&L.exc[EXC_PAD + tid]);
    int d = (int)L.carry[wave];
    *reinterpret_cast<uint2 *>(&L.start[2 * tid]) = make_uint2(0u, 0u);
    *reinterpret_cast<uint2 *>(&L.end[2 * tid]) = make_uint2(0u, 0u);
    L.exc[EXC_PAD + tid] = 0ull;
    if (lane == 0) L.carry[wave] = 0u;
    if (tid == 0) L.end[TILE / 4] = 0;
    const int mine = (int)(__builtin_amdgcn_udot4(st.x, 0x01010101u, __builtin_amdgcn_udot4(st.y, 0x01010101u, 0u, false), false)) -
                     (int)(__builtin_amdgcn_udot4(en.x, 0x01010101u, __builtin_amdgcn_udot4(en.y, 0x01010101u, 0u, false), false));
    d += wave_inclusive_scan(mine) - mine;                  // depth just left of my 8 positions
    // SWAR prefix sum: (d + starts - ends) * 0x01010101 holds the depth of 4 positions in its 4 bytes -- byte k of x * 0x01010101
    // is the sum of bytes 0..k of x, and the identity is exact mod 2^32 whatever the intermediate borrows because every true
    // depth is in [0, 255) (host bound).  Minus the not-counted bases = the per-sample coverage bytes = the spill words.
    const uint32_t v0 = ((uint32_t)d + st.x - en.x) * 0x01010101u;
    const uint32_t v1 = ((v0 >> 24) + st.y - en.y) * 0x01010101u;
    const uint32_t c0 = v0 - ex.x, c1 = v1 - ex.y;
    tc[0] += c0 & 0x00ff00ffu; tc[1] += (c0 >> 8) & 0x00ff00ffu;   // running totals, two u16 per register: positions (0,2) (1,3) (4,6) (5,7)
    tc[2] += c1 & 0x00ff00ffu; tc[3] += (c1 >> 8) & 0x00ff00ffu;
    if (!MERGED) *reinterpret_cast<uint2 *>(a.spill + (uint64_t)k * TILE + N_PPT * tid) = make_uint2(c0, c1);
    if (MERGED && FUSED) {                                      // whole-tile item: this group is the tile.  Its coverage bytes wait in the (zeroed) start bins,
        *reinterpret_cast<uint2 *>(&L.start[2 * tid]) = make_uint2(c0, c1);      // its allele totals stay in L.al, for fused_tile_gate behind the last barrier
        return;
    }
    // ---- allele events.  The allele bins are only looked at here, and word by word again when events are written: the
    // registers of the next chunk's column loads are live across this pass.
    if constexpr (DA && !MERGED) {
        uint32_t *row = reinterpret_cast<uint32_t *>(a.aspill) + (uint64_t)k * TILE + N_PPT * tid;      // the pair's row of allele words: A | C << 8 | G << 16 | T << 24 per position
        const uint32_t ge = (0x80u - min(a.min_snvs, 127u)) * 0x01010101u;
#pragma unroll
        for (uint32_t half = 0; half < 2u; ++half) {
            const uint4 v = *reinterpret_cast<uint4 *>(&L.al[N_PPT * tid + 4u * half]);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<uint4 *>(row + 4u * half) = v;         // the bins as they are: 16-byte stores, 8 KB per pair
            if (!(v.x | v.y | v.z | v.w)) continue;
            *reinterpret_cast<uint4 *>(&L.al[N_PPT * tid + 4u * half]) = make_uint4(0u, 0u, 0u, 0u);      // the bins are left zero for the next sample
            dirty = true;
            // the individual rule's marks (see below): some allele holds >= min_snvs reads (rare: SNV positions of this sample); a split
            // sample may reach the threshold only in sum: every position it holds an allele at is marked
            uint32_t todo = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) todo |= (split ? w4[j] != 0u : ((((w4[j] & 0x7f7f7f7fu) + ge) | w4[j]) & 0x80808080u) != 0u) ? 1u << j : 0u;
            while (todo) {
                const uint32_t j = (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1u;
                const uint32_t word = j == 0u ? w4[0] : j == 1u ? w4[1] : j == 2u ? w4[2] : w4[3];
                const uint32_t gpos = t0 + N_PPT * tid + 4u * half + j;
#pragma unroll
                for (uint32_t x = 0; x < 4u; ++x) {
                    const uint32_t n = (word >> (8u * x)) & 0xffu;
                    if (n >= a.min_snvs) atomicOr(&a.ind4[gpos >> 3], 1u << (4u * (gpos & 7u) + x));
                    else if (n && split) atomicOr(&a.unc_bits[gpos >> 5], 1u << (gpos & 31u));
                }
            }
        }
        return;
    }
    uint32_t pm = 0, myev;
    {
        const uint4 a0 = *reinterpret_cast<uint4 *>(&L.al[N_PPT * tid]);
        const uint4 a1 = *reinterpret_cast<uint4 *>(&L.al[N_PPT * tid + 4]);
        const uint32_t alw[N_PPT] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < N_PPT; ++j) pm |= (alw[j] ? 1u : 0u) << j;
        myev = count_nz_bytes(a0, a1);
    }
    if (!__any(pm != 0u)) return;                            // no mismatching allele in this wavefront's 512 positions
    dirty |= pm != 0u;                                       // my 8 positions lie in one 64-position block (store_part_row tells the gate kernel)
    if (MERGED) {
        while (pm) {
            const uint32_t j = (uint32_t)__builtin_ctz(pm);
            pm &= pm - 1u;
            const uint32_t word = L.al[N_PPT * tid + j];
            L.al[N_PPT * tid + j] = 0u;
            const uint32_t gpos = t0 + N_PPT * tid + j;
            for (uint32_t m = nz_bytes(word); m; m &= m - 1u) {      // (the non-zero alleles of the position, usually one: see below)
                const uint32_t x = (uint32_t)__builtin_ctz(m) >> 3;
                const uint32_t n = (word >> (8u * x)) & 0xffu;
                tot_add_one(a.tot + 4ull * t0, tmode, N_PPT * tid + j, x, n);
                if (n >= a.min_snvs) atomicOr(&a.unc_bits[gpos >> 5], 1u << (gpos & 31u));
            }
        }
        return;
    }
    const uint32_t ei = (uint32_t)wave_inclusive_scan((int)myev);   // exclusive prefix of the lanes' event counts = their slots
    uint32_t res = 0;
    if (lane == 63 && ei) {
        // A wavefront's slots in the staging buffer are reserved with compare-and-swap: they are taken or they are not.  (Add first, take
        // it back when the buffer is full -- the form until round 3 -- lets a second wavefront reserve ABOVE the first one's transient
        // count; when that is taken back the fill count lies below the second one's slots: its events are never flushed and the slots
        // that are flushed in their place still hold events of an earlier flush -- a per-sample count doubled at one site in 1 of ~10 000
        // fuzz cases once the buffer held 224 events instead of 256; profiles/stress_case.py shows it in seconds.)
        uint32_t cur = atomicAdd(&L.evn, 0u);
        for (;;) {
            if (cur + ei > (uint32_t)N_EVCAP) { res = 0x80000000u | atomicAdd(a.ev_count, ei); break; }   // straight into the list
            const uint32_t seen = atomicCAS(&L.evn, cur, cur + ei);
            if (seen == cur) { res = cur; break; }                                 // staged: flushed behind a later barrier (A)
            cur = seen;
        }
    }
    res = (uint32_t)__builtin_amdgcn_readlane((int)res, 63);
    const bool direct = (res & 0x80000000u) != 0u;
    uint32_t slot = (res & 0x7fffffffu) + ei - myev;
    while (pm) {
        const uint32_t j = (uint32_t)__builtin_ctz(pm);
        pm &= pm - 1u;
        const uint32_t word = L.al[N_PPT * tid + j];
        L.al[N_PPT * tid + j] = 0u;
        const uint32_t gpos = t0 + N_PPT * tid + j;
        // (a position with mismatches nearly always holds ONE allele: a walk over the non-zero bytes instead of four unrolled tests, each of
        // which the wavefront issues in full as soon as one of its lanes holds that allele -- round 4)
        for (uint32_t m = nz_bytes(word); m; m &= m - 1u) {
            const uint32_t x = (uint32_t)__builtin_ctz(m) >> 3;
            const uint32_t n = (word >> (8u * x)) & 0xffu;
            tot_add_one(a.tot + 4ull * t0, tmode, N_PPT * tid + j, x, n);
            // the individual rule's "some sample holds >= t reads of x"; a sample that was split into several pairs may reach
            // the threshold only in sum: those positions are marked and decided from the per-sample records (msnv_decide_sites)
            if (n >= a.min_snvs) atomicOr(&a.ind4[gpos >> 3], 1u << (4u * (gpos & 7u) + x));
            else if (split) atomicOr(&a.unc_bits[gpos >> 5], 1u << (gpos & 31u));
            const Pair32 e{gpos, sample << 18 | x << 16 | n};
            if (direct) { if (slot < a.cap_events) a.events[slot] = e; }
            else L.ev[slot] = e;
            ++slot;
        }
    }
}

// This item's coverage partial: 8 positions per thread, u8 when the host bounded the item's summed depth below 256 (bit 0 of
// part_lo), else u16 (an item holds <= 32 pairs of depth < 255).  tc: u16 pairs for positions (0,2) (1,3) (4,6) (5,7).
__device__ __forceinline__ void store_part_row(uint8_t *part, const WorkItem &w, const uint32_t (&tc)[N_PPT / 2], const int tid) {
    uint8_t *row = part + ((uint64_t)w.part_hi << 32 | (w.part_lo & ~15u));
    if (w.part_lo & 1u) {
        const uint32_t lo = (tc[0] & 0xffu) | (tc[1] & 0xffu) << 8 | (tc[0] >> 16 & 0xffu) << 16 | (tc[1] >> 16) << 24;
        const uint32_t hi = (tc[2] & 0xffu) | (tc[3] & 0xffu) << 8 | (tc[2] >> 16 & 0xffu) << 16 | (tc[3] >> 16) << 24;
        *reinterpret_cast<uint2 *>(row + N_PPT * tid) = make_uint2(lo, hi);
    } else {
        *reinterpret_cast<uint4 *>(row + 2u * N_PPT * tid) =
            make_uint4((tc[0] & 0xffffu) | tc[1] << 16, tc[0] >> 16 | (tc[1] & 0xffff0000u), (tc[2] & 0xffffu) | tc[3] << 16, tc[2] >> 16 | (tc[3] & 0xffff0000u));
    }
}

// 64-position blocks of the tile in which this item added to the allele totals: one word per work item, next to its coverage
// partial row in spirit (indexed by the row's slot); the gate kernel ORs the words of the tile's items and does not even read
// the totals of the other blocks.  Eight consecutive threads share a block, so a wavefront's ballot folds to 8 bits.
__device__ __forceinline__ void store_item_dirty(uint32_t *slot_dirty, const WorkItem &w, const bool dirty, const int tid) {
    if (tot_mode_of(w) == 0u) return;                           // the gate kernel reads such a tile's totals whatever they hold
    const unsigned long long b = __ballot(dirty);
    uint32_t byte = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) byte |= ((b >> (8 * i)) & 0xffull) ? 1u << i : 0u;
    if ((tid & 63) == 0 && byte) atomicOr(&slot_dirty[w.slot], byte << (8 * (tid >> 6)));      // four wavefronts, one word of THIS item: no contention
}

// Ring refill of the chunk descriptors, called by every thread right after barrier (A) of chunk c: every
// MAX_CHUNKS_PER_ITEM / 2 chunks the half of the ring that holds finished chunks [c - H, c) receives [c + H, c + 2H).
// Those slots are next read at chunk c + H - 1 (prefetch of c + H), i.e. after at least one more barrier.
__device__ __forceinline__ void desc_refill(ChunkDesc *ring, const ChunkDesc *src, const uint32_t c, const uint32_t nch, const int tid) {
    constexpr uint32_t H = MAX_CHUNKS_PER_ITEM / 2;
    if (c == 0u || (c % H) != 0u || c + H >= nch) return;
    const uint32_t first = c + H, n = min(H, nch - first);
    if ((uint32_t)tid < 2u * n) {
        const uint32_t d = first + (uint32_t)tid / 2u;
        reinterpret_cast<uint4 *>(ring + d % MAX_CHUNKS_PER_ITEM)[tid & 1] = reinterpret_cast<const uint4 *>(src + d)[tid & 1];
    }
}

// ------------------------------------------------------------------------------------------
// msnv_pileup_tiles_narrow32: FOUR lanes per piece, 32 bases per lane.
// Perturbation runs on the earlier 16-bases-per-lane kernel showed the vector-memory (TA/L1) path to be
// the most sensitive resource (+1 sixteen-byte load per round: +15 % time; +25 % VALU: +4 %).  32 bases
// per lane needed 3 wide loads (2 x 16 B quality, 1 x 16 B bases) where two 16-base lanes need 4, and the
// header decode / reference alignment / exception spreading are paid once per 32 bases (-9 % time).
// Round 3: the quality bytes became one bit per base (one 8-byte load) and the kernel, no longer waiting for memory, is bound by
// vector-instruction issue (205 M wavefront instructions per launch x 4 cycles on 1024 SIMDs: DESIGN.md section 8).
// ------------------------------------------------------------------------------------------
constexpr int N32_LANES = 4;
constexpr int N32_GROUPS = N_NT / N32_LANES;   // 64 pieces per round
constexpr int N32_ROUNDS = N_HCAP / N32_GROUPS;   // 2
static_assert(N32_LANES * 32 == SEG_MAX && N32_ROUNDS == 2, "narrow32 is written for 128-base pieces, 128-piece chunks");

__device__ __forceinline__ void narrow_classify32(NarrowLds &L, const uint32_t lq_all, const uint4 sq, const uint32_t P0, const int vhi) {
    const uint32_t wi = P0 >> 3, sh = (P0 & 7u) * 4u;
    const uint32_t w0 = L.ref[wi], w1 = L.ref[wi + 1], w2 = L.ref[wi + 2], w3 = L.ref[wi + 3], w4 = L.ref[wi + 4];
    const uint32_t vmask = (vhi >= 32) ? 0xffffffffu : ((1u << vhi) - 1u);
    const uint32_t lq = lq_all & vmask;                                       // bases below the BQ cutoff (one bit per base: the low-quality column)
    const unsigned long long m = (unsigned long long)lq << (P0 & 7u);      // <= 39 bits: five groups of 8 byte bins
#pragma unroll
    for (int w = 0; w < 5; ++w) {
        const uint32_t byte = (uint32_t)(m >> (8 * w)) & 0xffu;
        if (byte) atomicAdd(&L.exc[wi + w], spread_bits(byte));      // (one test per lane for the first four words instead of one per word: 0.395 vs 0.387 ms, round 4)
    }
    const uint32_t sw[4] = {sq.x, sq.y, sq.z, sq.w};
    const uint32_t rw[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
                            __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh)};
    // the host fills the alignment padding behind a piece (up to the next 16 bases) with the reference it is compared with
    // here, so mismatch flags are masked per 16 bases: words 0-1 exist when the lane has any base, words 2-3 beyond 16
    const uint32_t v01 = (vhi > 0) ? 0x88888888u : 0u, v23 = (vhi > 16) ? 0x88888888u : 0u;
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = sw[k] ^ rw[k];
        e[k] = (((x & 0x77777777u) + 0x77777777u) | x) & (k < 2 ? v01 : v23);
    }
    // mismatch flags of the four words in ONE register: bit 4 j + k <-> base 8 k + j (flag bits 4 j + 3 shifted down by 3 - k)
    uint32_t E = (e[0] >> 3) | (e[1] >> 2) | (e[2] >> 1) | e[3];
    // (pieces that start on 2 or 4 bytes: what follows the lane's last base inside its 16 is the next piece, not reference-filled padding)
    if constexpr (SEQ_ALIGN_LOG2 < 3) E &= L.emask[min(max(vhi, 0), 32)];
    while (E) {                                                      // mismatches (rare)
        const uint32_t b = (uint32_t)__builtin_ctz(E);
        E &= E - 1u;
        const uint32_t k = b & 3u, j = (b >> 2) + 8u * k;
        // the register PAIR that holds the base (one select of 64 bits), then one 64-bit shift: this loop body is what the kernel issues
        // most, and the kernel is bound by vector-instruction issue (bit-field inserts over the four words: 7 instructions more per
        // iteration, +1.8 % kernel time; a chain of ?: became three nested divergent branches: profiles/r03zar_ab_pair64.txt)
        const unsigned long long pair = (k & 2u) ? ((unsigned long long)sw[3] << 32 | sw[2]) : ((unsigned long long)sw[1] << 32 | sw[0]);
        if ((lq >> j) & 1u) continue;                                  // below the BQ cutoff: already in exc
        const uint32_t code = (uint32_t)(pair >> (((b & 1u) << 5) | (b & 28u))) & 0xfu;
        const uint32_t p = P0 + j;
        if ((code & (code - 1u)) == 0u) {
            atomicAdd(&L.al[p], __umul24(code, 0x00204081u) & 0x01010101u);      // one-hot code, bit x -> 1 << 8 x
        } else atomicAdd(&L.exc[p >> 3], 1ull << (8u * (p & 7u)));
    }
}

// Whole-tile work item (dataset.h): the single merged group of the item held every pair of the tile, so behind the last barrier the
// start bins hold the tile's coverage bytes and L.al its mismatching A / C / G / T totals (narrow_pass).  snpCall's gates
// (call_vC.cpp:545-552) and calling rule (:577-601) exactly as msnv_gate_sites applies them to a tile of merged pairs -- no pair
// can be credited with the individual rule here, every allele with >= t reads that is not a population call is left to the merged
// gather ("ask" + eligible alleles) -- and the candidates go to the tile's record list: the gate kernel reads that list instead of
// the tile's per-position state (4 B totals + partial row + rule bits + reference: ~19 KB a tile, as much as the reads themselves
// in a sparse cohort).
struct FusedGateArgs { TileStage *st; const uint32_t *ref_lc; uint32_t *counters; uint32_t vb, ve, min_snvs; int min_cov; double min_frac; uint32_t solo; };
// (two functions, two argument lists of <= 56 bytes: the ABI passes 16 registers of aggregates, a longer list goes through scratch memory -- in every lane of the kernel)
struct FusedSpillArgs { uint8_t *row; uint32_t *tot, *unc_bits, *item_dirty, *stage_ovf, *counters; uint32_t stage_idx, flags_snvs; };   // part_lo flags | min_snvs << 4; row, item_dirty: the work item's partial row and dirty word
__device__ __attribute__((noinline)) bool fused_tile_gate(NarrowLds &L, const FusedGateArgs a, const uint32_t lcb, const bool any, const int tid) {      // lcb: lower-case bits of my 8 positions (ref_lc)
    // (not inlined: the hot loop of the kernel sits exactly at the register step of its occupancy -- 64 for eight workgroups per CU)
    TileStage *const st = a.st;
    // (L.evn counts the candidates: merged items stage no events)
    if (any) {
        const uint2 cv = *reinterpret_cast<uint2 *>(&L.start[2 * tid]);
        const uint4 a0 = *reinterpret_cast<uint4 *>(&L.al[N_PPT * tid]);
        const uint4 a1 = *reinterpret_cast<uint4 *>(&L.al[N_PPT * tid + 4]);
        const uint32_t alw[N_PPT] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const uint32_t refw = L.ref[tid];                                        // nt16 codes of my 8 positions
        const uint32_t vb = a.vb, ve = a.ve;
        // A candidate holds an allele with >= t reads ("ok" below): a handful of a sparse tile's 2048 positions.  The lanes without one among
        // their eight leave here, the others look at a position's four bytes before anything else (round 5: the walk over all eight positions
        // by every lane was a quarter of the kernel's vector instructions on the configs[3] shard: profiles/r05_sparse_ablation.txt).
        // (bytes >= t for 1 <= t <= 127: bit 7 of (byte & 0x7f) + (0x80 - t), or bit 7 of the byte itself)
        const bool filt = a.min_snvs >= 1u && a.min_snvs <= 127u;
        const uint32_t ge = (0x80u - (filt ? a.min_snvs : 1u)) * 0x01010101u;
        const uint32_t orw = a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w;      // (>= each of the eight words, byte for byte)
        if (filt && !((((orw & 0x7f7f7f7fu) + ge) | orw) & 0x80808080u)) goto counted;
#pragma unroll
        for (int j = 0; j < N_PPT; ++j) {
            const uint32_t cov = ((j < 4 ? cv.x : cv.y) >> (8 * (j & 3))) & 0xffu, word = alw[j], p = (uint32_t)(N_PPT * tid + j);
            if (filt && !((((word & 0x7f7f7f7fu) + ge) | word) & 0x80808080u)) continue;
            const uint32_t n[4] = {word & 0xffu, (word >> 8) & 0xffu, (word >> 16) & 0xffu, word >> 24};
            if (cov == 0u || p < vb || p >= ve || (int)cov < a.min_cov || (int)(n[0] + n[1] + n[2] + n[3]) < (int)a.min_snvs) continue;
            const double lim = (double)(int)cov * a.min_frac;                     // call_vC.cpp:588
            const uint32_t rc = (refw >> (4 * j)) & 15u;
            const bool lc = (lcb >> j) & 1u;
            bool ok = false;
            uint32_t pop = 0, ind = 0, elig = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if ((int)n[x] < (int)a.min_snvs) continue;
                const bool is_pop = (double)n[x] >= lim;
                ok = true;                                                        // population call, or some sample may hold >= t reads: the gather decides
                if (lc && rc == (1u << x)) continue;                              // skip-same-base, case-sensitive (call_vC.cpp:580)
                if (is_pop) pop |= 1u << x;
                else if (a.solo) ind |= 1u << x;                                  // the tile's only sample holds them all: the individual rule, decided (call_vC.cpp:593-600)
                else elig |= 1u << x;
            }
            if (!ok) continue;
            const uint32_t slot = atomicAdd(&L.evn, 1u);
            if (slot < STAGE_CAP) st->rec[slot] = StageRec{p | (elig ? 1u << 11 : 0u) | (pop | ind << 4) << 16 | elig << 24, cov, word, 0u};
        }
    }
counted:
    __syncthreads();
    const uint32_t n_cand = L.evn;
    if (tid == 0) st->count = n_cand;                                             // (> STAGE_CAP: msnv_gate_staged leaves the tile alone)
    return n_cand > STAGE_CAP;                                                    // (uniform) more than the list holds: fused_tile_spill
}

// More candidates than the list holds (a handful of tiles of a large sparse cohort: a hypervariable stretch, a tile of noisy
// reads): THIS tile takes the unfused route -- coverage partial row, allele totals and "some sample may hold >= t reads" marks in
// global memory, exactly what narrow_pass<MERGED> and store_part_row leave behind -- and goes on the list msnv_gate_sites works
// through behind msnv_gate_staged.  (Until round 3 one such tile sent the whole dataset back to the unfused pass: 2-3 x the tail
// of BASELINE configs[3] at full scale.)
__device__ __attribute__((noinline)) void fused_tile_spill(NarrowLds &L, const FusedSpillArgs a, const uint32_t t0, const int tid) {
    {
        WorkItem w{};                                                             // (what store_part_row and store_item_dirty look at)
        w.slot = 0u; w.part_lo = a.flags_snvs & 15u; w.part_hi = 0u;
        const uint2 cv = *reinterpret_cast<uint2 *>(&L.start[2 * tid]);
        const uint32_t tc[N_PPT / 2] = {cv.x & 0x00ff00ffu, (cv.x >> 8) & 0x00ff00ffu, cv.y & 0x00ff00ffu, (cv.y >> 8) & 0x00ff00ffu};
        store_part_row(a.row, w, tc, tid);
        const uint32_t tmode = tot_mode_of(w);
        bool dirty = false;
#pragma unroll 1
        for (uint32_t j = 0; j < (uint32_t)N_PPT; ++j) {
            const uint32_t word = L.al[N_PPT * tid + j];
            if (!word) continue;
            dirty = true;
            const uint32_t gpos = t0 + N_PPT * tid + j;
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x) {
                const uint32_t n = (word >> (8u * x)) & 0xffu;
                if (n) {
                    tot_add_one(a.tot + 4ull * t0, tmode, N_PPT * tid + j, x, n);
                    if (n >= (a.flags_snvs >> 4)) atomicOr(&a.unc_bits[gpos >> 5], 1u << (gpos & 31u));
                }
            }
        }
        store_item_dirty(a.item_dirty, w, dirty, tid);
        if (tid == 0) a.stage_ovf[atomicAdd(&a.counters[CNT_STAGE], 1u)] = a.stage_idx;
    }
}

template <bool MERGED, bool FUSED = false, bool DA = false>
__device__ __forceinline__ void pileup_tiles_narrow32_body(PileupArgs a, NarrowLds &L) {
    ev_list(a);
    const WorkItem w = a.work[blockIdx.x];
    const uint32_t t0 = w.tile * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane4 = tid & (N32_LANES - 1), grp = tid / N32_LANES;
    const int b0 = 32 * lane4;

    for (int i = tid; i < (int)(TILE / 8 + 4); i += N_NT)
        L.ref[i] = (i < (int)(TILE / 8)) ? a.ref4[(t0 >> 3) + i] : 0xffffffffu;
    {   // the bins -- start, end, exc, al lie in a row -- are zeroed 16 bytes a lane (round 5: sixteen 4 / 8-byte stores a lane were a tenth of a sparse tile's instructions)
        static_assert(offsetof(NarrowLds, start) == 0 && offsetof(NarrowLds, ref) % 16 == 0 && alignof(NarrowLds) >= 16, "the bins are zeroed as 16-byte words");
        constexpr int n16 = (int)(offsetof(NarrowLds, ref) / 16);
        uint4 *z = reinterpret_cast<uint4 *>(&L);
#pragma unroll
        for (int i = 0; i < (n16 + N_NT - 1) / N_NT; ++i) if (i * N_NT + tid < n16) z[i * N_NT + tid] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) L.evn = 0;
    if (tid < N_NT / 64 + 1) L.carry[tid] = 0;
    if constexpr (SEQ_ALIGN_LOG2 < 3) {
        if (tid < 33) {                                         // flag bit 4 j + k <-> base 8 k + j: the flags of bases 0 .. tid - 1
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // word k holds bases 8 k .. 8 k + 7: the first c of them
                const int c = min(max(tid - 8 * k, 0), 8);
                m |= (c >= 8 ? 0xffffffffu : (1u << (4 * c)) - 1u) & (0x11111111u << k);
            }
            L.emask[tid] = m;
        }
    }
    uint32_t tc[N_PPT / 2] = {0u, 0u, 0u, 0u};                  // coverage totals of my 8 positions over the item's samples (u16 pairs)
    bool dirty = false;                                         // some pass of this item added to the allele totals of my 8 positions

    const uint32_t nch = w.chunk_hi - w.chunk_lo;
    constexpr bool fused = MERGED && FUSED;                       // (a compile-time fact: a run-time flag here costs the kernel its register step)
    // whole-tile items: what the gate behind the last barrier needs of global memory is asked for NOW, next to the first headers -- two
    // links less in the item's chain of dependent loads (a sparse cohort's item is little else than that chain)
    uint32_t g_sidx = 0, g_vb = 0, g_ve = 0, g_lcb = 0;
    if constexpr (fused) {
        g_sidx = a.tile_stage_idx[w.tile]; g_vb = a.tile_vbeg[w.tile]; g_ve = a.tile_vend[w.tile];
        g_lcb = reinterpret_cast<const uint8_t *>(a.ref_lc)[(t0 >> 3) + (uint32_t)tid];
    }
    // the chunk descriptors go through a ring of MAX_CHUNKS_PER_ITEM LDS slots: one deep (sample, tile) pair alone can
    // hold more chunks than that, so half of the ring is refilled every MAX_CHUNKS_PER_ITEM / 2 chunks (desc_refill)
    // the headers of chunk 0 do not wait for the descriptor stream: their descriptor came with the work item
    constexpr bool H4 = HDR4 && !MERGED;                          // ordinary work items: 4-byte headers, offsets relative to the chunk's base
    auto fetch_hdr = [&](const uint64_t hdr_base) -> uint2 {
        if constexpr (H4) return make_uint2(a.hdr4[hdr_base + (uint32_t)tid], 0u);
        else return *reinterpret_cast<const uint2 *>(a.hdr8 + hdr_base + tid);
    };
    // LDS staging of the headers: 4-byte slots for the 4-byte form (half the LDS traffic of the staging)
    auto put_hdr = [&](const uint32_t buf, const uint2 h) {
        if constexpr (H4) reinterpret_cast<uint32_t *>(L.hdr[buf])[tid] = h.x; else L.hdr[buf][tid] = h;
    };
    auto get_hdr = [&](const uint32_t buf, const uint32_t i) -> uint2 {
        if constexpr (H4) return make_uint2(reinterpret_cast<const uint32_t *>(L.hdr[buf])[i], 0u); else return L.hdr[buf][i];
    };
    uint2 hreg0 = make_uint2(0, 0);
    if (nch && tid < N_HCAP && (uint32_t)tid < (w.first.nrd_flags & 0xffffu)) hreg0 = fetch_hdr(w.first.hdr_base);
    for (uint32_t i = tid; i < min(nch, (uint32_t)MAX_CHUNKS_PER_ITEM) * 2; i += N_NT)
        reinterpret_cast<uint4 *>(L.desc)[i] = reinterpret_cast<const uint4 *>(a.chunks + w.chunk_lo)[i];
    __syncthreads();
    // Headers of chunk c sit in L.hdr[c & 1]: written at the end of iteration c - 1 (chunk 0: here) from the registers that
    // prefetched them one iteration earlier, visible behind barrier (B) of that iteration.  The column loads of chunk c + 1
    // are issued right behind that barrier: when chunk c closes its (sample, tile) pair they are in flight while the
    // per-sample pass runs (what a shallow pair needs: its chunk iteration is otherwise one exposed dependent-load latency).
    auto load_hdr = [&](const uint32_t c) -> uint2 {
        uint2 h = make_uint2(0, 0);                                   // slots beyond nrd hold length 0
        if (c < nch && tid < N_HCAP && (uint32_t)tid < (L.desc[c % MAX_CHUNKS_PER_ITEM].nrd_flags & 0xffffu))
            h = fetch_hdr(L.desc[c % MAX_CHUNKS_PER_ITEM].hdr_base);
        return h;
    };
    if (tid < N_HCAP) put_hdr(0u, hreg0);                         // chunk 0: fetched through the descriptor that came with the work item
    uint2 hreg = load_hdr(1);
    uint4 sq[N32_ROUNDS]; uint2 ql[N32_ROUNDS]; uint32_t P0[N32_ROUNDS], qsh[N32_ROUNDS]; int vh[N32_ROUNDS];      // ql >> qsh: the lane's 32 low-quality flags
    auto issue_loads = [&](const uint32_t c) {
        uint64_t sbase = L.desc[c % MAX_CHUNKS_PER_ITEM].seq_base;
        // (4-byte headers: the chunk's base is the same in every lane -- scalar registers, so that a lane's address is base + 32-bit offset)
        if constexpr (H4) sbase = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sbase);
        const uint8_t *seq = a.seq + sbase;
        // low-quality flags: one bit per base, flag of the column's nibble n at bit n (chunk bases are multiples of SEQ_ALIGN bytes = 4 bases)
        const uint8_t *qlow = a.qual + (sbase >> 2);
        const uint32_t qrem = 2u * ((uint32_t)sbase & 3u);
#pragma unroll
        for (int i = 0; i < N32_ROUNDS; ++i) {
            const uint2 h = get_hdr(c & 1u, (uint32_t)(grp + i * N32_GROUPS));     // all zero for empty slots
            const uint32_t len = (h.x >> 11) & 0xffu;                // (bits 19-26: index of the piece's pair inside a merged group, for the gather)
            const uint32_t s = h.x & (TILE - 1u);
            // seq byte offset of the piece inside the sample; a merged group's pieces come from all over the column: their ABSOLUTE offset / SEQ_ALIGN is 37 bits wide (bits 27+ of the first word)
            vh[i] = min(max((int)len - b0, 0), 32);
            sq[i] = any_uint4(); ql[i] = make_uint2(0u, 0u); qsh[i] = 0u;   // (sq never observed: vh masks every use)
            if constexpr (H4) {
                const uint32_t so = (h.x >> 19) << SEQ_ALIGN_LOG2;       // < 16 KB from the chunk's base
                if (vh[i] > 0) {                                     // lanes past the end of the piece load nothing
                    const uint32_t qbit = qrem + 2u * so + (uint32_t)b0, sq_o = so + (uint32_t)(b0 >> 1);
                    __builtin_memcpy(&ql[i], qlow + (qbit >> 3), 8);
                    qsh[i] = qbit & 7u;
                    __builtin_memcpy(&sq[i], seq + sq_o, 16);
                }
            } else {
                const uint64_t so = (MERGED ? ((uint64_t)(h.x >> 27) << 32 | h.y) : (uint64_t)h.y) << SEQ_ALIGN_LOG2;
                if (vh[i] > 0) {                                     // lanes past the end of the piece load nothing
                    const uint64_t qbit = (uint64_t)qrem + 2ull * so + (uint32_t)b0;
                    __builtin_memcpy(&ql[i], qlow + (qbit >> 3), 8);
                    qsh[i] = (uint32_t)qbit & 7u;
                    __builtin_memcpy(&sq[i], seq + so + (uint32_t)(b0 >> 1), 16);
                }
            }
            P0[i] = vh[i] > 0 ? s + (uint32_t)b0 : 0u;
        }
    };
    __syncthreads();
    if (nch) issue_loads(0u);
    bool prev_last = false;                                          // the bins start out zero

    for (uint32_t c = 0; c < nch; ++c) {
        const ChunkDesc cd = L.desc[c % MAX_CHUNKS_PER_ITEM];
        const bool last_chunk = (cd.nrd_flags >> 16) != 0u;
        if (prev_last) __syncthreads();                              // (A): the pass of the previous pair left every bin zero
        desc_refill(L.desc, a.chunks + w.chunk_lo, c, nch, tid);
        if (!MERGED && !DA && L.evn >= (uint32_t)(N_EVCAP / 2)) flush_events<NarrowLds, N_NT, N_EVCAP>(L, a, tid);
        if (tid < N_HCAP) {
            const uint32_t hx = get_hdr(c & 1u, (uint32_t)tid).x;
            const uint32_t s = hx & (TILE - 1u), sb = s + ((hx >> 11) & 0xffu);
            if (sb != s) {                                           // coverage difference array: +1 at the start, -1 behind the end
                atomicAdd(&L.start[s >> 2], 1u << (8u * (s & 3u)));
                atomicAdd(&L.end[sb >> 2], 1u << (8u * (sb & 3u)));
                // a piece that covers the last position before a wavefront's quarter of the tile feeds that quarter's carry
                if ((s >> 9) != (sb >> 9)) atomicAdd(&L.carry[sb >> 9], 1u);
            }
        }
#pragma unroll
        for (int i = 0; i < N32_ROUNDS; ++i)
            if (__any(vh[i] > 0)) narrow_classify32(L, lowq_bits(ql[i], qsh[i]), sq[i], P0[i], vh[i]);

        if (tid < N_HCAP) put_hdr((c + 1u) & 1u, hreg);               // headers of chunk c + 1 (zeros behind the last chunk)
        hreg = load_hdr(c + 2u);
        __syncthreads();                                            // (B): this chunk is in the bins; the next chunk's headers are visible
        if (c + 1u < nch) issue_loads(c + 1u);                       // in flight under the per-sample pass
        if (last_chunk) narrow_pass<NarrowLds, 0, MERGED, fused, DA>(L, a, tc, dirty, t0, tid, lane, wave, cd.sample, cd.pair, cd.pad, tot_mode_of(w));
        prev_last = last_chunk;
    }
    __syncthreads();
    if (MERGED && fused) {
        const uint32_t sidx = g_sidx;
        const FusedGateArgs fa{a.tile_stage + sidx, a.ref_lc, a.counters, g_vb, g_ve, a.min_snvs, a.min_cov, a.min_frac, w.pair_hi - w.pair_lo == 1u ? 1u : 0u};
        if (fused_tile_gate(L, fa, g_lcb, nch != 0u, tid)) {
            const FusedSpillArgs sa{a.part + ((uint64_t)w.part_hi << 32 | (w.part_lo & ~15u)), a.tot, a.unc_bits, a.slot_dirty + w.slot, a.stage_ovf, a.counters, sidx, (w.part_lo & 15u) | a.min_snvs << 4};
            fused_tile_spill(L, sa, t0, tid);
        }
        return;
    }
    if (!MERGED && !DA) flush_events<NarrowLds, N_NT, N_EVCAP>(L, a, tid);
    store_part_row(a.part, w, tc, tid);
    store_item_dirty(a.slot_dirty, w, dirty, tid);
}

// ------------------------------------------------------------------------------------------
// msnv_pileup_tiles_lean (round 6): a SPARSE tile's whole-tile item at a cost proportional to its pieces, not to its 2 048 positions.
// The item above zeroes four rows of bins, adds every piece's start / end and its low-quality bases to them, runs a prefix sum and the gates
// over all 2 048 positions -- for a tile of the configs[3] shard that holds ~100 pieces and ~4 positions anyone will ever ask about: half of the
// kernel's vector instructions were that per-tile work (profiles/r05_sparse_ablation.txt).  Here, for an item of at most LEAN_MAX_CHUNKS chunks:
//   * only the allele bins exist.  A piece's lanes XOR their 32 bases against the reference as before; a mismatching base above the -Q cutoff
//     is added to its position's allele bytes, and the add that is an allele's t-th read lists the position (the atomic's return value
//     says so) -- the tile's CANDIDATES, a handful;
//   * no start / end bins, no low-quality bins, no prefix sum: the coverage of a candidate is counted from the pieces that overlap it, out of
//     the REGISTERS that still hold them -- every lane tests its 32 bases' range against the handful of candidates; a piece that covers one
//     contributes unless its base there is below the cutoff (its flag bit) or neither the reference's nor one of A C G T (its nibble): what
//     the bins' arithmetic says, byte for byte.  (First form: a wavefront per candidate over the item's headers, the covering pieces' bytes
//     asked for from global memory -- two more links in every item's chain of dependent loads: 0.56 ms against the ordinary body's 0.43);
//   * the gates and the calling rule of fused_tile_gate on those few positions, into the same record list.
// More candidates than the list holds, or more than LEAN_MAX_CHUNKS chunks: the workgroup runs the item the ordinary way (the body above).
// The kernel is a launch of its own behind msnv_pileup_tiles_narrow32 (which keeps its registers: a run-time switch inside it costs the
// dominant kernel its eight workgroups per CU); MSNV_LEAN=0 sends the whole-tile items through the old launch.
__device__ __forceinline__ void lean_classify32(NarrowLds &L, const uint32_t lq_all, const uint4 sq, const uint32_t P0, const int vhi, const uint32_t t) {
    const uint32_t wi = P0 >> 3, sh = (P0 & 7u) * 4u;
    const uint32_t w0 = L.ref[wi], w1 = L.ref[wi + 1], w2 = L.ref[wi + 2], w3 = L.ref[wi + 3], w4 = L.ref[wi + 4];
    const uint32_t vmask = (vhi >= 32) ? 0xffffffffu : ((1u << vhi) - 1u);
    const uint32_t lq = lq_all & vmask;
    const uint32_t sw[4] = {sq.x, sq.y, sq.z, sq.w};
    const uint32_t rw[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
                            __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh)};
    const uint32_t v01 = (vhi > 0) ? 0x88888888u : 0u, v23 = (vhi > 16) ? 0x88888888u : 0u;      // (narrow_classify32: the padding behind a piece is reference-filled per 16 bases)
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = sw[k] ^ rw[k];
        e[k] = (((x & 0x77777777u) + 0x77777777u) | x) & (k < 2 ? v01 : v23);
    }
    uint32_t E = (e[0] >> 3) | (e[1] >> 2) | (e[2] >> 1) | e[3];
    if constexpr (SEQ_ALIGN_LOG2 < 3) E &= L.emask[min(max(vhi, 0), 32)];
    while (E) {
        const uint32_t b = (uint32_t)__builtin_ctz(E);
        E &= E - 1u;
        const uint32_t k = b & 3u, j = (b >> 2) + 8u * k;
        const unsigned long long pair = (k & 2u) ? ((unsigned long long)sw[3] << 32 | sw[2]) : ((unsigned long long)sw[1] << 32 | sw[0]);
        if ((lq >> j) & 1u) continue;                                  // below the BQ cutoff: not counted anywhere
        const uint32_t code = (uint32_t)(pair >> (((b & 1u) << 5) | (b & 28u))) & 0xfu;
        if ((code & (code - 1u)) != 0u || code == 0u) continue;        // neither A, C, G nor T: not counted (the candidates' coverage leaves it out); (0: never stored -- the packers write the reference's code for '=')
        const uint32_t p = P0 + j, ash = (uint32_t)__builtin_ctz(code) << 3;
        const uint32_t old = atomicAdd(&L.al[p], 1u << ash);
        if (((old >> ash) & 0xffu) + 1u == t) {                           // this allele's t-th read: the position is listed (twice when two alleles get there: the gate keeps the first entry)
            const uint32_t slot = atomicAdd(&L.evn, 1u);
            if (slot < (uint32_t)N_EVCAP) L.ev[slot] = Pair32{p, 0u};                // (.y: its coverage, counted behind the barrier)
        }
    }
}
// true: the item is done (its record list written); false: the caller runs it the ordinary way
constexpr uint32_t LEAN_MAX_CHUNKS = 2;          // the pieces of an item stay in the registers until its candidates are counted: two chunks' worth, 256 pieces (three: 64 VGPRs + 16 bytes
                                                // of scratch a lane, six rounds of header decoding for every item: 0.391 ms against 0.366; one: 0.388)
__device__ __forceinline__ bool lean_tile(const PileupArgs &a, NarrowLds &L, const WorkItem &w, const uint32_t nch) {
    const int tid = threadIdx.x;
    const int lane4 = tid & (N32_LANES - 1), grp = tid / N32_LANES, b0 = 32 * lane4;
    const uint32_t t0 = w.tile * TILE;
    const uint32_t sidx = a.tile_stage_idx[w.tile], vb = a.tile_vbeg[w.tile], ve = a.tile_vend[w.tile];      // (asked for now: the gate's links of the item's chain of dependent loads)
    for (int i = tid; i < (int)(TILE / 8 + 4); i += N_NT) L.ref[i] = (i < (int)(TILE / 8)) ? a.ref4[(t0 >> 3) + i] : 0xffffffffu;
    {
        uint4 *z = reinterpret_cast<uint4 *>(L.al);
#pragma unroll
        for (int i = 0; i < (int)(TILE / 4) / N_NT; ++i) z[i * N_NT + tid] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) { L.evn = 0u; L.ev_base = 0u; }
    if constexpr (SEQ_ALIGN_LOG2 < 3) {
        if (tid < 33) {
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = min(max(tid - 8 * k, 0), 8); m |= (c >= 8 ? 0xffffffffu : (1u << (4 * c)) - 1u) & (0x11111111u << k); }
            L.emask[tid] = m;
        }
    }
    // a lane's share of the item: for each of the (two per chunk) pieces it has a part in, 32 bases, their low-quality flags, where they lie in the tile
    constexpr int LR = N32_ROUNDS * (int)LEAN_MAX_CHUNKS;
    uint4 sq[LR]; uint32_t lq[LR], P0[LR]; int vh[LR];
    ChunkDesc cds[LEAN_MAX_CHUNKS];                                  // (a chunk's descriptor behind the first: one more link in that item's chain)
    cds[0] = w.first;
#pragma unroll
    for (uint32_t c = 1; c < LEAN_MAX_CHUNKS; ++c) { cds[c] = ChunkDesc{}; if (c < nch) cds[c] = a.chunks[w.chunk_lo + c]; }
    // the headers through LDS (a load per piece; straight from global memory, four lanes a header: 0.394 ms against 0.382): the rows of the
    // start / end bins this kernel does not use hold them, 128 to a chunk
    static_assert(sizeof(L.start) + sizeof(L.end) >= LEAN_MAX_CHUNKS * N_HCAP * sizeof(uint2) && offsetof(NarrowLds, end) == sizeof(L.start), "the lean kernel's header staging");
    uint2 *const hst = reinterpret_cast<uint2 *>(L.start);
    if (tid < N_HCAP) {
#pragma unroll
        for (uint32_t c = 0; c < LEAN_MAX_CHUNKS; ++c) {
            uint2 h = make_uint2(0u, 0u);
            if ((uint32_t)tid < (cds[c].nrd_flags & 0xffffu)) h = *reinterpret_cast<const uint2 *>(a.hdr8m + cds[c].hdr_base + (uint32_t)tid);
            hst[c * N_HCAP + (uint32_t)tid] = h;
        }
    }
    __syncthreads();                                                 // (the bins are zero, the reference and the headers staged)
    {
        uint2 ql[LR]; uint32_t qsh[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const uint64_t sbase = cds[r / N32_ROUNDS].seq_base;
            const uint8_t *seq = a.seq + sbase, *qlow = a.qual + (sbase >> 2);
            const uint32_t qrem = 2u * ((uint32_t)sbase & 3u);
            const uint2 h = hst[(r / N32_ROUNDS) * N_HCAP + grp + (r % N32_ROUNDS) * N32_GROUPS];      // (all zero behind a chunk's last piece, and for the chunks an item does not have)
            const uint32_t len = (h.x >> 11) & 0xffu, s = h.x & (TILE - 1u);
            vh[r] = min(max((int)len - b0, 0), 32);
            sq[r] = make_uint4(0u, 0u, 0u, 0u); ql[r] = make_uint2(0u, 0u); qsh[r] = 0u;
            if (vh[r] > 0) {
                const uint64_t so = ((uint64_t)(h.x >> 27) << 32 | h.y) << SEQ_ALIGN_LOG2;
                const uint64_t qbit = (uint64_t)qrem + 2ull * so + (uint32_t)b0;
                __builtin_memcpy(&ql[r], qlow + (qbit >> 3), 8);
                qsh[r] = (uint32_t)qbit & 7u;
                __builtin_memcpy(&sq[r], seq + so + (uint32_t)(b0 >> 1), 16);
            }
            P0[r] = vh[r] > 0 ? s + (uint32_t)b0 : 0u;
        }
#pragma unroll
        for (int r = 0; r < LR; ++r) lq[r] = lowq_bits(ql[r], qsh[r]);
    }
#pragma unroll
    for (int r = 0; r < LR; ++r)
        if (__any(vh[r] > 0)) lean_classify32(L, lq[r], sq[r], P0[r], vh[r], a.min_snvs);
    __syncthreads();
    const uint32_t n_cand = L.evn;
    if (n_cand > (uint32_t)N_EVCAP) return false;
    // ---- the candidates' coverage, from the registers that still hold the pieces: a lane whose 32 bases cover a candidate looks its base up
    for (uint32_t ci = 0; ci < n_cand; ++ci) {
        const uint32_t p = L.ev[ci].x;
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const uint32_t j = p - P0[i];
            if (j < (uint32_t)vh[i]) {                                   // (unsigned: P0 <= p < P0 + my bases; a lane without bases has none)
                const uint32_t word = j < 16u ? (j < 8u ? sq[i].x : sq[i].y) : (j < 24u ? sq[i].z : sq[i].w);
                const uint32_t nib = (word >> (4u * (j & 7u))) & 15u, rc = (L.ref[p >> 3] >> (4u * (p & 7u))) & 15u;
                if (!((lq[i] >> j) & 1u) && (nib == rc || (nib & (nib - 1u)) == 0u)) atomicAdd(&L.ev[ci].y, 1u);
            }
        }
    }
    __syncthreads();
    // ---- the gates and the calling rule (fused_tile_gate's, on the candidates alone): a thread per candidate
    TileStage *const st = a.tile_stage + sidx;
    if ((uint32_t)tid < n_cand) {
        const uint32_t p = L.ev[tid].x, cov = L.ev[tid].y, word = L.al[p];
        const uint32_t n[4] = {word & 0xffu, (word >> 8) & 0xffu, (word >> 16) & 0xffu, word >> 24};
        bool twice = false;                                                       // (a position two alleles of which reached t: its first entry speaks)
        for (uint32_t q = 0; q < (uint32_t)tid; ++q) twice |= L.ev[q].x == p;
        if (!(twice || cov == 0u || p < vb || p >= ve || (int)cov < a.min_cov || (int)(n[0] + n[1] + n[2] + n[3]) < (int)a.min_snvs)) {
            const double lim = (double)(int)cov * a.min_frac;                 // call_vC.cpp:588
            const uint32_t rc = (L.ref[p >> 3] >> (4u * (p & 7u))) & 15u;
            const bool lc = (reinterpret_cast<const uint8_t *>(a.ref_lc)[(t0 + p) >> 3] >> (p & 7u)) & 1u;
            const uint32_t solo = w.pair_hi - w.pair_lo == 1u ? 1u : 0u;
            bool ok = false;
            uint32_t pop = 0, ind = 0, elig = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if ((int)n[x] < (int)a.min_snvs) continue;
                const bool is_pop = (double)n[x] >= lim;
                ok = true;
                if (lc && rc == (1u << x)) continue;                          // skip-same-base, case-sensitive (call_vC.cpp:580)
                if (is_pop) pop |= 1u << x;
                else if (solo) ind |= 1u << x;
                else elig |= 1u << x;
            }
            if (ok) {
                const uint32_t slot = atomicAdd(&L.ev_base, 1u);
                if (slot < STAGE_CAP) st->rec[slot] = StageRec{p | (elig ? 1u << 11 : 0u) | (pop | ind << 4) << 16 | elig << 24, cov, word, 0u};
            }
        }
    }
    __syncthreads();
    const uint32_t n_rec = L.ev_base;
    if (n_rec > STAGE_CAP) return false;                                          // (the ordinary way counts again and takes the tile through fused_tile_spill)
    if (tid == 0) st->count = n_rec;
    return true;
}
__global__ __launch_bounds__(N_NT, MSNV_N32_WAVES) void msnv_pileup_tiles_lean(PileupArgs a) {
    __shared__ NarrowLds L;
    a.hdr8 = a.hdr8m;
    a.work += a.n_fused_lo;                                       // (the whole-tile items: the last ones of the work list)
    const WorkItem w = a.work[blockIdx.x];
    const uint32_t nch = w.chunk_hi - w.chunk_lo;
    if (nch >= 1u && nch <= LEAN_MAX_CHUNKS && a.min_snvs >= 1u && a.min_snvs <= 127u && lean_tile(a, L, w, nch)) return;
    __syncthreads();
    pileup_tiles_narrow32_body<true, true>(a, L);
}

// Work items [0, n_narrow) are ordinary ones; the items behind them hold MERGED groups of shallow (sample, tile) pairs: a chunk
// holds pieces of several samples, one pass per group instead of one per pair, no per-sample coverage spill and no allele
// events (narrow_pass).  A pair of ~20 pieces costs a chunk iteration and a pass over all 2048 positions whatever it holds
// (1600 samples at 1x ran at 24 % of the roofline).  One launch for both kinds: a handful of merged items (the partial last
// tile of every contig) would otherwise run as a launch of its own with the chip idle around it (17.8 us on the benchmark shape).
__global__ __launch_bounds__(N_NT, MSNV_N32_WAVES) void msnv_pileup_tiles_narrow32(PileupArgs a) {
    __shared__ NarrowLds L;                                     // ONE instance for both kinds of work item (8 workgroups per CU)
    if (blockIdx.x < a.n_narrow) pileup_tiles_narrow32_body<false>(a, L);
    else if (blockIdx.x < a.n_fused_lo) { a.hdr8 = a.hdr8m; pileup_tiles_narrow32_body<true>(a, L); }
    else { a.hdr8 = a.hdr8m; pileup_tiles_narrow32_body<true, true>(a, L); }      // whole-tile items (the last ones; none when the pass runs unfused)
}
// The same launch for noisy reads (pack.cpp: allele planes): the ordinary work items write their pairs' allele counts as byte planes;
// merged groups and whole-tile items keep their own bookkeeping (a group's few per-sample cells are recomputed from the pieces).
__global__ __launch_bounds__(N_NT, MSNV_N32_WAVES) void msnv_pileup_tiles_narrow32_planes(PileupArgs a) {
    __shared__ NarrowLds L;
    if (blockIdx.x < a.n_narrow) pileup_tiles_narrow32_body<false, false, true>(a, L);
    else if (blockIdx.x < a.n_fused_lo) { a.hdr8 = a.hdr8m; pileup_tiles_narrow32_body<true>(a, L); }
    else { a.hdr8 = a.hdr8m; pileup_tiles_narrow32_body<true, true>(a, L); }
}
// ------------------------------------------------------------------------------------------
// msnv_pileup_tiles_dense: the narrow algorithm over the DENSE layout (dataset.h: BLK_*, pack.cpp: relayout_dense).
// The pieces of a (sample, tile) pair form a stream of 32-base blocks without alignment padding; ONE lane owns ONE
// block: 2 x 16 B of qualities + 16 B of bases, all 16-byte aligned and contiguous across the wavefront, and a 4-byte
// descriptor (no LDS staging of headers).  A block holds bits [0, nA) of one piece and at most bits [sB, 32) of the
// next, so the classification runs on up to two (position of bit 0, bit range) segments; the position of bit 0 of the
// second segment may be up to 32 before the tile start, which is why the reference words and the exception bins carry
// 32 positions of front padding.  Versus the per-piece layout: no padding bytes (-6 % HBM traffic) and no idle lanes
// behind short pieces.  Measured (same box, pileup kernel, ms): reads of 50 bases: pieces 0.802 / dense 0.658;
// 100: 0.670 / 0.677; 150: 0.618 / 0.672; 250: 0.567 / 0.645 -- the second segment costs what the padding saves at
// 100 bases and more, so finalize picks this layout only for short pieces (pack.cpp: layout choice).
// ------------------------------------------------------------------------------------------
constexpr int D_PAD = 4;                       // 4 exception words / reference words = 32 positions of front padding
struct DenseLds {
    uint32_t start[TILE / 4 + 4];
    uint32_t end[TILE / 4 + 4];
    unsigned long long exc[TILE / 8 + D_PAD + 8];
    uint32_t al[TILE];
    uint32_t ref[TILE / 8 + D_PAD + 8];
    Pair32   ev[N_EVCAP];
    ChunkDesc desc[MAX_CHUNKS_PER_ITEM];
    uint32_t carry[N_NT / 64 + 1];             // depth at the left edge of each wavefront's quarter of the tile ([4]: scratch for pieces that end at the tile end)
    uint32_t evn, ev_base;
};

// nibble flags (bit 4j+3) of the bases j in [lo, hi) of the 8-base word k
__device__ __forceinline__ uint32_t range_nibbles(const int lo, const int hi, const int k) {
    const int h = min(max(hi - 8 * k, 0), 8), l = min(max(lo - 8 * k, 0), 8);
    const uint32_t below_h = (h == 8) ? 0xffffffffu : ((1u << (4 * h)) - 1u);
    const uint32_t below_l = (l == 8) ? 0xffffffffu : ((1u << (4 * l)) - 1u);
    return below_h & ~below_l & 0x88888888u;
}

// One segment of a block: bits [lo, hi) lie at padded positions P + j (padded = tile position + 32).
__device__ __forceinline__ void dense_segment(DenseLds &L, const uint4 sq, const uint32_t lq, const uint32_t P, const int lo, const int hi) {
    const uint32_t wi = P >> 3, sh = (P & 7u) * 4u;
    const uint32_t below_hi = (hi >= 32) ? 0xffffffffu : ((1u << hi) - 1u);
    const uint32_t below_lo = (lo >= 32) ? 0xffffffffu : ((1u << lo) - 1u);
    const uint32_t lqm = lq & below_hi & ~below_lo;
    const unsigned long long m = (unsigned long long)lqm << (P & 7u);
#pragma unroll
    for (int w = 0; w < 5; ++w) {
        const uint32_t byte = (uint32_t)(m >> (8 * w)) & 0xffu;
        if (byte) atomicAdd(&L.exc[wi + w], spread_bits(byte));
    }
    const uint32_t w0 = L.ref[wi], w1 = L.ref[wi + 1], w2 = L.ref[wi + 2], w3 = L.ref[wi + 3], w4 = L.ref[wi + 4];
    const uint32_t sw[4] = {sq.x, sq.y, sq.z, sq.w};
    const uint32_t rw[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
                            __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh)};
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = nz_nibbles(sw[k] ^ rw[k]) & range_nibbles(lo, hi, k);
    uint32_t E = (e[0] >> 3) | (e[1] >> 2) | (e[2] >> 1) | e[3];     // bit 4 j + k <-> base 8 k + j (narrow_classify32)
    while (E) {                                                      // mismatches (rare)
        const uint32_t b = (uint32_t)__builtin_ctz(E);
        E &= E - 1u;
        const uint32_t k = b & 3u, j = (b >> 2) + 8u * k;
        const unsigned long long pair = (k & 2u) ? ((unsigned long long)sw[3] << 32 | sw[2]) : ((unsigned long long)sw[1] << 32 | sw[0]);   // (as in narrow_classify32)
        if ((lq >> j) & 1u) continue;                                  // below the BQ cutoff: already in exc
        const uint32_t code = (uint32_t)(pair >> (((b & 1u) << 5) | (b & 28u))) & 0xfu;
        const uint32_t pp = P + j;                                     // padded position of the base
        if ((code & (code - 1u)) == 0u) atomicAdd(&L.al[pp - 32u], __umul24(code, 0x00204081u) & 0x01010101u);
        else atomicAdd(&L.exc[pp >> 3], 1ull << (8u * (pp & 7u)));
    }
}

__global__ __launch_bounds__(N_NT) void msnv_pileup_tiles_dense(PileupArgs a) {
    ev_list(a);
    __shared__ DenseLds L;
    const WorkItem w = a.work[blockIdx.x];
    const uint32_t t0 = w.tile * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < (int)(TILE / 8 + D_PAD + 8); i += N_NT) {
        const int r = i - D_PAD;
        L.ref[i] = (r >= 0 && r < (int)(TILE / 8)) ? a.ref4[(t0 >> 3) + r] : 0xffffffffu;
        L.exc[i] = 0;
    }
    for (int i = tid; i < (int)(TILE / 4 + 4); i += N_NT) { L.start[i] = 0; L.end[i] = 0; }
    for (int i = tid; i < (int)TILE; i += N_NT) L.al[i] = 0;
    if (tid == 0) L.evn = 0;
    if (tid < N_NT / 64 + 1) L.carry[tid] = 0;
    uint32_t tc[N_PPT / 2] = {0u, 0u, 0u, 0u};                  // coverage totals of my 8 positions over the item's samples (u16 pairs)
    bool dirty = false;                                         // some pass of this item added to the allele totals of my 8 positions

    const uint32_t nch = w.chunk_hi - w.chunk_lo;
    // the chunk descriptors go through a ring of MAX_CHUNKS_PER_ITEM LDS slots: one deep (sample, tile) pair alone can
    // hold more chunks than that, so half of the ring is refilled every MAX_CHUNKS_PER_ITEM / 2 chunks (desc_refill)
    for (uint32_t i = tid; i < min(nch, (uint32_t)MAX_CHUNKS_PER_ITEM) * 2; i += N_NT)
        reinterpret_cast<uint4 *>(L.desc)[i] = reinterpret_cast<const uint4 *>(a.chunks + w.chunk_lo)[i];
    __syncthreads();
    constexpr int ROUNDS = DENSE_CHUNK_BLOCKS / N_NT;           // 2
    uint32_t dnext[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        dnext[r] = BLK_EMPTY;
        if (nch && (uint32_t)(r * N_NT + tid) < (L.desc[0].nrd_flags & 0xffffu)) dnext[r] = a.blk[L.desc[0].hdr_base + (uint32_t)(r * N_NT + tid)];
    }

    for (uint32_t c = 0; c < nch; ++c) {
        const ChunkDesc cd = L.desc[c % MAX_CHUNKS_PER_ITEM];
        const bool last_chunk = (cd.nrd_flags >> 16) != 0u;
        const uint32_t nblk = cd.nrd_flags & 0xffffu;
        uint32_t dcur[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            dcur[r] = dnext[r];
            dnext[r] = BLK_EMPTY;
            if (c + 1 < nch && (uint32_t)(r * N_NT + tid) < (L.desc[(c + 1) % MAX_CHUNKS_PER_ITEM].nrd_flags & 0xffffu))
                dnext[r] = a.blk[L.desc[(c + 1) % MAX_CHUNKS_PER_ITEM].hdr_base + (uint32_t)(r * N_NT + tid)];
        }
        __syncthreads();                                            // (A) bins of the previous sample are zeroed
        desc_refill(L.desc, a.chunks + w.chunk_lo, c, nch, tid);
        if (L.evn >= (uint32_t)(N_EVCAP / 2)) flush_events<DenseLds, N_NT, N_EVCAP>(L, a, tid);

        const uint8_t *seq = a.seq + cd.seq_base;
        const uint64_t qbit0 = 2ull * cd.seq_base;                  // the chunk's first flag in the low-quality column (one bit per base)
        uint4 sq[ROUNDS]; uint2 ql[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t b = (uint32_t)(r * N_NT + tid);
            sq[r] = any_uint4(); ql[r] = make_uint2(0u, 0u);   // (sq never observed: an empty descriptor masks every use)
            if (b < nblk) {
                ql[r] = lowq_fetch(a.qual, qbit0 + 32ull * b);
                sq[r] = *reinterpret_cast<const uint4 *>(seq + 16ull * b);
            }
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (!__any((uint32_t)(r * N_NT + tid) < nblk)) continue;
            const uint32_t d = dcur[r];
            const uint32_t P0A = d & 2047u, nA = (d >> 11) & 63u, PBv = (d >> 17) & 0xfffu;
            const bool hasB = PBv != BLK_NO_B;
            const uint32_t sB = (nA + 1u) & ~1u;
            if (d & BLK_START_A) atomicAdd(&L.start[P0A >> 2], 1u << (8u * (P0A & 3u)));
            if (d & BLK_END_A) { const uint32_t e = P0A + nA; atomicAdd(&L.end[e >> 2], 1u << (8u * (e & 3u))); }
            // a segment that covers the last position before a wavefront's quarter of the tile feeds that quarter's carry
            if (d != BLK_EMPTY && (P0A >> 9) != ((P0A + nA) >> 9)) atomicAdd(&L.carry[(P0A + nA) >> 9], 1u);
            if (hasB) {
                const uint32_t s = PBv + sB - 32u;
                if ((s >> 9) != (PBv >> 9)) atomicAdd(&L.carry[PBv >> 9], 1u);
                atomicAdd(&L.start[s >> 2], 1u << (8u * (s & 3u)));
                if (d & BLK_END_B) { const uint32_t e = PBv; atomicAdd(&L.end[e >> 2], 1u << (8u * (e & 3u))); }   // s + (32 - sB) = PBv
            }
            const uint32_t lq = lowq_bits(ql[r], (uint32_t)qbit0 & 7u);
            dense_segment(L, sq[r], lq, P0A + 32u, 0, (int)nA);
            if (__any(hasB)) dense_segment(L, sq[r], lq, hasB ? PBv : 32u, hasB ? (int)sB : 32, 32);
        }
        if (last_chunk) {
            __syncthreads();                                        // (B)
            narrow_pass<DenseLds, D_PAD>(L, a, tc, dirty, t0, tid, lane, wave, cd.sample, cd.pair, cd.pad, tot_mode_of(w));
        }
    }
    __syncthreads();
    flush_events<DenseLds, N_NT, N_EVCAP>(L, a, tid);
    store_part_row(a.part, w, tc, tid);
    store_item_dirty(a.slot_dirty, w, dirty, tid);
}

// ------------------------------------------------------------------------------------------
// msnv_gate_sites: snpCall's two gates (call_vC.cpp:545-552) plus what either call kind needs of some allele x
// (call_vC.cpp:588,593-600): n_x >= calling_threshold and (n_x >= cov * min_fraction  -- the population rule --  or
// the position's individual-candidate bit, set by the pileup kernels when one sample holds >= calling_threshold reads of
// a mismatching allele).  Only the case-sensitive skip-same-base rule is left to msnv_decide_sites, so the candidates
// are essentially the called positions (10 773 -> 6 2xx on the benchmark shape; 51 903 -> 6 2xx with 320 samples, where
// error alleles alone reach the threshold at many positions).  One workgroup per tile; survivors
// are written in position order to a contiguous range reserved with one atomic per tile.
// ------------------------------------------------------------------------------------------
constexpr int GATE_NT = 256;
constexpr int GATE_PPT = TILE / GATE_NT;       // 8 consecutive positions per thread: one 8 / 16 / 32-byte load per row and thread
static_assert(GATE_PPT == 8, "the gate kernel is written for 8 positions per thread (one site_bits word per 8 lanes)");

// Per-sample records of the called positions are stored per TILE SLOT, not per sample: a tile's sites own rows of
// tile_nslots[tile] cells -- one per sample that has reads in the tile (pack.cpp numbers them; the pairs of a split sample share
// one) -- so a sparse cohort (BASELINE configs[3]: 500 samples, a species carried by a handful) does not pay 500 cells per
// site.  The host expands to all samples when it fetches.  Cell of (site, slot): tile_cell_base[tile] + (site - first site of
// the tile) * slots + slot; the gate kernel reserves a tile's cells with one 64-bit atomic (counters[6..7]).
struct CellMap { const uint32_t *tile_site_base; const unsigned long long *tile_cell_base; const uint32_t *tile_nslots; unsigned long long cap_cells;
                 const unsigned long long *block_row; };     // per 64 positions: first cell of the first site in them (gate kernel)
__device__ __forceinline__ uint64_t cell_of(const CellMap &m, const uint32_t tile, const uint32_t site, const uint32_t slot) {
    return m.tile_cell_base[tile] + (uint64_t)(site - m.tile_site_base[tile]) * m.tile_nslots[tile] + slot;
}

// Zeroes n bytes at p (2-byte aligned, n even) with the NT threads of the caller: halfword stores up to the first 16-byte boundary
// and behind the last, 16-byte stores in between (two-byte stores cost a wavefront ~540 cycles each: DESIGN.md "tail").
template <int NT>
__device__ __forceinline__ void zero_span(uint8_t *p, const uint64_t n, const uint32_t tid) {
    const uint64_t head = min(n, (uint64_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u));
    if ((uint64_t)2u * tid < head) *reinterpret_cast<uint16_t *>(p + 2u * tid) = 0;                     // (head <= 14 bytes)
    uint8_t *q = p + head;
    const uint64_t body = (n - head) >> 4;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (uint64_t i = tid; i < body; i += NT) reinterpret_cast<uint4 *>(q)[i] = z;
    const uint64_t tail0 = head + (body << 4);
    if (tail0 + 2u * tid < n) *reinterpret_cast<uint16_t *>(p + tail0 + 2u * tid) = 0;                   // (tail <= 14 bytes)
}

// The per-sample cells of newly reserved sites start out zero (gather and scatter only add to them): n cells from `first` on in the five
// u16 columns -- coverage and the four allele counts (structure of arrays: a site's row of cells is contiguous in every column)
template <int NT>
__device__ __forceinline__ void zero_cells(uint16_t *ncol, uint16_t *cov_col, const unsigned long long cap_cells, const unsigned long long first, const unsigned long long n, const uint32_t tid) {
    if (n <= (unsigned long long)NT) {                              // a handful of cells (a sparse cohort's tile): one store per column and lane, no alignment bookkeeping
        if (tid < (uint32_t)n) {
#pragma unroll
            for (int x = 0; x < 4; ++x) ncol[(uint64_t)x * cap_cells + first + tid] = 0;
            cov_col[first + tid] = 0;
        }
        return;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) zero_span<NT>(reinterpret_cast<uint8_t *>(ncol + (uint64_t)x * cap_cells + first), n * sizeof(uint16_t), tid);
    zero_span<NT>(reinterpret_cast<uint8_t *>(cov_col + first), n * sizeof(uint16_t), tid);      // samples without reads at a position keep coverage 0
}

struct GateTile { uint32_t tile, slot_lo, slot_16, slot_w, slot_hi, vbeg, vend, n_slots; uint64_t row0; uint32_t tot_mode, staged, pair_lo, n_plane_pairs, pad0, pad1; };   // 64 B (pack.cpp); staged: a whole-tile work item leaves the tile's candidates in a record list; pair_lo / n_plane_pairs: the tile's pairs that write allele planes

struct GateArgs {
    uint32_t *tot; const uint8_t *part; const uint64_t *slot_off; const uint32_t *tile_slot_start, *tile_slot_u16, *tile_slot_wide; uint64_t npos;
    const uint32_t *tile_vbeg, *tile_vend; int min_cov, min_snvs; double min_frac;
    uint32_t *ind4, *unc_bits;                 // read and left zero for the next pass
    const uint32_t *ref4, *ref_lc;
    unsigned long long *site_bits; uint32_t *site_rank; SiteRec *sites; uint32_t cap_sites; uint32_t *counters, *counters_next;
    uint32_t *tile_site_base, *tile_site_cnt; const uint32_t *active_tiles;
    uint16_t *ncol; uint16_t *cov_col; uint8_t *site_flags; uint32_t cap_out;      // ncol: four columns of cap_cells u16 (A, C, G, T counts of every cell), like cov_col
    const uint32_t *tile_nslots; unsigned long long *tile_cell_base; unsigned long long cap_cells;
    const GateTile *gate_tiles; uint32_t *tile_dirty; uint32_t *unc_sites; uint32_t use_dirty; unsigned long long *block_row;
    uint8_t *site_elig; uint32_t any_split;
    uint32_t n_active, tiles_per_wg;           // active tiles; consecutive ones per workgroup (<= GATE_MAX_TILES)
    const TileStage *tile_stage;               // record lists of the whole-tile work items (msnv_gate_staged)
    const uint8_t *aspill;                     // allele planes of the pairs (noisy reads: msnv_gate_sites<.., .., true> sums them), else NULL
    uint32_t zero_next;                        // this launch zeroes the counter block of the next pass (one of the two gate kernels does)
    // the launch behind msnv_gate_staged: tiles whose whole-tile work item found more candidates than its record list holds (fused_tile_gate);
    // tile_list[i] = index into gate_tiles, counters[CNT_STAGE] of them; the workgroups stride over the list
    const uint32_t *tile_list;
    uint32_t solo_cells;                       // the merged gather skips tiles with ONE pair (their cell is the tile's totals): write it here
};

// Site slots and per-sample cells are handed out with returning atomics on two device-wide counters, and same-address atomics are
// served one after the other (~6 ns each, measured: doubling them doubled the kernel).  With one tile per workgroup that is all a
// sparse cohort's launch waits for (BASELINE configs[3] shard: 80 k active tiles of a pair or two, 1.05 ms against 0.6 ms of
// pileup), so a workgroup takes tiles_per_wg consecutive active tiles, stages their few sites in LDS and reserves once for all of
// them; a tile with more sites than the stage holds is written straight from the registers as before.
constexpr uint32_t GATE_STAGE = 384;           // staged sites per reservation
constexpr uint32_t GATE_MAX_TILES = 8;         // tiles per workgroup (GATE_MAX_TILES x 32 blocks of 64 positions <= GATE_NT)
struct GateStageTile { uint32_t tile, site_rel, total, n_slots; unsigned long long cell_rel; };
struct GateLds {
    SiteRec  rec[GATE_STAGE];
    uint16_t fl[GATE_STAGE];                   // site_flags | site_elig << 8
    uint8_t  unc[GATE_STAGE];
    uint32_t blk_rel[GATE_MAX_TILES][TILE / 64];
    GateStageTile tiles[GATE_MAX_TILES];
    uint32_t wave[GATE_NT / 64];
    uint32_t pop, ind, base;
    unsigned long long cell;
};

// reserve for the staged tiles and write everything that needed the bases; called by all threads, staged data complete (barrier inside)
__device__ __forceinline__ void gate_flush(GateLds &L, const GateArgs &a, const uint32_t n_tiles, const uint32_t n_sites, const unsigned long long n_cells, const int tid) {
    if (n_tiles == 0u) return;                                   // (uniform)
    if (tid == 0) {
        L.base = atomicAdd(&a.counters[2], n_sites);
        L.cell = atomicAdd(reinterpret_cast<unsigned long long *>(&a.counters[CNT_CELLS]), (n_cells + 7ull) & ~7ull);   // (the counter stays a multiple of 8: gather_cov_wide)
    }
    __syncthreads();
    const uint32_t base = L.base; const unsigned long long cb = L.cell;
    if ((uint32_t)tid < n_tiles) {
        const GateStageTile t = L.tiles[tid];
        a.tile_site_base[t.tile] = base + t.site_rel; a.tile_site_cnt[t.tile] = t.total; a.tile_cell_base[t.tile] = cb + t.cell_rel;
    }
    if ((unsigned long long)base + n_sites <= a.cap_out && cb + n_cells <= a.cap_cells) {   // else: the host sees the counts and runs again with larger buffers
        // the per-sample cells of these sites start out zero: gather and scatter (one launch, side by side) only add to them
        zero_cells<GATE_NT>(a.ncol, a.cov_col, a.cap_cells, cb, n_cells, (uint32_t)tid);
    }
    // index of the first site of every 64 positions: an event finds its site as rank + popcount of the lower bits
    if ((uint32_t)tid < n_tiles * (TILE / 64)) {
        const uint32_t k = (uint32_t)tid / (TILE / 64), blk = (uint32_t)tid % (TILE / 64);
        const GateStageTile t = L.tiles[k];
        const uint32_t rel = L.blk_rel[k][blk];
        a.site_rank[(uint64_t)t.tile * (TILE / 64) + blk] = base + rel;
        a.block_row[(uint64_t)t.tile * (TILE / 64) + blk] = cb + t.cell_rel + (unsigned long long)(rel - t.site_rel) * t.n_slots;   // first cell of the block's first site
    }
    for (uint32_t i = (uint32_t)tid; i < n_sites; i += GATE_NT) {
        const uint32_t idx = base + i;
        if (idx >= a.cap_sites) continue;
        a.sites[idx] = L.rec[i];
        if (idx < a.cap_out) { a.site_flags[idx] = (uint8_t)(L.fl[i] & 0xffu); a.site_elig[idx] = (uint8_t)(L.fl[i] >> 8); }
        if (L.unc[i]) {                                             // decided from the per-sample records behind the scatter
            const uint32_t u = atomicAdd(&a.counters[CNT_UNC], 1u);
            if (u < a.cap_sites) a.unc_sites[u] = idx;
        }
    }
    __syncthreads();                                                // the stage is free again
}

// MULTI = false: one tile per workgroup, the loop below runs once and the compiler sees it (80 registers, 6 workgroups per CU: the
// benchmark shape); true: the loop is a loop (loop-invariant addresses and constants pile up: 126 registers, 4 workgroups per CU).
template <bool MULTI, bool WIDE_TOT, bool PLANES = false>
__global__ __launch_bounds__(GATE_NT) void msnv_gate_sites(const GateArgs a) {
    constexpr int NN = WIDE_TOT ? 4 * GATE_PPT : 2 * GATE_PPT;
    constexpr int ROWS = MULTI ? 8 : 16;                        // partial rows in flight per thread (a sparse cohort's tile has a row or two)
    uint32_t *const tot = a.tot; const uint8_t *const part = a.part;
    const int min_cov = a.min_cov, min_snvs = a.min_snvs; const double min_frac = a.min_frac;
    unsigned long long *const site_bits = a.site_bits; SiteRec *const sites = a.sites;
    const uint32_t cap_sites = a.cap_sites; uint32_t *const counters = a.counters;
    uint16_t *const cov_col = a.cov_col;
    const uint32_t cap_out = a.cap_out;
    __shared__ GateLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p0 = (uint32_t)GATE_PPT * (uint32_t)tid;    // my positions of every tile: p0 .. p0 + 7
    if (tid == 0) { L.pop = 0; L.ind = 0; }
    // the counter block of the NEXT pass (the passes of a dataset alternate between two blocks: no memset between passes)
    if (blockIdx.x == 0 && a.zero_next) for (uint32_t i = threadIdx.x; i < CNT_WORDS; i += GATE_NT) a.counters_next[i] = 0u;
    const bool listed = MULTI && a.tile_list != nullptr;       // (uniform) the tiles of a device-side list, workgroups stride over it
    const uint32_t ti_step = listed ? gridDim.x : 1u;
    const uint32_t ti_lo = listed ? blockIdx.x : MULTI ? blockIdx.x * a.tiles_per_wg : blockIdx.x;
    const uint32_t ti_hi = listed ? min(a.counters[CNT_STAGE], a.n_active) : MULTI ? min(a.n_active, ti_lo + a.tiles_per_wg) : ti_lo + 1u;
    if (MULTI && ti_lo >= ti_hi) return;                        // (uniform; before any barrier)
    auto tile_at = [&](const uint32_t ti) -> GateTile { return a.gate_tiles[listed ? a.tile_list[ti] : ti]; };
    uint32_t st_tiles = 0, st_sites = 0; unsigned long long st_cells = 0;      // staged so far (uniform)
    // everything the workgroup needs to know about a tile in ONE load (the kernel is a chain of dependent loads; with a sparse
    // cohort -- BASELINE configs[3]: a pair or two per tile -- the chain is all there is); the next tile's is fetched a tile ahead
    GateTile gt_next = tile_at(ti_lo);                          // tiles that hold work items; the others have no coverage
    for (uint32_t ti = ti_lo; ti < ti_hi; ti += ti_step) {
        const GateTile gt = gt_next;
        if (ti + ti_step < ti_hi) gt_next = tile_at(ti + ti_step);
        const uint32_t tile = gt.tile;
        const uint32_t t0 = tile * TILE;
        const uint32_t vb = gt.vbeg, ve = gt.vend;
        const uint64_t g0 = (uint64_t)t0 + p0;

        const uint32_t slot_lo = gt.slot_lo, slot_hi = gt.slot_hi;
        const uint32_t slot_16 = gt.slot_16, slot_w = gt.slot_w;
        // a listed tile with ONE (sample, tile) pair: the merged gather is not launched for it (msnv_gate_staged writes such a tile's cells), so the
        // individual rule is decided here -- the pair holds every read of the tile -- and the cells are written here
        const bool solo = MULTI && a.solo_cells && gt.staged == 2u;
        // 64-position blocks of the tile in which some pass added to the allele totals (narrow_pass / wide kernel): the other blocks'
        // totals and individual-rule bits are zero and are not even read (16.6 B per position against ~8 B of reads at 5x)
        // (only consulted for sparse cohorts, use_dirty: with many work items per tile every block is dirty anyway and the totals'
        // loads would wait for the words for nothing)
        // (nor for a tile whose totals are one word per position -- tot_add mode 0, what a sparse cohort's tiles are: 4 B per
        // position are not worth a round trip, and the pileup kernels do not mark such tiles: store_item_dirty)
        const bool consult = a.use_dirty && gt.tot_mode != 0u;
        uint32_t dirty = 0;
        if (consult) for (uint32_t sl = slot_lo; sl < slot_hi; ++sl) dirty |= a.tile_dirty[sl];      // (uniform addresses, independent loads)
        const bool my_dirty = !consult || ((dirty >> (p0 >> 6)) & 1u);
        // Everything else a thread reads about its 8 positions is requested here, ahead of the partial rows: the tile costs ONE round
        // trip to memory (the kernel is latency-bound: its time goes with 1 / resident workgroups).
        uint32_t *const tb = tot + 4ull * t0;
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        uint4 e0 = z4, e1 = z4;                                    // mode 0: the allele totals of my positions
        if (gt.tot_mode == 0u) { e0 = reinterpret_cast<const uint4 *>(tb + p0)[0]; e1 = reinterpret_cast<const uint4 *>(tb + p0)[1]; }
        // individual-rule bits of my 8 positions (4 per position) and the "split sample" marks; consumed here and here only, so they
        // are left zero for the next pass like the allele totals
        const uint32_t ind4w = my_dirty ? a.ind4[g0 >> 3] : 0u;
        const uint32_t uncb = my_dirty ? reinterpret_cast<const uint8_t *>(a.unc_bits)[g0 >> 3] : 0u;
        const uint32_t refw = a.ref4[g0 >> 3];                                  // nt16 codes of my 8 positions
        const uint32_t lcb = reinterpret_cast<const uint8_t *>(a.ref_lc)[g0 >> 3];   // FASTA character is a lower-case a / c / g / t
        uint32_t covs[GATE_PPT];
    #pragma unroll
        for (int j = 0; j < GATE_PPT; ++j) covs[j] = 0;
        // coverage = sum of the tile's work-item partials.  The rows of a tile are contiguous (u8 rows, then u16 rows, then the
        // u32 rows of wide items), so the addresses need no per-row lookup and the loads of several rows are in flight together.
        if (slot_16 > slot_lo) {
            const uint8_t *p8 = part + gt.row0 + p0;
            const uint32_t n8 = slot_16 - slot_lo;
            // u8 rows are summed two positions per register (u16 halves: positions (0,2) (1,3) (4,6) (5,7)); widened every 255 rows
            for (uint32_t r0 = 0; r0 < n8; r0 += 255u) {
                uint32_t h[4] = {0u, 0u, 0u, 0u};
                const uint32_t r1 = min(n8, r0 + 255u);
                // the kernel is latency-bound (under two workgroups per CU): 16 row loads in flight per thread
                uint32_t s = r0;
                for (; s + (uint32_t)ROWS <= r1; s += (uint32_t)ROWS) {
                    uint2 v[ROWS];
    #pragma unroll
                    for (int u = 0; u < ROWS; ++u) v[u] = *reinterpret_cast<const uint2 *>(p8 + (uint64_t)(s + (uint32_t)u) * TILE);
    #pragma unroll
                    for (int u = 0; u < ROWS; ++u) {
                        h[0] += v[u].x & 0x00ff00ffu; h[1] += (v[u].x >> 8) & 0x00ff00ffu;
                        h[2] += v[u].y & 0x00ff00ffu; h[3] += (v[u].y >> 8) & 0x00ff00ffu;
                    }
                }
    #pragma unroll 4
                for (; s < r1; ++s) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(p8 + (uint64_t)s * TILE);
                    h[0] += v.x & 0x00ff00ffu; h[1] += (v.x >> 8) & 0x00ff00ffu;
                    h[2] += v.y & 0x00ff00ffu; h[3] += (v.y >> 8) & 0x00ff00ffu;
                }
                covs[0] += h[0] & 0xffffu; covs[2] += h[0] >> 16; covs[1] += h[1] & 0xffffu; covs[3] += h[1] >> 16;
                covs[4] += h[2] & 0xffffu; covs[6] += h[2] >> 16; covs[5] += h[3] & 0xffffu; covs[7] += h[3] >> 16;
            }
        }
        if (slot_w > slot_16) {
            const uint8_t *p16 = part + gt.row0 + (uint64_t)(slot_16 - slot_lo) * TILE + 2u * p0;
            const uint32_t n16 = slot_w - slot_16;
    #pragma unroll 2
            for (uint32_t s = 0; s < n16; ++s) {
                const uint4 v = *reinterpret_cast<const uint4 *>(p16 + (uint64_t)s * 2u * TILE);
                covs[0] += v.x & 0xffffu; covs[1] += v.x >> 16; covs[2] += v.y & 0xffffu; covs[3] += v.y >> 16;
                covs[4] += v.z & 0xffffu; covs[5] += v.z >> 16; covs[6] += v.w & 0xffffu; covs[7] += v.w >> 16;
            }
        }
        if (slot_hi > slot_w) {
            const uint8_t *p32 = part + gt.row0 + (uint64_t)(slot_16 - slot_lo) * TILE + (uint64_t)(slot_w - slot_16) * 2u * TILE + 4u * p0;
            const uint32_t n32 = slot_hi - slot_w;
            for (uint32_t s = 0; s < n32; ++s) {
                const uint4 v0 = *reinterpret_cast<const uint4 *>(p32 + (uint64_t)s * 4u * TILE);
                const uint4 v1 = *reinterpret_cast<const uint4 *>(p32 + (uint64_t)s * 4u * TILE + 16);
                covs[0] += v0.x; covs[1] += v0.y; covs[2] += v0.z; covs[3] += v0.w;
                covs[4] += v1.x; covs[5] += v1.y; covs[6] += v1.z; covs[7] += v1.w;
            }
        }
        // allele totals of my positions; consumed here and here only, so they are left zero for the next pass (no 16 B/position
        // memset per pass, which is what a large sparse reference would mostly pay for).  The tile's mode -- tot_add -- says how wide
        // they are in memory: 4, 8 or 16 bytes per position; in registers two u16 per word (A | C << 16, G | T << 16) unless some
        // tile of the dataset needs all 32 bits (WIDE_TOT: 32 registers instead of 16).
        uint32_t nn[NN];
    #pragma unroll
        for (int k = 0; k < NN; ++k) nn[k] = 0u;
        if (my_dirty) {                                            // (allele planes: only merged groups add to the totals)
            if (gt.tot_mode == 0u) {
                uint4 *tp = reinterpret_cast<uint4 *>(tb + p0);
                const uint32_t wd[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    #pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (WIDE_TOT) { nn[4 * k] = wd[k] & 0xffu; nn[4 * k + 1] = (wd[k] >> 8) & 0xffu; nn[4 * k + 2] = (wd[k] >> 16) & 0xffu; nn[4 * k + 3] = wd[k] >> 24; }
                    else { nn[2 * k] = (wd[k] & 0xffu) | (wd[k] & 0xff00u) << 8; nn[2 * k + 1] = ((wd[k] >> 16) & 0xffu) | (wd[k] >> 24) << 16; }
                }
                if (e0.x | e0.y | e0.z | e0.w) tp[0] = z4;
                if (e1.x | e1.y | e1.z | e1.w) tp[1] = z4;
            } else if (gt.tot_mode == 1u) {
                uint4 *tp = reinterpret_cast<uint4 *>(tb + 2u * p0);
                const uint4 v[4] = {tp[0], tp[1], tp[2], tp[3]};
    #pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const uint32_t wd[4] = {v[h].x, v[h].y, v[h].z, v[h].w};
    #pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (WIDE_TOT) { nn[8 * h + 2 * k] = wd[k] & 0xffffu; nn[8 * h + 2 * k + 1] = wd[k] >> 16; }
                        else nn[4 * h + k] = wd[k];
                    }
                    if (v[h].x | v[h].y | v[h].z | v[h].w) tp[h] = z4;
                }
            } else if (WIDE_TOT) {
                uint4 *tp = reinterpret_cast<uint4 *>(tb + 4u * p0);
    #pragma unroll
                for (int k = 0; k < GATE_PPT; ++k) {
                    const uint4 v = tp[k];
                    nn[4 * k] = v.x; nn[4 * k + 1] = v.y; nn[4 * k + 2] = v.z; nn[4 * k + 3] = v.w;
                    if (v.x | v.y | v.z | v.w) tp[k] = z4;
                }
            }
        }
        if constexpr (PLANES) {
            // noisy reads: the pairs of the ordinary work items wrote their allele counts as byte planes (narrow_pass<.., DA>): summed like the
            // u8 coverage rows, two positions per register, widened every 255 rows; merged groups still came through the totals above
            const uint32_t npl = gt.n_plane_pairs;
            if (npl) {
                // even bytes (A, G) and odd bytes (C, T) of the words as u16 pairs: ae[j] = A | G << 16, ao[j] = C | T << 16 of my position j
                const uint32_t *p32 = reinterpret_cast<const uint32_t *>(a.aspill) + (uint64_t)gt.pair_lo * TILE + p0;
                for (uint32_t r0 = 0; r0 < npl; r0 += 256u) {                 // (a u16 half holds 256 rows of bytes)
                    uint32_t ae[GATE_PPT], ao[GATE_PPT];
#pragma unroll
                    for (int j = 0; j < GATE_PPT; ++j) { ae[j] = 0u; ao[j] = 0u; }
                    const uint32_t r1 = min(npl, r0 + 256u);
                    uint32_t sidx = r0;
                    for (; sidx + 4u <= r1; sidx += 4u) {
                        uint4 v[4][2];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint4 *q = reinterpret_cast<const uint4 *>(p32 + (uint64_t)(sidx + (uint32_t)u) * TILE);
                            v[u][0] = q[0]; v[u][1] = q[1];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t w8[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
#pragma unroll
                            for (int j = 0; j < GATE_PPT; ++j) { ae[j] += w8[j] & 0x00ff00ffu; ao[j] += (w8[j] >> 8) & 0x00ff00ffu; }
                        }
                    }
                    for (; sidx < r1; ++sidx) {
                        const uint4 *q = reinterpret_cast<const uint4 *>(p32 + (uint64_t)sidx * TILE);
                        const uint4 v0 = q[0], v1 = q[1];
                        const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                        for (int j = 0; j < GATE_PPT; ++j) { ae[j] += w8[j] & 0x00ff00ffu; ao[j] += (w8[j] >> 8) & 0x00ff00ffu; }
                    }
#pragma unroll
                    for (int j = 0; j < GATE_PPT; ++j) {
                        if (WIDE_TOT) { nn[4 * j] += ae[j] & 0xffffu; nn[4 * j + 1] += ao[j] & 0xffffu; nn[4 * j + 2] += ae[j] >> 16; nn[4 * j + 3] += ao[j] >> 16; }
                        else { nn[2 * j] += (ae[j] & 0xffffu) | ao[j] << 16; nn[2 * j + 1] += (ae[j] >> 16) | (ao[j] & 0xffff0000u); }
                    }
                }
            }
        }
        // mismatching A / C / G / T reads at my position j
        auto n_of = [&](const int j, const int x) -> uint32_t { return WIDE_TOT ? nn[4 * j + x] : (nn[2 * j + (x >> 1)] >> (16 * (x & 1))) & 0xffffu; };
        if (ind4w) a.ind4[g0 >> 3] = 0u;
        if (uncb) reinterpret_cast<uint8_t *>(a.unc_bits)[g0 >> 3] = 0;
        uint32_t okm = 0, uncm = 0, flw[2] = {0u, 0u}, elig[2] = {0u, 0u};      // site mask; "ask the per-sample records" mask; pop | ind << 4 and the alleles
                                                                                // still open to the individual rule, one byte per position
    #pragma unroll
        for (int j = 0; j < GATE_PPT; ++j) {
            const uint32_t p = p0 + (uint32_t)j, cov = covs[j];
            // mismatching bases are counted bases: no coverage, no allele totals (a stale total can not exist: they are zeroed above)
            if (cov != 0u && p >= vb && p < ve && (int)cov >= min_cov &&
                (int)(n_of(j, 0) + n_of(j, 1) + n_of(j, 2) + n_of(j, 3)) >= min_snvs) {                          // call_vC.cpp:547,550
                const uint32_t indx = (ind4w >> (4 * j)) & 15u;
                const bool unc = (uncb >> j) & 1u;
                const double lim = (double)(int)cov * min_frac;                // call_vC.cpp:588
                const uint32_t rc = (refw >> (4 * j)) & 15u;
                const bool lc = (lcb >> j) & 1u;
                bool ok = false;
                uint32_t pop = 0, ind = 0;
    #pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if ((int)n_of(j, x) < min_snvs) continue;                   // neither rule can fire (call_vC.cpp:588,593-600)
                    const bool is_pop = (double)n_of(j, x) >= lim, is_ind = ((indx >> x) & 1u) || (solo && unc);
                    ok |= is_pop || is_ind || unc;
                    if (lc && rc == (1u << x)) continue;                       // skip-same-base, case-sensitive (call_vC.cpp:580)
                    if (is_pop) pop |= 1u << x; else if (is_ind) ind |= 1u << x;
                    else if (unc) { uncm |= 1u << j; elig[j >> 2] |= (1u << x) << (8 * (j & 3)); }   // a split / merged sample may hold >= t reads in sum
                }
                okm |= (ok ? 1u : 0u) << j;
                flw[j >> 2] |= (pop | ind << 4) << (8 * (j & 3));
            }
        }
        // 1 bit per position: is a site (the scatter half of msnv_gather_scatter filters on it).  One 64-bit word = 8 lanes.
        {
            unsigned long long w = (unsigned long long)okm << (8u * ((uint32_t)lane & 7u));
            w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
            if ((lane & 7) == 0) site_bits[g0 >> 6] = w;
        }
        const uint32_t mycnt = (uint32_t)__popc(okm);
        const uint32_t incl = (uint32_t)wave_inclusive_scan((int)mycnt);
        if (lane == 63) L.wave[wave] = incl;
        __syncthreads();
        uint32_t total = 0, mybase = incl - mycnt;                 // exclusive prefix in position order
#pragma unroll
        for (int wv = 0; wv < GATE_NT / 64; ++wv) {
            if (wv < wave) mybase += L.wave[wv];
            total += L.wave[wv];
        }
        const uint32_t n_slots = gt.n_slots;
        if (tid == 0 && dirty) for (uint32_t sl = slot_lo; sl < slot_hi; ++sl) a.tile_dirty[sl] = 0u;     // (every thread read them before the barrier above)
        // output-line tallies (before the first-line drop); the sites of the "ask" mask are tallied by msnv_decide_sites
        // (a split sample: msnv_decide_sites re-decides and tallies those sites; merged groups only: the merged gather adds the
        // individual calls it finds, so everything decided here is tallied here)
        const uint32_t okm_certain = a.any_split ? okm & ~uncm : okm;
        if (okm_certain) {
            uint32_t np = 0, ni = 0;
#pragma unroll
            for (int j = 0; j < GATE_PPT; ++j) { const uint32_t f = ((okm_certain >> j) & 1u) ? (flw[j >> 2] >> (8 * (j & 3))) & 0xffu : 0u; np += (f & 15u) ? 1u : 0u; ni += (f >> 4) ? 1u : 0u; }
            if (np) atomicAdd(&L.pop, np);
            if (ni) atomicAdd(&L.ind, ni);
        }
        if (total == 0u) {                                          // (uniform)
            if (tid == 0) { a.tile_site_base[tile] = 0u; a.tile_site_cnt[tile] = 0u; a.tile_cell_base[tile] = 0ull; }
            __syncthreads();                                        // L.wave is written again by the next tile
            continue;
        }
        if (st_sites + total > GATE_STAGE || st_tiles == GATE_MAX_TILES) {     // (uniform) no room: hand out what is staged
            gate_flush(L, a, st_tiles, st_sites, st_cells, tid);
            st_tiles = 0; st_sites = 0; st_cells = 0;
        }
        if (total <= GATE_STAGE && !solo) {
            if ((n_slots & 7u) == 0u) st_cells = (st_cells + 7ull) & ~7ull;      // a tile whose rows are multiples of 16 bytes starts on 16 bytes (gather_cov_wide)
            if (tid == 0) L.tiles[st_tiles] = GateStageTile{tile, st_sites, total, n_slots, st_cells};
            if ((lane & 7) == 0) L.blk_rel[st_tiles][p0 >> 6] = st_sites + mybase;
            uint32_t i = st_sites + mybase;
#pragma unroll
            for (int j = 0; j < GATE_PPT; ++j) {
                if (okm & (1u << j)) {
                    SiteRec s;
                    s.gpos = (uint32_t)(g0 + (uint32_t)j); s.cov = covs[j];
                    s.n[0] = n_of(j, 0); s.n[1] = n_of(j, 1); s.n[2] = n_of(j, 2); s.n[3] = n_of(j, 3);
                    L.rec[i] = s;
                    L.fl[i] = (uint16_t)(((flw[j >> 2] >> (8 * (j & 3))) & 0xffu) | ((elig[j >> 2] >> (8 * (j & 3))) & 0xffu) << 8);
                    L.unc[i] = (uint8_t)((a.any_split && ((uncm >> j) & 1u)) ? 1u : 0u);
                    ++i;
                }
            }
            ++st_tiles; st_sites += total; st_cells += (unsigned long long)total * n_slots;
            __syncthreads();                                        // L.wave is written again by the next tile
            continue;
        }
        // ---- a tile with more sites than the stage holds: its own reservation, written from the registers
        if (tid == 0) {
            const uint32_t base = atomicAdd(&counters[2], total);
            L.base = base;
            a.tile_site_base[tile] = base;
            a.tile_site_cnt[tile] = total;
            const unsigned long long cb = atomicAdd(reinterpret_cast<unsigned long long *>(&counters[CNT_CELLS]), ((unsigned long long)total * n_slots + 7ull) & ~7ull);
            L.cell = cb;
            a.tile_cell_base[tile] = cb;
        }
        __syncthreads();
        const uint32_t base = L.base; const unsigned long long s_cell = L.cell;
        const bool fits = (uint64_t)base + total <= cap_out && s_cell + (unsigned long long)total * n_slots <= a.cap_cells;
        if (fits && !solo) {
            const uint64_t n_cells = (uint64_t)total * n_slots;
            zero_cells<GATE_NT>(a.ncol, cov_col, a.cap_cells, s_cell, n_cells, (uint32_t)tid);
        }
        if ((lane & 7) == 0) {
            a.site_rank[g0 >> 6] = base + mybase;
            a.block_row[g0 >> 6] = s_cell + (unsigned long long)mybase * n_slots;
        }
        uint32_t idx = base + mybase;
#pragma unroll
        for (int j = 0; j < GATE_PPT; ++j) {
            if (okm & (1u << j)) {
                if (idx < cap_sites) {
                    SiteRec s;
                    s.gpos = (uint32_t)(g0 + (uint32_t)j); s.cov = covs[j];
                    s.n[0] = n_of(j, 0); s.n[1] = n_of(j, 1); s.n[2] = n_of(j, 2); s.n[3] = n_of(j, 3);
                    sites[idx] = s;
                    if (idx < cap_out) a.site_flags[idx] = (uint8_t)((flw[j >> 2] >> (8 * (j & 3))) & 0xffu);
                    if (idx < cap_out) a.site_elig[idx] = (uint8_t)((elig[j >> 2] >> (8 * (j & 3))) & 0xffu);
                    if (a.any_split && ((uncm >> j) & 1u)) {
                        const uint32_t u = atomicAdd(&counters[CNT_UNC], 1u);
                        if (u < cap_sites) a.unc_sites[u] = idx;
                    }
                    if (solo && fits) {                                // one slot per site: the cell is the tile's totals
                        const unsigned long long cell = s_cell + (idx - base);
#pragma unroll
                        for (int x = 0; x < 4; ++x) a.ncol[(uint64_t)x * a.cap_cells + cell] = (uint16_t)s.n[x];
                        cov_col[cell] = (uint16_t)s.cov;
                    }
                }
                ++idx;
            }
        }
        __syncthreads();                                            // L.wave, L.base and L.cell are written again
    }
    gate_flush(L, a, st_tiles, st_sites, st_cells, tid);
    __syncthreads();
    if (tid == 0 && (L.pop | L.ind)) atomicAdd(reinterpret_cast<unsigned long long *>(&counters[CNT_TALLY]), (unsigned long long)L.pop | (unsigned long long)L.ind << 32);
}

// ------------------------------------------------------------------------------------------
// msnv_gate_staged: the gate kernel's part for tiles piled up by whole-tile work items (fused_tile_gate): their candidates arrive as
// records with the decision taken, so what is left is the bookkeeping msnv_gate_sites does behind its decision -- site slots and
// cells, site bitmap, per-block ranks, site records -- and none of it needs a workgroup: ONE WAVEFRONT takes GS_TILES consecutive
// tiles, reads their record counts (known before anything is computed: one reservation for all of them, no same-address atomic
// per tile), then walks the tiles: lane q holds record q, its rank among the tile's records is its site; lanes 0-31 build the
// 32 bitmap words and the per-block site ranks from the same loop over the records.  A tile with ONE pair gets its per-sample
// cell written right here (the cell is the tile's totals), so the merged gather is not launched for it.
// (the same tiles through msnv_gate_sites, reading records instead of per-position state: 0.38 ms for 85 k tiles -- three
// workgroup barriers and a 127-register kernel per handful of records)
// ------------------------------------------------------------------------------------------
constexpr int GS_WAVES = 4;                    // wavefronts per workgroup: ONE reservation of site slots and cells per 64 tiles (sixteen wavefronts, a reservation per 256 tiles: 70 -> 79 us on the configs[3] shard -- the barrier waits for the slowest of sixteen load chains; the 1 400 pairs of same-address returning atomics are 20 us of the 70)
constexpr int GS_TILES = 16;                   // tiles per wavefront; a workgroup of four wavefronts reserves ONCE for its 64 tiles
__global__ __launch_bounds__(64 * GS_WAVES) void msnv_gate_staged(const GateArgs a, const GateTile *__restrict__ tiles, const uint32_t n_tiles) {
    __shared__ uint32_t s_sites[GS_WAVES]; __shared__ unsigned long long s_cells[GS_WAVES];
    __shared__ uint32_t s_base, s_np, s_ni; __shared__ unsigned long long s_cb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w0 = (blockIdx.x * (uint32_t)GS_WAVES + (uint32_t)wave) * (uint32_t)GS_TILES;
    if (blockIdx.x == 0 && a.zero_next) for (uint32_t i = threadIdx.x; i < CNT_WORDS; i += 64 * GS_WAVES) a.counters_next[i] = 0u;
    if (threadIdx.x == 0) { s_np = 0; s_ni = 0; }
    const uint32_t n_here = w0 < n_tiles ? min((uint32_t)GS_TILES, n_tiles - w0) : 0u;
    // lanes 0 .. n_here - 1: one tile each
    GateTile gt{};
    uint32_t cnt = 0;
    bool mine = (uint32_t)lane < n_here;                              // a tile whose candidates did not fit its list is not mine: fused_tile_gate sent it to msnv_gate_sites
    if (mine) {
        gt = tiles[w0 + (uint32_t)lane]; cnt = a.tile_stage[gt.row0].count;
        if (cnt > STAGE_CAP) { mine = false; cnt = 0u; }
    }
    const unsigned long long cells = (unsigned long long)cnt * gt.n_slots;
    uint32_t site_rel = cnt; unsigned long long cell_rel = cells;
#pragma unroll
    for (int o = 1; o < GS_TILES; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)site_rel, o); const unsigned long long u = (unsigned long long)__shfl_up((long long)cell_rel, o);
        if (lane >= o) { site_rel += t; cell_rel += u; }
    }
    const uint32_t tot_sites = (uint32_t)__shfl((int)site_rel, GS_TILES - 1); const unsigned long long tot_cells = (unsigned long long)__shfl((long long)cell_rel, GS_TILES - 1);      // (lanes behind n_here hold zeros: the prefix stays)
    site_rel -= cnt; cell_rel -= cells;                               // exclusive
    // one reservation per workgroup: same-address returning atomics are served one after the other (~6 ns each: one per wavefront of
    // 8 tiles was 140 us of this kernel's 290)
    if (lane == 0) { s_sites[wave] = tot_sites; s_cells[wave] = tot_cells; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ts = 0; unsigned long long tc = 0;
        for (int k = 0; k < GS_WAVES; ++k) { ts += s_sites[k]; tc += s_cells[k]; }
        s_base = ts ? atomicAdd(&a.counters[2], ts) : 0u;
        s_cb = ts ? atomicAdd(reinterpret_cast<unsigned long long *>(&a.counters[CNT_CELLS]), (tc + 7ull) & ~7ull) : 0ull;
    }
    __syncthreads();
    uint32_t base = s_base; unsigned long long cb = s_cb;
    uint32_t wg_sites = 0; unsigned long long wg_cells = 0;
    for (int k = 0; k < GS_WAVES; ++k) { if (k < wave) { base += s_sites[k]; cb += s_cells[k]; } wg_sites += s_sites[k]; wg_cells += s_cells[k]; }
    if (mine) { a.tile_site_base[gt.tile] = cnt ? base + site_rel : 0u; a.tile_site_cnt[gt.tile] = cnt; a.tile_cell_base[gt.tile] = cnt ? cb + cell_rel : 0ull; }
    const bool fits = (unsigned long long)s_base + wg_sites <= a.cap_out && s_cb + wg_cells <= a.cap_cells;   // else: the host sees the counts and runs again with larger buffers
    uint32_t np = 0, ni = 0;
    const unsigned long long mine_mask = __ballot(mine);
    for (uint32_t t = 0; t < n_here; ++t) {
        if (!((mine_mask >> t) & 1ull)) continue;                       // (uniform)
        const uint32_t tile = (uint32_t)__shfl((int)gt.tile, (int)t), n = (uint32_t)__shfl((int)cnt, (int)t), n_slots = (uint32_t)__shfl((int)gt.n_slots, (int)t);
        const uint32_t kind = (uint32_t)__shfl((int)gt.staged, (int)t), srel = (uint32_t)__shfl((int)site_rel, (int)t);
        const unsigned long long crel = (unsigned long long)__shfl((long long)cell_rel, (int)t), sidx = (unsigned long long)__shfl((long long)gt.row0, (int)t);
        StageRec r{0xffffffffu, 0u, 0u, 0u};
        if ((uint32_t)lane < n) r = a.tile_stage[sidx].rec[lane];
        const uint32_t pos = r.pos_flags & (TILE - 1u);
        uint32_t rank = 0, before = 0; unsigned long long word = 0;      // my record's rank; (lanes 0-31) records in front of my 64 positions, my bitmap word
        for (uint32_t q = 0; q < n; ++q) {
            const uint32_t pq = (uint32_t)__shfl((int)pos, (int)q);
            rank += pq < pos ? 1u : 0u;
            before += pq < 64u * (uint32_t)lane ? 1u : 0u;
            if ((pq >> 6) == (uint32_t)lane) word |= 1ull << (pq & 63u);
        }
        if (lane < (int)(TILE / 64)) {
            a.site_bits[(uint64_t)tile * (TILE / 64) + lane] = word;
            if (n) {
                a.site_rank[(uint64_t)tile * (TILE / 64) + lane] = base + srel + before;
                a.block_row[(uint64_t)tile * (TILE / 64) + lane] = cb + crel + (unsigned long long)before * n_slots;
            }
        }
        if ((uint32_t)lane < n) {
            const uint32_t fl = (r.pos_flags >> 16) & 0xffu, el = r.pos_flags >> 24;
            const bool ask = (r.pos_flags >> 11) & 1u;
            if (!(a.any_split && ask)) { np += (fl & 15u) ? 1u : 0u; ni += (fl >> 4) ? 1u : 0u; }      // (those sites are tallied by msnv_decide_sites)
            const uint32_t idx = base + srel + rank;
            if (idx < a.cap_sites) {
                SiteRec sr;
                sr.gpos = tile * TILE + pos; sr.cov = r.cov;
                sr.n[0] = r.nword & 0xffu; sr.n[1] = (r.nword >> 8) & 0xffu; sr.n[2] = (r.nword >> 16) & 0xffu; sr.n[3] = r.nword >> 24;
                a.sites[idx] = sr;
                if (idx < a.cap_out) { a.site_flags[idx] = (uint8_t)fl; a.site_elig[idx] = (uint8_t)el; }
                if (a.any_split && ask) {
                    const uint32_t u = atomicAdd(&a.counters[CNT_UNC], 1u);
                    if (u < a.cap_sites) a.unc_sites[u] = idx;
                }
                if (kind == 2u && fits) {                              // the tile's only pair: its cell is the tile's totals
#pragma unroll
                    for (int x = 0; x < 4; ++x) a.ncol[(uint64_t)x * a.cap_cells + cb + crel + rank] = (uint16_t)sr.n[x];
                    a.cov_col[cb + crel + rank] = (uint16_t)r.cov;
                }
            }
        }
        if (kind != 2u && fits && n) {                                  // the per-sample cells of these sites start out zero (gather / scatter only add to them)
            const unsigned long long n_cells = (unsigned long long)n * n_slots;
            zero_cells<64>(a.ncol, a.cov_col, a.cap_cells, cb + crel, n_cells, (uint32_t)lane);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { np += (uint32_t)__shfl_xor((int)np, o); ni += (uint32_t)__shfl_xor((int)ni, o); }
    if (lane == 0 && (np | ni)) { atomicAdd(&s_np, np); atomicAdd(&s_ni, ni); }
    __syncthreads();
    if (threadIdx.x == 0 && (s_np | s_ni)) atomicAdd(reinterpret_cast<unsigned long long *>(&a.counters[CNT_TALLY]), (unsigned long long)s_np | (unsigned long long)s_ni << 32);
}

// ------------------------------------------------------------------------------------------
// First cell of allele column x (A, C, G, T): x * cap_cells without a 64-bit vector multiply (x differs from lane to lane, the
// capacity is uniform: two selects and an add)
__device__ __forceinline__ unsigned long long ncol_base(const uint32_t x, const unsigned long long cap_cells) {
    return ((x & 1u) ? cap_cells : 0ull) + ((x & 2u) ? cap_cells << 1 : 0ull);
}
// The u16 fields of the site records are summed with 32-bit atomics on the aligned word that holds them (sums stay below
// 65536: the depth cap is 8000), so that a sample which was split into several (sample, tile) pairs adds up.
__device__ __forceinline__ void add_u16(uint16_t *field, uint32_t v) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(field);
    atomicAdd(reinterpret_cast<uint32_t *>(addr & ~(uintptr_t)3), v << (8u * (uint32_t)(addr & 2u)));
}

// tail of a pass: per-sample coverages (gather half) and allele counts (scatter half) of the surviving sites.
// ------------------------------------------------------------------------------------------
constexpr uint32_t SCATTER_BLOCKS_PER_LIST = 16;   // default (gather || scatter launch on the benchmark shape: 32 -> 50 us, 16 -> 45, 8 -> 48, 4 -> 57; profiles/r03r_tail_tune.txt); TailArgs::scatter_blocks is what a launch uses (MSNV_SCATTER_BLOCKS: tuning experiments)
struct TailArgs {
    const SiteRec *sites; const uint32_t *tile_site_base, *tile_site_cnt, *tile_pair_start, *tile_pair_merged; const TilePair *pairs; const uint8_t *spill;
    // merged groups of shallow pairs: their per-sample coverage at the called positions is recomputed from the pieces
    const MergedGroupDev *merged_groups; const ChunkDesc *chunks; const PieceHdr *hdr8m; const uint8_t *seq, *qual; const uint32_t *ref4;
    uint32_t n_merged_blocks, min_baseq;
    uint32_t n_merged_run, merged_per_block;    // groups the merged gather runs over; groups per workgroup (gather_merged_block)
    // individual rule inside the merged gather (a sample's reads at a site all sit in ONE group): unless a split sample needs msnv_decide_sites anyway
    uint8_t *site_flags; const uint8_t *site_elig; uint32_t ind_in_gather, min_snvs;
    uint16_t *ncol; uint16_t *cov_col; uint32_t cap_out; const uint32_t *active_tiles; uint32_t n_gather_blocks;
    CellMap cells; uint32_t gather_split;
    uint32_t scatter_blocks;                    // workgroups per event sub-list
    uint32_t debug_skip;                        // MSNV_TAIL_SKIP (profiling only, results are wrong): 1 = no scatter blocks, 2 = no gather blocks, 4 = no merged blocks
    const uint8_t *aspill;                      // allele rows of the pairs (noisy reads: a word per position), else NULL: gathered into the four allele columns like the coverage bytes
    uint32_t has_wide;                          // some work item runs the wide kernel (coverage bytes of 255 stand for an overflow-list entry the scatter half writes)
    const Pair32 *events, *overflow; uint32_t *counters; uint32_t cap_list, cap_overflow;
    const unsigned long long *site_bits; const uint32_t *site_rank;
};

constexpr uint32_t GD_MIN_SITES = 32;          // sites of a tile per gather workgroup from which the cells go through LDS (gather_cov_block)
constexpr uint32_t GD_ROW = 68;                // bytes per LDS row of 64 pairs (17 words: a wavefront's column writes hit 64 banks)
constexpr uint32_t GW_ROW = 34;                // words per LDS row of 64 slots (u16 each) + 2: 8-byte aligned rows, a wavefront's column updates meet two to a bank

// Dense gather with 16-BYTE STORES (round 3).  The two-byte cell stores of the form below cost a wavefront ~540 cycles each and were all
// the many-site gather waited for (stores off: 1.90 -> 0.38 ms on the sigma = 2 cohort).  Here a block of 64 sites x 64 SLOTS is
// assembled in LDS -- a wavefront reads one pair's bytes at the 64 sites and adds them to the pair's slot column, so the pairs of a
// split sample sum up right there, no global atomic -- and written out as rows of cells, eight cells (16 bytes) per lane.  That needs
// rows that start on 16 bytes: the tile's cell stride is a multiple of 8 (pack.cpp pads tiles of >= 16 slots) and so is its first
// cell (the gate kernels reserve in multiples of 8).  Slots of merged pairs (behind the others) belong to gather_merged_block: the
// group of eight that straddles their first slot is written cell by cell.
constexpr uint32_t GW_MAX_BLOCKS = 64;         // 64-slot blocks of a tile this form takes (<= 4096 slots; larger cohorts keep the two-byte form)
constexpr uint32_t GW_SITES = 64;              // sites per LDS block: 64 x 64 slots x u16 = 8.7 KB (32: the many-site gather 8 % slower, the benchmark shape's tail the same: profiles/r03l_phase_times.txt)
__device__ __forceinline__ void gather_cov_wide(const TailArgs &a, uint32_t *s_acc, uint32_t *s_off, uint32_t *s_blk, const uint32_t t0, const uint32_t base, const uint32_t stride,
                                                const unsigned long long cell0, const uint32_t ps, const uint32_t np, const bool tile_has_merged, const uint32_t j_lo, const uint32_t dense_n) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // slots are numbered in pair order (the pairs of a split sample share one): first pair of every block of 64 slots
    const uint32_t n_real = (a.pairs[ps + np - 1u].pad >> 8) + 1u, nblk = (n_real + 63u) >> 6;
    for (uint32_t kk = tid; kk < np; kk += 256u) {
        const uint32_t sl = a.pairs[ps + kk].pad >> 8, prev = kk ? a.pairs[ps + kk - 1u].pad >> 8 : 0xffffffffu;
        if (kk == 0u || (prev >> 6) != (sl >> 6)) s_blk[sl >> 6] = kk;
    }
    if (tid == 0u) s_blk[nblk] = np;
    // cells of the last group of eight that this path may write as a whole: everything when no merged pair follows, else up to the first merged slot
    const uint32_t wide_end = tile_has_merged ? (n_real & ~7u) : stride;
    for (uint32_t jj0 = 0; jj0 < dense_n; jj0 += GW_SITES) {
        const uint32_t nS = min(GW_SITES, dense_n - jj0);
        __syncthreads();                                            // (the block before has been written out; s_blk is visible)
        if (tid < GW_SITES) s_off[tid] = a.sites[base + j_lo + jj0 + min(tid, nS - 1u)].gpos - t0;   // (idle lanes: a valid position; their rows are not written out)
        for (uint32_t b = 0; b < nblk; ++b) {
            const uint8_t *src = a.spill + (uint64_t)ps * TILE; const uint64_t src_stride = TILE; uint16_t *col = a.cov_col;
            __syncthreads();                                        // s_off is visible; the previous 64 slots have been written out
            for (uint32_t i = tid; i < GW_SITES * GW_ROW; i += 256u) s_acc[i] = 0u;
            __syncthreads();
            const uint32_t kA = s_blk[b], kB = s_blk[b + 1u];
            const uint32_t site = lane;                              // (GW_SITES == 64: a lane per site, a wavefront per pair)
            static_assert(GW_SITES == 64, "the fill loop deals one site to every lane of a wavefront");
            const uint32_t off = s_off[site];
            for (uint32_t kl = kA + wave; kl < kB; kl += 16u) {      // four loads in flight per lane
                uint32_t v[4], c[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t k = ps + min(kl + 4u * u, kB - 1u);
                    v[u] = src[(uint64_t)(k - ps) * src_stride + off];
                    c[u] = (a.pairs[k].pad >> 8) - 64u * b;          // (one address for the wavefront)
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u)
                    if (kl + 4u * u < kB) atomicAdd(&s_acc[site * GW_ROW + (c[u] >> 1)], v[u] << (16u * (c[u] & 1u)));
            }
            __syncthreads();
            for (uint32_t idx = tid; idx < nS * 8u; idx += 256u) {
                const uint32_t row = idx >> 3, seg = idx & 7u, c0 = 64u * b + 8u * seg;
                if (c0 >= n_real) continue;                              // (slots of merged pairs, padding behind the last slot: zeroed by the gate kernel)
                const uint2 lo = *reinterpret_cast<const uint2 *>(&s_acc[row * GW_ROW + 4u * seg]);
                const uint2 hi = *reinterpret_cast<const uint2 *>(&s_acc[row * GW_ROW + 4u * seg + 2u]);
                uint16_t *dst = col + cell0 + (uint64_t)(j_lo + jj0 + row) * stride + c0;
                if (c0 + 8u <= wide_end) *reinterpret_cast<uint4 *>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);      // (columns behind the last slot are zero)
                else {
                    const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
                    for (uint32_t q = 0; q < 8u && c0 + q < n_real; ++q) dst[q] = (uint16_t)(w[q >> 1] >> (16u * (q & 1u)));
                }
            }
        }
    }
}

// The same for the ALLELE ROWS of noisy reads (pack.cpp: allele planes): a pair's word A | C << 8 | G << 16 | T << 24 at a site goes
// into four u16 column blocks in LDS -- [site][allele][32 slots], one LDS atomic per allele the word holds (mostly one) -- and every
// allele column is written as rows of 32 slots, eight cells (16 bytes) per lane.  One load per (site, pair) for all four alleles.
constexpr uint32_t GA_SLOTS = 32;              // slots per LDS block of the allele gather
constexpr uint32_t GA_ROW = 4 * GA_SLOTS / 2 + 2;   // words per site: 4 alleles x 32 u16 + 2 (8-byte aligned rows, column updates spread over the banks)
__device__ __forceinline__ void gather_alleles_wide(const TailArgs &a, uint32_t *s_acc, uint32_t *s_off, uint32_t *s_blk, const uint32_t t0, const uint32_t base, const uint32_t stride,
                                                    const unsigned long long cell0, const uint32_t ps, const uint32_t np, const bool tile_has_merged, const uint32_t j_lo, const uint32_t dense_n) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_real = (a.pairs[ps + np - 1u].pad >> 8) + 1u, nblk = (n_real + GA_SLOTS - 1u) / GA_SLOTS;
    __syncthreads();                                                // (s_blk is re-used: the coverage gather is done with it)
    for (uint32_t kk = tid; kk < np; kk += 256u) {
        const uint32_t sl = a.pairs[ps + kk].pad >> 8, prev = kk ? a.pairs[ps + kk - 1u].pad >> 8 : 0xffffffffu;
        if (kk == 0u || (prev / GA_SLOTS) != (sl / GA_SLOTS)) s_blk[sl / GA_SLOTS] = kk;
    }
    if (tid == 0u) s_blk[nblk] = np;
    const uint32_t wide_end = tile_has_merged ? (n_real & ~7u) : stride;
    const uint32_t *rows = reinterpret_cast<const uint32_t *>(a.aspill) + (uint64_t)ps * TILE;
    for (uint32_t jj0 = 0; jj0 < dense_n; jj0 += 64u) {
        const uint32_t nS = min(64u, dense_n - jj0);
        __syncthreads();
        if (tid < 64u) s_off[tid] = a.sites[base + j_lo + jj0 + min(tid, nS - 1u)].gpos - t0;
        for (uint32_t b = 0; b < nblk; ++b) {
            __syncthreads();
            for (uint32_t i = tid; i < 64u * GA_ROW; i += 256u) s_acc[i] = 0u;
            __syncthreads();
            const uint32_t kA = s_blk[b], kB = s_blk[b + 1u];
            const uint32_t off = s_off[lane];
            for (uint32_t kl = kA + wave; kl < kB; kl += 16u) {      // four loads in flight per lane
                uint32_t v[4], c[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t k = min(kl + 4u * u, kB - 1u);
                    v[u] = rows[(uint64_t)k * TILE + off];
                    c[u] = (a.pairs[ps + k].pad >> 8) - GA_SLOTS * b;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    if (kl + 4u * u >= kB || v[u] == 0u) continue;
#pragma unroll
                    for (uint32_t x = 0; x < 4u; ++x) {
                        const uint32_t n = (v[u] >> (8u * x)) & 0xffu;
                        if (n) atomicAdd(&s_acc[lane * GA_ROW + x * (GA_SLOTS / 2u) + (c[u] >> 1)], n << (16u * (c[u] & 1u)));
                    }
                }
            }
            __syncthreads();
            // a site's block: 4 alleles x 4 segments of eight slots
            for (uint32_t idx = tid; idx < nS * 16u; idx += 256u) {
                const uint32_t row = idx >> 4, x = (idx >> 2) & 3u, seg = idx & 3u, c0 = GA_SLOTS * b + 8u * seg;
                if (c0 >= n_real) continue;
                const uint2 lo = *reinterpret_cast<const uint2 *>(&s_acc[row * GA_ROW + x * (GA_SLOTS / 2u) + 4u * seg]);
                const uint2 hi = *reinterpret_cast<const uint2 *>(&s_acc[row * GA_ROW + x * (GA_SLOTS / 2u) + 4u * seg + 2u]);
                uint16_t *dst = a.ncol + (uint64_t)x * a.cells.cap_cells + cell0 + (uint64_t)(j_lo + jj0 + row) * stride + c0;
                if (c0 + 8u <= wide_end) *reinterpret_cast<uint4 *>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                else {
                    const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
                    for (uint32_t q = 0; q < 8u && c0 + q < n_real; ++q) dst[q] = (uint16_t)(w[q >> 1] >> (16u * (q & 1u)));
                }
            }
        }
    }
}

// One cell at a time (tiles with few sites, or rows that do not start on 16 bytes): the pair's allele word at the site -> its cell in
// the allele columns.  add: the pair is one of several of a split sample (their counts add up).
__device__ __forceinline__ void put_allele_word(const TailArgs &a, const uint32_t word, const unsigned long long cell, const bool add) {
    if (word == 0u) return;                                         // (what the gate kernel left there)
#pragma unroll
    for (uint32_t x = 0; x < 4u; ++x) {
        const uint32_t n = (word >> (8u * x)) & 0xffu;
        if (!n) continue;
        uint16_t *dst = a.ncol + (uint64_t)x * a.cells.cap_cells + cell;
        if (add) add_u16(dst, n); else *dst = (uint16_t)n;
    }
}

// LDS of the launch: ONE pool that the block types carve up (a workgroup is one of them): the dense gathers need 64 x GA_ROW words + three
// small tables, the merged gather 13.7 KB.  Separate __shared__ arrays per block type add up in every workgroup of the launch (31.6 KB:
// five workgroups per CU where the latency-bound scatter and merged blocks want eight).
constexpr uint32_t TAIL_POOL_WORDS = 64 * GA_ROW + 64 + 64 + 2 * GW_MAX_BLOCKS + 2;
__device__ __forceinline__ void gather_cov_block(const TailArgs &a, const uint32_t bid, uint32_t *s_pool) {
    const uint32_t GATHER_SPLIT = a.gather_split;
    const uint32_t tile = a.active_tiles[bid / GATHER_SPLIT], part = bid % GATHER_SPLIT;   // a tile's sites are dealt to GATHER_SPLIT workgroups
    const uint32_t n = a.tile_site_cnt[tile];
    if (n <= part) return;
    const uint32_t base = a.tile_site_base[tile];
    if (base + n > a.cap_out) return;                   // the host sees the site count and retries with a larger buffer
    const uint32_t n_slots = a.cells.tile_nslots[tile];
    const unsigned long long cell0 = a.cells.tile_cell_base[tile];
    if (cell0 + (unsigned long long)n * n_slots > a.cells.cap_cells) return;
    const uint32_t ps = a.tile_pair_start[tile], np = a.tile_pair_merged[tile] - ps;      // the merged pairs (behind the others) spill nothing
    if (np == 0u) return;
    const uint32_t t0 = tile * TILE;
    const uint32_t mine = (n - part + GATHER_SPLIT - 1) / GATHER_SPLIT;
    const uint32_t share = (n + GATHER_SPLIT - 1) / GATHER_SPLIT;          // (the same for every workgroup of the tile: they all take the same path)
    const uint32_t *arows = a.aspill ? reinterpret_cast<const uint32_t *>(a.aspill) + (uint64_t)ps * TILE : nullptr;      // allele rows of the tile's pairs (noisy reads)
    if (share >= GD_MIN_SITES) {
        // MANY sites in the tile (deep or divergent data: every other position of a cohort with LogNormal sigma = 2 abundances is a
        // site).  One cell per thread in (site, pair) order reads one byte of a different spill row per lane -- a cache line per cell,
        // 10.7 GB for 84 M cells, and the next site's bytes of the same lines come after the lines are gone -- so the block moves
        // 64 sites x 64 pairs at a time through LDS: a wavefront READS one pair's bytes at 64 of its sites (a line or a few) and WRITES
        // one site's cells of 64 pairs (consecutive slots).  A workgroup takes a CONTIGUOUS range of the tile's sites here (every
        // GATHER_SPLIT-th one, as below, spreads each 128-byte line of the cell rows over workgroups on different XCDs).
        const uint32_t j_lo = min(n, part * share), dense_n = min(n, j_lo + share) - j_lo;
        uint32_t *const s_lds = s_pool;                              // one block for every form of the dense gather (coverage: 64 x 34 words; the older form: 64 rows of 68 bytes; alleles: 64 x 66 words)
        static_assert(GA_ROW >= GW_ROW && GW_SITES == 64 && 64 * GW_ROW * 4 >= 64 * GD_ROW, "the coverage forms live in the allele gather's LDS block");
        uint32_t *const s_off = s_pool + 64 * GA_ROW, *const s_pad = s_off + 64, *const s_blk = s_pad + 64;      // (s_blk: first pair of every block of 64 slots -- coverage -- or 32 slots -- alleles)
        if ((n_slots & 7u) == 0u && (cell0 & 7ull) == 0ull && !a.has_wide && n_slots <= 64u * GW_MAX_BLOCKS) {      // (uniform)
            const bool has_merged = a.tile_pair_merged[tile] < a.tile_pair_start[tile + 1u];
            gather_cov_wide(a, s_lds, s_off, s_blk, t0, base, n_slots, cell0, ps, np, has_merged, j_lo, dense_n);
            if (arows) gather_alleles_wide(a, s_lds, s_off, s_blk, t0, base, n_slots, cell0, ps, np, has_merged, j_lo, dense_n);
            return;
        }
        uint8_t (*s_t)[GD_ROW] = reinterpret_cast<uint8_t (*)[GD_ROW]>(s_lds);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (uint32_t jj0 = 0; jj0 < dense_n; jj0 += 64) {
            const uint32_t nS = min(64u, dense_n - jj0);
            __syncthreads();                                        // (the rows of the previous group have been written out)
            if (threadIdx.x < 64) s_off[threadIdx.x] = a.sites[base + j_lo + jj0 + min(threadIdx.x, nS - 1u)].gpos - t0;   // (idle lanes: a valid position, their row is not written out)
            for (uint32_t kk0 = 0; kk0 < np; kk0 += 64u) {
                const uint32_t nP = min(64u, np - kk0);
                __syncthreads();                                    // s_off is visible; the previous 64 pairs have been written out
                if (threadIdx.x < nP) s_pad[threadIdx.x] = a.pairs[ps + kk0 + threadIdx.x].pad;
                const uint32_t off = s_off[lane];
                for (uint32_t kl = (uint32_t)wave; kl < nP; kl += 16) {      // four loads in flight per lane
                    uint8_t v[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) v[u] = a.spill[(uint64_t)(ps + kk0 + min(kl + 4u * u, nP - 1u)) * TILE + off];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) if (kl + 4u * u < nP) s_t[lane][kl + 4u * u] = v[u];
                }
                __syncthreads();
                if ((uint32_t)lane < nP) {
                    const uint32_t pad = s_pad[lane];
                    for (uint32_t sl = (uint32_t)wave; sl < nS; sl += 4) {
                        const uint32_t cov = s_t[sl][lane];
                        uint16_t *dst = &a.cov_col[cell0 + (uint64_t)(j_lo + jj0 + sl) * n_slots + (pad >> 8)];
                        if (pad & 0xffu) add_u16(dst, cov);
                        else if (cov != 255u) *dst = (uint16_t)cov;
                    }
                }
            }
        }
        if (arows) {                                                  // (rows that do not start on 16 bytes: small tiles; one cell per thread)
            const uint64_t work = (uint64_t)dense_n * np;
            for (uint64_t i = threadIdx.x; i < work; i += blockDim.x) {
                const uint32_t j = j_lo + (uint32_t)(i / np), kk = (uint32_t)(i % np);
                const uint32_t pad = a.pairs[ps + kk].pad;
                put_allele_word(a, arows[(uint64_t)kk * TILE + (a.sites[base + j].gpos - t0)], cell0 + (uint64_t)j * n_slots + (pad >> 8), (pad & 0xffu) != 0u);
            }
        }
        return;
    }
    const uint64_t work = (uint64_t)mine * np;
    for (uint64_t i = threadIdx.x; i < work; i += blockDim.x) {
        const uint32_t j = part + (uint32_t)(i / np) * GATHER_SPLIT, kk = (uint32_t)(i % np);
        const uint32_t off = a.sites[base + j].gpos - t0;
        const uint32_t pad = a.pairs[ps + kk].pad;          // kind | slot << 8
        const uint32_t cov = a.spill[(uint64_t)(ps + kk) * TILE + off];
        uint16_t *dst = &a.cov_col[cell0 + (uint64_t)j * n_slots + (pad >> 8)];
        if (pad & 0xffu) add_u16(dst, cov);                 // one of several pairs of this sample: the groups add up
        else if (cov != 255u && cov) *dst = (uint16_t)cov;  // (zero: what the gate kernel left there) 255 (wide kernel only): the overflow list holds the value, the scatter half writes it
        if (arows) put_allele_word(a, arows[(uint64_t)kk * TILE + off], cell0 + (uint64_t)j * n_slots + (pad >> 8), (pad & 0xffu) != 0u);
    }
}

__device__ __forceinline__ void scatter_events_block(const TailArgs &a, const uint32_t bx, const uint32_t k) {
    if (a.counters[2] > a.cap_out || *reinterpret_cast<const unsigned long long *>(&a.counters[CNT_CELLS]) > a.cells.cap_cells) return;
    // an event finds its cell from three INDEPENDENT lookups by position -- site bitmap word, first cell of the 64-position
    // block's first site, slots of the tile -- so the loop is two levels of dependent loads deep (event -> tables -> atomic)
    auto apply = [&](const Pair32 e, const bool allele) {
        const unsigned long long w = a.site_bits[e.x >> 6], bit = 1ull << (e.x & 63u);
        const unsigned long long row0 = a.cells.block_row[e.x >> 6];
        const uint32_t ns = a.cells.tile_nslots[e.x / TILE];
        if (!(w & bit)) return;                    // most events are sequencing errors at positions that are not sites
        const unsigned long long row = row0 + (unsigned long long)__popcll(w & (bit - 1ull)) * ns;
        // events carry the SLOT of their sample in the tile (pack.cpp)
        if (allele) add_u16(&a.ncol[ncol_base((e.y >> 16) & 3u, a.cells.cap_cells) + row + (e.y >> 18)], e.y & 0xffffu);   // one event per (site, pair, allele)
        else a.cov_col[row + (e.y >> 16)] = (uint16_t)(e.y & 0xffffu);
    };
    const uint32_t n_k = min(a.counters[16u + k * EV_CNT_STRIDE], a.cap_list);
    const Pair32 *list = a.events + (uint64_t)k * a.cap_list;
    {   // four events per trip: their loads, then their table loads, are in flight together
        constexpr uint32_t U = 4;
        const uint32_t stride = a.scatter_blocks * blockDim.x;
        uint32_t i = bx * blockDim.x + threadIdx.x;
        for (; i + (U - 1u) * stride < n_k; i += U * stride) {
            Pair32 e[U]; unsigned long long w[U], row0[U]; uint32_t ns[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) e[u] = list[i + u * stride];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) { w[u] = a.site_bits[e[u].x >> 6]; row0[u] = a.cells.block_row[e[u].x >> 6]; ns[u] = a.cells.tile_nslots[e[u].x / TILE]; }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const unsigned long long bit = 1ull << (e[u].x & 63u);
                if (!(w[u] & bit)) continue;
                const unsigned long long row = row0[u] + (unsigned long long)__popcll(w[u] & (bit - 1ull)) * ns[u];
                add_u16(&a.ncol[ncol_base((e[u].y >> 16) & 3u, a.cells.cap_cells) + row + (e[u].y >> 18)], e[u].y & 0xffffu);
            }
        }
        for (; i < n_k; i += stride) apply(list[i], true);
    }
    const uint32_t n_overflow = min(a.counters[1], a.cap_overflow);
    for (uint32_t i = (k * a.scatter_blocks + bx) * blockDim.x + threadIdx.x; i < n_overflow; i += a.scatter_blocks * EV_LISTS * blockDim.x)
        apply(a.overflow[i], false);
}

// Per-sample coverage of the called positions for the pairs of MERGED groups (no spill row exists for them): one workgroup
// per merged work item walks the item's groups; the pieces of a group are checked against the tile's site bitmap, and every
// counted base at a called position -- quality at or above the cutoff and either a match or one of A, C, G, T: the bases the
// pileup kernel did not put into its exception bins -- adds one to the group's LDS table [site][pair of the group], which
// is then written out with plain stores (every (site, sample) cell belongs to exactly one group).
// MSNV_MERGED_GATHER=block|wave forces a form (tests); default: a wavefront per group whenever every group is small enough
static bool merged_wave_form(const DeviceCols &d);
constexpr uint32_t GM_CELLS = 2048;           // cells of the LDS tables (6 B each); a tile with more sites x pairs is done in batches of sites
constexpr uint32_t GMW_CELLS = 256, GMW_PAIRS = 16, GMW_GROUPS = 4;     // the one-wavefront form: groups of <= 16 pairs, four of them per workgroup
constexpr uint32_t GMW_HITS = 240, GM_HITS = 480;       // (piece, site) overlaps listed per round (more: looked up where they are found)
constexpr uint32_t GMW_WORDS = 2 * (TILE / 64) + TILE / 64 + GMW_CELLS / 2 + GMW_CELLS + GMW_PAIRS + 2 + 2 * GMW_HITS;
constexpr uint32_t GM_WORDS = 2 * (TILE / 64) + TILE / 64 + GM_CELLS / 2 + GM_CELLS + MERGE_MAX_PAIRS + 2 + 2 * GM_HITS;
// NT = 256: the workgroup works on the group; NT = 64 (round 5): ONE WAVEFRONT does, four groups per workgroup side by side.  A group of a sparse
// cohort is ~250 pieces and ~5 sites -- no arithmetic to speak of but a chain of six to eight dependent loads (descriptor -> site tables and
// headers -> slots -> bases / flags / reference of every hit -> rule bits), and a launch of one group per workgroup kept eight chains per CU
// in flight where the wavefront slots allow thirty-two: 153 us of the configs[3] shard's 680 us pass (profiles/r05_sparse_ablation.txt).
template <uint32_t NT, uint32_t CELLS, uint32_t PAIRS, uint32_t HITS>
__device__ __forceinline__ void gather_merged_group(const TailArgs &a, const MergedGroupDev w, uint32_t *s_tab, const uint32_t tid) {
    static_assert(GM_WORDS <= TAIL_POOL_WORDS && GMW_GROUPS * GMW_WORDS <= TAIL_POOL_WORDS && GMW_WORDS % 2 == 0, "the merged gather's tables fit the launch's LDS pool");
    static_assert((2 * (TILE / 64) + TILE / 64 + CELLS / 2 + CELLS + PAIRS + 2) % 2 == 0, "the overlap list starts on 8 bytes");
    unsigned long long *const s_bits = reinterpret_cast<unsigned long long *>(s_tab);       // [TILE / 64]
    uint32_t *const s_rank = s_tab + 2 * (TILE / 64);                                        // [TILE / 64]
    uint32_t *const s_cov = s_rank + TILE / 64;         // [CELLS / 2]: u16 per (site, pair): counted bases
    uint32_t *const s_al = s_cov + CELLS / 2;           // [CELLS]: 4 x u8 per (site, pair): mismatching A, C, G, T (a shallow pair is < 81 deep)
    uint32_t *const s_gsample = s_al + CELLS;           // [PAIRS]
    uint32_t *const s_nhit = s_gsample + PAIRS;         // overlaps listed in this round
    unsigned long long *const s_hit = reinterpret_cast<unsigned long long *>(s_nhit + 2);      // [HITS]
    // (one wavefront: its LDS operations are served in order; the fences keep the compiler from moving them across)
    auto sync = [] {
        if constexpr (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
    };
    // everything that hangs on the descriptor alone is requested together -- the tile's site tables, its bitmap and ranks, the first
    // round(s) of piece headers, the pairs' slots: the group is a chain of dependent loads (one group of a sparse cohort is ~250 pieces), not arithmetic
    const uint32_t tile = w.tile, t0 = tile * TILE;
    const uint32_t n = a.tile_site_cnt[tile], base = a.tile_site_base[tile], n_slots = a.cells.tile_nslots[tile];
    const unsigned long long cell0 = a.cells.tile_cell_base[tile];
    unsigned long long my_bits = 0; uint32_t my_rank = 0;
    if (tid < TILE / 64) { my_bits = a.site_bits[(t0 >> 6) + tid]; my_rank = a.site_rank[(t0 >> 6) + tid]; }
    constexpr uint32_t PF = NT == 64 ? 4u : 1u;         // header rounds asked for up front
    PieceHdr h_first[PF];
#pragma unroll
    for (uint32_t k = 0; k < PF; ++k) { h_first[k] = PieceHdr{0u, 0u}; if (tid + k * NT < w.n_pieces) h_first[k] = a.hdr8m[w.hdr_base + tid + k * NT]; }
    const uint32_t m = w.n_pairs;                       // pairs of the group
    uint32_t my_slot = 0;
    if (tid < m) my_slot = a.pairs[w.pair_lo + tid].pad >> 8;      // the pair's slot in the tile
    if (n == 0u) return;
    if (base + n > a.cap_out) return;                   // the host sees the site count and retries with a larger buffer
    if (cell0 + (unsigned long long)n * n_slots > a.cells.cap_cells) return;
    if (tid < TILE / 64) { s_bits[tid] = my_bits; s_rank[tid] = my_rank - base; }
    if (tid < m) s_gsample[tid] = my_slot;
    uint32_t my_tally = 0;                              // individual-call lines this lane adds
    // one (piece, site) overlap: the base's flag, the base, the reference -> the (site, pair) cell
    auto count_hit = [&](const unsigned long long rec) {
        const unsigned long long qbit = rec & ((1ull << 40) - 1ull);
        const uint32_t gp = t0 + ((uint32_t)(rec >> 40) & (TILE - 1u)), cell = (uint32_t)(rec >> 51);
        const bool counted_q = !((a.qual[qbit >> 3] >> (qbit & 7u)) & 1u);
        const uint32_t code = (a.seq[qbit >> 1] >> (4u * (uint32_t)(qbit & 1u))) & 0xfu;       // (the flag's index is the base's index in the column)
        const uint32_t rc = (a.ref4[gp >> 3] >> (4u * (gp & 7u))) & 0xfu;
        if (counted_q && (code == rc || (code != 0u && (code & (code - 1u)) == 0u))) {
            atomicAdd(&s_cov[cell >> 1], 1u << (16u * (cell & 1u)));
            if (code != rc) atomicAdd(&s_al[cell], 1u << (8u * (uint32_t)__builtin_ctz(code)));
        }
    };
    {
        const uint32_t n_pieces = w.n_pieces;
        const uint32_t batch = max(1u, CELLS / m);        // sites per round
        for (uint32_t j0 = 0; j0 < n; j0 += batch) {
            const uint32_t nj = min(batch, n - j0);
            if (tid == 0u) *s_nhit = 0u;
            for (uint32_t i = tid; i < (nj * m + 1u) / 2u; i += NT) s_cov[i] = 0u;
            for (uint32_t i = tid; i < nj * m; i += NT) s_al[i] = 0u;
            sync();
            // phase 1: the lanes list the (piece, site) overlaps; phase 2: one overlap per lane -- the three loads of every overlap of the round are
            // in flight together (round 5: looked up where they were found, a lane's overlaps were as many memory round trips in a row)
            for (uint32_t pi = tid, k = 0; pi < n_pieces; pi += NT, ++k) {
                PieceHdr h;
                if (PF == 4u) h = k == 0u ? h_first[0] : k == 1u ? h_first[PF > 1 ? 1 : 0] : k == 2u ? h_first[PF > 2 ? 2 : 0] : k == 3u ? h_first[PF > 3 ? 3 : 0] : a.hdr8m[w.hdr_base + pi];
                else h = k == 0u ? h_first[0] : a.hdr8m[w.hdr_base + pi];
                const uint32_t s = h.w0 & (TILE - 1u), len = (h.w0 >> 11) & 0xffu, pidx = (h.w0 >> 19) & 0xffu;
                const uint64_t so = ((uint64_t)(h.w0 >> 27) << 32 | h.seqoff8) << SEQ_ALIGN_LOG2;
                for (uint32_t wd = s >> 6; wd <= (s + len - 1u) >> 6 && wd < TILE / 64; ++wd) {
                    unsigned long long bits = s_bits[wd];
                    if (wd == (s >> 6)) bits &= ~0ull << (s & 63u);
                    if (wd == ((s + len - 1u) >> 6)) bits &= ~0ull >> (63u - ((s + len - 1u) & 63u));
                    while (bits) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(bits);
                        const unsigned long long low = bits & (0ull - bits);
                        bits ^= low;
                        const uint32_t j = s_rank[wd] + (uint32_t)__popcll(s_bits[wd] & (low - 1ull));   // site index inside the tile
                        if (j < j0 || j >= j0 + nj) continue;
                        const uint32_t q = (wd << 6) + b, o = q - s;
                        const unsigned long long rec = (2ull * so + o) | (unsigned long long)q << 40 | (unsigned long long)((j - j0) * m + pidx) << 51;      // flag bit (40) | position (11) | cell (<= 11 bits)
                        const uint32_t slot = atomicAdd(s_nhit, 1u);
                        if (slot < HITS) s_hit[slot] = rec;
                        else count_hit(rec);                          // (more overlaps than the list holds: looked up here)
                    }
                }
            }
            sync();
            for (uint32_t i = tid, nh = min(*s_nhit, HITS); i < nh; i += NT) count_hit(s_hit[i]);
            sync();
            for (uint32_t i = tid; i < nj * m; i += NT) {
                const uint32_t v = (s_cov[i >> 1] >> (16u * (i & 1u))) & 0xffffu;
                const uint64_t cell = cell0 + (uint64_t)(j0 + i / m) * n_slots + s_gsample[i % m];
                if (v) a.cov_col[cell] = (uint16_t)v;
                const uint32_t al = s_al[i];
                if (al) {
                    uint32_t ind = 0;
#pragma unroll
                    for (uint32_t x = 0; x < 4; ++x) {
                        const uint32_t cnt = (al >> (8u * x)) & 0xffu;
                        if (cnt) a.ncol[(uint64_t)x * a.cells.cap_cells + cell] = (uint16_t)cnt;
                        if (cnt >= a.min_snvs) ind |= 1u << x;
                    }
                    // call_vC.cpp:593-600: this sample holds >= t reads of an allele the gate kernel left open -> individual call
                    if (ind && a.ind_in_gather) {
                        const uint32_t site = base + j0 + i / m;
                        ind &= a.site_elig[site];
                        if (ind) {
                            uint32_t *wordp = reinterpret_cast<uint32_t *>(a.site_flags + (site & ~3u));
                            const uint32_t sh = 8u * (site & 3u);
                            const uint32_t old = atomicOr(wordp, (ind << 4) << sh);
                            if ((((old >> sh) & 0xffu) >> 4) == 0u) ++my_tally;      // the site's first individual call: one more indiv_called line
                        }
                    }
                }
            }
            sync();
        }
    }
    // one add per wavefront (round 5: one per site -- 125 k adds to ONE word on the configs[3] shard, served one after the other by that
    // word's L2 channel -- was most of this launch's 150 us)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_tally += (uint32_t)__shfl_xor((int)my_tally, o);
    if ((tid & 63u) == 0u && my_tally) atomicAdd(reinterpret_cast<unsigned long long *>(&a.counters[CNT_TALLY]), (unsigned long long)my_tally << 32);
}
// merged_per_block = 1: ONE group per workgroup (a work item's groups in a row made this block the launch's long pole); GMW_GROUPS when every
// group of the dataset holds <= GMW_PAIRS pairs (a sparse cohort): a wavefront each
__device__ __forceinline__ void gather_merged_block(const TailArgs &a, const uint32_t bid, uint32_t *s_pool) {
    if (a.merged_per_block == 1u) { gather_merged_group<256, GM_CELLS, MERGE_MAX_PAIRS, GM_HITS>(a, a.merged_groups[bid], s_pool, threadIdx.x); return; }
    const uint32_t gi = bid * GMW_GROUPS + (threadIdx.x >> 6);
    if (gi < a.n_merged_run) gather_merged_group<64, GMW_CELLS, GMW_PAIRS, GMW_HITS>(a, a.merged_groups[gi], s_pool + (threadIdx.x >> 6) * GMW_WORDS, threadIdx.x & 63u);
}
static bool merged_wave_form(const DeviceCols &d) {
    const int forced = [] { const char *e = getenv("MSNV_MERGED_GATHER"); return !e ? 0 : e[0] == 'b' ? 1 : e[0] == 'w' ? 2 : 0; }();      // (per pass: tests switch it)
    return d.max_group_pairs <= GMW_PAIRS && forced != 1;
}

// msnv_gather_scatter: one launch, two independent halves working on the zeroed (msnv_gate_sites) per-sample records.
//   blocks [0, n_gather_blocks): per-sample coverage of every surviving site, from the spilled bytes;
//   the other SCATTER_BLOCKS_PER_LIST x EV_LISTS blocks: per-sample allele counts from the event sub-lists (sparse) and the
//   >= 255 coverages of the wide kernel from the overflow list.
// HAS_MERGED = false: no merged groups in the dataset -- the instantiation without the merged gather's LDS tables (with them
// every workgroup of the launch allocates 14 KB it never touches: 33.8 -> 43 us on the benchmark shape)
template <bool HAS_MERGED>
__global__ __launch_bounds__(256) void msnv_gather_scatter(TailArgs a) {
    const uint32_t n_scatter = a.scatter_blocks * EV_LISTS;     // dispatched first: the longer-running half
    if (blockIdx.x == gridDim.x - 1u) {
        // (one workgroup of its own at the END of the grid: in front of a scatter block -- the long pole of this launch -- the 32
        // counter loads delayed the whole kernel by ~10 us)
        // counters[0] for the host: the number of allele events, or -- when a sub-list overflowed -- the total capacity
        // that would have held the fullest one (the host grows the list to that and runs the pass again)
        if (threadIdx.x < 64) {
            const uint32_t c = threadIdx.x < EV_LISTS ? a.counters[16u + threadIdx.x * EV_CNT_STRIDE] : 0u;
            unsigned long long total = c; uint32_t fullest = c;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { total += __shfl_xor(total, o); fullest = max(fullest, (uint32_t)__shfl_xor((int)fullest, o)); }
            if (fullest > a.cap_list) total = (unsigned long long)fullest * EV_LISTS;
            if (threadIdx.x == 0) a.counters[0] = (uint32_t)min(total, 0xffffffffull);
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[TAIL_POOL_WORDS];
    if (blockIdx.x < n_scatter) { if (!(a.debug_skip & 1u)) scatter_events_block(a, blockIdx.x % a.scatter_blocks, blockIdx.x / a.scatter_blocks); }
    else if (HAS_MERGED && blockIdx.x < n_scatter + a.n_merged_blocks) { if (!(a.debug_skip & 4u)) gather_merged_block(a, blockIdx.x - n_scatter, s_pool); }
    else if (!(a.debug_skip & 2u)) gather_cov_block(a, blockIdx.x - n_scatter - a.n_merged_blocks, s_pool);
}

// ------------------------------------------------------------------------------------------
// msnv_decide_sites: call_vC.cpp:577-601.  One wavefront per site.
//   skip allele x when the FASTA character equals the lower-case letter x (:580)
//   population  iff n_x >= t and (double)n_x >= cov * min_fraction          (:588)
//   individual  iff not population and some sample has x_s >= t              (:593-600)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msnv_decide_sites(const SiteRec *sites, const uint32_t *unc_sites, uint32_t *counters, uint32_t cap_sites, uint32_t cap_out,
                                                         const uint32_t *ref4, const uint32_t *ref_lc, const uint16_t *ncol,
                                                         const CellMap cells, int min_snvs, double min_frac, uint8_t *site_flags) {
    // Only the sites the gate kernel could not decide: some allele x has n_x >= t, is no population call, no single pair holds
    // >= t reads of it, but a sample that is split into several pairs / sits in a merged group holds some -- the individual rule
    // then asks the summed per-sample records.  One wavefront per site.
    __shared__ uint32_t s_pop, s_ind;
    if (threadIdx.x == 0) { s_pop = 0; s_ind = 0; }
    __syncthreads();
    const uint32_t n_unc = min(counters[CNT_UNC], cap_sites), n_sites = counters[2];
    const int lane = threadIdx.x & 63;
    if (n_sites <= cap_sites && n_sites <= cap_out && *reinterpret_cast<const unsigned long long *>(&counters[CNT_CELLS]) <= cells.cap_cells) {
        for (uint32_t u = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < n_unc; u += gridDim.x * (blockDim.x >> 6)) {
            const uint32_t site = unc_sites[u];
            const SiteRec s = sites[site];
            const uint32_t rc = (ref4[s.gpos >> 3] >> (4 * (s.gpos & 7))) & 0xfu;
            const bool lc = (ref_lc[s.gpos >> 5] >> (s.gpos & 31)) & 1u;
            uint32_t pop = 0, ind = 0;
            for (int x = 0; x < 4; ++x) {
                if (lc && rc == (1u << x)) continue;
                const uint32_t nx = s.n[x];
                if ((int)nx < min_snvs) continue;           // neither rule can fire
                if ((double)nx >= (double)(int)s.cov * min_frac) { pop |= 1u << x; continue; }
                bool any = false;
                const uint32_t tile = s.gpos / TILE, n_slots = cells.tile_nslots[tile];
                const uint64_t row = cell_of(cells, tile, site, 0u);
                for (uint32_t i = lane; i < n_slots; i += 64)
                    any |= (int)ncol[(uint64_t)x * cells.cap_cells + row + i] >= min_snvs;
                if (__any(any)) ind |= 1u << x;
            }
            if (lane == 0) {
                site_flags[site] = (uint8_t)(pop | ind << 4);
                if (pop) atomicAdd(&s_pop, 1u);
                if (ind) atomicAdd(&s_ind, 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                 // output-line tallies: one global atomic per workgroup
        if (s_pop) atomicAdd(&counters[4], s_pop);
        if (s_ind) atomicAdd(&counters[5], s_ind);
    }
}

// ------------------------------------------------------------------------------------------
// msnv_coverage_tiles: qaCompute's per-contig coverage arithmetic (qaCompute.cpp:530-552 scatter,
// :142-165 prefix sum + histogram) for every sample, tile by tile.
//   per (tile, sample): +1/-1 of every M interval into an LDS difference array (intervals that
//   started in an earlier tile enter at index 0), then per RUN of constant depth (i < contig length)
//   covSum += depth x length and hist[min(depth, max_cov)] += length, reduced per wavefront and added
//   to the (sample, contig) accumulators with 64-bit atomics.
// Algorithmic HBM bytes: 8 B per M interval.  The kernel is bound by vector-instruction issue
// (DESIGN.md section 4 "Coverage kernel"), so its form follows the instruction count per pair.
// ------------------------------------------------------------------------------------------
constexpr int C_NT = 64;                     // one wavefront per workgroup: nothing is shared, and the chip's wave slots refill one by one (workgroups of four: -9 %)
static_assert(C_NT / 64 * COV_PW == COV_ITEM_PAIRS, "coverage work items are sized for COV_PW pairs per wavefront");

__device__ __forceinline__ int wave_reduce_add(int x) {
    x = wave_inclusive_scan(x);
    return __builtin_amdgcn_readlane(x, 63);
}
// sum over the 16 lanes of a row, left in all of them (xor 1, xor 2, mirror of 8, mirror of 16)
__device__ __forceinline__ uint32_t row_sum16(uint32_t x) {
    x = (uint32_t)dpp_add<0xB1, 0xf>((int)x);    // quad_perm [1, 0, 3, 2]
    x = (uint32_t)dpp_add<0x4E, 0xf>((int)x);    // quad_perm [2, 3, 0, 1]
    x = (uint32_t)dpp_add<0x141, 0xf>((int)x);   // row_half_mirror
    x = (uint32_t)dpp_add<0x140, 0xf>((int)x);   // row_mirror
    return x;
}

// WIDE = false: the difference array holds two positions per word (16-bit halves biased by 0x8000, so that a half never borrows
// from or carries into its neighbour) -- good for pairs of at most 32 767 intervals, which pack.cpp guarantees for the work items it hands
// to this variant; WIDE = true: one word per position, any number of intervals (a pile-up of tens of thousands of reads with one
// start: a handful of work items at most).
template <bool WIDE>
__global__ __launch_bounds__(C_NT) void msnv_coverage_tiles(const Pair32 *iv, const TilePair *pairs,
                                                            const WorkItem *work, const uint32_t *tile_len,
                                                            unsigned long long *acc, uint32_t n_rows, int max_cov, uint32_t n_copies,
                                                            const uint64_t n_iv) {
    // One WAVEFRONT per (tile, sample) pair, no workgroup barrier at all, and the work follows the BREAKPOINTS of the coverage
    // instead of the positions: an interval adds +1 / -1 to the difference array in LDS and sets the bit of either end in a
    // 2048-bit mask; lane l owns positions [32 l, 32 l + 32) = one mask word and the 64 (128) bytes of differences behind it.  It
    // sums ALL its differences with four (eight) 16-byte reads -- one wave scan then gives every lane the depth at its left edge --,
    // walks its ~7 breakpoints (runs of constant depth: covSum += depth x length, hist[min(depth, max_cov)] += length in byte
    // fields of two registers) and clears its words with four (eight) 16-byte writes.  The wavefront's histogram is reduced by
    // halving: two lane-swap steps fold the four byte-field registers into ONE whose rows of 16 lanes hold four bins each, then
    // four row steps on 16-bit fields.  ~445 vector instructions per pair at ~215 intervals (the first wavefront-per-pair form:
    // ~915 -- three divergent loops over the breakpoints and nine wave scans; the workgroup-per-pair form before it: ~3200).
    constexpr int WPL = WIDE ? 32 : 16;                      // LDS words per lane ...
    constexpr int LSTRIDE = WPL + 4;                         // ... 16 bytes apart from the next lane's: the 16-byte reads and writes of the 64 lanes then spread over all banks (back to back they met 4 to a bank)
    constexpr uint32_t EMPTY = WIDE ? 0u : 0x80008000u;      // both halves biased: neither ever borrows from or carries into its neighbour, and a half reads as difference + 0x8000
    __shared__ __attribute__((aligned(16))) uint32_t s_d[C_NT / 64][64 * LSTRIDE];
    __shared__ uint32_t s_mask[C_NT / 64][TILE / 32];
    const WorkItem w = work[blockIdx.x];
    const uint32_t t0 = w.tile * TILE, tl = tile_len[w.tile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *const d = s_d[wave];
    uint32_t *const mask = s_mask[wave];
    static_assert(TILE / 32 == 64, "one mask word per lane");
    auto base_of = [](const TilePair &p) -> uint64_t { return (uint64_t)p.nblk << 32 | p.blk_lo; };      // absolute index of the sample's first interval
    uint4 *const mine = reinterpret_cast<uint4 *>(d + LSTRIDE * lane);
    auto diff_at = [&](const uint32_t bpos) -> int {         // difference at my position bpos
        if constexpr (WIDE) return (int)d[LSTRIDE * lane + (int)bpos];
        else {
            return (int)reinterpret_cast<const uint16_t *>(d + LSTRIDE * lane)[bpos] - 0x8000;
        }
    };
    const uint32_t lim = (uint32_t)min(max((int)tl - 32 * lane, 0), 32);      // scanned positions among my 32 (i < contig length)
    const uint32_t lim_mask = lim >= 32u ? 0xffffffffu : (1u << lim) - 1u;
    // A work item holds at most COV_PW pairs (pack.cpp): ALL their descriptors are loaded up front, and the first 256 intervals of
    // a pair while the pair before it is worked on (the kernel waited on one dependent descriptor -> intervals chain per pair, and
    // then on the intervals beyond the first 128).  A lane takes FOUR CONSECUTIVE intervals (two 16-byte loads): the 64 lanes of one
    // scatter step are then four intervals apart in the sorted list and seldom meet on an LDS word (64 consecutive intervals at 10x
    // lie within ~600 positions = 19 mask words)
    TilePair prs[COV_PW];
    uint4 xa[COV_PW], xb[COV_PW];
#pragma unroll
    for (int j = 0; j < COV_PW; ++j) {
        const uint32_t kk = w.pair_lo + (uint32_t)(wave * COV_PW + j);
        prs[j] = kk < w.pair_hi ? pairs[kk] : TilePair{0, 0, 0, 0, 0, 0, 0, 0};
    }
    auto load4 = [&](const Pair32 *v, const uint32_t first, const uint32_t n, uint4 &a, uint4 &b) {
        // lanes without an interval load the {0, 0} entries behind the last one (pack.cpp; they touch nothing): a load under a lane
        // condition gets its own wait right behind it, and the round trips then run one after the other instead of under the work
        // on the pair before.  The entries past a pair's last interval are somebody else's: scatter4 leaves them alone
        const Pair32 *p = first < n ? v + first : iv + n_iv;
        __builtin_memcpy(&a, p, 16);
        __builtin_memcpy(&b, p + 2, 16);
    };
    // (a pair of at most 64 intervals -- a cohort of many shallow samples -- takes ONE per lane: one scatter step instead of four
    // with a handful of lanes each)
    auto load_ahead = [&](const int j) {
        const uint32_t n = prs[j].read_hi - prs[j].read_lo;
        load4(iv + base_of(prs[j]) + prs[j].read_lo, n <= 64u ? (uint32_t)lane : 4u * (uint32_t)lane, n, xa[j], xb[j]);
    };
    load_ahead(0);
    const uint4 empty4 = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
#pragma unroll
    for (int i = 0; i < WPL / 4; ++i) mine[i] = empty4;                      // (under the loads)
    mask[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int j = 0; j < COV_PW; ++j) {
        const TilePair pr = prs[j];
        if (w.pair_lo + (uint32_t)(wave * COV_PW + j) >= w.pair_hi) break;
        if (j + 1 < COV_PW) load_ahead(j + 1);
        auto add_at = [&](const uint32_t p, const bool up) {
            uint32_t *const word = d + 4u * (p >> 5) + (WIDE ? p : p >> 1);         // (lane p >> 5, LSTRIDE words each)
            const uint32_t one = WIDE ? 1u : 1u << ((p & 1u) << 4);
            if (up) atomicAdd(word, one); else atomicSub(word, one);
            atomicOr(&mask[p >> 5], 1u << (p & 31u));
        };
        auto scatter = [&](const Pair32 x) {
            // one form for every case: an interval that started in an earlier tile enters at 0; an end left of the tile wraps
            // (unsigned) and leaves it on the right like one beyond it; {end, end - 1} -- qaCompute's `--entireChr[chrSize-1]`
            // without a matching `++` inside the scanned range, an M op whose cursor is at or beyond the contig end
            // (qaCompute.cpp:542-549; pack.cpp stores it that way) -- has no start and its -1 at x.y; the {0, 0} of an idle lane
            // touches nothing (in the tile at 0: +1 and -1 on position 0)
            const uint32_t m = max(x.x, t0), e = x.y - t0;
            if (x.y >= m && m - t0 < TILE) add_at(m - t0, true);
            if (e < TILE) add_at(e, false);
        };
        const uint32_t n_iv_pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pr.read_hi - pr.read_lo));
        auto scatter4 = [&](const uint4 a, const uint4 b, const uint32_t first) {
            scatter(Pair32{a.x, a.y});                       // (mine, or {0, 0})
            if (first + 1u < n_iv_pair) scatter(Pair32{a.z, a.w});
            if (first + 2u < n_iv_pair) scatter(Pair32{b.x, b.y});
            if (first + 3u < n_iv_pair) scatter(Pair32{b.z, b.w});
        };
        if (n_iv_pair <= 64u) scatter(Pair32{xa[j].x, xa[j].y});
        else scatter4(xa[j], xb[j], 4u * (uint32_t)lane);
        if (n_iv_pair > 256u) {                              // a deep pair: 256 intervals per step
            const Pair32 *v = iv + base_of(pr) + pr.read_lo;
            for (uint32_t first = 256u + 4u * (uint32_t)lane; first < n_iv_pair + 4u * (uint32_t)lane; first += 256u) {
                uint4 ya, yb;
                load4(v, first, n_iv_pair, ya, yb);
                scatter4(ya, yb, first);
            }
        }
        // the LDS executes one wavefront's instructions in order: the atomics above are done before the reads below are served
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t m = mask[lane] & lim_mask;           // breakpoints behind the contig end are not scanned
        mask[lane] = 0u;
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < WPL / 4; ++i) { const uint4 q = mine[i]; sum += q.x + q.y + q.z + q.w; }
        int delta;
        if constexpr (WIDE) delta = (int)sum;
        else {
            // sum = 65536 H + L + 16 x 0x8000 (+ 16 x 0x8000 x 65536 = 0 mod 2^32) with L, H = my even / odd positions'
            // differences, each within +-32 767
            const uint32_t s2 = sum - 16u * 0x8000u;
            const int L = (int)(short)(s2 & 0xffffu);
            delta = L + (((int)s2 - L) >> 16);
        }
        int cur = wave_inclusive_scan(delta) - delta;      // depth at my left edge
        // runs of constant depth inside my scanned positions
        unsigned long long ha = 0, hb = 0;                 // bins 0-7 / 8-15, one byte each (a lane adds at most 32)
        int csum = 0;
        uint32_t prev = 0;
        auto account = [&](const int depth, const uint32_t len) {
            csum += depth * (int)len;
            // -1 at the last position of a contig (see above): the reference then increments coverageHist[-1], out of bounds --
            // the position lands in no bin, the sum takes the -1 (and wraps, unsigned, exactly as covSum does)
            const uint32_t bin = (uint32_t)min(depth, max_cov), sh = (8u * bin) & 63u;
            const uint32_t counted = depth < 0 ? 0u : len, la = bin < 8u ? counted : 0u;
            ha += (unsigned long long)la << sh;
            hb += (unsigned long long)(counted - la) << sh;
        };
        for (uint32_t mm = m; mm; mm &= mm - 1u) {
            const uint32_t bpos = (uint32_t)__builtin_ctz(mm);
            account(cur, bpos - prev);
            cur += diff_at(bpos);
            prev = bpos;
        }
        account(cur, lim - prev);
#pragma unroll
        for (int i = 0; i < WPL / 4; ++i) mine[i] = empty4;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // reduce over the wavefront.  Histogram: G[g] = bins 4 g .. 4 g + 3 in bytes; lanes l and l + 32 meet in the first swap
        // (G0 | G1 in the two halves of one register), rows r and r + 1 in the second: one register whose rows hold G0, G2, G1, G3
        // summed over 4 lanes (bytes <= 128); then 16-bit fields and the four steps inside a row (a wavefront adds <= 2048 per bin)
        const int ws = wave_reduce_add(csum);
        const auto a01 = __builtin_amdgcn_permlane32_swap((uint32_t)ha, (uint32_t)(ha >> 32), false, false);
        const auto a23 = __builtin_amdgcn_permlane32_swap((uint32_t)hb, (uint32_t)(hb >> 32), false, false);
        const auto c = __builtin_amdgcn_permlane16_swap(a01[0] + a01[1], a23[0] + a23[1], false, false);
        const uint32_t c4 = c[0] + c[1];
        const uint32_t u0 = row_sum16(__builtin_amdgcn_perm(c4, c4, 0x0c010c00u));     // bins 4 g, 4 g + 1
        const uint32_t u1 = row_sum16(__builtin_amdgcn_perm(c4, c4, 0x0c030c02u));     // bins 4 g + 2, 4 g + 3
        // lanes 0-3 of row r hold the bins of group g = {0, 2, 1, 3}[r]; lane 4 adds covSum
        const uint32_t k = (uint32_t)lane & 15u, r = (uint32_t)lane >> 4;
        const uint32_t g = ((r & 1u) << 1) | (r >> 1);
        const uint32_t two = (k & 2u) ? u1 : u0;
        const uint32_t bin = 4u * g + k;
        int val = (int)((k & 1u) ? two >> 16 : two & 0xffffu);
        uint32_t idx = 1u + bin;
        if (lane == 4) { val = ws; idx = 0u; }
        if ((k < 4u || lane == 4) && val && idx <= (uint32_t)max_cov + 1u) {
            // one accumulator row per (sample, contig) that HAS intervals (pack.cpp numbers them; TilePair::max_depth carries the row): a
            // 500-sample cohort over a million contigs would need 100 GB as a dense [sample][contig] table, and holds a few million rows
            unsigned long long *dst = acc + ((uint64_t)(w.tile % n_copies) * n_rows + pr.max_depth) * (1 + COV_BINS);
            atomicAdd(&dst[idx], (unsigned long long)(long long)val);
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
uint32_t dev_resident_workgroups(uint32_t per_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256u * per_cu;
    return (uint32_t)prop.multiProcessorCount * per_cu;
}

int dev_set_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MSNV_ENODEV, "no HIP device is available (this path has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MSNV_ENODEV, "HIP device %d does not exist (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    return MSNV_OK;
}

// MSNV_GUARD_ALLOC=1 (debugging; tests/test_gpu_guard.py): every device buffer is mapped through the virtual-memory API so that it ENDS at
// the end of its mapping, with unmapped address space reserved behind it -- a read or write past the end of a buffer is then a GPU memory
// fault wherever the allocator would otherwise have put a neighbour (round 3: a rocprofv3 counter pass moved the neighbours and faulted).
struct GuardedAlloc { void *va; size_t reserved, mapped; hipMemGenericAllocationHandle_t handle; };
static std::mutex g_guard_mu;
static std::unordered_map<void *, GuardedAlloc> g_guarded;
static bool guard_alloc_enabled() { static const bool on = [] { const char *e = getenv("MSNV_GUARD_ALLOC"); return e && e[0] == '1'; }(); return on; }

// Device memory goes through a small caching allocator: a buffer that is given back keeps its mapping and serves the next request of about
// its size.  hipFree of the rounds' work buffers and columns was a quarter of finalize's wall time (110 us a call on average, 0.7 ms for the
// gigabyte-sized ones; a dataset's build makes ~270 of them), and every dataset of a process asks for the same sizes again.  Blocks are
// keyed by device; a request takes the smallest free block of at least its size and at most 1.5 x (+ 1 MB) of it.  Free blocks are handed
// back to the runtime when they exceed MSNV_DEV_CACHE_MB in total (default: a sixteenth of the device's memory, at most 16 GB; 0 = no cache),
// when an allocation fails (then everything cached goes and the allocation is tried again), and by dev_cache_trim() (context destroy).
// Stream order: a block is handed on only after hipDeviceSynchronize() in dev_free -- what hipFree does itself -- so no kernel of its
// previous owner is still running when the next owner's first write (possibly on another stream) arrives.
struct CacheBlock { void *p; uint64_t bytes; int device; };
static std::mutex g_cache_mu;
static std::vector<CacheBlock> g_cache_free;
static std::unordered_map<void *, CacheBlock> g_cache_live;
static uint64_t g_cache_free_bytes = 0;
static uint64_t cache_cap_bytes(int dev) {                          // per device: the default is a share of THAT device's memory (the caller has made it current)
    static const long long env_mb = [] { const char *e = getenv("MSNV_DEV_CACHE_MB"); return e ? std::max<long long>(0, atoll(e)) : -1ll; }();
    if (env_mb >= 0) return (uint64_t)env_mb << 20;
    static std::mutex mu;
    static std::unordered_map<int, uint64_t> caps;
    std::lock_guard<std::mutex> lk(mu);
    auto it = caps.find(dev);
    if (it != caps.end()) return it->second;
    size_t fr = 0, tot = 0;
    const uint64_t cap = hipMemGetInfo(&fr, &tot) != hipSuccess ? (uint64_t)1 << 30 : std::min<uint64_t>((uint64_t)tot / 16, (uint64_t)16 << 30);
    caps[dev] = cap;
    return cap;
}
static uint64_t cache_round(uint64_t bytes) {
    if (bytes <= 4096) return 4096;
    if (bytes < (1u << 20)) { uint64_t r = 4096; while (r < bytes) r <<= 1; return r; }     // powers of two below 1 MB
    return (bytes + (1u << 20) - 1) & ~(uint64_t)((1u << 20) - 1);                          // whole megabytes above
}
void dev_cache_trim() {
    std::vector<CacheBlock> out;
    { std::lock_guard<std::mutex> lk(g_cache_mu); out.swap(g_cache_free); g_cache_free_bytes = 0; }
    for (const CacheBlock &b : out) (void)hipFree(b.p);
}

int dev_alloc(void **p, uint64_t bytes, uint64_t *acct) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (guard_alloc_enabled()) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        size_t gran = 0;
        HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
        if (gran == 0) gran = 2u << 20;
        GuardedAlloc g{};
        g.mapped = (size_t)((bytes + gran - 1) / gran * gran);
        g.reserved = g.mapped + gran;                               // one granule behind the buffer stays unmapped
        HIP_TRY(hipMemAddressReserve(&g.va, g.reserved, gran, nullptr, 0));
        HIP_TRY(hipMemCreate(&g.handle, g.mapped, &prop, 0));
        HIP_TRY(hipMemMap(g.va, g.mapped, 0, g.handle, 0));
        hipMemAccessDesc ad = {};
        ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        HIP_TRY(hipMemSetAccess(g.va, g.mapped, &ad, 1));
        const size_t used = (size_t)((bytes + 15) & ~(uint64_t)15);  // (the buffers are read with 16-byte loads: keep that alignment)
        *p = static_cast<char *>(g.va) + (g.mapped - used);
        if (const char *e = getenv("MSNV_GUARD_FILL")) { HIP_TRY(hipMemset(g.va, atoi(e), g.mapped)); HIP_TRY(hipStreamSynchronize(nullptr)); }      // (fresh mappings are not defined to be zero: 0 or 255 tells a read of unwritten memory)
        if (getenv("MSNV_GUARD_LOG")) fprintf(stderr, "[guard] %p .. %p (%llu bytes; mapping %p + %zu)\n", *p, static_cast<char *>(*p) + bytes, (unsigned long long)bytes, g.va, g.mapped);
        std::lock_guard<std::mutex> lk(g_guard_mu);
        g_guarded[*p] = g;
    } else {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (cache_cap_bytes(dev) == 0) { HIP_TRY(hipMalloc(p, bytes)); if (acct) *acct += bytes; return MSNV_OK; }
        const uint64_t want = cache_round(bytes);
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            size_t best = SIZE_MAX;
            for (size_t i = 0; i < g_cache_free.size(); ++i) {
                const CacheBlock &b = g_cache_free[i];
                if (b.device != dev || b.bytes < want || b.bytes > want + want / 2 + (1u << 20)) continue;
                if (best == SIZE_MAX || b.bytes < g_cache_free[best].bytes) best = i;
            }
            if (best != SIZE_MAX) {
                const CacheBlock b = g_cache_free[best];
                g_cache_free[best] = g_cache_free.back(); g_cache_free.pop_back();
                g_cache_free_bytes -= b.bytes;
                g_cache_live[b.p] = b;
                *p = b.p;
            }
        }
        if (!*p) {
            hipError_t e = hipMalloc(p, want);
            if (e != hipSuccess) { (void)hipGetLastError(); dev_cache_trim(); e = hipMalloc(p, want); }       // (the cache must never be the reason an allocation fails)
            if (e != hipSuccess) { *p = nullptr; return fail(MSNV_EHIP, "hipMalloc of %llu bytes failed: %s", (unsigned long long)want, hipGetErrorString(e)); }
            std::lock_guard<std::mutex> lk(g_cache_mu);
            g_cache_live[*p] = CacheBlock{*p, want, dev};
        }
    }
    if (acct) *acct += bytes;
    return MSNV_OK;
}
void dev_free(void *p) {
    if (!p) return;
    if (guard_alloc_enabled()) {
        GuardedAlloc g{};
        {
            std::lock_guard<std::mutex> lk(g_guard_mu);
            auto it = g_guarded.find(p);
            if (it == g_guarded.end()) { (void)hipFree(p); return; }
            g = it->second; g_guarded.erase(it);
        }
        (void)hipDeviceSynchronize();                               // (what hipFree does by itself)
        // the address range stays reserved: a freed buffer's addresses are unmapped for good, so a stale pointer faults too.  (Handing the
        // range back -- hipMemAddressFree -- and mapping the next buffer over it gave kernels wrong bytes on this runtime, with every
        // kernel serialised and no stale pointer anywhere: profiles/r03zq notes in MEASURED.md)
        (void)hipMemUnmap(g.va, g.mapped); (void)hipMemRelease(g.handle);
        return;
    }
    CacheBlock b{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_cache_live.find(p);
        if (it != g_cache_live.end()) { b = it->second; g_cache_live.erase(it); }
    }
    if (!b.p) { (void)hipFree(p); return; }                         // (not one of ours: allocated while the cache was off)
    // no kernel of the old owner is running when the next owner gets the block: the BLOCK's device is the one to wait for (a dataset or a
    // context may be destroyed while another device is current)
    int cur = b.device;
    (void)hipGetDevice(&cur);
    if (cur != b.device) (void)hipSetDevice(b.device);
    (void)hipDeviceSynchronize();
    const uint64_t cap = cache_cap_bytes(b.device);
    if (cur != b.device) (void)hipSetDevice(cur);
    std::vector<CacheBlock> evict;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_cache_free.push_back(b); g_cache_free_bytes += b.bytes;
        while (g_cache_free_bytes > cap && !g_cache_free.empty()) {  // over the cap: the largest blocks go first
            size_t big = 0;
            for (size_t i = 1; i < g_cache_free.size(); ++i) if (g_cache_free[i].bytes > g_cache_free[big].bytes) big = i;
            evict.push_back(g_cache_free[big]); g_cache_free_bytes -= g_cache_free[big].bytes;
            g_cache_free[big] = g_cache_free.back(); g_cache_free.pop_back();
        }
    }
    for (const CacheBlock &e : evict) (void)hipFree(e.p);
}
// Many buffers of ONE owner at once (the device pack's tables and work buffers at the end of finalize): one wait for the device, not one per buffer.
void dev_free_batch(const std::vector<void *> &ptrs) {
    if (ptrs.empty()) return;
    if (guard_alloc_enabled()) { for (void *p : ptrs) dev_free(p); return; }
    (void)hipDeviceSynchronize();                                   // (the owner's device is current: devpack_finish / msnv_dataset_destroy set it)
    std::vector<CacheBlock> evict; std::vector<void *> foreign;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (void *p : ptrs) {
            if (!p) continue;
            auto it = g_cache_live.find(p);
            if (it == g_cache_live.end()) { foreign.push_back(p); continue; }
            g_cache_free.push_back(it->second); g_cache_free_bytes += it->second.bytes;
            g_cache_live.erase(it);
        }
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t cap = cache_cap_bytes(dev);
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        while (g_cache_free_bytes > cap && !g_cache_free.empty()) {
            size_t big = 0;
            for (size_t i = 1; i < g_cache_free.size(); ++i) if (g_cache_free[i].bytes > g_cache_free[big].bytes) big = i;
            evict.push_back(g_cache_free[big]); g_cache_free_bytes -= g_cache_free[big].bytes;
            g_cache_free[big] = g_cache_free.back(); g_cache_free.pop_back();
        }
    }
    for (const CacheBlock &e : evict) (void)hipFree(e.p);
    for (void *p : foreign) (void)hipFree(p);
}
int dev_upload(void *dst, const void *src, uint64_t bytes) {
    if (!bytes) return MSNV_OK;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MSNV_OK;
}
int dev_download(void *dst, const void *src, uint64_t bytes) { if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return MSNV_OK; }
int dev_copy_bytes(void *dst_device, const void *src, uint64_t bytes, bool src_on_device, void *stream) {
    if (!bytes) return MSNV_OK;
    HIP_TRY(hipMemcpyAsync(dst_device, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MSNV_OK;
}
// hipMemset of device memory is enqueued on the null stream and returns before it has run, and the passes run on a NON-BLOCKING stream
// that the null stream does not order: without the wait below the first kernels of a pass can meet buffers that are not zero yet (fresh
// hipMalloc memory happens to be zero, so nothing showed -- until a profiler's counter pass delayed the fill kernels and the first pass
// faulted on garbage counters, and the guarded allocator's fresh mappings gave wrong counts; tests/test_gpu_guard.py)
int dev_memset(void *dst, int v, uint64_t bytes) {
    if (bytes) { HIP_TRY(hipMemset(dst, v, bytes)); HIP_TRY(hipStreamSynchronize(nullptr)); }
    return MSNV_OK;
}
// ... or on the stream the buffer's first user runs on: ordered before it, nothing to wait for
int dev_memset_async(void *dst, int v, uint64_t bytes, void *stream) {
    if (bytes) HIP_TRY(hipMemsetAsync(dst, v, bytes, (hipStream_t)stream));
    return MSNV_OK;
}
int dev_stream_wait(void *stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return MSNV_OK; }
int dev_stream_create(void **stream) { hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = s; return MSNV_OK; }
void dev_stream_destroy(void *stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

void dev_free_all(DeviceCols &d) {
    void *ptrs[] = {d.hdr, d.hdr8, d.hdr4, d.hdr8m, d.merged_groups, d.tile_pair_merged, d.blk, d.seq, d.qual, d.s_read_base, d.s_seq_base, d.ref4, d.ref_lc, d.pairs,
                    d.tile_pair_start, d.work, d.chunks, d.tile_vbeg, d.tile_vend, d.tot, d.part, d.tile_slot_start, d.tile_slot_u16, d.tile_slot_wide, d.slot_off, d.spill, d.events, d.overflow, d.counters, d.ind4, d.tile_dirty, d.unc_sites, d.site_row, d.gate_tiles,
                    d.sites, d.tile_site_base, d.tile_site_cnt, d.tile_cell_base, d.tile_nslots, d.ncol, d.cov_col, d.site_flags, d.site_elig,
                    d.cov_iv, d.s_cov_base, d.cov_pairs, d.cov_work, d.tile_len, d.tile_contig_dev, d.cov_acc, d.tile_stage, d.tile_stage_idx, d.alt.tile_stage, d.gate_tiles_dense, d.gate_tiles_staged, d.gather_tiles};
    auto in_block = [&](void *p) { for (const auto &b : d.blocks) if ((char *)p >= (char *)b.first && (char *)p < (char *)b.first + b.second) return true; return false; };
    for (void *p : ptrs) if (p && !in_block(p)) dev_free(p);
    void *aptrs[] = {d.ann.seg_beg, d.ann.seg_end, d.ann.seg_gene, d.ann.genes, d.ann.contigs, d.ann.codons, d.ann.out, d.ann.err};
    for (void *p : aptrs) dev_free(p);
    for (void *e : d.timing_events) if (e) (void)hipEventDestroy((hipEvent_t)e);
    for (void *e : d.event_pool) (void)hipEventDestroy((hipEvent_t)e);
    if (d.pinned_cnt) (void)hipHostFree(d.pinned_cnt);
    void *alts[] = {d.aspill, d.alt.aspill, d.alt.tot, d.alt.part, d.alt.spill, d.alt.events, d.alt.overflow, d.alt.counters, d.alt.ind4, d.alt.tile_dirty, d.alt.unc_sites, d.alt.site_row, d.alt.sites, d.alt.tile_site_base,
                    d.alt.tile_site_cnt, d.alt.tile_cell_base, d.alt.ncol, d.alt.cov_col, d.alt.site_flags, d.alt.site_elig, d.alt.site_bits, d.site_bits, d.alt.site_rank, d.site_rank, d.active_tiles};
    for (void *p : alts) if (p && !in_block(p)) dev_free(p);
    for (const auto &b : d.blocks) dev_free(b.first);
    if (d.stream2) (void)hipStreamDestroy((hipStream_t)d.stream2);
    d = DeviceCols{};
}

static int ensure_out(DeviceCols &d, uint64_t n_sites, uint64_t n_cells) {
    if (n_sites > d.cap_out_sites) {
        dev_free(d.site_flags); d.site_flags = nullptr;
        dev_free(d.site_elig); d.site_elig = nullptr;
        const uint64_t cap = std::max<uint64_t>(n_sites + n_sites / 4, 1024);
        if (int rc = dev_alloc((void **)&d.site_flags, cap + 4, &d.device_bytes)) return rc;      // (+4: 32-bit atomics on the byte's aligned word)
        if (int rc = dev_alloc((void **)&d.site_elig, cap + 4, &d.device_bytes)) return rc;
        d.cap_out_sites = cap;
    }
    if (n_cells > d.cap_cells) {
        dev_free(d.ncol); dev_free(d.cov_col);
        d.ncol = nullptr; d.cov_col = nullptr;
        const uint64_t cap = (std::max<uint64_t>(n_cells + n_cells / 4, 1u << 16) + 7) & ~7ull;      // (a multiple of 8: every column starts on 16 bytes)
        if (int rc = dev_alloc((void **)&d.ncol, 4 * cap * sizeof(uint16_t) + 16, &d.device_bytes)) return rc;
        if (int rc = dev_alloc((void **)&d.cov_col, cap * sizeof(uint16_t) + 16, &d.device_bytes)) return rc;
        d.cap_cells = cap;
    }
    return MSNV_OK;
}

// (the list lives behind the record lists, in the same allocation: the two sets of intermediates swap it with them)
static inline uint32_t *stage_ovf_list(const DeviceCols &d) { return d.tile_stage ? reinterpret_cast<uint32_t *>(d.tile_stage + d.n_active_tiles) : nullptr; }

// Enqueues one pass (kernels + readback of the pass' counter block into host_cnt[CNT_WORDS]) without waiting for it.
// ev_begin / ev_pile0 / ev_pile1 are recorded before the pass, before and after the pileup kernel(s); ev3 / ev4 (optional)
// split the tail.  Buffers must have been sized by ensure_out before.
static int enqueue_pass(DeviceCols &d, const msnv_params &p, hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_pile0, hipEvent_t ev_pile1,
                        hipEvent_t ev3, hipEvent_t ev4, uint32_t *host_cnt, hipEvent_t wait_before_pileup = nullptr) {
    const uint64_t npos = (uint64_t)d.n_tiles * TILE;
    if (p.min_baseq != d.qlow_cutoff) return fail(MSNV_EINVAL, "the dataset's quality column was packed for -Q %d, the pass asks for -Q %d", d.qlow_cutoff, p.min_baseq);
    if (ev_begin) HIP_TRY(hipEventRecord(ev_begin, st));
    // nothing to clear: d.tot and the individual-rule bits are zero after finalize and msnv_gate_sites zeroes what a pass has
    // written; the counters live in two blocks that consecutive passes alternate between (the gate kernel zeroes the other one)
    uint32_t *const counters = d.counters + d.cnt_parity * CNT_WORDS, *const counters_next = d.counters + (d.cnt_parity ^ 1u) * CNT_WORDS;
    d.cnt_parity ^= 1u;
    if (wait_before_pileup) HIP_TRY(hipStreamWaitEvent(st, wait_before_pileup, 0));   // the previous pass' pileup kernel (other stream)
    // whole-tile work items apply the gates themselves (fused_tile_gate) unless a pass found a tile with too many candidates (check_counts);
    // with a calling threshold below 1 every covered position is a candidate and the unfused path is the one that is defined for it
    const uint32_t use_stage = (d.n_fused_tiles && !d.fuse_disabled && p.calling_threshold >= 1) ? 1u : 0u;
    HIP_TRY(hipEventRecord(ev_pile0, st));
    if (d.n_work) {
        PileupArgs a;
        a.hdr = d.hdr; a.hdr8 = d.hdr8; a.hdr4 = d.hdr4; a.blk = d.blk; a.seq = d.seq; a.qual = d.qual;
        a.s_read_base = d.s_read_base; a.s_seq_base = d.s_seq_base;
        a.ref4 = d.ref4; a.pairs = d.pairs; a.work = d.work; a.chunks = d.chunks; a.tot = d.tot; a.part = d.part; a.npos = npos; a.spill = d.spill; a.aspill = d.aspill;
        a.events = d.events; a.cap_events = d.cap_events / EV_LISTS; a.ev_count = nullptr; a.overflow = d.overflow; a.cap_overflow = d.cap_overflow;
        a.counters = counters; a.min_baseq = (uint32_t)std::max(0, p.min_baseq);
        a.ind4 = d.ind4; a.unc_bits = d.unc_bits; a.slot_dirty = d.tile_dirty; a.min_snvs = (uint32_t)std::max(0, p.calling_threshold);
        a.n_fused_lo = d.n_work_narrow + d.n_work_merged - (use_stage ? d.n_work_fused : 0u); a.min_cov = p.min_coverage; a.min_frac = p.min_fraction; a.ref_lc = d.ref_lc; a.tile_vbeg = d.tile_vbeg; a.tile_vend = d.tile_vend;
        a.tile_stage = d.tile_stage; a.tile_stage_idx = d.tile_stage_idx; a.stage_ovf = stage_ovf_list(d);
        const uint32_t n_narrow = d.n_work_narrow, n_merged = d.n_work_merged;
        // narrow work items (byte bins), merged groups of shallow pairs and wide items (16-bit bins) touch disjoint (tile, sample) pairs
        a.hdr8m = d.hdr8m; a.n_narrow = n_narrow;
        if (n_narrow && d.dense) hipLaunchKernelGGL(msnv_pileup_tiles_dense, dim3(n_narrow), dim3(N_NT), 0, st, a);      // (the dense layout never merges)
        else {
            // (whole-tile items of a sparse cohort are the last items of the work list: a launch of their own since round 6, msnv_pileup_tiles_lean)
            const bool lean_off = [] { const char *e = getenv("MSNV_LEAN"); return e && e[0] == '0'; }();      // (read per pass: the tests switch it)
            const uint32_t n_lean = (use_stage && !lean_off) ? d.n_work_fused : 0u, n_all = n_narrow + n_merged - n_lean;
            if (n_all && d.allele_planes) hipLaunchKernelGGL(msnv_pileup_tiles_narrow32_planes, dim3(n_all), dim3(N_NT), 0, st, a);
            else if (n_all) hipLaunchKernelGGL(msnv_pileup_tiles_narrow32, dim3(n_all), dim3(N_NT), 0, st, a);
            if (n_lean) hipLaunchKernelGGL(msnv_pileup_tiles_lean, dim3(n_lean), dim3(N_NT), 0, st, a);
        }
        if (d.n_work > n_narrow + n_merged) {
            PileupArgs b = a;
            b.work = d.work + n_narrow + n_merged;
            hipLaunchKernelGGL(msnv_pileup_tiles_wide, dim3(d.n_work - n_narrow - n_merged), dim3(W_NT), 0, st, b);
        }
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ev_pile1, st));
    const uint32_t cap_out = (uint32_t)std::min<uint64_t>(d.cap_out_sites, 0xffffffffull);
    const bool need_decide = d.any_split;                           // some sample's reads sit in several pairs of a tile: sites whose call depends
                                                                    // on its summed counts are decided behind the scatter (msnv_decide_sites);
                                                                    // merged groups alone are handled inside the merged gather
    if (d.n_active_tiles) {
        GateArgs g;
        g.tot = d.tot; g.part = d.part; g.slot_off = d.slot_off; g.tile_slot_start = d.tile_slot_start; g.tile_slot_u16 = d.tile_slot_u16; g.tile_slot_wide = d.tile_slot_wide;
        g.npos = npos; g.tile_vbeg = d.tile_vbeg; g.tile_vend = d.tile_vend; g.min_cov = p.min_coverage; g.min_snvs = p.calling_threshold; g.min_frac = p.min_fraction;
        g.ind4 = d.ind4; g.unc_bits = d.unc_bits; g.ref4 = d.ref4; g.ref_lc = d.ref_lc;
        g.site_bits = d.site_bits; g.site_rank = d.site_rank; g.sites = d.sites; g.cap_sites = d.cap_sites; g.counters = counters; g.counters_next = counters_next;
        g.tile_site_base = d.tile_site_base; g.tile_site_cnt = d.tile_site_cnt; g.active_tiles = d.active_tiles;
        g.ncol = d.ncol; g.cov_col = d.cov_col; g.site_flags = d.site_flags; g.cap_out = cap_out;
        g.gate_tiles = reinterpret_cast<const GateTile *>(d.gate_tiles); g.tile_dirty = d.tile_dirty; g.unc_sites = d.unc_sites;
        g.use_dirty = d.use_dirty ? 1u : 0u; g.block_row = d.site_row; g.site_elig = d.site_elig; g.any_split = d.any_split ? 1u : 0u;
        static_assert(sizeof(GateTile) == sizeof(DeviceCols::GateTileH) && sizeof(GateTile) == 64, "gate tile descriptor");
        g.tile_nslots = d.tile_nslots; g.tile_cell_base = d.tile_cell_base; g.cap_cells = d.cap_cells;
        // several tiles per workgroup once the tiles outnumber what the device holds at a time several times over (one reservation of
        // site slots per workgroup: msnv_gate_sites); MSNV_GATE_TILES overrides (tests run every size)
        g.tile_stage = d.tile_stage; g.tiles_per_wg = 1u;
        // whole-tile work items left record lists: their tiles go through msnv_gate_staged, the others through msnv_gate_sites
        const uint32_t n_staged = use_stage ? d.n_fused_tiles : 0u, n_dense = d.n_active_tiles - n_staged;
        if (use_stage) g.gate_tiles = reinterpret_cast<const GateTile *>(d.gate_tiles_dense);
        g.n_active = n_dense;
        g.zero_next = 1u; g.tile_list = nullptr; g.solo_cells = 0u;
        if (n_dense) {
            g.tiles_per_wg = n_dense >= 32768u ? 8u : n_dense >= 8192u ? 4u : 1u;
            if (const char *e = getenv("MSNV_GATE_TILES")) g.tiles_per_wg = (uint32_t)std::min<int>((int)GATE_MAX_TILES, std::max(1, atoi(e)));
            const dim3 grid((n_dense + g.tiles_per_wg - 1) / g.tiles_per_wg);
            g.aspill = d.aspill;
            if (d.allele_planes) {
                if (g.tiles_per_wg == 1u) {
                    if (d.wide_tot) hipLaunchKernelGGL((msnv_gate_sites<false, true, true>), grid, dim3(GATE_NT), 0, st, g);
                    else hipLaunchKernelGGL((msnv_gate_sites<false, false, true>), grid, dim3(GATE_NT), 0, st, g);
                } else {
                    if (d.wide_tot) hipLaunchKernelGGL((msnv_gate_sites<true, true, true>), grid, dim3(GATE_NT), 0, st, g);
                    else hipLaunchKernelGGL((msnv_gate_sites<true, false, true>), grid, dim3(GATE_NT), 0, st, g);
                }
            } else if (g.tiles_per_wg == 1u) {
                if (d.wide_tot) hipLaunchKernelGGL((msnv_gate_sites<false, true>), grid, dim3(GATE_NT), 0, st, g);
                else hipLaunchKernelGGL((msnv_gate_sites<false, false>), grid, dim3(GATE_NT), 0, st, g);
            } else {
                if (d.wide_tot) hipLaunchKernelGGL((msnv_gate_sites<true, true>), grid, dim3(GATE_NT), 0, st, g);
                else hipLaunchKernelGGL((msnv_gate_sites<true, false>), grid, dim3(GATE_NT), 0, st, g);
            }
            g.zero_next = 0u;
        }
        if (n_staged) {
            hipLaunchKernelGGL(msnv_gate_staged, dim3((n_staged + GS_WAVES * GS_TILES - 1) / (GS_WAVES * GS_TILES)), dim3(64 * GS_WAVES), 0, st, g, reinterpret_cast<const GateTile *>(d.gate_tiles_staged), n_staged);
            // the tiles whose candidates did not fit their record list (counted and listed on the device: fused_tile_gate) through the ordinary
            // gate; the workgroups stride over the list -- as many as the previous pass would have kept busy, a handful when it listed none
            g.gate_tiles = reinterpret_cast<const GateTile *>(d.gate_tiles); g.tile_list = stage_ovf_list(d); g.n_active = n_staged; g.solo_cells = 1u;
            g.tiles_per_wg = 1u; g.zero_next = 0u; g.aspill = d.aspill;
            const dim3 grid(std::min<uint32_t>(n_staged, std::max<uint32_t>(64u, std::min<uint32_t>(4096u, d.last_ovf_tiles))));
            if (d.allele_planes) {
                if (d.wide_tot) hipLaunchKernelGGL((msnv_gate_sites<true, true, true>), grid, dim3(GATE_NT), 0, st, g);
                else hipLaunchKernelGGL((msnv_gate_sites<true, false, true>), grid, dim3(GATE_NT), 0, st, g);
            } else {
                if (d.wide_tot) hipLaunchKernelGGL((msnv_gate_sites<true, true>), grid, dim3(GATE_NT), 0, st, g);
                else hipLaunchKernelGGL((msnv_gate_sites<true, false>), grid, dim3(GATE_NT), 0, st, g);
            }
        }
        HIP_TRY(hipGetLastError());
    } else HIP_TRY(hipMemsetAsync(counters_next, 0, CNT_WORDS * sizeof(uint32_t), st));   // nobody else would
    if (ev3) HIP_TRY(hipEventRecord(ev3, st));
    // the tail runs on device-side counts: no host round trip inside a pass
    if (d.n_active_tiles) {
        TailArgs ta;
        ta.sites = d.sites; ta.tile_site_base = d.tile_site_base; ta.tile_site_cnt = d.tile_site_cnt; ta.tile_pair_start = d.tile_pair_start;
        ta.pairs = d.pairs; ta.spill = d.spill; ta.ncol = d.ncol; ta.cov_col = d.cov_col; ta.cap_out = cap_out; ta.active_tiles = d.gather_tiles;
        ta.cells = CellMap{d.tile_site_base, d.tile_cell_base, d.tile_nslots, d.cap_cells, d.site_row};
        ta.gather_split = d.gather_split;
        ta.has_wide = d.n_work > d.n_work_narrow + d.n_work_merged ? 1u : 0u;
        ta.aspill = d.allele_planes ? d.aspill : nullptr;
        static const uint32_t tail_skip = [] { const char *e = getenv("MSNV_TAIL_SKIP"); return e ? (uint32_t)atoi(e) : 0u; }();
        ta.debug_skip = tail_skip;
        ta.n_gather_blocks = d.n_gather_tiles * d.gather_split;
        ta.tile_pair_merged = d.tile_pair_merged; ta.merged_groups = d.merged_groups; ta.chunks = d.chunks; ta.hdr8m = d.hdr8m;
        ta.seq = d.seq; ta.qual = d.qual; ta.ref4 = d.ref4; ta.n_merged_run = d.n_merged_groups - (use_stage ? d.n_groups_solo : 0u);
        ta.merged_per_block = merged_wave_form(d) ? GMW_GROUPS : 1u; ta.n_merged_blocks = (ta.n_merged_run + ta.merged_per_block - 1u) / ta.merged_per_block; ta.min_baseq = (uint32_t)std::max(0, p.min_baseq);
        ta.site_flags = d.site_flags; ta.site_elig = d.site_elig; ta.ind_in_gather = d.any_split ? 0u : 1u; ta.min_snvs = (uint32_t)std::max(1, p.calling_threshold);
        ta.events = d.events; ta.overflow = d.overflow; ta.counters = counters; ta.cap_list = d.cap_events / EV_LISTS; ta.cap_overflow = d.cap_overflow;
        ta.site_bits = d.site_bits; ta.site_rank = d.site_rank;
        static const uint32_t scatter_blocks = [] { const char *e = getenv("MSNV_SCATTER_BLOCKS"); return e ? (uint32_t)std::max(1, atoi(e)) : SCATTER_BLOCKS_PER_LIST; }();
        ta.scatter_blocks = scatter_blocks;
        const dim3 grid(ta.n_gather_blocks + ta.n_merged_blocks + ta.scatter_blocks * EV_LISTS + 1u);      // + 1: the event total
        if (ta.n_merged_blocks) hipLaunchKernelGGL(msnv_gather_scatter<true>, grid, dim3(256), 0, st, ta);
        else hipLaunchKernelGGL(msnv_gather_scatter<false>, grid, dim3(256), 0, st, ta);
        HIP_TRY(hipGetLastError());
    }
    if (ev4) HIP_TRY(hipEventRecord(ev4, st));
    if (need_decide && d.n_active_tiles) {
        // (one wavefront per listed site, grid-stride: a launch that fills the chip -- with 256 workgroups half a million listed sites
        // of a deep, uneven cohort took 1.6 ms)
        static const uint32_t decide_grid = dev_resident_workgroups(8);          // (queried once: the device properties call is slow)
        hipLaunchKernelGGL(msnv_decide_sites, dim3(decide_grid), dim3(256), 0, st, d.sites, d.unc_sites, counters, d.cap_sites, cap_out, d.ref4, d.ref_lc,
                           d.ncol, CellMap{d.tile_site_base, d.tile_cell_base, d.tile_nslots, d.cap_cells, d.site_row}, p.calling_threshold, p.min_fraction, d.site_flags);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(host_cnt, counters, CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    return MSNV_OK;
}

static int check_counts(DeviceCols &d, const uint32_t *cnt, RunCounts *counts) {
    RunCounts c{cnt[0], cnt[1], cnt[2], cnt[3], (uint64_t)cnt[CNT_CELLS] | (uint64_t)cnt[CNT_CELLS + 1] << 32};
    d.last_sites = c.n_sites; d.last_cells = c.n_cells;
    if (counts) *counts = c;
    d.last_ovf_tiles = cnt[CNT_STAGE];                         // whole-tile work items whose candidates did not fit a record list (their tiles went through msnv_gate_sites)
    if (c.n_events > d.cap_events / EV_LISTS * EV_LISTS || c.n_overflow > d.cap_overflow || c.n_sites > d.cap_sites || c.n_sites > d.cap_out_sites || c.n_cells > d.cap_cells)
        return fail_quiet(MSNV_ECAPACITY, "device buffer too small: events %u/%u overflow %u/%u sites %u/%u out %u/%llu cells %llu/%llu",
                          c.n_events, d.cap_events, c.n_overflow, d.cap_overflow, c.n_sites, d.cap_sites, c.n_sites,
                          (unsigned long long)d.cap_out_sites, (unsigned long long)c.n_cells, (unsigned long long)d.cap_cells);
    return MSNV_OK;
}

int dev_run_pipeline(DeviceCols &d, const msnv_params &p, void *stream_, msnv_run_stats *stats, RunCounts *counts) {
    hipStream_t st = (hipStream_t)stream_;
    hipEvent_t ev[6];                                      // created once per dataset: event create / destroy costs host time in every pass
    for (int i = 0; i < 6; ++i) {
        if (!d.timing_events[i]) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); d.timing_events[i] = e; }
        ev[i] = (hipEvent_t)d.timing_events[i];
    }
    // every event record costs ~6 us of stream time (the next kernel waits for the marker): the per-phase split of
    // the tail is only recorded on request (MSNV_PHASE_TIMES=1, profiles/phase_times.py)
    static const bool phase_times = [] { const char *e = getenv("MSNV_PHASE_TIMES"); return e && e[0] == '1'; }();
    if (int rc = ensure_out(d, std::max<uint64_t>(d.last_sites + d.last_sites / 2, 4096), std::max<uint64_t>(d.last_cells + d.last_cells / 2, 1u << 18))) return rc;
    uint32_t cnt[CNT_WORDS] = {0};
    if (int rc = enqueue_pass(d, p, st, ev[0], ev[1], ev[2], phase_times ? ev[3] : nullptr, phase_times ? ev[4] : nullptr, cnt)) return rc;
    HIP_TRY(hipEventRecord(ev[5], st));
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = check_counts(d, cnt, counts)) return rc;
    if (stats) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[5])); stats->ms_total = ms;
        HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2])); stats->ms_pileup = ms;
        if (phase_times) {
            HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); stats->ms_gate = ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev[3], ev[4])); stats->ms_gather = ms;
            HIP_TRY(hipEventElapsedTime(&ms, ev[4], ev[5])); stats->ms_decide = ms;
        }
        stats->n_sites = cnt[2]; stats->n_events = cnt[0]; stats->n_overflow = cnt[1];
        stats->n_called_pop = cnt[4] + cnt[CNT_TALLY]; stats->n_called_indiv = cnt[5] + cnt[CNT_TALLY + 1];    // msnv_decide_sites' lines + the gate kernel's (and the merged gather's)
        stats->algorithmic_bytes = d.algorithmic_bytes;
    }
    return MSNV_OK;
}

// Second set of per-pass intermediates, sized like the first (re-made when the first set was grown).
static int ensure_alt(DeviceCols &d) {
    DeviceCols::AltBufs &a = d.alt;
    const uint64_t npos = (uint64_t)d.n_tiles * TILE;
    if (a.tot && a.cap_events == d.cap_events && a.cap_overflow == d.cap_overflow && a.cap_sites == d.cap_sites && a.cap_out_sites == d.cap_out_sites && a.cap_cells == d.cap_cells) return MSNV_OK;
    void *old[] = {a.aspill, a.tot, a.part, a.spill, a.events, a.overflow, a.counters, a.ind4, a.tile_dirty, a.unc_sites, a.site_row, a.sites, a.tile_site_base, a.tile_site_cnt, a.tile_cell_base, a.ncol, a.cov_col, a.site_flags, a.site_elig, a.site_bits, a.site_rank, a.tile_stage};
    for (void *p : old) dev_free(p);
    a = DeviceCols::AltBufs{};
    if (int rc = dev_alloc((void **)&a.tot, std::max<uint64_t>(1, 4 * npos) * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_memset(a.tot, 0, std::max<uint64_t>(1, 4 * npos) * sizeof(uint32_t))) return rc;
    if (int rc = dev_alloc((void **)&a.part, d.part_bytes, &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.spill, std::max<uint64_t>(1, d.n_pairs) * TILE, &d.device_bytes)) return rc;
    if (d.allele_planes) {
        if (int rc = dev_alloc((void **)&a.aspill, (uint64_t)d.n_pairs * 4 * TILE, &d.device_bytes)) return rc;
        if (int rc = dev_memset(a.aspill, 0, (uint64_t)d.n_pairs * 4 * TILE)) return rc;
    }
    if (int rc = dev_alloc((void **)&a.events, (uint64_t)d.cap_events * sizeof(Pair32), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.overflow, (uint64_t)d.cap_overflow * sizeof(Pair32), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.sites, (uint64_t)d.cap_sites * sizeof(SiteRec), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.unc_sites, (uint64_t)d.cap_sites * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.site_row, (npos / 64 + 1) * sizeof(unsigned long long), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.tile_dirty, ((uint64_t)d.n_work + 1) * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_memset(a.tile_dirty, 0, ((uint64_t)d.n_work + 1) * sizeof(uint32_t))) return rc;
    if (int rc = dev_alloc((void **)&a.counters, 2 * CNT_WORDS * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_memset(a.counters, 0, 2 * CNT_WORDS * sizeof(uint32_t))) return rc;
    if (int rc = dev_alloc((void **)&a.ind4, (npos / 8 + npos / 32 + 2) * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_memset(a.ind4, 0, (npos / 8 + npos / 32 + 2) * sizeof(uint32_t))) return rc;
    a.unc_bits = a.ind4 + npos / 8 + 1; a.cnt_parity = 0;
    if (int rc = dev_alloc((void **)&a.site_bits, (npos / 64 + 1) * sizeof(unsigned long long), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.site_rank, (npos / 64 + 1) * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.tile_site_base, ((uint64_t)d.n_tiles + 1) * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.tile_site_cnt, ((uint64_t)d.n_tiles + 1) * sizeof(uint32_t), &d.device_bytes)) return rc;
    if (int rc = dev_memset(a.tile_site_cnt, 0, ((uint64_t)d.n_tiles + 1) * sizeof(uint32_t))) return rc;
    if (int rc = dev_memset(a.tile_site_base, 0, ((uint64_t)d.n_tiles + 1) * sizeof(uint32_t))) return rc;
    if (int rc = dev_alloc((void **)&a.tile_cell_base, ((uint64_t)d.n_tiles + 1) * sizeof(unsigned long long), &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.ncol, 4 * d.cap_cells * sizeof(uint16_t) + 16, &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.cov_col, d.cap_cells * sizeof(uint16_t) + 16, &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.site_flags, d.cap_out_sites + 4, &d.device_bytes)) return rc;
    if (int rc = dev_alloc((void **)&a.site_elig, d.cap_out_sites + 4, &d.device_bytes)) return rc;
    if (d.n_fused_tiles && !a.tile_stage) {
        if (int rc = dev_alloc((void **)&a.tile_stage, (uint64_t)d.n_active_tiles * sizeof(TileStage) + (uint64_t)d.n_fused_tiles * sizeof(uint32_t), &d.device_bytes)) return rc;   // (+ the overflow list: stage_ovf_list)
        if (int rc = dev_memset(a.tile_stage, 0, (uint64_t)d.n_active_tiles * sizeof(TileStage))) return rc;
    }
    a.cap_events = d.cap_events; a.cap_overflow = d.cap_overflow; a.cap_sites = d.cap_sites; a.cap_out_sites = d.cap_out_sites; a.cap_cells = d.cap_cells;
    return MSNV_OK;
}
static void swap_sets(DeviceCols &d) {
    DeviceCols::AltBufs &a = d.alt;
    std::swap(d.tot, a.tot); std::swap(d.part, a.part); std::swap(d.spill, a.spill); std::swap(d.events, a.events);
    if (a.aspill) std::swap(d.aspill, a.aspill);
    std::swap(d.overflow, a.overflow); std::swap(d.counters, a.counters); std::swap(d.sites, a.sites);
    std::swap(d.tile_site_base, a.tile_site_base); std::swap(d.tile_site_cnt, a.tile_site_cnt); std::swap(d.tile_cell_base, a.tile_cell_base); std::swap(d.ncol, a.ncol); std::swap(d.cov_col, a.cov_col);
    std::swap(d.site_flags, a.site_flags); std::swap(d.site_elig, a.site_elig); std::swap(d.ind4, a.ind4); std::swap(d.unc_bits, a.unc_bits); std::swap(d.cnt_parity, a.cnt_parity);
    if (a.tile_stage) std::swap(d.tile_stage, a.tile_stage);
    std::swap(d.tile_dirty, a.tile_dirty); std::swap(d.unc_sites, a.unc_sites); std::swap(d.site_row, a.site_row); std::swap(d.site_bits, a.site_bits); std::swap(d.site_rank, a.site_rank);
}

// n passes, ONE host synchronisation at the end; with `overlap` they are in flight on two streams (a queue of shards / repeated passes keeps the
// GPU busy: the host round trip of the single-pass form costs ~40 us of idle GPU per pass).  Consecutive passes use
// alternating sets of intermediates and alternating streams; the pileup kernels are serialised among themselves by an
// event (they saturate the chip), so what overlaps is the latency-bound tail of pass i (gate, gather, scatter, decide,
// ~0.12 ms at a fraction of the chip) with the memsets and the pileup kernel of pass i+1.  Measured (one box, 30 passes):
// step time 0.810 -> 0.770 ms (+5 % bases/s), while the pileup kernel itself reads 0.673 -> 0.736 ms because it shares
// the chip with the tail kernels -- which is why bench.py measures the roofline on the non-overlapped form.  stats[i]: pileup-kernel time
// of pass i from its own event pair (recorded after the cross-stream wait); ms_total = batch time / n.  On return the
// primary set holds the last pass.
// events and the pinned counter blocks of a batch of n passes are pooled in the dataset: creating ~4n events and a pinned buffer per call
// costs about a millisecond of host time before the first pass is even enqueued (msnv_pileup_reserve does it ahead of a timed batch)
int dev_reserve_passes(DeviceCols &d, int n) {
    while (d.event_pool.size() < (size_t)4 * n + 2) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); d.event_pool.push_back(e); }
    if (d.pinned_cnt_cap < (size_t)n) {
        if (d.pinned_cnt) (void)hipHostFree(d.pinned_cnt);
        d.pinned_cnt = nullptr; d.pinned_cnt_cap = 0;
        if (hipHostMalloc((void **)&d.pinned_cnt, (size_t)n * CNT_WORDS * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return fail(MSNV_ENOMEM, "pinned host memory for %d counter blocks", n);
        d.pinned_cnt_cap = (size_t)n;
    }
    return MSNV_OK;
}

int dev_run_pipeline_many(DeviceCols &d, const msnv_params &p, void *stream_, int n, bool overlap, msnv_run_stats *stats, RunCounts *counts) {
    if (n <= 0) return MSNV_OK;
    hipStream_t s0 = (hipStream_t)stream_, s1 = s0;
    if (int rc = ensure_out(d, std::max<uint64_t>(d.last_sites + d.last_sites / 2, 4096), std::max<uint64_t>(d.last_cells + d.last_cells / 2, 1u << 18))) return rc;
    if (overlap && n > 1) {
        if (!d.stream2) { hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); d.stream2 = s; }
        s1 = (hipStream_t)d.stream2;
        if (int rc = ensure_alt(d)) return rc;
    }
    const bool two = s1 != s0;
    if (int rc = dev_reserve_passes(d, n)) return rc;
    hipEvent_t *ev = reinterpret_cast<hipEvent_t *>(d.event_pool.data());
    auto cleanup = [] {};
    uint32_t *cnt = d.pinned_cnt;
    int rc = MSNV_OK;
    hipError_t he = hipSuccess;
    bool swapped = false;                                     // true while d's primary fields hold the second set
    for (int i = 0; i < n && !rc && he == hipSuccess; ++i) {
        hipStream_t st = (i & 1) ? s1 : s0;
        if (two && (i & 1) != (swapped ? 1 : 0)) { swap_sets(d); swapped = !swapped; }
        // the pileup kernel of pass i starts after the one of pass i-1 (other stream) has finished; its memsets do not wait
        // (only the first pass records a begin event: every event record costs stream time)
        rc = enqueue_pass(d, p, st, i == 0 ? ev[0] : nullptr, ev[4 * i + 1], ev[4 * i + 2], nullptr, nullptr, cnt + (size_t)CNT_WORDS * i, (two && i > 0) ? ev[4 * (i - 1) + 2] : nullptr);
    }
    if (!rc && he == hipSuccess) he = hipEventRecord(ev[4 * n], s0);
    if (!rc && he == hipSuccess) he = hipEventRecord(ev[4 * n + 1], s1);
    if (he == hipSuccess) he = hipStreamSynchronize(s0);
    if (he == hipSuccess && two) he = hipStreamSynchronize(s1);
    // leave the set of the LAST pass in the primary fields (results, annotation and filters read them)
    if (two && swapped != (((n - 1) & 1) != 0)) { swap_sets(d); swapped = !swapped; }
    if (!rc && he != hipSuccess) rc = fail(MSNV_EHIP, "batched passes: %s", hipGetErrorString(he));
    float total = 0, t1 = 0;
    if (!rc) {
        (void)hipEventElapsedTime(&total, ev[0], ev[4 * n]);
        if (two && hipEventElapsedTime(&t1, ev[0], ev[4 * n + 1]) == hipSuccess) total = std::max(total, t1);
    }
    for (int i = 0; i < n && !rc; ++i) {
        rc = check_counts(d, cnt + (size_t)CNT_WORDS * i, counts);
        if (rc || !stats) continue;
        float ms = 0;
        msnv_run_stats &s = stats[i];
        s = msnv_run_stats{};
        s.ms_total = total / (float)n;
        if (hipEventElapsedTime(&ms, ev[4 * i + 1], ev[4 * i + 2]) == hipSuccess) s.ms_pileup = ms;
        const uint32_t *c = cnt + (size_t)CNT_WORDS * i;
        s.n_sites = c[2]; s.n_events = c[0]; s.n_overflow = c[1]; s.n_called_pop = c[4] + c[CNT_TALLY]; s.n_called_indiv = c[5] + c[CNT_TALLY + 1];
        s.algorithmic_bytes = d.algorithmic_bytes;
    }
    cleanup();
    return rc;
}

int dev_run_coverage(DeviceCols &d, int max_cov, void *stream_, msnv_run_stats *stats) {
    hipStream_t st = (hipStream_t)stream_;
    if (max_cov < 1 || max_cov >= COV_BINS) return fail(MSNV_EINVAL, "coverage histogram cutoff must be in [1, %d]", COV_BINS - 1);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    HIP_TRY(hipMemsetAsync(d.cov_acc, 0, (uint64_t)d.cov_copies * std::max<uint64_t>(1, d.n_cov_rows) * (1 + COV_BINS) * sizeof(unsigned long long), st));
    if (d.n_cov_work) {
        // work items that hold a pair of more than 32 767 intervals are the last n_cov_work_wide of the list (pack.cpp)
        const uint32_t n_narrow = d.n_cov_work - d.n_cov_work_wide;
        if (n_narrow) hipLaunchKernelGGL(msnv_coverage_tiles<false>, dim3(n_narrow), dim3(C_NT), 0, st, d.cov_iv, d.cov_pairs, d.cov_work,
                                         d.tile_len, d.cov_acc, (uint32_t)d.n_cov_rows, max_cov, d.cov_copies, d.n_cov_iv);
        if (d.n_cov_work_wide) hipLaunchKernelGGL(msnv_coverage_tiles<true>, dim3(d.n_cov_work_wide), dim3(C_NT), 0, st, d.cov_iv, d.cov_pairs, d.cov_work + n_narrow,
                                                  d.tile_len, d.cov_acc, (uint32_t)d.n_cov_rows, max_cov, d.cov_copies, d.n_cov_iv);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (stats) stats->ms_coverage = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return MSNV_OK;
}

// The runtime loads a translation unit's code object when its first kernel is launched (~10 ms): msnv_ctx_create does that here, on the
// thread that brings the context up, instead of inside the first timed stage.
__global__ void msnv_warm_kernels() {}
void warm_kernels(void *stream) { hipLaunchKernelGGL(msnv_warm_kernels, dim3(1), dim3(1), 0, (hipStream_t)stream); (void)hipGetLastError(); }

}  // namespace msnv

extern "C" int msnv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
