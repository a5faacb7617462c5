// metasnv_amd/csrc/dataset.h -- packed per-read columns ("read table") and their HBM layout.
//
// Coordinate system.  The contigs of a shard are laid side by side in one linear "gpos"
// space; every contig starts on a TILE boundary and owns ceil(max(L, furthest read end)/TILE)
// tiles, so a tile never straddles two contigs and a read never leaves its contig's tiles.
// gpos = tile_base[contig] * TILE + pos.   All device kernels work in gpos; the host maps
// results back to (contig, pos).
//
// Per sample, in BAM (coordinate) order, concatenated over samples:
//   hdr[]   16 B  {gpos, seqoff, cig, meta}       one per stored read
//   cig[]    4 B  BAM-encoded ops                  only for reads with n_cigar != 1
//   seq[]   4 bit nt16 codes, LOW nibble first     (host swaps BAM's high-nibble-first order)
//   qual[]   1 B  phred                            qual offset = 2 * seqoff
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "msnv_internal.h"

namespace msnv {

constexpr uint32_t TILE = 2048;          // reference positions per tile (LDS bins per workgroup)

// meta: bits 0-15 n_cigar | 16-23 mapq | 24 pileup_ok | 25 cov_ok | 26 fast
constexpr uint32_t META_PILEUP_OK = 1u << 24;
constexpr uint32_t META_COV_OK    = 1u << 25;
// Piece alignment in the seq column (the quality column is aligned to twice that): 2 bytes = 4 bases, so a 100-base piece carries no
// padding at all (8-byte starts: 6 + 12 B per piece = 0.27 GB of the benchmark launch's 3.04 GB).  The wide loads are then not
// naturally aligned; same-box A/B, pileup kernel (profiles/r03b_ab.txt): 8 B 0.573 ms, 4 B 0.548, 2 B 0.547.
// MSNV_SEQ_ALIGN_LOG2 (build-time, profiles/build_variant.sh) sets 4- or 8-byte piece starts.
#ifndef MSNV_SEQ_ALIGN_LOG2
#define MSNV_SEQ_ALIGN_LOG2 1
#endif
constexpr uint32_t SEQ_ALIGN_LOG2 = MSNV_SEQ_ALIGN_LOG2;
constexpr uint32_t SEQ_ALIGN = 1u << SEQ_ALIGN_LOG2;
static_assert(SEQ_ALIGN_LOG2 >= 1 && SEQ_ALIGN_LOG2 <= 3, "piece starts on 2, 4 or 8 bytes of the seq column");
// Piece headers of the ordinary narrow work items: 4 bytes -- start in the tile (11 bits) | length (8) | seq offset relative to the
// CHUNK's first byte in SEQ_ALIGN units (13 bits: a chunk spans < 16 KB of the seq column, pack.cpp closes it early otherwise); the
// chunk descriptor carries the absolute base (uniform per chunk: scalar registers, the loads take base + 32-bit lane offset).
// MSNV_HDR4=0 (build-time) keeps the 8-byte form {start | length << 11, offset in the sample / SEQ_ALIGN} for A/B runs.
#ifndef MSNV_HDR4
#define MSNV_HDR4 1
#endif
constexpr bool HDR4 = MSNV_HDR4 != 0;
constexpr uint32_t HDR4_OFF_BITS = 13;
constexpr uint32_t SEG_MAX = 128;            // bases per segment piece = 8 lanes x 16 bases
constexpr uint32_t NARROW_MAX_DEPTH = 255;   // (tile, sample) pairs below this depth use byte-wide LDS bins

struct ReadHdr { uint32_t gpos, seqoff, cig, meta; };
struct PieceHdr { uint32_t w0, seqoff8; };      // seqoff8: seq byte offset / SEQ_ALIGN
// Dense layout: the pieces of one (sample, tile) pair are packed back to back (each starts on an even base) into a
// stream of 32-base blocks; a block holds the tail / middle of one piece (segment A, bits [0, nA)) and at most the
// head of the next one (segment B, bits [sB, 32), sB = nA rounded up to even, always running to the block end).
//   bits  0-10  P0A   tile-relative position of bit 0 of segment A
//   bits 11-16  nA    bases of segment A (0..32)
//   bits 17-28  PBv   (position of B's first base) - sB + 32, or BLK_NO_B
//   bit  29/30/31     segment A starts a piece here / segment A ends its piece here / segment B ends its piece here
constexpr uint32_t BLK_NO_B = 0xfffu, BLK_START_A = 1u << 29, BLK_END_A = 1u << 30, BLK_END_B = 1u << 31;
constexpr uint32_t BLK_EMPTY = BLK_NO_B << 17;
constexpr int DENSE_CHUNK_BLOCKS = 512;   // blocks per chunk descriptor (two rounds of 256 lanes)                             // narrow kernel: start in tile | length << 11; seq byte offset / 8

struct TilePair { uint32_t sample, read_lo, read_hi, max_depth, blk_lo, nblk, seq0, pad; };    // reads of `sample` that may overlap the tile;
                                                                        // max_depth = upper bound of the per-position depth
// One chunk = up to CHUNK_READS consecutive reads of one (tile, sample) pair, with everything the
// kernel needs to start loading (no dependent scalar loads on the critical path).
// Whole-tile work items (sparse cohorts, pack.cpp): ONE merged group holds every pair of the tile, so the workgroup that piles it up
// has the tile's coverage and allele totals in its LDS bins and applies snpCall's gates and calling rule itself (kernels.hip:
// fused_tile_gate); what the gate kernel then reads of such a tile is this record list instead of ~19 KB of per-position state.
constexpr uint32_t STAGE_CAP = 24;           // candidate positions of one tile kept here; a tile with more sends the dataset back to the unfused path
struct StageRec { uint32_t pos_flags, cov, nword, pad; };      // position in the tile | "ask the per-sample records" << 11 | (pop | ind << 4) << 16 | eligible alleles << 24;
                                                               // coverage; mismatching A, C, G, T totals, one byte each (< 256: the group's depth bound)
struct TileStage { uint32_t count, pad[3]; StageRec rec[STAGE_CAP]; };
constexpr uint32_t WORK_FUSED = 8u;          // WorkItem::part_lo bit: whole-tile item (bit 0: u8 partial row, bits 1-2: allele-total mode)
constexpr uint32_t MERGE_MAX_PAIRS = 256;    // pairs per merged group of shallow (sample, tile) pairs (pack.cpp; kernels.hip: msnv_pileup_tiles_merged)
constexpr uint32_t MERGE_MAX_DEPTH = 240;    // their depth bounds add up to at most this (byte bins)
constexpr int COV_PW = 4;                    // (tile, sample) pairs per wavefront of msnv_coverage_tiles
constexpr uint32_t COV_ITEM_PAIRS = 4;       // pairs per coverage work item (one wavefront x COV_PW)
constexpr uint32_t CHUNK_READS = 128;
constexpr uint32_t MAX_CHUNKS_PER_ITEM = 32;
struct ChunkDesc { uint64_t hdr_base, seq_base; uint32_t sample, pair, nrd_flags, pad; };   // nrd | last_chunk << 16
// first: the item's first chunk descriptor, so that a workgroup can fetch its first headers without waiting for the
// descriptor stream (one dependent load less at start-up: what a sparse cohort's one-chunk work items are made of)
struct WorkItem { uint32_t tile, pair_lo, pair_hi, chunk_lo, chunk_hi, slot, part_lo, part_hi; ChunkDesc first; };   // slot: row of the coverage partials (tile-major); part_lo/hi: byte offset of the row
static_assert(sizeof(WorkItem) == 64, "work items are loaded as four 16-byte words");

struct SiteRec { uint32_t gpos, cov, n[4]; };                          // gate kernel output (24 B)

// What a round of the device pack leaves in HBM besides the columns: its piece headers (tile order, positions still contig-relative) and
// qaCompute intervals, sample after sample.  finalize builds the tile index from them on the device (devpack.hip: devfin_*); the host sees
// one DevPair per (sample, contig, tile) run of pieces.
struct DevRound { void *buf = nullptr; ReadHdr *hdr = nullptr; int32_t *tid = nullptr, *end = nullptr; uint16_t *depth = nullptr;
                  int32_t *cov_tid = nullptr, *cov_beg = nullptr, *cov_end = nullptr; uint64_t n_pieces = 0, n_iv = 0; size_t first_sample = 0;
                  // the round's columns as devpack_add_round laid them out: sample after sample exactly as finalize lays the dataset's (every
                  // sample's share rounded up to 16 bytes), seq_total bytes of bases, COL_PAD bytes of N, then seq_total / 4 (+ 64) bytes of flags
                  void *col_buf = nullptr; uint8_t *col_seq = nullptr, *col_qual = nullptr; uint64_t seq_total = 0; size_t n_samples = 0; };
constexpr uint64_t COL_PAD = 256;          // readable bytes of N behind a seq column (lanes past the last piece read on)
struct DevPair { int32_t tid; uint32_t tile, lo, hi, maxd, grp; };      // tile: inside the contig; [lo, hi): pieces of the sample; maxd: bound of the per-position depth; grp: 0, or 1 + group of a deep run dealt into groups

// host staging of one sample
struct SampleCols {
    std::vector<ReadHdr>  hdr;
    std::vector<int32_t>  tid;       // per read
    std::vector<int32_t>  end;       // per read: contig-relative end of everything the kernels may touch
    std::vector<uint16_t> depth;     // per read: pileup reads alive when this one starts (saturating)
    std::vector<uint16_t> grp;       // per piece: 0, or 1 + group of a deep (contig, tile) run that was split (pack.cpp: split_deep_runs)
    std::vector<uint8_t>  seq, qual;
    // dense layout (pack.cpp: relayout_dense): per (contig, tile) run of pieces a stream of 32-base blocks
    std::vector<uint32_t> blk;            // one descriptor per block (BLK_* fields)
    std::vector<uint32_t> run_blk_lo, run_nblk, run_seq0;   // per run, in run order: first block, block count, seq byte offset
    std::vector<int32_t>  cov_tid, cov_beg, cov_end;   // qaCompute M intervals (index space), reads that pass its filter
    uint64_t n_pileup_bases = 0, n_pileup_reads = 0;
    uint64_t mm_sampled_bases = 0, mm_sampled = 0;   // every 16th piece: aligned bases compared with the reference / how many differ (finalize picks the allele bookkeeping by it)
    uint64_t alg_seq_bytes = 0, alg_qual_bytes = 0;   // shipped bytes without alignment padding
    uint64_t alg_8d_bytes = 0, alg_cigar_bytes = 0;   // SURVEY.md section 8d accounting (per pileup read)
    int32_t  first_tid = -1, first_beg = 0, first_end = 0;   // first pileup_ok read (first-line quirk)
    std::vector<int32_t> first_any, first_from1;             // per contig (empty = no pileup read): first pileup line without -l / with `name 1 LEN`
    bool     warned_beyond_end = false;   // qaCompute cursor beyond a contig end: reported once per sample
    // packed on the device (devpack.hip): bases and quality FLAGS (one bit per base, as the kernels read them) live in HBM, seq / qual above stay empty
    bool     on_device = false;
    uint8_t *d_seq = nullptr, *d_qual = nullptr;      // into a round buffer of msnv_dataset::dp
    uint64_t d_seq_bytes = 0;                         // bytes of the seq column incl. the 32 tail bytes (what seq.size() is for a host-packed sample)
    bool     dev_index = false;                       // headers and intervals are still in HBM only (DevPackTables::rounds): hdr / tid / end / depth / cov_* above are empty
    int32_t  dev_round = -1;                          // ... in that round, from these offsets
    uint64_t dev_piece0 = 0, dev_iv0 = 0, n_dev_pieces = 0, n_dev_iv = 0;
    uint32_t *d_blk = nullptr; uint64_t n_dev_blk = 0;  // dense layout: its block descriptors in HBM (devpack.hip: devfin_dense)
    std::vector<DevPair> dev_pairs;                   // its (contig, tile) runs of pieces, in order
    msnv_sample_stats st{};          // qaCompute "Other" statistics (qaCompute.cpp:642-654), counted over every record of the BAM
};

struct DeviceCols;   // kernels.hip

// Dataset-level tables of the device pack, built when the first sample is packed on the device (BED and contig mask are fixed by then).
struct DevPackTables {
    void     *contigs = nullptr;      // DpContig[n_contigs]
    uint32_t *pref4 = nullptr;        // nt16 codes of the FASTA records of the selected contigs, 8 per word, contig c from nibble pref_off[c]
    uint64_t  pref_words = 0;
    // ... and on the host, for finalize (the reference is converted ONCE per dataset): the same words, per contig its first word (~0: none),
    // and one bit per base "FASTA char is a lower-case a/c/g/t" (call_vC.cpp:580), 32 per word, contig c from word h_lc_off[c]
    std::vector<uint32_t> h_codes, h_lc;
    std::vector<uint64_t> h_code_off, h_lc_off;
    std::vector<void *> round_bufs;   // packed seq / quality-bit buffers of the rounds (SampleCols::d_seq / d_qual point into them): freed by finalize
    std::vector<std::pair<void *, uint64_t>> scratch;   // per-round work buffers, kept (grow-only) from round to round: {pointer, capacity}
    std::vector<DevRound> rounds;
    int32_t  *overhang = nullptr;     // per contig: furthest end of a piece beyond the contig's length (0: none), device
    uint32_t *any_overhang = nullptr; // device flag
    bool      ready = false;
    // a round whose last kernels (msnv_emit_block: bases and headers out) may still be running when devpack_add_round returns: what is left
    // to read of it -- the mismatch sample of its samples, the kernels' time -- is taken by devpack_sync_pending, which everything that
    // needs the round finished calls first (the next round, finalize before it decides the allele bookkeeping, the statistics, release)
    struct Pending { bool active = false; size_t first = 0, n = 0; void *ev0 = nullptr, *ev1 = nullptr, *evh = nullptr, *evd = nullptr, *evd2 = nullptr, *evw = nullptr; bool has_evd = false; } pending;      // evd / evd2: around the depth stage on the second stream; evw: the records' tables are written      // evh: behind the small results the host takes while the emit kernels run
    bool      any_overhang_h = false; // some read of some round runs past its contig (msnv_measure_reads): finalize fetches `overhang`
    void     *cov_event = nullptr;   // recorded behind those kernels: what devfin_coverage waits for
    // (round 6) the coverage index's pair tables cut on the device: the work memory of their counts, what the second step reads of it
    struct CovT { uint32_t *tcont, *rid_t, *rowid, *row_sample, *row_contig, *item_off, *lo, *hi; unsigned long long *iv_start; };
    void     *cov_tables = nullptr; CovT cov_t{}; uint32_t cov_item_intervals = 16384; bool cov_tables_done = false;
    void     *fin_chunk_event = nullptr; bool fin_chunk_pending = false;      // behind the words devfin_chunks_launch's kernels leave in pinned memory
    void     *cov_job = nullptr, *cov_tmp = nullptr, *cov_runs = nullptr; bool cov_launched = false;   // finalize: the coverage index's kernels launched ahead of their results (devfin_coverage_launch)
    void     *fin_tile_base = nullptr;                     // finalize on the device: the contigs' first tiles (devfin_headers)
    void     *fin_list = nullptr, *fin_cbase = nullptr;   // finalize on the device: the narrow pairs' list and their chunk counts / scan, between devfin_chunk_counts and devfin_chunk_fill
    std::vector<void *> fin_keep;                          // small device tables of finalize's kernels, which are queued, not waited for: freed with the pack's tables (devpack_finish)
    uint64_t  fin_chunk_cap = 0; uint32_t fin_chunk_base = 0; bool fin_chunks_async = false;   // the narrow chunks cut without a wait (devfin_chunks_launch): room for them in d.chunks, where they start
    // pinned words of THIS dataset (taken from its context's pool, given back by devpack_release): first half = what a round's last kernels
    // leave for the host (DpAcc per sample: devpack_sync_pending), second half = finalize's small results (coverage index, chunk count)
    void     *pin = nullptr; uint64_t pin_cap = 0;
    bool      pad_in_emit = false;    // the rounds' emit kernels wrote the alignment nibbles behind the pieces (reference inside the tile, N beyond): msnv_fill_padding has nothing to do
    std::vector<uint8_t> h_contigs;                        // the contig table as it went up (the copy is not waited for)
    // cumulative device-pack accounting (msnv_host_timers: pack_device_wall_s; msnv_devpack_stats)
    double    ms_scan = 0, ms_measure = 0, ms_depth = 0, ms_emit = 0, ms_sort = 0, wall_upload_s = 0, wall_download_s = 0, wall_prepass_s = 0;
    uint64_t  raw_bytes = 0, n_records = 0, n_pieces = 0, n_prepass_samples = 0, n_scan_redone = 0, n_deep_runs_split = 0, n_dense_samples = 0;
    uint64_t  n_device_edit_samples = 0;     // samples whose depth cap / token limit ran as kernels (msnv_cap_reads, msnv_token_cut) instead of the host pre-pass
    uint64_t  n_quick_redone = 0;    // rounds the quick route had launched and the careful route took over (a sample needs the host pre-pass, far-reaching reads)
};


}  // namespace msnv
struct msnv_dataset;
namespace msnv {
// The per-sample records of fetched site i, one entry per sample: a pointer into the dense form when the dataset holds it, else the
// site's cells expanded into `scratch` (S entries, all zero between calls: the entries a call touches are zeroed again by the next).
struct SiteRowView {
    std::vector<msnv_site_sample> scratch;
    std::vector<uint32_t> touched;
    const msnv_site_sample *row(const msnv_dataset &ds, size_t i, size_t S);
};
}  // namespace msnv

struct msnv_ctx {
    int device = 0;
    void *stream = nullptr;   // hipStream_t
    void *stream2 = nullptr;  // a second stream of the context, created when first needed: the depth stage of a device-pack round beside its emit kernels (devpack.hip)
    // device BGZF inflate (inflate_k.hip): pinned host staging and device buffers, grown on demand and kept for the life of the context
    // (pinning memory costs ~0.25 s per GB: paid once, not per batch of BAMs)
    void *pin_in = nullptr, *pin_out = nullptr, *dev_in = nullptr, *dev_out = nullptr;
    uint64_t pin_in_cap = 0, pin_out_cap = 0, dev_in_cap = 0, dev_out_cap = 0;
    // pinned blocks for the small results kernels leave behind asynchronous copies (devpack.hip): a dataset takes one for its lifetime and
    // gives it back (two datasets of one context never share a landing area); pinning costs ~0.1 ms, so the blocks are kept
    std::vector<std::pair<void *, uint64_t>> pin_pool;
};

struct msnv_dataset {
    msnv_ctx *ctx = nullptr;
    msnv_ctx *feed_ctx = nullptr;     // msnv_dataset_set_feed_ctx: stream and staging of msnv_dataset_deal_bams_device / msnv_dataset_inflate_bams_device (NULL: ctx)
    msnv_params params{};
    // reference
    std::vector<std::string> names;
    std::vector<int64_t>     lengths;
    std::vector<std::string> seqs;         // FASTA characters; empty + !has_seq = contig absent
    std::vector<uint8_t>     has_seq;
    // shard selection
    std::vector<uint8_t>     sel;          // per contig
    std::vector<int64_t>     bed_beg, bed_end;   // per contig valid range (no BED: [0, INT64_MAX))
    bool has_bed = false;
    // samples
    std::vector<msnv::SampleCols> samples;
    bool finalized = false;
    bool poisoned = false;          // an add_* call failed AFTER rounds of the device pack had been appended (api.cpp: add_streams_device): nothing more is added or finalized
    // layout
    std::vector<uint32_t> tile_base;       // per contig (selected only; others = UINT32_MAX)
    std::vector<uint32_t> tile_contig;     // per tile
    std::vector<uint64_t> tile_slot_base;  // per tile: first entry of slot_sample (n_tiles + 1)
    std::vector<uint32_t> tile_slot_stride; // per tile: cells per site row on the device (>= its slots: rows of big tiles are padded to multiples of 8 cells)
    std::vector<uint32_t> slot_sample;     // sample of every (tile, slot): the device stores per-sample cells per slot (kernels.hip: CellMap)
    uint32_t n_tiles = 0;
    // first pileup line of the invocation (call_vC.cpp:423)
    int32_t first_tid = -1; int64_t first_pos = -1;
    msnv_dataset_info info{};
    msnv::DeviceCols *dev = nullptr;
    msnv::DevPackTables dp;
    // record streams read and inflated before the dataset had its device (msnv_dataset_stage_sample_bams): packed by finalize
    std::vector<msnv::ByteBuf> staged; std::vector<uint64_t> staged_off;
    // results of the last run (host copies)
    bool have_results = false, results_fetched = false;
    uint32_t last_counts_sites = 0;
    msnv_run_stats last_stats{};
    std::vector<msnv_site> sites;
    std::vector<msnv_site_sample> site_samples;   // dense [sites][samples] form: only datasets that were handed dense records hold it (msnv_write_calls_records, text entry)
    // per-sample records of the fetched sites as ROWS OF CELLS (CSR): site i owns cells [site_row[i], site_row[i + 1]), a cell = one sample
    // that has counted bases or alleles at the site.  What the device stores per tile slot (kernels.hip: CellMap) stays compact on the
    // host: a 500-sample cohort in which a species is carried by a handful of samples costs a handful of cells per site, not 500.
    std::vector<uint64_t> site_row;
    std::vector<uint32_t> site_cell_sample;
    std::vector<msnv_site_sample> site_cells;
    std::vector<uint32_t> site_dev_index;      // host record -> device site record
    bool ann_valid = false;                    // the device holds annotation records of the last run
    bool have_coverage = false;
    // coverage accumulators, one row per (sample, contig) that has intervals, in (sample, contig) order
    std::vector<uint32_t> cov_row_sample, cov_row_contig;
    std::vector<uint64_t> cov_row_start;       // per sample: its first row (n_samples + 1)
    std::vector<unsigned long long> cov_acc;   // [row][1 + COV_BINS] of the last coverage run
};

inline const msnv_site_sample *msnv::SiteRowView::row(const msnv_dataset &ds, size_t i, size_t S) {
    if (ds.site_row.empty()) return &ds.site_samples[i * S];
    if (scratch.size() != S) { scratch.assign(S, msnv_site_sample{}); touched.clear(); }
    for (uint32_t k : touched) scratch[k] = msnv_site_sample{};
    touched.clear();
    for (uint64_t c = ds.site_row[i]; c < ds.site_row[i + 1]; ++c) {
        const uint32_t k = ds.site_cell_sample[(size_t)c];
        scratch[k] = ds.site_cells[(size_t)c];
        touched.push_back(k);
    }
    return scratch.data();
}
