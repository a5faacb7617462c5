// metasnv_amd/csrc/device.h -- HBM-resident state of one dataset and the kernel pipeline entry.
#pragma once

#include <cstdint>

#include <string>
#include <vector>

#include "dataset.h"

namespace msnv {

struct Pair32 { uint32_t x, y; };
struct MergedGroupDev { uint64_t hdr_base; uint32_t tile, pair_lo, n_pairs, n_pieces; };   // one merged group of shallow pairs (gather_merged_block): its
                                                                                        // headers in hdr8m, its pairs

// ---- gene / codon annotation tables (ann_tables.cpp builds them, msnv_annotate_sites reads them)
constexpr uint8_t ANN_GENE_MINUS = 1, ANN_GENE_LINEAR = 2;       // strand '-', start < end (call_vC.cpp:609)
struct AnnGene   { int64_t start; int32_t contig; uint32_t flags; };          // start is contig-relative (may be -1)
struct AnnContig { int64_t cg_base, cg_len; uint32_t goff, pad; };            // codon genome: base index, length (-1: absent)
struct AnnHost {
    std::vector<uint32_t> seg_beg, seg_end;     // disjoint, sorted runs of the linear position space
    std::vector<int32_t>  seg_gene;
    std::vector<AnnGene>  genes;
    std::vector<AnnContig> contigs;
    std::vector<uint8_t>  codons;               // 4-bit codes, gene.h numbering A0 T1 C2 G3 N4
};
struct AnnDev {
    uint32_t *seg_beg = nullptr, *seg_end = nullptr; int32_t *seg_gene = nullptr;
    AnnGene *genes = nullptr; AnnContig *contigs = nullptr; uint8_t *codons = nullptr;
    uint32_t n_seg = 0;
    msnv_site_ann *out = nullptr; uint64_t cap_out = 0;
    uint32_t *err = nullptr;                    // [0] first position whose contig has genes but no FASTA record, [1] codon past the end
    bool ready = false;
    std::string key;                            // ann_path + '\n' + fasta_path the tables were built from
    std::vector<std::string> gene_names;
};
int  ann_build(msnv_dataset &ds, const Annotation &an, AnnHost &h);
void ann_gene_names(const Annotation &an, const std::vector<std::string> &contigs, std::vector<std::string> &out);
int  dev_ann_upload(DeviceCols &d, const AnnHost &h);
// Annotates the d.last_sites device site records; err_gpos[k] = UINT32_MAX when error kind k did not occur.
int  dev_annotate(DeviceCols &d, uint32_t n_sites, uint32_t drop_gpos, void *stream, double *ms, uint32_t err_gpos[2]);

struct DeviceCols {
    // ---- inputs (uploaded once by finalize)
    ReadHdr  *hdr = nullptr;         // 16-byte piece headers (wide kernel)
    PieceHdr *hdr8 = nullptr;        // 8-byte tile-local piece headers (narrow32 kernel, MSNV_LAYOUT=pieces; MSNV_HDR4=0 builds)
    uint32_t *hdr4 = nullptr;        // 4-byte chunk-relative piece headers (narrow32 kernel, default; dataset.h: HDR4)
    PieceHdr *hdr8m = nullptr;       // headers of the merged groups of shallow pairs, group by group: pair index in bits 19+, ABSOLUTE seq offset / 8
    uint32_t *tile_pair_merged = nullptr;   // per tile: first merged pair (they sit behind the tile's other pairs)
    uint32_t  n_work_merged = 0;     // work[n_work_narrow .. + n_work_merged) = merged items
    MergedGroupDev *merged_groups = nullptr; uint32_t n_merged_groups = 0;   // every merged group, for the gather
    uint32_t *blk = nullptr;         // dense layout: one descriptor per 32-base block (dense kernel)
    bool      dense = true;
    uint8_t  *seq = nullptr;
    uint8_t  *qual = nullptr;
    uint64_t *s_read_base = nullptr, *s_seq_base = nullptr;   // per sample
    uint32_t *ref4 = nullptr;        // nt16 codes, 8 positions per word, low nibble = lowest position
    uint32_t *ref_lc = nullptr;      // 1 bit per position: FASTA char is a lower-case a/c/g/t
    TilePair *pairs = nullptr;
    uint32_t *tile_pair_start = nullptr;   // n_tiles + 1
    WorkItem *work = nullptr;
    ChunkDesc *chunks = nullptr;
    uint32_t *tile_vbeg = nullptr, *tile_vend = nullptr;   // callable range inside each tile (BED / contig)
    uint32_t  n_tiles = 0, n_pairs = 0, n_work = 0, n_work_narrow = 0, n_samples = 0;   // work[0..n_work_narrow) = narrow items
    uint64_t  n_reads = 0, n_seq_bytes = 0, n_chunks = 0, n_hdr8m = 0, n_blk = 0;
    // ---- intermediates
    uint32_t *tot = nullptr;         // [4][n_tiles*TILE]: mismatching A, C, G, T summed over samples
    uint32_t *active_tiles = nullptr; // tiles that hold work items (gate / gather run over these only)
    uint32_t  n_active_tiles = 0;
    uint8_t  *part = nullptr;        // coverage partial row of every work item (tile-major; u16 per position for narrow items, u32 for wide)
    uint64_t *slot_off = nullptr;    // byte offset of every row; n_work + 1
    uint32_t *tile_slot_u16 = nullptr;    // first u16 row of every tile (u8 rows come first)
    uint32_t *tile_slot_wide = nullptr;   // first wide (u32) row of every tile
    uint64_t  part_bytes = 0;
    uint32_t *tile_slot_start = nullptr;   // n_tiles + 1
    uint8_t  *spill = nullptr;       // [n_pairs][TILE] per-sample coverage, saturating at 255
    uint8_t  *aspill = nullptr;      // [n_pairs][4][TILE] per-sample mismatching A, C, G, T counts (allele planes: noisy reads, pack.cpp); else NULL
    bool      allele_planes = false;
    Pair32   *events = nullptr;      // {gpos, sample<<18 | allele<<16 | count}
    Pair32   *overflow = nullptr;    // {gpos, sample<<16 | cov}
    uint32_t *counters = nullptr;    // two blocks of CNT_WORDS ([0] events [1] overflow [2] sites [4] pop lines [5] indiv lines, then the event
                                     // sub-list counters): consecutive passes alternate, the gate kernel of a pass zeroes the other block
    uint32_t  cnt_parity = 0;        // block the NEXT pass uses
    uint32_t *ind4 = nullptr;        // 4 bits per position: some sample holds >= calling_threshold reads of mismatching A / C / G / T
    uint32_t *unc_bits = nullptr;    // 1 bit per position: a sample split into several pairs holds a mismatching allele (follows ind4 in its allocation)
    uint32_t *tile_dirty = nullptr;  // per work item (by the slot of its coverage row), 1 bit per 64 positions of the tile: the item added to the allele totals there
                                     // (set by the pileup kernels, consumed and cleared by the gate)
    uint32_t *unc_sites = nullptr;   // [cap_sites]: sites whose call depends on a split / merged sample's summed counts (msnv_decide_sites)
    struct GateTileH { uint32_t tile, slot_lo, slot_16, slot_w, slot_hi, vbeg, vend, n_slots; uint64_t row0; uint32_t tot_mode, staged, pair_lo, n_plane_pairs, pad0, pad1; } *gate_tiles = nullptr;   // per active tile (kernels.hip: GateTile)
    unsigned long long *site_row = nullptr;   // per 64 positions: first cell of the first site in them (gate kernel; what an event finds its cell with)
    GateTileH *gate_tiles_dense = nullptr, *gate_tiles_staged = nullptr;   // gate_tiles without / only the tiles of whole-tile work items (staged: row0 = index of the record list)
    uint32_t *gather_tiles = nullptr; uint32_t n_gather_tiles = 0;   // active tiles that hold pairs outside merged groups (spill gather)
    TileStage *tile_stage = nullptr;   // per active tile (index of its GateTile): candidate records of whole-tile work items
    uint32_t *tile_stage_idx = nullptr;   // per tile: that index
    uint32_t  max_group_pairs = 0;   // most pairs a merged group holds (<= GMW_PAIRS: the merged gather runs a wavefront per group, kernels.hip)
    uint32_t  n_groups_solo = 0;     // the last n_groups_solo merged groups are whole-tile groups of ONE pair (no gather needed when the pass is fused)
    uint32_t  n_work_fused = 0;      // the last n_work_fused merged work items are whole-tile items
    uint32_t  n_fused_tiles = 0;     // tiles handled by whole-tile work items
    bool      fuse_disabled = false; // the passes of this dataset do not use the record lists (tests, experiments)
    int       qlow_cutoff = 0;       // the -Q cutoff the one-bit quality column was packed with (pack.cpp: pack_lowq); a pass must ask for the same
    uint32_t  last_ovf_tiles = 0;    // whole-tile work items of the last pass whose candidates did not fit a record list (their tiles took the unfused route)
    bool      wide_tot = false;      // some tile's allele totals need 32 bits per allele (tot_add mode 2): the gate kernel's wide instantiation
    bool      use_dirty = false;     // sparse cohort (few work items per tile): the gate kernel consults tile_dirty before it reads the allele totals
    uint32_t  gather_split = 4;      // workgroups per tile in the spill gather (fewer for sparse cohorts: a pair or two per tile)
    bool      any_split = false;     // some (sample, tile) run was dealt into several pairs: the calling rule then needs the summed per-sample records
    unsigned long long *site_bits = nullptr;   // 1 bit per position: is a site (written by the gate kernel for every tile)
    uint32_t *site_rank = nullptr;   // per 64 positions: index of their first site (tiles with sites only)
    uint32_t  cap_events = 0, cap_overflow = 0, cap_sites = 0;
    SiteRec  *sites = nullptr;
    uint32_t *tile_site_base = nullptr, *tile_site_cnt = nullptr;
    uint32_t *tile_nslots = nullptr; // per tile: samples that have reads in it = cells per site of that tile (kernels.hip: CellMap)
    unsigned long long *tile_cell_base = nullptr;   // per tile and pass: first cell of its sites' rows (gate kernel)
    uint64_t  cap_cells = 0, last_cells = 0;
    uint16_t *ncol = nullptr;        // [4][cap_cells]: mismatching A, C, G, T counts per (site, slot), one column per allele
    uint16_t *cov_col = nullptr;     // [cap_cells]: per-sample coverage.  Structure of arrays: a site's row of cells is contiguous in every column,
                                     // so rows can be zeroed and written 16 bytes at a time; the host zips the five columns into msnv_site_sample records
    uint8_t  *site_flags = nullptr;  // pop_mask | ind_mask << 4
    uint8_t  *site_elig = nullptr;   // alleles still open to the individual rule when the gate kernel has decided what it can (merged gather)
    uint64_t  cap_out_sites = 0, last_sites = 0;
    // ---- genome coverage (qaCompute path)
    Pair32   *cov_iv = nullptr;          // {gbeg, gend}: +1 at gbeg, -1 at gend (index space of qaCompute.cpp:530-552)
    uint64_t *s_cov_base = nullptr;      // per sample
    TilePair *cov_pairs = nullptr;
    WorkItem *cov_work = nullptr;
    uint32_t *tile_len = nullptr;        // scanned indices of each tile (i < contig length)
    uint32_t *tile_contig_dev = nullptr;
    unsigned long long *cov_acc = nullptr;   // [copy][row][1 + COV_BINS]: covSum, hist[0..] of every (sample, contig) that has intervals (rows: dataset.h); tile t adds to copy t % cov_copies
    uint64_t  n_cov_rows = 0;
    uint32_t  cov_copies = 1;                // (the tiles of a long contig would otherwise queue up on one 64-byte line); summed on the host
    uint32_t  n_cov_pairs = 0, n_cov_work = 0, n_cov_work_wide = 0, n_contigs = 0;   // (the last n_cov_work_wide items hold a pair of > 32 767 intervals)
    uint64_t  n_cov_iv = 0;
    uint64_t  device_bytes = 0;
    uint64_t  algorithmic_bytes = 0;
    // allocations that hold SEVERAL of the tables above (the index arena of finalize, an adopted round buffer of the device pack): pointers
    // into one of them are not freed by themselves (dev_free_all)
    std::vector<std::pair<void *, uint64_t>> blocks;
    AnnDev    ann;
    // second set of per-pass intermediates + second stream: msnv_pileup_run_many alternates passes between the two sets so
    // that the small tail kernels of pass i overlap with the pileup kernel of pass i+1 (allocated on first use)
    struct AltBufs {
        uint32_t *tot = nullptr; uint8_t *part = nullptr; uint8_t *spill = nullptr, *aspill = nullptr; Pair32 *events = nullptr, *overflow = nullptr;
        TileStage *tile_stage = nullptr;
        uint32_t *counters = nullptr; SiteRec *sites = nullptr; uint32_t *tile_site_base = nullptr, *tile_site_cnt = nullptr; unsigned long long *tile_cell_base = nullptr;
        uint16_t *ncol = nullptr; uint16_t *cov_col = nullptr; uint8_t *site_flags = nullptr, *site_elig = nullptr; uint32_t *ind4 = nullptr, *unc_bits = nullptr, *tile_dirty = nullptr, *unc_sites = nullptr; unsigned long long *site_row = nullptr; unsigned long long *site_bits = nullptr; uint32_t *site_rank = nullptr;
        uint32_t cap_events = 0, cap_overflow = 0, cap_sites = 0, cnt_parity = 0; uint64_t cap_out_sites = 0, cap_cells = 0;
    } alt;
    void     *stream2 = nullptr;
    std::vector<void *> event_pool;          // hipEvent_t of msnv_pileup_run_many
    uint32_t *pinned_cnt = nullptr; size_t pinned_cnt_cap = 0;   // pinned host blocks for the per-pass counters
    void     *timing_events[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // hipEvent_t, reused by every pass
};

// host copies of the counters after a run
// The allele-event list is cut into EV_LISTS equal sub-lists, each with its own fill counter on its own 64-byte line
// (work item i appends to sub-list i % EV_LISTS): returning atomics on ONE address serialise device-wide, and reads with
// several per cent of mismatches append often enough for that to dominate the pileup kernel.
constexpr uint32_t EV_LISTS = 32, EV_CNT_STRIDE = 16;
// counters[0..15] (what the host reads back: [0] events [1] overflow [2] sites [4] pop lines [5] indiv lines [6..7] cells), the
// event sub-list counters, then the counters every gate workgroup adds to -- each on a 64-byte line of its own: with a sparse
// cohort 10^5 workgroups add to them, and atomics on one line serialise at the memory side (BASELINE configs[3] shard: the gate
// kernel went from 2.0 to 2.9 ms when two more counters shared the line of the site counter)
constexpr uint32_t CNT_CELLS = 16 + EV_LISTS * EV_CNT_STRIDE;   // 64-bit: cells of the per-sample records
constexpr uint32_t CNT_TALLY = CNT_CELLS + 16;                  // 64-bit: population lines | individual lines << 32 (gate kernel's share)
constexpr uint32_t CNT_UNC = CNT_TALLY + 16;                    // sites left to msnv_decide_sites
constexpr uint32_t CNT_STAGE = CNT_UNC + 8;                   // a whole-tile work item found more than STAGE_CAP candidate positions
constexpr uint32_t CNT_WORDS = CNT_UNC + 16;

struct RunCounts { uint32_t n_events, n_overflow, n_sites, err; uint64_t n_cells; };

int  dev_set_device(int device);
uint32_t dev_resident_workgroups(uint32_t per_cu);   // compute units of the current device x per_cu (256 CUs when the query fails)
int  dev_alloc(void **p, uint64_t bytes, uint64_t *acct);
void dev_free(void *p);
void dev_free_batch(const std::vector<void *> &ptrs);
int dev_memset_async(void *dst, int v, uint64_t bytes, void *stream);
void dev_cache_trim();                               // cached free blocks back to the runtime (kernels.hip: dev_alloc)
int  dev_upload(void *dst, const void *src, uint64_t bytes);
int  dev_download(void *dst, const void *src, uint64_t bytes);
int  dev_copy_bytes(void *dst_device, const void *src, uint64_t bytes, bool src_on_device, void *stream);      // device <- device or host, on the stream, waited for
int  dev_memset(void *dst, int v, uint64_t bytes);
int  dev_stream_wait(void *stream);
int  dev_stream_create(void **stream);
void dev_stream_destroy(void *stream);

// One pass of the pipeline (pileup -> gate -> gather -> decide) on `stream`.
constexpr int COV_BINS = 16;            // histogram bins kept on the device (qaCompute -c <= 15)
int  dev_run_coverage(DeviceCols &d, int max_cov, void *stream, msnv_run_stats *stats);
int  dev_run_pipeline(DeviceCols &d, const msnv_params &p, void *stream, msnv_run_stats *stats, RunCounts *counts);
int  dev_run_pipeline_many(DeviceCols &d, const msnv_params &p, void *stream, int n, bool overlap, msnv_run_stats *stats, RunCounts *counts);
void dev_free_all(DeviceCols &d);

}  // namespace msnv
