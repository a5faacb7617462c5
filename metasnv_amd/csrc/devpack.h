// metasnv_amd/csrc/devpack.h -- the per-read stage on the device (devpack.hip): raw BAM alignment records resident in HBM ->
// packed read columns (dataset.h).  What the reference's tools do per read before anything is counted -- samtools' read filters
// (bam_plcmd.c mplp_func [EXT]), its CIGAR walk, the -Q test of every base, qaCompute's read filter and its
// walk over the M blocks (qaCompute.cpp:441-593, the walk :530-552) -- as kernels over a record stream that never visits a host core.
// pack.cpp holds the same stage as host code (MSNV_PACK=host); both produce the same bytes (tests/test_gpu_devpack.py).
#pragma once

#include <cstdint>
#include <vector>

#include "dataset.h"

namespace msnv {

// One round of samples: streams[i] (n_bytes[i] bytes of alignment records, on the host or -- on_device -- in HBM of the dataset's device)
// become ds.samples[first + i].  The headers, intervals and per-sample summaries come back to the host (finalize_dataset builds the tile
// index from them); bases and quality bits stay in HBM.
// in_place_base != NULL (streams on the device, all inside [in_place_base, + in_place_capacity), 16-byte aligned base, 256 readable bytes behind
// the last stream): the records are read where they lie, no copy into a round buffer (qualities may be edited there).
// waits for the round devpack_add_round left running, if any, and takes its last results (dataset.h: DevPackTables::Pending)
int devpack_sync_pending(msnv_dataset &ds);
void devpack_ctx_release(msnv_ctx *ctx);                         // the context's pinned words (msnv_ctx_destroy)
int devpack_add_round(msnv_dataset &ds, size_t first, const uint8_t *const *streams, const uint64_t *n_bytes, int n, bool on_device, const uint8_t *in_place_base = nullptr, uint64_t in_place_capacity = 0);
// A device-packed sample's bases and quality flags as host staging (SampleCols::seq / qual), for the two re-layouts that still run on
// the host (pack.cpp: relayout_dense, split_deep_runs' relocation).
int devpack_sample_to_host(SampleCols &sc);
int devpack_download_pieces(msnv_dataset &ds);
void devpack_release(msnv_dataset &ds);                       // the pack's tables, round buffers, work buffers; the pinned block back to the context
// finalize: copies of the rounds' columns into the dataset's, and the alignment padding behind every piece set to the reference
int devpack_copy_columns(const SampleCols &sc, uint8_t *dst_seq, uint8_t *dst_qual_bits, void *stream);
int devpack_place_columns(msnv_dataset &ds, DeviceCols &d, const std::vector<uint64_t> &sbase);      // fast finalize, piece layout: adopt one round's buffer / copy round by round
int devpack_fill_padding(DeviceCols &d, const std::vector<uint8_t> &sample_on_device, void *stream);
// finalize on the device (all samples device-packed; devpack.hip "finalize on the device"): the per-piece / per-interval loops of
// finalize_dataset as kernels over the rounds' headers and intervals
struct DevMergedSrc { uint32_t pair, in_group; unsigned long long h_base; };   // a pair of a merged group: its index in the group, first slot of its headers in hdr8m
struct DevCovPair { uint32_t tile, sample, lo, hi; };                            // intervals [lo, hi) of `sample` (kept ones, sample-relative) may touch `tile`
int devfin_overhang(msnv_dataset &ds, std::vector<int64_t> &maxend);
// deep (sample, tile) runs (pack.cpp: split_deep_runs) on the device: exact depth of every run whose bound reaches split_at, runs that are
// really that deep dealt round robin into groups of pieces that each stay below the byte bins' limit; the sample's headers are permuted group
// by group and its columns re-laid in header order.  Updates SampleCols::dev_pairs.  *fallback: a run needs more groups than the kernel holds.
int devfin_deep_runs(msnv_dataset &ds, uint32_t split_at, uint32_t group_depth, bool *fallback);
// pack.cpp: relayout_dense for device-packed samples: every sample's columns as dense block streams, its descriptors (d_blk) and run_* tables
int devfin_dense(msnv_dataset &ds);
// msnv_records_deal_device (msnv.h): pack.cpp's records_partition for several streams at once, on the device
int records_deal_device(msnv_ctx *ctx, const uint8_t *const *streams, const uint64_t *n_bytes, int n, bool on_device, const int32_t *owner, int n_contigs, int n_parts,
                        int cov_min_mapq, uint8_t *out, uint64_t capacity, uint64_t gap, uint64_t *part_bytes, msnv_sample_stats *stats, uint64_t *contig_bases);
int devpack_copy_blocks(const SampleCols &sc, uint32_t *dst, void *stream);
int devfin_headers(msnv_dataset &ds, DeviceCols &d, const std::vector<uint64_t> &rbase);
int devfin_chunk_counts(msnv_dataset &ds, DeviceCols &d, const std::vector<uint32_t> &narrow_pairs, std::vector<uint32_t> &cbase);   // cbase: first chunk of every listed pair, total behind them
int devfin_chunk_fill(msnv_dataset &ds, DeviceCols &d, size_t n_pairs_listed, uint32_t base);        // d.chunks[base .. base + total), d.hdr4
// the same without a wait (devpack.hip): item_first[wi] = index of narrow work item wi's first pair in the list (n_work_narrow + 1 entries); the chunks
// go behind the `base` chunks the host wrote, `cap` of them at most; devfin_chunks_result (behind a wait for the stream) says how many there were
int devfin_chunks_launch(msnv_dataset &ds, DeviceCols &d, const std::vector<uint32_t> &narrow_pairs, const std::vector<uint32_t> &item_first, uint32_t base, uint64_t cap);
int devfin_chunks_result(msnv_dataset &ds, uint64_t *n_chunks, bool *overflow);
int devfin_work_first(msnv_dataset &ds, DeviceCols &d, uint32_t n_items);                            // WorkItem::first of the narrow and merged items, from d.chunks
int devfin_merged_headers(msnv_dataset &ds, DeviceCols &d, const std::vector<DevMergedSrc> &list);
int devfin_coverage_launch(msnv_dataset &ds, DeviceCols &d);   // needs ds.tile_base / n_tiles; the kernels only (their results: devfin_coverage)
int devfin_coverage(msnv_dataset &ds, DeviceCols &d, std::vector<uint64_t> &cvbase, std::vector<DevCovPair> &cp);
// waits for the copies, releases the round buffers and the tables (no sample can be added after finalize)
int devpack_finish(msnv_dataset &ds);

// pack.cpp: the three sequential edits of the host stage (depth cap, overlapping-mate tweak, snpCall's token limit) for ONE sample whose
// records need them (devpack.hip decides that): per record 1 | pile_ok << 1 | cov_ok << 2 | depth << 16, the stream with the edited
// qualities (empty: nothing was edited), and whether it carries QUAL_CUT marks.
int host_prepass(const msnv_dataset &ds, const uint8_t *rec, uint64_t n_bytes, std::vector<uint32_t> &ovr, std::vector<uint8_t> &patched, bool &cut_marks);

}  // namespace msnv
