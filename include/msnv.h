/*
 * include/msnv.h -- C ABI of libmsnv.so, the MI355X-native pileup SNV caller.
 *
 * The reference (metaSNV) has no plugin API on this path: its hot path is two PROCESS
 * boundaries driven by metaSNV.py.  Each entry point below names the process invocation
 * (argv + files + exit status) it replaces, so a maintainer can swap the subprocess call
 * for one ctypes call (the stub is shown in INTEGRATION.md).
 *
 *   msnv_coverage()  <-  `qaCompute -c 10 -d -i BAM OUT`            metaSNV.py:55-78
 *                        (src/qaTools/qaCompute.cpp:286-681)
 *   msnv_call()      <-  `samtools mpileup -f REF [-l SPLIT] -B -b LIST |
 *                         snpCall -f REF [-g ANN] -i INDIV -c C -t T > CALLED`
 *                                                                    metaSNV.py:153-176
 *                        (src/snpCaller/call_vC.cpp:330-679)
 *
 * The staged msnv_dataset_* / msnv_pileup_run / msnv_write_calls entry points are the same
 * path cut at its natural seams (decode+pack | device kernels | text), so that a caller
 * can keep the packed columns resident in HBM (bench.py times msnv_pileup_run only).
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on
 * success and a positive MSNV_E* code otherwise (the reference's drivers treat any
 * non-zero exit status as fatal: metaSNV.py:75-78,212-221); the message is available from
 * msnv_last_error() (thread-local) and echoed to stderr.  The library never abort()s.
 * All compute runs on the GPU: there is no CPU fallback, and every entry point fails with
 * MSNV_ENODEV when no HIP device is usable.
 */
#ifndef MSNV_H
#define MSNV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSNV_OK        0
#define MSNV_EINVAL    1   /* bad argument                                              */
#define MSNV_EIO       2   /* cannot open / read / write a file                         */
#define MSNV_EFORMAT   3   /* malformed BAM / FASTA / BED / annotation                  */
#define MSNV_ENODEV    4   /* no usable HIP device (the path never falls back to CPU)   */
#define MSNV_EHIP      5   /* a HIP runtime call or kernel launch failed                */
#define MSNV_ENOMEM    6
#define MSNV_EDOMAIN   7   /* input outside the domain of the reference tools (SURVEY.md
                              Appendix A "D" items), e.g. BAM without mapped reads       */
#define MSNV_ECAPACITY 8   /* internal device buffer too small (caller may retry)        */

typedef struct msnv_ctx     msnv_ctx;      /* one per GPU / per rank          */
typedef struct msnv_dataset msnv_dataset;  /* packed read columns of one shard */

int         msnv_abi_version(void);
const char *msnv_last_error(void);

/* Number of visible HIP devices (hipGetDeviceCount; 0 when the runtime has none). */
int  msnv_device_count(void);
int  msnv_ctx_create(int device_id, msnv_ctx **out);
void msnv_ctx_destroy(msnv_ctx *ctx);

/* ------------------------------------------------------------------------------------
 * Calling parameters = snpCall's options (call_vC.cpp:26-36,382-390) + the mpileup
 * defaults metaSNV relies on (metaSNV.py:160-165 passes none).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    int32_t min_coverage;        /* snpCall -c, default 4                                 */
    int32_t calling_threshold;   /* snpCall -t, default 4                                 */
    double  min_fraction;        /* snpCall -p, default 0.01 (never passed by metaSNV.py) */
    int32_t min_baseq;           /* mpileup -Q, default 13                                */
    int32_t flag_filter;         /* mpileup --ff, default 0x704                           */
    int32_t count_orphans;       /* mpileup -A, default 0                                 */
    int32_t max_depth;           /* mpileup -d, default 8000 (per file)                   */
    int32_t min_mapq;            /* mpileup -q, default 0                                 */
    int32_t drop_first_line;     /* 1 = reproduce call_vC.cpp:423 (first pileup position of
                                    the invocation is consumed for sample counting and never
                                    called); 0 = call every position                      */
    int32_t cov_max;             /* qaCompute -c, default 10                              */
    int32_t cov_min_mapq;        /* qaCompute -q, default 1                               */
    int32_t ignore_overlaps;     /* mpileup -x, default 0: the overlapping-mate quality tweak is ON, as in the
                                    reference's command line (metaSNV.py:160-165 passes no -x)             */
    int32_t token_limit;         /* snpCall's per-token character limit (call_vC.cpp:482: 10000): a sample's
                                    base string is cut there, bases behind the cut are not counted; 0 = unlimited */
} msnv_params;

void msnv_params_default(msnv_params *p);

/* ------------------------------------------------------------------------------------
 * msnv_coverage  <-  qaCompute -c <max_cov> -d -i <bam_path> <out_cov_path>
 * Writes <out_cov_path> and <out_detail_path> byte-for-byte as qaCompute.cpp:192-217,
 * 226-263,439,623-657 would.  Return: 0 ok (qaCompute.cpp:680); MSNV_EIO mirrors its
 * exit 1 (:53-57,376-379); MSNV_EDOMAIN for the inputs on which it has undefined
 * behaviour (no mapped reads: :596).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const char *bam_path;
    int32_t     max_cov;          /* -c, metaSNV passes 10 */
    int32_t     min_mapq;         /* -q, default 1         */
    const char *out_cov_path;     /* OUT                   */
    const char *out_detail_path;  /* OUT.detail (-d)       */
} msnv_cov_args;

int msnv_coverage(msnv_ctx *ctx, const msnv_cov_args *args);

/* ------------------------------------------------------------------------------------
 * msnv_call  <-  samtools mpileup -f ref_fasta [-l bed_split_path] -B -b <bam list> |
 *                snpCall -f ref_fasta [-g ann_path] -i out_indiv_path -c .. -t .. [-p ..]
 *                > out_called_path
 * Sample order = bam_paths order = all_samples order (call_vC.cpp:530).
 * contig_rank_mask: optional (NULL = all): one byte per BAM-header contig, non-zero =
 * this rank's shard (multi-GPU contig sharding); output then holds
 * only those contigs.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const char *const *bam_paths;
    int32_t     n_bams;
    const char *ref_fasta;        /* -f                                                  */
    const char *ann_path;         /* -g or NULL                                          */
    const char *bed_split_path;   /* -l or NULL: 3-column `name\t1\tLEN` (metaSNV.py:92)  */
    const char *out_called_path;  /* stdout of snpCall                                   */
    const char *out_indiv_path;   /* -i or NULL                                          */
    const uint8_t *contig_rank_mask;
    int32_t     n_contig_rank_mask;
    int32_t     host_threads;     /* BAM decode threads, 0 = auto                        */
    msnv_params params;
} msnv_call_args;

int msnv_call(msnv_ctx *ctx, const msnv_call_args *args);

/* msnv_call_from_mpileup  <-  snpCall -f ref_fasta [-g ann_path] -i out_indiv_path -c -t [-p] < mpileup.txt > out_called_path
 * (call_vC.cpp:346-410 options, :423-668 main loop): snpCall on its own input, the TEXT `samtools mpileup -f REF -B -b LIST`
 * writes -- the reference's literal process boundary, for pipelines that already hold pileup text and for
 * A/B runs against real samtools output.  The text is parsed and called on the device (csrc/textcall.hip): first line = sample
 * count, never processed (:423-431); toksplit's 10 000-character tokens (:92-111); ^x / +n / -n / * $ N n (:503-535); gates and
 * calling rule (:545-601); gene / codon annotation with both ann_path and ref_fasta (:448).
 *   text / text_bytes: the pileup text in memory; or text = NULL and mpileup_path names a file ("-" or NULL: stdin).
 *   out_indiv_path may be NULL (individual SNVs are then dropped, :653-660).  Of params only min_coverage, calling_threshold and
 *   min_fraction apply (the mpileup options were spent by whoever wrote the text).
 *   stats (may be NULL): [0] lines read, [1] samples, [2] called_SNPs lines, [3] indiv_called lines, [4] kernel microseconds,
 *   [5] text bytes parsed on the device, [6] base-string characters parsed, [7] reserved.
 * Input the reference crashes on (a pileup symbol outside its ten keys, e.g. '>' '<' from CIGAR N or an IUPAC letter; more samples
 * in a line than in the first) is MSNV_EDOMAIN and nothing is written. */
typedef struct {
    const char *text; uint64_t text_bytes;
    const char *mpileup_path;
    const char *ref_fasta;        /* may be NULL without ann_path */
    const char *ann_path;         /* may be NULL                  */
    const char *out_called_path;
    const char *out_indiv_path;   /* may be NULL                  */
    msnv_params params;
} msnv_mpileup_args;
int msnv_call_from_mpileup(msnv_ctx *ctx, const msnv_mpileup_args *args, uint64_t stats[8]);

/* ------------------------------------------------------------------------------------
 * Next row of the path: metaSNV_Filtering.py filter_two
 * (metaSNV_Filtering.py:156-242) -- position filter and allele frequencies of the called
 * positions, computed on the device for all species of interest in one pass over the files.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const char        *species;     /* TaxID = contig name up to the first '.' (:172)              */
    int32_t            n_soi;       /* samples of interest of this species (relevant_taxa, :111-145) */
    const int32_t     *soi;         /* their indices in all_samples order (:180-181)                */
    const char *const *soi_names;   /* header of <species>.filtered.freq (:203)                     */
} msnv_filter_species;

/* snp_paths: the snpCaller/called* (or indiv*) files, in the order they are to be read.  Writes
 * out_dir/<species>.filtered.freq for every species with at least one surviving position (-c min_cov_c,
 * -p min_prop_p; defaults 5.0 and 0.50).  n_positions_kept / ms_kernel may be NULL. */
/* repr() of a Python float, the text metaSNV_Filtering.py:231 prints for a frequency (shortest digits that
 * round-trip; exponent form below 1e-4 and from 1e16).  Returns the length, or -1 when cap is too small. */
int msnv_format_float(double x, char *buf, int32_t cap);

int msnv_filter_files(msnv_ctx *ctx, const char *const *snp_paths, int32_t n_paths, int32_t n_samples,
                      const msnv_filter_species *species, int32_t n_species, double min_cov_c, double min_prop_p,
                      const char *out_dir, uint64_t *n_positions_kept, double *ms_kernel);
/* The same filter fed from the records of the dataset's last msnv_pileup_run instead of the text of called_SNPs /
 * indiv_called (no S-wide lines are written, read back and split): which = 0 the population calls (what called_SNPs
 * holds), 1 the individual calls (indiv_called, metaSNV_Filtering.py --ind).  ann_path / fasta_path as for msnv_write_calls
 * (gene column and codon tags of the row ids), or NULL.  Writes the same bytes as msnv_filter_files on the written files. */
struct msnv_dataset;
int msnv_filter_resident(struct msnv_dataset *ds, int32_t which, const msnv_filter_species *species, int32_t n_species,
                         double min_cov_c, double min_prop_p, const char *out_dir, const char *ann_path, const char *fasta_path,
                         uint64_t *n_positions_kept, double *ms_kernel);

/* metaSNV_DistDiv.py --dist (computeDist, metaSNV_DistDiv.py:105-124): pairwise sample
 * distances of one species' *.filtered.freq table on the device, bit-exact with pandas (NaN-skipping mean whose sum is
 * numpy's pairwise summation).  Writes the manhattan (mean |f1 - f2|) and the allele (share of positions with
 * |f1 - f2| > threshold, 0.6 in the reference) matrices as DataFrame.to_csv(sep='\t') does.  Out pointers may be NULL. */
/* The value pandas' default CSV float converter (precise_xstrtod) gives `text` -- what computeDist sees after
 * pd.read_table; not always the correctly rounded one.  Returns 0, or MSNV_EFORMAT when text is not a plain number. */
int msnv_parse_float(const char *text, double *value);

int msnv_dist_file(msnv_ctx *ctx, const char *freq_path, const char *mann_path, const char *allele_path, double threshold,
                   int32_t *n_samples, uint64_t *n_positions, double *ms_kernel);

/* subpopr's raw-SNV consumers.
 *
 * msnv_genotyping_subset -- src/subpopr/inst/getGenotypingSNVSubset.py:20-48: the positions listed in every
 * <species>_hap_positions.tab (field 2 = contig:gene:pos:base), then one scan of the called_SNPs* files that copies each
 * line whose contig:pos is wanted into out_dir/<species>.pos of every species that listed it.  The path lists are read
 * in the order given (the reference walks glob.glob's directory order).  Text only: runs without a device.
 *
 * msnv_snv_allele_freq -- src/subpopr/inst/convertSNVtoAlleleFreq.py:7-24: writes <pos_path>.freq, one row per allele
 * of every line: id contig:gene:pos:base, then per sample -1 (coverage below min_depth) or count / coverage * 100,
 * computed on the device and printed like str(float).  Every line of one file must list the same number of samples. */
int msnv_genotyping_subset(const char *const *hap_paths, int32_t n_hap, const char *const *snp_paths, int32_t n_snp,
                           const char *out_dir, uint64_t *n_positions, uint64_t *n_lines_written);
int msnv_snv_allele_freq(msnv_ctx *ctx, const char *pos_path, int32_t min_depth, uint64_t *n_rows, double *ms_kernel);

/* ------------------------------------------------------------------------------------
 * Staged form of the same path.
 * ------------------------------------------------------------------------------------ */

/* Reference description: contigs in BAM-header order.  seqs[i] may be NULL (contig absent
 * from the FASTA: mpileup then prints 'N').  The library copies everything. */
typedef struct {
    int32_t            n_contigs;
    const char *const *names;
    const int64_t     *lengths;    /* @SQ LN                                   */
    const char *const *seqs;       /* FASTA characters, case preserved, or NULL */
    const int64_t     *seq_lens;
} msnv_ref_desc;

/* ctx may be NULL: the dataset then serves the host-stage entry points only (add_sample_*, pileup_qualities,
 * sample_stats, first_lines) and msnv_dataset_finalize fails with MSNV_ENODEV -- nothing is ever computed on the CPU. */
int  msnv_dataset_create(msnv_ctx *ctx, const msnv_ref_desc *ref, const msnv_params *params,
                         msnv_dataset **out);
/* Convenience: contigs from the header of `bam_path`, sequences from `fasta_path`. */
int  msnv_dataset_create_from_files(msnv_ctx *ctx, const char *bam_path, const char *fasta_path,
                                    const msnv_params *params, msnv_dataset **out);
/* Gives a dataset that was created without a context (ctx = NULL: host-stage entry points only) its device, before msnv_dataset_finalize.
 * A one-shot driver creates the HIP context (~0.5 s of runtime start-up) on a thread of its own while the BAMs are read, inflated and
 * packed by the host threads (metasnv_amd/cli.py); the reference's counterpart is process start-up of its tools. */
int  msnv_dataset_attach_ctx(msnv_dataset *ds, msnv_ctx *ctx);
void msnv_dataset_destroy(msnv_dataset *ds);

/* Restrict the shard: BED regions (mpileup -l; 0-based half-open, at most one region per
 * contig) and/or a contig mask.  Must precede the first add_sample call. */
int  msnv_dataset_set_bed(msnv_dataset *ds, int32_t n, const int32_t *tid, const int64_t *beg, const int64_t *end);
int  msnv_dataset_set_bed_file(msnv_dataset *ds, const char *bed_path);
int  msnv_dataset_set_contig_mask(msnv_dataset *ds, const uint8_t *mask, int32_t n);

/* Append one sample (= one BAM of all_samples), in order.  `records` = concatenated raw
 * uncompressed BAM alignment records (each starting with its int32 block_size), i.e. what
 * sam_read1() yields (qaCompute.cpp:441). */
int  msnv_dataset_add_sample_records(msnv_dataset *ds, const uint8_t *records, uint64_t n_bytes);
/* n record streams appended as n samples, in order, packed by a pool of host_threads threads (0 = all cores): what the N-rank
 * driver does with the streams of one exchange round (metasnv_amd/parallel.py: feed_sharded). */
int  msnv_dataset_add_sample_records_many(msnv_dataset *ds, const uint8_t *const *records, const uint64_t *n_bytes, int32_t n, int32_t host_threads);
/* The same for record streams that already lie in the HBM of the dataset's device -- what an all-to-all over RCCL has just received
 * (metasnv_amd/parallel.py: exchange_records), what the device inflate has written: the records never visit a host core.  The per-read
 * stage of both reference tools runs as kernels over them (csrc/devpack.hip): record boundaries, the read loop and CIGAR walk of
 * qaCompute (qaCompute.cpp:441-593) and, for `samtools mpileup` (metaSNV.py:160-165), its read filters, the CIGAR walk to aligned
 * segments and the -Q test of every base.  The streams are copied; the caller may release them when the call returns.
 * Every add_sample_* entry point takes this route when the dataset has a device context (MSNV_PACK=host keeps the stage on the
 * host threads: pack.cpp, the same bytes).  The three sequential edits -- depth cap, overlapping-mate tweak, snpCall's token
 * limit -- are a host pre-pass over the samples whose records can trigger them. */
int  msnv_dataset_add_sample_records_device(msnv_dataset *ds, const void *const *dev_records, const uint64_t *n_bytes, int32_t n);
/* ... and WITHOUT the copy (round 5), for streams that lie in ONE device buffer the caller hands over for the duration of the call:
 * stream i = bytes [offsets[i], offsets[i] + n_bytes[i]) of `dev_buffer` (capacity bytes).  The buffer must start on 16 bytes and hold
 * at least 256 readable bytes behind the last stream's end (the kernels read whole 16-byte pieces around a record); the streams must not
 * overlap and must come in ascending order.  The library reads the records where they lie and MAY EDIT base qualities in place (htslib's
 * overlapping-mate tweak, snpCall's token limit: what the reference tools do to their own copy of a record, sam.c [EXT],
 * call_vC.cpp:481-483).  The last kernels of the call may still be READING the buffer when it returns (the host goes on with the
 * tile index meanwhile): the buffer is the library's -- valid, not written by anyone else -- until the NEXT call on this dataset that
 * adds samples, finalizes or destroys it has returned.  What the device inflate leaves in HBM goes this way, and the bench's "records
 * resident in HBM -> calls" region starts here. */
int  msnv_dataset_add_sample_records_resident(msnv_dataset *ds, void *dev_buffer, uint64_t capacity, const uint64_t *offsets, const uint64_t *n_bytes, int32_t n);
int  msnv_dataset_add_sample_bam(msnv_dataset *ds, const char *bam_path);
/* Decode many BAMs with a host thread pool, preserving order. */
int  msnv_dataset_add_sample_bams(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads);

/* The files are read, inflated and checked now (host threads), packed by msnv_dataset_finalize: a dataset created without a context
 * (msnv_dataset_attach_ctx later) still gets the per-read stage as kernels.  No other add_sample_* call may follow: every one of them
 * returns MSNV_EINVAL on a dataset that holds staged streams (more staging is fine).
 * Failure of any add_sample_* call leaves the dataset as it was before the call -- except when the call had already packed part of its
 * samples on the device (a later round of the same call failed: out of memory, a malformed record): such a dataset answers MSNV_EINVAL
 * to every further add_sample_* and to msnv_dataset_finalize and can only be destroyed. */
int  msnv_dataset_stage_sample_bams(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads);

/* Host-stage seam (tests, A/B against `samtools mpileup` text): writes to `out` (n_bytes) the same record stream with the
 * base qualities as the pileup engine sees them -- after the overlapping-mate tweak (unless params.ignore_overlaps) and
 * with the bases behind params.token_limit characters of a sample's base string set to quality 0 -- under this dataset's
 * read filters, BED and contig mask.  Does not add a sample.  (With params.min_baseq = 0 a quality of 0 is not below the
 * cutoff, so this FORM cannot tell a cut base from a counted one; the dataset itself marks cut bases below every cutoff:
 * pack.cpp QUAL_CUT.) */
int  msnv_dataset_pileup_qualities(const msnv_dataset *ds, const uint8_t *records, uint64_t n_bytes, uint8_t *out);

/* Builds the tile index and uploads the packed columns to HBM. */
int  msnv_dataset_finalize(msnv_dataset *ds);

typedef struct {
    uint64_t n_samples, n_contigs, n_positions;   /* positions = sum of shard contig lengths      */
    uint64_t n_reads, n_reads_pileup;             /* stored reads / reads that pass mpileup filters */
    uint64_t n_pileup_bases;                      /* (read, ref position) pairs from M/=/X ops of
                                                     reads that pass the read-level filters        */
    uint64_t bytes_headers, bytes_cigar, bytes_seq, bytes_qual, bytes_ref, bytes_index;
    uint64_t n_tiles, n_pairs, n_work;
    uint64_t device_bytes;                        /* total HBM held by the dataset                 */
    uint64_t allele_planes;                       /* 1: the per-sample allele counts go through byte planes (noisy reads), 0: through
                                                     sparse events (clean reads); picked by finalize from the sampled mismatch rate */
    uint64_t sampled_mismatch_ppm;                /* aligned bases that differ from the reference, per million (every 16th piece)   */
    uint64_t n_whole_tile_items;                  /* tiles piled up AND gated by one workgroup (sparse cohorts) */
    uint64_t n_listed_tiles;                      /* ... of which the last pass sent through the ordinary gate kernel: more candidate
                                                     positions than a tile's record list holds                                      */
} msnv_dataset_info;

int  msnv_dataset_info_get(const msnv_dataset *ds, msnv_dataset_info *out);
/* Cost of the per-read stage on the device for this dataset so far (csrc/devpack.hip), cumulative over its rounds:
 * out[0..4] kernel ms (HIP events on the context's stream) of record scan (the quick route, round 6: record boundaries AND the per-record measure,
 * one walk), per-record measure (careful route only), depth, emit (headers + bases + flags), tile-order sort; [5..7] wall seconds of uploading host streams, downloading headers / intervals, the host pre-pass (depth cap,
 * overlapping mates, token limit); [8] record bytes resident in HBM, [9] records, [10] pieces, [11] samples that took the pre-pass,
 * [12] scan segments whose guessed entry point was wrong and that were walked again, [13] deep (sample, tile) runs dealt into groups by
 * the device form of finalize, [14] samples whose dense block streams (short reads) were laid out by it, [15] rounds the quick route had
 * launched and the careful route took over (a sample that needs the sequential edits, more far-reaching reads than the list holds), [16] samples
 * whose depth cap / token limit ran as kernels (round 6; [11] counts the ones the host pre-pass still takes: MSNV_PREPASS=host, a template with
 * more alignments than the overlap kernel's slots, far-reaching reads). */
#define MSNV_PACK_STATS 17
int  msnv_dataset_pack_stats(const msnv_dataset *ds, double *out, int32_t n);
/* Inspection hook: the bytes of one device column / index table of a finalized dataset ("hdr", "hdr4", "hdr8m", "blk", "seq", "qual",
 * "s_read_base", "s_seq_base", "ref4", "pairs", "work", "chunks", "cov_iv", "cov_pairs", "cov_work").  out = NULL: size query.  The tests
 * use it to show that the device pack (csrc/devpack.hip) and the host pack (csrc/pack.cpp) build the same dataset byte for byte. */
int  msnv_dataset_fetch_column(msnv_dataset *ds, const char *name, uint8_t *out, uint64_t capacity, uint64_t *n_bytes);

typedef struct {
    float    ms_total;           /* all kernels of one pass, HIP events on the launch stream */
    float    ms_pileup;          /* the dominant kernel (msnv_pileup_tiles)                  */
    float    ms_gate, ms_gather, ms_decide, ms_coverage;   /* the tail split is recorded only with MSNV_PHASE_TIMES=1 (event records cost stream time) */
    uint64_t n_sites;            /* candidate positions that reached the decision kernel     */
    uint64_t n_called_pop, n_called_indiv;   /* output lines (before the first-line drop)    */
    uint64_t n_events, n_overflow;
    uint64_t algorithmic_bytes;  /* bytes the pileup kernel must read  */
} msnv_run_stats;

/* One pass of the hot path over the resident dataset: per-position / per-sample allele
 * histogram, calling rule, per-sample gather.  Results stay on the device until
 * msnv_write_calls / msnv_results_*.  Safe to call repeatedly (bench.py). */
int  msnv_pileup_run(msnv_dataset *ds, msnv_run_stats *stats);

/* Per-sample genome coverage (qaCompute arithmetic) over the resident dataset. */
/* n passes enqueued back to back with ONE host synchronisation at the end (stats: n entries, may be NULL): how a queue of
 * shards keeps the GPU busy -- the host round trip of msnv_pileup_run costs ~40 us of idle GPU per pass.  With overlap != 0
 * consecutive passes alternate between two HIP streams and two sets of intermediates, so the small tail kernels of one
 * pass run under the pileup kernel of the next (+5 % passes/s; per-kernel times are then those of kernels sharing the
 * chip).  The results are those of the last pass.  msnv_pileup_run must have run once before (it sizes the buffers). */
int  msnv_pileup_run_many(msnv_dataset *ds, int32_t n, int32_t overlap, msnv_run_stats *stats);
/* The HIP events and the pinned counter blocks a batch of n passes needs, created now (the first msnv_pileup_run_many call of that size
 * otherwise creates them inside its own time: ~0.5-1 ms of host work in front of the first pass). */
int  msnv_pileup_reserve(msnv_dataset *ds, int32_t n);
int  msnv_coverage_run(msnv_dataset *ds, msnv_run_stats *stats);
/* Both passes over the same resident columns (BASELINE configs[2]: qaCompute + snpCall fused on the device):
 * replaces running `qaCompute` per BAM (metaSNV.py:63-65) and then `samtools mpileup | snpCall` (:160-176) on
 * the same files.  Either stats pointer may be NULL. */
int  msnv_fused_run(msnv_dataset *ds, msnv_run_stats *pileup_stats, msnv_run_stats *coverage_stats);
/* Writes OUT / OUT.detail for sample `sample_idx` from the last msnv_coverage_run. */
int  msnv_write_coverage(msnv_dataset *ds, int32_t sample_idx, const char *cov_path, const char *detail_path);

/* Formats the last run as called_SNPs / indiv_called (call_vC.cpp:641-667).
 * ann_path / fasta_path feed the codon annotation (call_vC.cpp:604-633) and may be NULL. */
int  msnv_write_calls(msnv_dataset *ds, const char *called_path, const char *indiv_path,
                      const char *ann_path, const char *fasta_path);

/* Raw results of the last run (for tests and for the multi-GPU gather).
 * A site record is fixed width:  msnv_site header + n_samples x msnv_site_sample. */
typedef struct {
    int32_t  tid;        /* BAM-header contig index              */
    int32_t  pos;        /* 0-based position                     */
    uint32_t cov;        /* total coverage over all samples      */
    uint32_t n[4];       /* total A, C, G, T                     */
    uint8_t  pop_mask;   /* bit i: allele i (A,C,G,T) is a population SNV */
    uint8_t  ind_mask;   /* bit i: allele i is an individual SNV          */
    uint8_t  refchar;    /* FASTA character                      */
    uint8_t  dropped;    /* 1 = this is the first pileup line (call_vC.cpp:423) */
} msnv_site;

typedef struct { uint16_t cov; uint16_t n[4]; } msnv_site_sample;

/* Gene / codon annotation of one site (snpCall -g, call_vC.cpp:567-574,604-633), computed on the device.
 * gene = index of the annotation row (file order, rows with start > end excluded) or -1.
 * codon[x] for allele x (A,C,G,T): {flags, len_old | len_new << 4, old[3], new[3]}. */
#define MSNV_ANN_VALID      1
#define MSNV_ANN_CIRCULAR   2   /* gene with start == end: the reference drops the allele */
#define MSNV_ANN_SYNONYMOUS 4
typedef struct { int32_t gene; uint8_t codon[4][8]; } msnv_site_ann;

/* ------------------------------------------------------------------------------------
 * Multi-GPU form of the two invocations above (the reference's counterpart is its pool of
 * processes, metaSNV.py:55-78,196-215, where every split process inflates every BAM in full).  The BAMs are dealt to the
 * ranks for decoding -- every file is inflated exactly once in the whole job -- each decoder deals the records to the
 * rank that owns their contig (msnv_records_partition; the exchange is an all-to-all over RCCL, metasnv_amd/parallel.py),
 * every rank runs the kernels over its contigs, and rank 0 receives the small tables: site records (msnv_results_fetch ->
 * msnv_write_calls_records) and coverage accumulators (msnv_coverage_fetch -> msnv_write_coverage_records).
 * ------------------------------------------------------------------------------------ */
/* qaCompute's read bookkeeping for the "Other" block of OUT (qaCompute.cpp:461-473,518-526,642-654). */
typedef struct { uint32_t total_reads, unmapped, zero_quality, proper_pairs, duplicates, any_mapped; } msnv_sample_stats;
#define MSNV_COV_WORDS 17   /* per (sample, contig): covSum, hist[0..15] (qaCompute.cpp:142-165) */

/* Deals one sample's raw record stream to n_parts parts: a mapped record goes to part contig_owner[tid] (-1 = nobody),
 * order preserved; unmapped records are counted and dropped.  out: n_bytes bytes, receives the parts back to back;
 * part_bytes[n_parts] their sizes; stats (may be NULL) the sample's qaCompute statistics over ALL records. */
int  msnv_records_partition(const uint8_t *records, uint64_t n_bytes, const int32_t *contig_owner, int32_t n_contigs,
                            int32_t n_parts, int32_t cov_min_mapq, uint8_t *out, uint64_t *part_bytes, msnv_sample_stats *stats);
/* msnv_records_partition for n streams at once, ON THE DEVICE (csrc/devpack.hip): the streams (host memory, or device memory when on_device
 * != 0) are dealt by kernels into `out` -- DEVICE memory of `capacity` bytes, e.g. the send tensor of an all-to-all over RCCL -- destination-
 * major: part 0 of stream 0, of stream 1, ..., part 1 of stream 0, ..., with `gap` bytes left free in front of every part (the caller's size
 * table; capacity >= sum of n_bytes + n_parts * gap).  part_bytes[i * n_parts + k] = bytes of stream i in part k; stats[i] = stream i's
 * qaCompute statistics; contig_bases (may be NULL; n_contigs entries, not cleared) += aligned bases per contig (msnv_records_contig_bases). */
/* ... for BAM FILES: their BGZF blocks are inflated and CRC-checked on the device and the record streams dealt from there (no host copy of the
 * inflated bytes); the headers are checked against the dataset's contigs.  record_bytes[i] = bytes of file i's record stream.  The files of a
 * call must fit ONE batch of the device inflate (1 GB of BAM; MSNV_EDOMAIN otherwise: the caller takes the host route). */
int  msnv_dataset_deal_bams_device(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads, const int32_t *contig_owner, int32_t n_parts,
                                   int32_t cov_min_mapq, uint8_t *out, uint64_t capacity, uint64_t gap, uint64_t *part_bytes, msnv_sample_stats *stats,
                                   uint64_t *record_bytes);
/* ... before the owners are known (the split planner holds decoded rounds): the files' record streams, inflated and checked on the device, are
 * left in `out` (DEVICE memory) -- stream i at out + rec_off[i], rec_bytes[i] long, 16 readable bytes behind it -- with their statistics and the
 * aligned bases per contig (contig_bases, n_contigs of the dataset, not cleared); msnv_records_deal_device (on_device != 0) deals them later. */
int  msnv_dataset_inflate_bams_device(msnv_dataset *ds, const char *const *bam_paths, int32_t n, int32_t host_threads, uint8_t *out, uint64_t capacity,
                                      uint64_t *rec_off, uint64_t *rec_bytes, msnv_sample_stats *stats, uint64_t *contig_bases);
/* A second context OF THE SAME DEVICE for the two calls above (NULL: back to the dataset's own): its stream, staging buffers and pinned
 * words carry the upload, inflate, CRC check and dealing of round k + 1 of the N-rank feed while the dataset's own context packs round k
 * (one thread per context) -- the reference overlaps its splits as concurrent processes, each of which reads every BAM
 * (/root/reference/metaSNV.py:196-215); here a round's files are read once, by the feed.  Nothing of the dataset is written through it. */
int  msnv_dataset_set_feed_ctx(msnv_dataset *ds, msnv_ctx *ctx);
int  msnv_records_deal_device(msnv_ctx *ctx, const uint8_t *const *streams, const uint64_t *n_bytes, int32_t n, int32_t on_device, const int32_t *contig_owner,
                              int32_t n_contigs, int32_t n_parts, int32_t cov_min_mapq, uint8_t *out, uint64_t capacity, uint64_t gap, uint64_t *part_bytes,
                              msnv_sample_stats *stats, uint64_t *contig_bases);
/* Adds the aligned (M/=/X) bases of every mapped record of a raw record stream to bases[tid] (n_contigs entries, not cleared):
 * the weight of the reference's split rule, genome length x coverage = aligned bases (src/createOptimumSplit.py:46-50), which
 * the N-rank driver takes from its first round of decoded BAMs before it fixes the contig owners. */
int  msnv_records_contig_bases(const uint8_t *records, uint64_t n_bytes, int32_t n_contigs, uint64_t *bases);
/* Statistics of sample `sample_idx` as counted while it was packed (only over the records this dataset was given). */
int  msnv_dataset_sample_stats(const msnv_dataset *ds, int32_t sample_idx, msnv_sample_stats *out);
/* Accumulators of the last msnv_coverage_run: acc[n_samples][n_contigs][MSNV_COV_WORDS], zeros for contigs outside the shard. */
int  msnv_coverage_fetch(msnv_dataset *ds, uint64_t *acc, uint64_t capacity_words);
/* The same accumulators as ROWS: one per (sample, contig) that has reads passing qaCompute's filter on this rank, in (sample, contig)
 * order -- what the device keeps and what a rank ships to rank 0.  The dense table above is n_samples x n_contigs x 136 bytes whatever
 * it holds (500 BAMs over a database of a million contigs: 68 GB); qaCompute itself prints a row per header contig per BAM
 * (qaCompute.cpp:226-263 for the ones without reads), which the writers below still do. */
int  msnv_coverage_rows_count(const msnv_dataset *ds, uint64_t *n_rows);
int  msnv_coverage_fetch_rows(msnv_dataset *ds, uint32_t *sample, uint32_t *contig, uint64_t *acc, uint64_t capacity_rows);
/* Writes OUT / OUT.detail of one sample from gathered accumulators acc[n_contigs][MSNV_COV_WORDS] exactly as
 * msnv_write_coverage does (qaCompute.cpp:192-217,226-263,439,623-657). */
int  msnv_write_coverage_records(const msnv_ref_desc *ref, int32_t max_cov, const msnv_sample_stats *stats, const uint64_t *acc,
                                 const char *cov_path, const char *detail_path);

/* First pileup line of this dataset's invocation (the one call_vC.cpp:423 drops); tid = -1 if no read passes. */
int  msnv_dataset_first_line(const msnv_dataset *ds, int32_t *tid, int32_t *pos);
/* Per contig (n = header contigs): position of the first pileup line of an invocation that starts at that contig, -1 when
 * no read of this dataset piles up there -- without -l (first_any) and under metaSNV's split BED `name 1 LEN`, which
 * excludes position 0 (first_from1; metaSNV.py:92).  Lets a driver that holds ONE resident dataset write every
 * best_split_K output with the first line that split's own snpCall process would have dropped (call_vC.cpp:423).
 * Only valid for datasets created without a BED restriction. */
int  msnv_dataset_first_lines(const msnv_dataset *ds, int32_t *first_any, int32_t *first_from1, int32_t n);
/* Formats site records that did not come from a local run (multi-GPU: records gathered from the
 * ranks over RCCL) exactly as msnv_write_calls does.  Records must be in (tid, pos) order. */
int  msnv_write_calls_records(const msnv_ref_desc *ref, int32_t n_samples, const msnv_site *sites,
                              const msnv_site_sample *samples, uint64_t n_sites,
                              const char *called_path, const char *indiv_path,
                              const char *ann_path, const char *fasta_path, const msnv_site_ann *ann);

/* Runs the annotation kernel over the sites of the last msnv_pileup_run.  The gene table and the codon genome are
 * parsed and uploaded on the first call for a given (ann_path, fasta_path) and stay resident; ms_kernel may be NULL. */
int  msnv_annotate_run(msnv_dataset *ds, const char *ann_path, const char *fasta_path, double *ms_kernel);
/* ann[n_sites] in the order of msnv_results_fetch. */
int  msnv_results_fetch_ann(msnv_dataset *ds, msnv_site_ann *ann, uint64_t capacity);

int  msnv_results_count(const msnv_dataset *ds, uint64_t *n_sites);
/* sites[n_sites], samples[n_sites * n_samples], both in (tid, pos) order. */
int  msnv_results_fetch(msnv_dataset *ds, msnv_site *sites, msnv_site_sample *samples, uint64_t capacity);
/* The same records with the per-sample part as ROWS OF CELLS (CSR) instead of n_samples entries per site: site i owns
 * cells [row_off[i], row_off[i + 1]) (row_off has n_sites + 1 entries), cell c belongs to sample cell_sample[c] (ascending
 * inside a row) and samples without a cell hold zeros.  This is what the device keeps (a cell per sample that has reads in the
 * site's tile) and what the ranks ship to rank 0: a 500-sample cohort whose species are each carried by a handful of samples
 * (BASELINE configs[3]) costs a handful of cells per site, where the dense form costs 5 KB.  The reference's counterpart is
 * the `c1|c2|...|cS` text of every called line (call_vC.cpp:316-325,635), S numbers whatever they are. */
int  msnv_results_cells_count(const msnv_dataset *ds, uint64_t *n_sites, uint64_t *n_cells);
int  msnv_results_fetch_cells(msnv_dataset *ds, msnv_site *sites, uint64_t *row_off, uint32_t *cell_sample, msnv_site_sample *cells,
                              uint64_t cap_sites, uint64_t cap_cells);
/* msnv_write_calls_records for records in the cell form (same text, byte for byte). */
int  msnv_write_calls_cells(const msnv_ref_desc *ref, int32_t n_samples, const msnv_site *sites, const uint64_t *row_off,
                            const uint32_t *cell_sample, const msnv_site_sample *cells, uint64_t n_sites,
                            const char *called_path, const char *indiv_path,
                            const char *ann_path, const char *fasta_path, const msnv_site_ann *ann);

/* ------------------------------------------------------------------------------------
 * Host I/O helpers used by the Python CLI and the tests (BGZF/BAM on zlib: htslib is not
 * available in the build image).
 * ------------------------------------------------------------------------------------ */
/* Host-stage counters since the library was loaded: BGZF blocks that the library's own DEFLATE decoder (csrc/inflate.cpp)
 * handed to zlib (0 for well-formed files; tests). */
int  msnv_host_stats(uint64_t *zlib_fallbacks);
/* Host-stage timers, cumulative seconds since the library was loaded or the last reset: [0] file reads, [1] host inflate + CRC checks,
 * [2] device inflate (wall: H2D, kernel, D2H), [3] parse + pack, [4] msnv_dataset_finalize (index build + upload, wall),
 * [5] text formatting of called_SNPs / indiv_called (wall), [6] msnv_dataset_add_sample_bams (wall).  [0], [1] and [3] are summed over
 * the host threads that did the work.  What the end-to-end block of bench.py splits a metaSNV.py run into; the reference's
 * counterpart is the wall time of its process pool (metaSNV.py:55-78,196-221). */
int  msnv_host_timers(double *seconds, int32_t n, int32_t reset);
/* The alignment-record streams of several BAM files in one call (the N-rank driver reads a round of files and deals their records to
 * the ranks that own the contigs: msnv_records_partition): through the device BGZF inflate when ctx is given and the files bring
 * >= 64 MB (MSNV_INFLATE overrides), else one host thread per file.  records[i], n_bytes[i]: the records behind the header of
 * bam_paths[i]; every records[i] is released with msnv_free.  Replaces the read loop `sam_read1` of qaCompute.cpp:441 / samtools. */
int  msnv_bam_records_many(msnv_ctx *ctx, const char *const *bam_paths, int32_t n, int32_t host_threads, uint8_t **records, uint64_t *n_bytes);
/* Test / measurement hook of the device BGZF inflate (csrc/inflate_k.hip): the inflated bytes of one BGZF
 * file, through the device (on_device != 0, needs ctx) or the host decoder.  *out is released with msnv_free.  counters (may be
 * NULL): [0] blocks, [1] blocks the device refused and the host inflated, [2] kernel microseconds, [3] inflated bytes. */
int  msnv_bgzf_inflate(msnv_ctx *ctx, const char *path, int32_t on_device, uint8_t **out, uint64_t *n_out, uint64_t counters[4]);
/* `samtools view -H` replacement for bed_header (metaSNV.py:81-94): writes SN\t1\tLN lines. */
int  msnv_bam_write_bed_header(const char *bam_path, const char *out_path);
/* Cores' worth of CPU time this process may use: the hardware threads, or the container's quota (cgroup cpu.max) when that is less. */
int32_t msnv_host_cores(void);
/* Reads a whole BAM: header text, contigs and the raw record stream.  Free with msnv_free. */
typedef struct {
    int32_t   n_contigs;
    char    **names;
    int64_t  *lengths;
    uint8_t  *records;
    uint64_t  n_record_bytes;
    char     *header_text;
} msnv_bam_data;
int  msnv_bam_read(const char *bam_path, msnv_bam_data *out);
/* ... the header alone (records = NULL): only the leading BGZF blocks are inflated (what `samtools view -H` reads, metaSNV.py:88). */
int  msnv_bam_read_header(const char *bam_path, msnv_bam_data *out);
void msnv_bam_data_free(msnv_bam_data *d);
/* Writes a BAM (BGZF) from a header and a raw record stream (synthetic inputs, tests). */
int  msnv_bam_write(const char *bam_path, const char *header_text, int32_t n_contigs,
                    const char *const *names, const int64_t *lengths,
                    const uint8_t *records, uint64_t n_record_bytes, int32_t compress_level);

/* ------------------------------------------------------------------------------------
 * Synthetic workload generator (SURVEY.md section 8d "testdata" shape).  Deterministic in
 * `seed`.  Produces the reference sequences and, per sample, a raw BAM record stream.
 * ------------------------------------------------------------------------------------ */
typedef struct {
    int32_t  n_species;          /* contigs refGenome{1..n}clus, one contig each            */
    int64_t  contig_len;
    int32_t  n_samples;
    int32_t  read_len;
    double   mean_cov;           /* per (sample, species): LogNormal(ln mean_cov, sigma)     */
    double   sigma_cov;
    double   frac_absent;        /* (sample, species) pairs with zero coverage               */
    double   snv_density;        /* subspecies SNV sites per position                        */
    double   error_rate;
    double   frac_lowq;          /* bases with BQ in [2,12]                                  */
    double   frac_indel_reads;   /* reads carrying one small I or D                          */
    double   frac_clip_reads;    /* reads with a leading soft/hard clip                      */
    double   frac_flagged;       /* DUP / SECONDARY / QCFAIL / mapq 0 reads (each)           */
    int32_t  lowercase_ref;      /* 1 = soft-mask 5 % of the reference (lower-case)           */
    uint64_t seed;
    double   frac_paired;        /* 0 (default, the BASELINE testdata shape is single-end): fraction of read starts that
                                    become a proper pair whose mates mostly overlap on the reference               */
    int32_t  contigs_per_species_max;   /* 0 / 1: one contig per species.  > 1 (BASELINE configs[2] / [3]): a species has
                                    1 .. max contigs `refGenome<k>clus.c<j>` that share contig_len bases           */
    int32_t  species_per_sample; /* 0: presence by frac_absent.  > 0: every sample draws that many random species and keeps
                                    each of them with probability 1 - frac_absent (fractional species counts of scaled shards) */
    double   frac_aux;           /* records that carry auxiliary fields behind the qualities, as every aligner writes them
                                    (NM:C, MD:Z, AS:i -- 18 bytes); 0 = none (the default workload's bytes do not change)       */
    double   frac_noseq;         /* records with SEQ `*` (l_seq = 0, CIGAR kept): in the pileup every base of such a read is N
                                    with quality 0; qaCompute counts its M blocks like any other                               */
} msnv_synth_params;

void msnv_synth_params_default(msnv_synth_params *p);
/* Number of contigs the parameters describe (= n_species unless contigs_per_species_max > 1). */
int  msnv_synth_contig_count(const msnv_synth_params *p);
/* Reference sequences: caller frees each seqs[i] and the arrays with msnv_free. */
int  msnv_synth_reference(const msnv_synth_params *p, char ***names, int64_t **lengths, char ***seqs);
/* One sample's raw BAM record stream (coordinate sorted). */
int  msnv_synth_sample(const msnv_synth_params *p, int32_t sample_idx, char *const *seqs,
                       uint8_t **records, uint64_t *n_bytes);
/* Generates samples [first, first+count) with a host thread pool and appends them to `ds`
 * (same bytes as msnv_synth_sample + msnv_dataset_add_sample_records, without the copies). */
int  msnv_dataset_add_synth_samples(msnv_dataset *ds, const msnv_synth_params *p, int32_t first, int32_t count,
                                    int32_t host_threads);
void msnv_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
