/*
 * oracle/orc.h -- CPU restatement ("oracle") of metaSNV's pileup / SNV-calling hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * there only as the checker.  The product path (metasnv_amd/ + libmsnv.so) never links,
 * imports or executes it.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - orc_snpcall_*   restates /root/reference/src/snpCaller/call_vC.cpp + gene.h.
 *                     The reference needs boost::icl, which this image lacks, so it is
 *                     UNBUILDABLE here and the restatement is checked only against the
 *                     known-answer vectors recorded in SURVEY.md Appendix E
 *                     (tests/golden/snpcall_E*).                    -> parity unpinned
 *   - orc_mpileup     restates `samtools mpileup -f REF [-l BED] -B -b LIST` (samtools is
 *                     a third-party binary, version unpinned by the reference, absent
 *                     here; call site metaSNV.py:160-165).           -> parity unpinned
 *   - orc_qacompute   restates /root/reference/src/qaTools/qaCompute.cpp for the only
 *                     invocation metaSNV uses (-c 10 -d -i, metaSNV.py:63-65); needs
 *                     htslib, absent here.                           -> parity unpinned
 *
 * Input convention: one sample = the concatenated *uncompressed* BAM alignment records
 * of a coordinate-sorted BAM file (every record starts with its int32 block_size, SAM
 * spec section 4.2), i.e. exactly what sam_read1() would hand to the reference tools.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors */
#define ORC_OK            0
#define ORC_ERR_IO        1
#define ORC_ERR_FORMAT    2
#define ORC_ERR_DOMAIN    3   /* input on which the reference itself crashes / has UB */
#define ORC_ERR_NOMEM     4

const char *orc_last_error(void);

/* --------------------------------------------------------------- reference */
typedef struct {
    int            n_contigs;
    const char   **names;     /* BAM header @SQ order                                  */
    const int64_t *lengths;   /* BAM header LN                                         */
    const char   **seqs;      /* FASTA characters (case preserved) or NULL if the contig
                                 is missing from the FASTA; seq_lens[i] valid chars    */
    const int64_t *seq_lens;
} orc_ref;

/* ------------------------------------------------------------------ sample */
typedef struct {
    const uint8_t *records;   /* concatenated raw BAM records */
    uint64_t       n_bytes;
} orc_sample;

/* ------------------------------------------------------- mpileup restatement */
typedef struct {
    int  min_baseq;      /* -Q, default 13                                             */
    int  flag_filter;    /* --ff, default 0x704 (UNMAP,SECONDARY,QCFAIL,DUP)            */
    int  count_orphans;  /* -A, default 0: PAIRED && !PROPER_PAIR reads are dropped     */
    int  max_depth;      /* -d, default 8000 per file                                   */
    int  min_mapq;       /* -q, default 0                                               */
    int  ignore_overlaps;/* -x, default 0: overlapping mates are quality-tweaked        */
    /* optional BED (-l): regions are 0-based half-open; NULL/0 = none                 */
    int             n_bed;
    const int      *bed_tid;
    const int64_t  *bed_beg;
    const int64_t  *bed_end;
} orc_mpileup_opts;

void orc_mpileup_default_opts(orc_mpileup_opts *o);

/* line sink: return non-zero to abort */
typedef int (*orc_line_cb)(void *user, const char *line, size_t len);

/* Emits the mpileup text, one callback per line (line includes the trailing '\n'). */
int orc_mpileup(const orc_ref *ref, const orc_sample *samples, int n_samples,
                const orc_mpileup_opts *opts, orc_line_cb cb, void *user);

/* ------------------------------------------------------- snpCall restatement */
typedef struct {
    int    min_coverage;          /* -c, default 4     call_vC.cpp:28  */
    int    calling_threshold;     /* -t, default 4     call_vC.cpp:29  */
    double calling_min_fraction;  /* -p, default 0.01  call_vC.cpp:30  */
    const char *fasta_path;       /* -f or NULL */
    const char *genes_path;       /* -g or NULL */
    int    token_cap;             /* 0 = the reference's 10000 (call_vC.cpp:482); a smaller value is "snpCall with a shorter token buffer":
                                     the parity fuzz uses it to reach the cut with shallow pileups (tests/fuzz_parity.py); never the reference's */
} orc_snpcall_opts;

void orc_snpcall_default_opts(orc_snpcall_opts *o);

typedef struct orc_snpcall orc_snpcall;

/* pop_out = the reference's stdout, indiv_out = its -i file (may be NULL). */
int  orc_snpcall_begin(orc_snpcall **out, const orc_snpcall_opts *opts,
                       FILE *pop_out, FILE *indiv_out);
/* Feed one mpileup line (with or without the trailing '\n' exactly as fgets would
 * deliver it).  The very first line fed is the one the reference drops. */
int  orc_snpcall_line(orc_snpcall *sc, const char *line, size_t len);
int  orc_snpcall_end(orc_snpcall *sc);

/* Convenience: whole stream from a FILE (what `snpCall < mpileup` does). */
int orc_snpcall_stream(const orc_snpcall_opts *opts, FILE *in, FILE *pop_out, FILE *indiv_out);

/* ----------------------------------------- fused: mpileup | snpCall in memory */
int orc_call(const orc_ref *ref, const orc_sample *samples, int n_samples,
             const orc_mpileup_opts *mopts, const orc_snpcall_opts *sopts,
             const char *pop_path, const char *indiv_path,
             uint64_t *n_lines_out, uint64_t *n_pileup_bases_out);

/* ---------------------------------------------------- qaCompute restatement */
/* `qaCompute -c max_cov -d -i BAM OUT` : writes OUT and OUT.detail. */
int orc_qacompute(const orc_ref *ref /* names+lengths only */, const orc_sample *sample,
                  int max_cov, int min_mapq, const char *cov_path, const char *detail_path);

#ifdef __cplusplus
}
#endif

/* A CIGAR of more than 65535 operations does not fit the record's 16-bit count: it travels in the auxiliary field CG:B,I behind a
 * placeholder CIGAR `<l_seq>S<ref_len>N`, and htslib puts it back when it reads the record (sam.c bam_tag2cigar, called by bam_read1
 * [EXT]) -- so both `samtools mpileup` and qaCompute (sam_read1, qaCompute.cpp:441) see the real one.  Restated: if the record is
 * mapped (tid, pos >= 0), has at least one operation, its first operation is a soft clip of exactly l_seq bases, and a CG field of type
 * B with subtype I or i holds at least as many operations as the placeholder (and fewer than 2^29), *cigar / *n_cigar become the
 * field's.  aux .. end: the bytes behind the qualities. */
static inline void orc_resolve_cg(const unsigned char *aux, const unsigned char *end, int tid, int pos, int l_seq, const unsigned char **cigar, unsigned *n_cigar) {
    const unsigned char *c0 = *cigar;
    unsigned first;
    if (*n_cigar == 0 || tid < 0 || pos < 0) return;
    first = (unsigned)c0[0] | (unsigned)c0[1] << 8 | (unsigned)c0[2] << 16 | (unsigned)c0[3] << 24;
    if ((first & 15u) != 4u || (int)(first >> 4) != l_seq) return;
    while (aux + 3 <= end) {                                   /* tag[2] type[1] value */
        const unsigned char t = aux[2];
        const unsigned char *v = aux + 3;
        unsigned long sz;
        if (t == 'A' || t == 'c' || t == 'C') sz = 1;
        else if (t == 's' || t == 'S') sz = 2;
        else if (t == 'i' || t == 'I' || t == 'f') sz = 4;
        else if (t == 'd') sz = 8;
        else if (t == 'Z' || t == 'H') { const unsigned char *q = v; while (q < end && *q) ++q; sz = (unsigned long)(q - v) + 1; }
        else if (t == 'B') {
            unsigned long es, n;
            if (v + 5 > end) return;
            es = (v[0] == 'c' || v[0] == 'C') ? 1 : (v[0] == 's' || v[0] == 'S') ? 2 : 4;
            n = (unsigned long)v[1] | (unsigned long)v[2] << 8 | (unsigned long)v[3] << 16 | (unsigned long)v[4] << 24;
            if (aux[0] == 'C' && aux[1] == 'G') {
                if ((v[0] != 'I' && v[0] != 'i') || n < *n_cigar || n >= (1ul << 29) || v + 5 + 4 * n > end) return;
                *cigar = v + 5; *n_cigar = (unsigned)n;
                return;
            }
            sz = 5 + es * n;
        } else return;                                         /* unknown type: the walk cannot go on */
        if (aux[0] == 'C' && aux[1] == 'G') return;             /* a CG field of another type: no real CIGAR there */
        if (v + sz > end) return;
        aux = v + sz;
    }
}

#endif
