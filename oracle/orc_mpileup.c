/*
 * oracle/orc_mpileup.c -- TEST INFRASTRUCTURE (see orc.h).  PARITY UNPINNED.
 *
 * CPU restatement of the text that
 *     samtools mpileup -f REF [-l BED] -B -b ALL_SAMPLES          (metaSNV.py:160-165)
 * pipes into snpCall.  samtools/htslib are third-party, absent from /root/reference and
 * from this image, and the reference pins no version; this file restates the published
 * algorithm of samtools/htslib >= 1.10 (bam_plcmd.c: mplp_func, mpileup text loop,
 * pileup_seq; htslib sam.c: bam_plp_push, bam_plp_next, resolve_cigar2, bam_endpos) as
 * summarised in SURVEY.md Appendix C, including the overlapping-mate quality tweak that
 * mpileup applies unless -x is given (bam_mplp_init_overlaps; sam.c overlap_push,
 * tweak_overlap_quality, cigar_iref2iseq_set / _next) -- metaSNV.py:160-165 passes no -x.
 * The tweak is restated LITERALLY here (cursor functions and loop as in sam.c); the product
 * (metasnv_amd/csrc/pack.cpp) uses an independent closed form of the same loop.
 * Not modelled: htslib applies the tweak when the SECOND mate is pushed, which (one read of look-ahead) can be before or
 * after a deletion element of the first mate was printed; such an element prints the quality of the base behind the
 * deletion, so its quality character -- and whether -Q keeps the '*' -- may differ from samtools' text.  snpCall ignores
 * '*' (call_vC.cpp:517-521), so no count depends on it.
 */
#include "orc.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

void orc_set_error(const char *msg);

#define BAM_FPAIRED        1
#define BAM_FPROPER_PAIR   2
#define BAM_FUNMAP         4
#define BAM_FREVERSE      16

enum { CM = 0, CI = 1, CD = 2, CN = 3, CS = 4, CH = 5, CP = 6, CEQ = 7, CX = 8 };

static const char k_nt16_str[] = "=ACMGRSVTWYHKDBN";

/* htslib hts.c seq_nt16_table */
static unsigned nt16_of_char(unsigned char c) {
    switch (c) {
        case '=': return 0;
        case '0': return 1; case '1': return 2; case '2': return 4; case '3': return 8;
        case 'A': case 'a': return 1;  case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;  case 'T': case 't': return 8;
        case 'M': case 'm': return 3;  case 'R': case 'r': return 5;
        case 'S': case 's': return 6;  case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;  case 'Y': case 'y': return 10;
        case 'H': case 'h': return 11; case 'K': case 'k': return 12;
        case 'D': case 'd': return 13; case 'B': case 'b': return 14;
        default: return 15;
    }
}

static int32_t  rd_i32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }

/* One alignment record, fields of SAM spec 4.2 */
typedef struct {
    int32_t tid, pos, l_seq;
    uint16_t flag; uint32_t n_cigar;    /* (the count of the CG field when the CIGAR lives there: orc_resolve_cg) */
    uint8_t mapq;
    const uint8_t *cigar, *seq, *qual;
    const char *qname; int32_t mtid, mpos, isize;
    int in_olap;            /* this read is the value of its qname's entry in the overlap hash */
    int64_t end;            /* bam_endpos */
    /* resolve_cigar2 cursor */
    int k; int64_t x; int32_t y;
} read_t;

static uint32_t cig(const read_t *r, int k) { return rd_u32(r->cigar + 4 * (size_t)k); }

static int parse_record(const uint8_t *p, uint64_t avail, read_t *r, uint64_t *consumed) {
    int32_t bs; uint32_t l_name;
    int k; int64_t rlen = 0;
    if (avail < 4) return -1;
    bs = rd_i32(p);
    if (bs < 32 || (uint64_t)bs + 4 > avail) return -1;
    r->tid = rd_i32(p + 4); r->pos = rd_i32(p + 8);
    l_name = p[12]; r->mapq = p[13];
    r->n_cigar = rd_u16(p + 16); r->flag = rd_u16(p + 18);
    r->l_seq = rd_i32(p + 20);
    r->mtid = rd_i32(p + 24); r->mpos = rd_i32(p + 28); r->isize = rd_i32(p + 32);
    r->qname = (const char *)(p + 36); r->in_olap = 0;
    r->cigar = p + 36 + l_name;
    r->seq = r->cigar + 4 * (size_t)r->n_cigar;
    r->qual = r->seq + ((size_t)r->l_seq + 1) / 2;
    if ((uint64_t)(r->qual + r->l_seq - p) > (uint64_t)bs + 4) return -1;
    { unsigned nc = r->n_cigar; orc_resolve_cg(r->qual + r->l_seq, p + 4 + (size_t)bs, r->tid, r->pos, r->l_seq, &r->cigar, &nc); r->n_cigar = nc; }   /* bam_read1 -> bam_tag2cigar */
    for (k = 0; k < (int)r->n_cigar; ++k) {          /* bam_cigar2rlen */
        uint32_t c = cig(r, k); int op = c & 15;
        if (op == CM || op == CD || op == CN || op == CEQ || op == CX) rlen += c >> 4;
    }
    if (r->flag & BAM_FUNMAP) rlen = 0;                                  /* bam_endpos */
    if (rlen == 0) rlen = 1;
    r->end = (int64_t)r->pos + rlen;
    r->k = -1; r->x = r->pos; r->y = 0;
    *consumed = (uint64_t)bs + 4;
    return 0;
}

static int has_ref_op(const read_t *r) {
    int k;
    for (k = 0; k < (int)r->n_cigar; ++k) { int op = cig(r, k) & 15; if (op == CM || op == CD || op == CN || op == CEQ || op == CX) return 1; }
    return 0;
}

/* ------------------------------------------------------------ per-sample iterator */
typedef struct {
    const orc_sample *s;
    const uint8_t *rec;      /* the sample's records: a private mutable copy when the overlap tweak is on */
    uint8_t *copy;
    uint64_t off;
    read_t   peek; int has_peek;
    read_t  *act; int n_act, cap_act;
    int      first_push_done;
} iter_t;

/* ------------------------------------------------------------ overlapping mates (sam.c) */
/* cigar_iref2iseq_set: find the first CMATCH at or after *iref, set the cursor.  Returns 0 (BAM_CMATCH), -1 no more / not covered. */
static int iref2iseq_set(const read_t *r, int *kc, int64_t *icig, int64_t *iseq, int64_t *iref) {
    int64_t pos = *iref;
    if (pos < 0) return -1;
    *icig = 0; *iseq = 0; *iref = 0; *kc = 0;
    while (*kc < (int)r->n_cigar) {
        uint32_t c = cig(r, *kc); int op = c & 15; int64_t n = c >> 4;
        if (op == CS) { (*kc)++; *iseq += n; *icig = 0; continue; }
        if (op == CH || op == CP) { (*kc)++; *icig = 0; continue; }
        if (op == CM || op == CEQ || op == CX) {
            pos -= n;
            if (pos < 0) { *icig = n + pos; *iseq += *icig; *iref += *icig; return 0; }
            (*kc)++; *iseq += n; *icig = 0; *iref += n;
            continue;
        }
        if (op == CI) { (*kc)++; *iseq += n; *icig = 0; continue; }
        if (op == CD || op == CN) {
            pos -= n;
            if (pos < 0) pos = 0;
            (*kc)++; *icig = 0; *iref += n;
            continue;
        }
        return -2;
    }
    *iseq = -1;
    return -1;
}
/* cigar_iref2iseq_next: the next CMATCH base */
static int iref2iseq_next(const read_t *r, int *kc, int64_t *icig, int64_t *iseq, int64_t *iref) {
    while (*kc < (int)r->n_cigar) {
        uint32_t c = cig(r, *kc); int op = c & 15; int64_t n = c >> 4;
        if (op == CM || op == CEQ || op == CX) {
            if (*icig >= n - 1) { *icig = -1; (*kc)++; continue; }
            (*iseq)++; (*icig)++; (*iref)++;
            return 0;
        }
        if (op == CD || op == CN) { (*kc)++; *iref += n; *icig = -1; continue; }
        if (op == CI) { (*kc)++; *iseq += n; *icig = -1; continue; }
        if (op == CS) { (*kc)++; *iseq += n; *icig = -1; continue; }
        if (op == CH || op == CP) { (*kc)++; *icig = -1; continue; }
        return -2;
    }
    *iseq = -1; *iref = -1;
    return -1;
}
static int seqi_raw(const read_t *r, int64_t i) { return (r->seq[i >> 1] >> ((~i & 1) << 2)) & 0xf; }

/* tweak_overlap_quality(a, b): a = the mate that was pushed first, b = the one being pushed */
static void tweak_overlap_quality(read_t *a, read_t *b) {
    uint8_t *a_qual = (uint8_t *)a->qual, *b_qual = (uint8_t *)b->qual;
    int a_k, b_k; int64_t a_icig, a_iseq, b_icig, b_iseq;
    int64_t iref = b->pos, a_iref = iref - a->pos, b_iref = iref - b->pos;
    int a_ret = iref2iseq_set(a, &a_k, &a_icig, &a_iseq, &a_iref);
    int b_ret;
    if (a_ret < 0) return;
    b_ret = iref2iseq_set(b, &b_k, &b_icig, &b_iseq, &b_iref);
    if (b_ret < 0) return;
    for (;;) {
        while (a_ret >= 0 && a_iref >= 0 && a_iref < iref - a->pos) a_ret = iref2iseq_next(a, &a_k, &a_icig, &a_iseq, &a_iref);
        if (a_ret < 0) break;
        if (iref < a_iref + a->pos) iref = a_iref + a->pos;
        while (b_ret >= 0 && b_iref >= 0 && b_iref < iref - b->pos) b_ret = iref2iseq_next(b, &b_k, &b_icig, &b_iseq, &b_iref);
        if (b_ret < 0) break;
        if (iref < b_iref + b->pos) iref = b_iref + b->pos;
        iref++;
        if (a_iref + a->pos != b_iref + b->pos) continue;      /* only CMATCH positions */
        if (a_iseq > a->l_seq || b_iseq > b->l_seq) return;
        if (a_iseq >= a->l_seq || b_iseq >= b->l_seq) return;  /* (SEQ '*': nothing to tweak) */
        if (seqi_raw(a, a_iseq) == seqi_raw(b, b_iseq)) {
            int qual = a_qual[a_iseq] + b_qual[b_iseq];        /* very confident about this base */
            a_qual[a_iseq] = (uint8_t)(qual > 200 ? 200 : qual);
            b_qual[b_iseq] = 0;
        } else if (a_qual[a_iseq] >= b_qual[b_iseq]) {
            a_qual[a_iseq] = (uint8_t)(0.8 * a_qual[a_iseq]);  /* not so confident about a_qual anymore given the mismatch */
            b_qual[b_iseq] = 0;
        } else {
            b_qual[b_iseq] = (uint8_t)(0.8 * b_qual[b_iseq]);
            a_qual[a_iseq] = 0;
        }
    }
}

/* overlap_push: called for the node bam_plp_push has just accepted (t->act[t->n_act - 1]).
 * The hash (qname -> node) is modelled as holding a read while it is alive in this restatement's pileup buffer, i.e.
 * while end > start of the read being pushed; htslib drops the entry a few pushes later (when the position behind the
 * read's end has been emitted), which can only matter for templates with three or more alignments in the file. */
static void overlap_push(iter_t *t) {
    read_t *node = &t->act[t->n_act - 1];
    int a;
    if ((node->flag & 8 /* MUNMAP */) || !(node->flag & BAM_FPROPER_PAIR)) return;
    if ((node->mtid >= 0 && node->tid != node->mtid) ||
        (llabs((long long)node->isize) >= 2 * (long long)node->l_seq && node->mpos >= node->end)) return;   /* no overlap possible */
    for (a = 0; a < t->n_act - 1; ++a) {
        read_t *x = &t->act[a];
        if (x->in_olap && strcmp(x->qname, node->qname) == 0) {
            tweak_overlap_quality(x, node);
            x->in_olap = 0;
            return;
        }
    }
    if (node->mpos >= node->pos || ((node->flag & BAM_FPAIRED) && node->mpos == -1)) node->in_olap = 1;   /* only reads whose mate is still to arrive */
}

static int bed_overlap(const orc_mpileup_opts *o, int tid, int64_t beg, int64_t end) {
    int i;
    for (i = 0; i < o->n_bed; ++i)
        if (o->bed_tid[i] == tid && o->bed_beg[i] < end && beg < o->bed_end[i]) return 1;
    return 0;
}

/* bam_plcmd.c mplp_func: fetch the next read that survives the read-level filters */
static int fetch(iter_t *it, const orc_ref *ref, const orc_mpileup_opts *o) {
    it->has_peek = 0;
    while (it->off < it->s->n_bytes) {
        read_t r; uint64_t used;
        if (parse_record(it->rec + it->off, it->s->n_bytes - it->off, &r, &used)) { orc_set_error("corrupt BAM record"); return ORC_ERR_FORMAT; }
        it->off += used;
        if (r.tid < 0 || (r.flag & BAM_FUNMAP)) continue;
        if (o->flag_filter & r.flag) continue;
        if (o->n_bed && !bed_overlap(o, r.tid, r.pos, r.end)) continue;
        if (r.tid < ref->n_contigs && ref->seqs && ref->seqs[r.tid] && ref->seq_lens[r.tid] <= r.pos) continue;   /* "outside of the reference" */
        if (r.mapq < o->min_mapq) continue;
        if (!o->count_orphans && (r.flag & BAM_FPAIRED) && !(r.flag & BAM_FPROPER_PAIR)) continue;
        if (!has_ref_op(&r)) continue;      /* no pileup element can ever be produced (resolve_cigar2 asserts) */
        it->peek = r; it->has_peek = 1;
        return ORC_OK;
    }
    return ORC_OK;
}

/* htslib sam.c resolve_cigar2, restated for one (read, pos).  Returns 1 if an element exists. */
typedef struct { int qpos, indel, is_del, is_head, is_tail, is_refskip; } elem_t;

static int resolve(read_t *r, int64_t pos, elem_t *p) {
    int k;
    if (r->k == -1) {                        /* first ref-consuming op */
        r->x = r->pos; r->y = 0;
        for (k = 0; k < (int)r->n_cigar; ++k) {
            uint32_t c = cig(r, k); int op = c & 15;
            if (op == CM || op == CD || op == CN || op == CEQ || op == CX) break;
            else if (op == CI || op == CS) r->y += (int32_t)(c >> 4);
        }
        r->k = k;
    }
    /* advance to the op that contains pos */
    for (;;) {
        uint32_t c = cig(r, r->k); int op = c & 15; int64_t l = c >> 4;
        int refc = (op == CM || op == CD || op == CN || op == CEQ || op == CX);
        if (refc && pos < r->x + l) break;
        if (refc) r->x += l;
        if (op == CM || op == CI || op == CS || op == CEQ || op == CX) r->y += (int32_t)l;
        r->k++;
        if (r->k >= (int)r->n_cigar) return 0;
    }
    {
        uint32_t c = cig(r, r->k); int op = c & 15; int64_t l = c >> 4;
        memset(p, 0, sizeof *p);
        if (op == CM || op == CEQ || op == CX) {
            p->qpos = r->y + (int)(pos - r->x);
            if (r->x + l - 1 == pos && r->k + 1 < (int)r->n_cigar) {
                uint32_t c2 = cig(r, r->k + 1); int op2 = c2 & 15; int l2 = (int)(c2 >> 4);
                if (op2 == CD) p->indel = -l2;
                else if (op2 == CI) p->indel = l2;
                else if (op2 == CP && r->k + 2 < (int)r->n_cigar) {
                    int l3 = 0;
                    for (k = r->k + 2; k < (int)r->n_cigar; ++k) {
                        c2 = cig(r, k); op2 = c2 & 15; l2 = (int)(c2 >> 4);
                        if (op2 == CI) l3 += l2;
                        else if (op2 == CD || op2 == CM || op2 == CN || op2 == CEQ || op2 == CX) break;
                    }
                    if (l3 > 0) p->indel = l3;
                }
            }
        } else {                              /* D or N */
            p->is_del = 1; p->qpos = r->y; p->is_refskip = (op == CN);
        }
        p->is_head = (pos == r->pos);
        p->is_tail = (pos == r->end - 1);
    }
    return 1;
}

typedef struct { char *p; size_t n, cap; } lbuf;
static void lb_need(lbuf *b, size_t extra) {
    if (b->n + extra + 1 > b->cap) { size_t nc = b->cap ? b->cap * 2 : 4096; while (nc < b->n + extra + 1) nc *= 2; b->p = (char *)realloc(b->p, nc); b->cap = nc; }
}
static void lb_putc(lbuf *b, char c) { lb_need(b, 1); b->p[b->n++] = c; }
static void lb_puts(lbuf *b, const char *s) { size_t l = strlen(s); lb_need(b, l); memcpy(b->p + b->n, s, l); b->n += l; }
static void lb_putd(lbuf *b, long v) { char t[32]; snprintf(t, sizeof t, "%ld", v); lb_puts(b, t); }

static int seqi(const read_t *r, int i) { return (r->seq[i >> 1] >> ((~i & 1) << 2)) & 0xf; }     /* bam_seqi */

/* bam_plcmd.c pileup_seq */
static void pileup_seq(lbuf *b, const read_t *r, const elem_t *p, int64_t pos, const char *ref, int64_t ref_len) {
    int rev = (r->flag & BAM_FREVERSE) != 0; int j;
    if (p->is_head) { lb_putc(b, '^'); lb_putc(b, (char)(r->mapq > 93 ? 126 : r->mapq + 33)); }
    if (!p->is_del) {
        int c = p->qpos < r->l_seq ? k_nt16_str[seqi(r, p->qpos)] : 'N';
        if (ref) {
            int rb = pos < ref_len ? ref[pos] : 'N';
            if (c == '=' || nt16_of_char((unsigned char)c) == nt16_of_char((unsigned char)rb)) c = rev ? ',' : '.';
            else c = rev ? tolower(c) : toupper(c);
        } else {
            if (c == '=') c = rev ? ',' : '.';
            else c = rev ? tolower(c) : toupper(c);
        }
        lb_putc(b, (char)c);
    } else lb_putc(b, p->is_refskip ? (rev ? '<' : '>') : '*');
    if (p->indel > 0) {
        lb_putc(b, '+'); lb_putd(b, p->indel);
        for (j = 1; j <= p->indel; ++j) {
            int c = (p->qpos + j < r->l_seq) ? k_nt16_str[seqi(r, p->qpos + j)] : 'N';
            lb_putc(b, (char)(rev ? tolower(c) : toupper(c)));
        }
    } else if (p->indel < 0) {
        lb_putc(b, '-'); lb_putd(b, -p->indel);
        for (j = 1; j <= -p->indel; ++j) {
            int c = (ref && pos + j < ref_len) ? ref[pos + j] : 'N';
            lb_putc(b, (char)(rev ? tolower(c) : toupper(c)));
        }
    }
    if (p->is_tail) lb_putc(b, '$');
}

void orc_mpileup_default_opts(orc_mpileup_opts *o) {
    memset(o, 0, sizeof *o);
    o->min_baseq = 13; o->flag_filter = 0x704; o->count_orphans = 0; o->max_depth = 8000; o->min_mapq = 0;
    o->ignore_overlaps = 0;      /* mpileup -x is not passed (metaSNV.py:160-165) */
}

/* statistics of the last orc_mpileup call (test/bench bookkeeping, not reference behaviour) */
static __thread uint64_t g_n_lines, g_n_bases;      /* per calling thread: bench.py times the restatement on every host core */
uint64_t orc_mpileup_last_lines(void) { return g_n_lines; }
uint64_t orc_mpileup_last_bases(void) { return g_n_bases; }

int orc_mpileup(const orc_ref *ref, const orc_sample *samples, int n_samples,
                const orc_mpileup_opts *opts, orc_line_cb cb, void *user) {
    iter_t *it = (iter_t *)calloc((size_t)n_samples, sizeof(iter_t));
    lbuf line = {0}, quals = {0};
    int i, rc = ORC_OK;
    int cur_tid = -1; int64_t cur_pos = -1;
    g_n_lines = g_n_bases = 0;
    if (!it) return ORC_ERR_NOMEM;
    for (i = 0; i < n_samples; ++i) {
        it[i].s = &samples[i]; it[i].rec = samples[i].records;
        if (!opts->ignore_overlaps && samples[i].n_bytes) {           /* the tweak edits qualities: work on a copy (bam_copy1 in bam_plp_push) */
            it[i].copy = (uint8_t *)malloc(samples[i].n_bytes);
            if (!it[i].copy) { rc = ORC_ERR_NOMEM; goto done; }
            memcpy(it[i].copy, samples[i].records, samples[i].n_bytes);
            it[i].rec = it[i].copy;
        }
        if ((rc = fetch(&it[i], ref, opts))) goto done;
    }

    for (;;) {
        /* next position = bam_mplp_auto: the smallest (tid,pos) any iterator reports */
        int have = 0, nt = 0; int64_t np = 0;
        for (i = 0; i < n_samples; ++i) {
            int a;
            for (a = 0; a < it[i].n_act; ++a)
                if (it[i].act[a].tid == cur_tid && it[i].act[a].end > cur_pos + 1) {
                    if (!have || cur_tid < nt || (cur_tid == nt && cur_pos + 1 < np)) { nt = cur_tid; np = cur_pos + 1; have = 1; }
                    break;
                }
            if (it[i].has_peek) {
                if (!have || it[i].peek.tid < nt || (it[i].peek.tid == nt && it[i].peek.pos < np)) { nt = it[i].peek.tid; np = it[i].peek.pos; have = 1; }
            }
        }
        if (!have) break;
        cur_tid = nt; cur_pos = np;

        {
            const char *rseq = (ref->seqs && cur_tid < ref->n_contigs) ? ref->seqs[cur_tid] : NULL;
            int64_t rlen = rseq ? ref->seq_lens[cur_tid] : 0;
            int in_bed = !opts->n_bed || bed_overlap(opts, cur_tid, cur_pos, cur_pos + 1);
            line.n = 0;
            lb_puts(&line, ref->names[cur_tid]); lb_putc(&line, '\t'); lb_putd(&line, (long)cur_pos + 1); lb_putc(&line, '\t');
            lb_putc(&line, (rseq && cur_pos < rlen) ? rseq[cur_pos] : 'N');

            for (i = 0; i < n_samples; ++i) {
                iter_t *t = &it[i];
                int a, w = 0, n_plp = 0, cnt = 0, nth = 0;
                /* drop finished reads (bam_plp_next) */
                for (a = 0; a < t->n_act; ++a)
                    if (!(t->act[a].tid < cur_tid || (t->act[a].tid == cur_tid && t->act[a].end <= cur_pos))) t->act[w++] = t->act[a];
                t->n_act = w;
                /* bam_plp_push: pull in every read that starts here */
                while (t->has_peek && t->peek.tid == cur_tid && t->peek.pos == cur_pos) {
                    /* sam.c bam_plp_push: "iter->tid == b->core.tid && iter->pos == b->core.pos &&
                       iter->mp->cnt > iter->maxcnt" -> read dropped.  The first read of a position was
                       pushed as look-ahead while iter->pos was still smaller, so it is never capped.
                       mp->cnt is restated sample-locally as "reads of this file still alive at this
                       position" (htslib frees finished nodes lazily, so its count can run slightly
                       higher); this only matters beyond 8000x per-sample depth. */
                    int capped = (nth > 0 || !t->first_push_done) && opts->max_depth > 0 && t->n_act > opts->max_depth;
                    if (!capped) {
                        if (t->n_act == t->cap_act) { t->cap_act = t->cap_act ? t->cap_act * 2 : 64; t->act = (read_t *)realloc(t->act, (size_t)t->cap_act * sizeof(read_t)); }
                        t->act[t->n_act++] = t->peek;
                        if (!opts->ignore_overlaps) overlap_push(t);
                    } else if (!opts->ignore_overlaps) {                 /* overlap_remove(iter, b) of a capped read: its qname leaves the hash */
                        for (a = 0; a < t->n_act; ++a) if (t->act[a].in_olap && strcmp(t->act[a].qname, t->peek.qname) == 0) t->act[a].in_olap = 0;
                    }
                    t->first_push_done = 1; ++nth;
                    if ((rc = fetch(t, ref, opts))) goto done;
                    if (t->has_peek && (t->peek.tid < cur_tid || (t->peek.tid == cur_tid && t->peek.pos < cur_pos))) { orc_set_error("BAM is not coordinate sorted"); rc = ORC_ERR_FORMAT; goto done; }
                }
                /* elements */
                lb_putc(&line, '\t');
                quals.n = 0;
                {
                    lbuf bases = {0};
                    for (a = 0; a < t->n_act; ++a) {
                        read_t *r = &t->act[a]; elem_t e;
                        if (r->tid != cur_tid || r->pos > cur_pos) continue;
                        if (!resolve(r, cur_pos, &e)) continue;
                        ++n_plp;
                        if (!e.is_del && in_bed) ++g_n_bases;
                        {
                            int q = e.qpos < r->l_seq ? r->qual[e.qpos] : 0;
                            if (q >= opts->min_baseq) {
                                int c = q + 33; if (c > 126) c = 126;
                                ++cnt;
                                pileup_seq(&bases, r, &e, cur_pos, rseq, rlen);
                                lb_putc(&quals, (char)c);
                            }
                        }
                    }
                    lb_putd(&line, cnt); lb_putc(&line, '\t');
                    if (n_plp == 0) lb_puts(&line, "*\t*");
                    else {
                        if (cnt == 0) lb_putc(&line, '*'); else { lb_need(&line, bases.n); memcpy(line.p + line.n, bases.p, bases.n); line.n += bases.n; }
                        lb_putc(&line, '\t');
                        if (cnt == 0) lb_putc(&line, '*'); else { lb_need(&line, quals.n); memcpy(line.p + line.n, quals.p, quals.n); line.n += quals.n; }
                    }
                    free(bases.p);
                }
            }
            lb_putc(&line, '\n');
            lb_need(&line, 1); line.p[line.n] = 0;
            if (in_bed) {
                ++g_n_lines;
                if (cb(user, line.p, line.n)) { rc = ORC_ERR_IO; goto done; }
            }
        }
    }
done:
    for (i = 0; i < n_samples; ++i) { free(it[i].act); free(it[i].copy); }
    free(it); free(line.p); free(quals.p);
    return rc;
}
