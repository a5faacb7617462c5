/*
 * oracle/orc_snpcall.c -- TEST INFRASTRUCTURE (see orc.h).  PARITY UNPINNED.
 *
 * Line-by-line CPU restatement of the reference SNV caller
 *     /root/reference/src/snpCaller/call_vC.cpp  (+ gene.h)
 * mpileup text in -> called_SNPs (stdout of the reference) and indiv_called (-i file).
 * Every function cites the reference lines it follows.  Behaviour that is undefined or
 * crashes in the reference is reported as ORC_ERR_DOMAIN instead of being imitated.
 */
#include "orc.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#define TOK_CAP 10000          /* call_vC.cpp:482 "new char[10000]", toksplit lgh */
#define LINE_CAP 10000         /* call_vC.cpp:117,216 fgets(line,10000,...) */

/* the reference chats on stderr (call_vC.cpp:195,434,449,...); non-contractual, off by default */
static int orc_verbose(void) { static int v = -1; if (v < 0) v = getenv("ORC_VERBOSE") != NULL; return v; }
#define CHAT(...) do { if (orc_verbose()) fprintf(stderr, __VA_ARGS__); } while (0)

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
void orc_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

/* ---------------------------------------------------------------- helpers */
typedef struct { char *p; size_t n, cap; } sbuf;

static int sb_reserve(sbuf *s, size_t extra) {
    if (s->n + extra + 1 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 256;
        while (nc < s->n + extra + 1) nc *= 2;
        char *np = (char *)realloc(s->p, nc);
        if (!np) return -1;
        s->p = np; s->cap = nc;
    }
    return 0;
}
static void sb_clear(sbuf *s) { s->n = 0; if (s->p) s->p[0] = 0; }
static void sb_putc(sbuf *s, char c) { if (sb_reserve(s, 1)) return; s->p[s->n++] = c; s->p[s->n] = 0; }
static void sb_puts(sbuf *s, const char *t) { size_t l = strlen(t); if (sb_reserve(s, l)) return; memcpy(s->p + s->n, t, l); s->n += l; s->p[s->n] = 0; }
static void sb_putl(sbuf *s, long v) { char b[32]; snprintf(b, sizeof b, "%ld", v); sb_puts(s, b); }
static void sb_free(sbuf *s) { free(s->p); s->p = NULL; s->n = s->cap = 0; }

/* call_vC.cpp:92-111 toksplit: skips leading blanks, copies at most lgh chars, stops at
 * tokchar or NUL, steps over the separator. */
static const char *toksplit(const char *src, char tokchar, char *token, size_t lgh) {
    if (src) {
        while (' ' == *src) src++;
        while (*src && (tokchar != *src)) {
            if (lgh) { *token++ = *src; --lgh; }
            src++;
        }
        if (*src && (tokchar == *src)) src++;
    }
    *token = '\0';
    return src;
}

/* ------------------------------------------------ gene.h:3-25 codon table */
static const struct { const char *c; char aa; } k_codons[] = {
    {"TAA",'X'},{"TGA",'X'},{"TAG",'X'},
    {"GCT",'A'},{"GCC",'A'},{"GCA",'A'},{"GCG",'A'},
    {"CGT",'R'},{"CGC",'R'},{"CGA",'R'},{"CGG",'R'},{"AGA",'R'},{"AGG",'R'},
    {"AAT",'N'},{"AAC",'N'},
    {"GAT",'D'},{"GAC",'D'},
    {"TGT",'C'},{"TGC",'C'},
    {"CAA",'Q'},{"CAG",'Q'},
    {"GAA",'E'},{"GAG",'E'},
    {"GGT",'G'},{"GGC",'G'},{"GGA",'G'},{"GGG",'G'},
    {"CAT",'H'},{"CAC",'H'},
    {"ATT",'I'},{"ATC",'I'},{"ATA",'I'},
    {"TTA",'L'},{"TTG",'L'},{"CTT",'L'},{"CTC",'L'},{"CTA",'L'},{"CTG",'L'},
    {"AAA",'K'},{"AAG",'K'},
    {"ATG",'M'},
    {"TTT",'F'},{"TTC",'F'},
    {"CCT",'P'},{"CCC",'P'},{"CCA",'P'},{"CCG",'P'},
    {"TCT",'S'},{"TCC",'S'},{"TCA",'S'},{"TCG",'S'},{"AGT",'S'},{"AGC",'S'},
    {"ACT",'T'},{"ACC",'T'},{"ACA",'T'},{"ACG",'T'},
    {"TGG",'W'},
    {"TAT",'Y'},{"TAC",'Y'},
    {"GTA",'V'},{"GTG",'V'},{"GTT",'V'},{"GTC",'V'},
};
/* call_vC.cpp:455,627: std::map<std::string,char>::operator[] -> '\0' for unknown keys */
static char codon_aa(const char *codon) {
    size_t i;
    for (i = 0; i < sizeof k_codons / sizeof k_codons[0]; ++i)
        if (strcmp(k_codons[i].c, codon) == 0) return k_codons[i].aa;
    return '\0';
}

/* ------------------------------------------------ gene.h:42-102 Genome */
typedef struct {
    long      length;
    uint32_t *sequence;      /* (length/10)+1 words, 10 bases x 3 bits each */
} genome_t;

/* gene.h:28-36,67: map_baseToShort[] with operator[] -> unknown chars become 0 ('A') */
static unsigned base_to_short(char c) {
    switch (c) { case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3; case 'N': return 4; default: return 0; }
}
static const char k_int_to_base[8] = {'A','T','C','G','N','?','?','?'};   /* gene.h:28 */

static genome_t *genome_new(const char *seq, size_t len) {           /* gene.h:49-74 */
    genome_t *g = (genome_t *)calloc(1, sizeof *g);
    size_t i, ins = 0; int pos = 0; uint32_t repr = 0;
    if (!g) return NULL;
    g->length = (long)len;
    g->sequence = (uint32_t *)calloc(len / 10 + 1, sizeof(uint32_t));
    if (!g->sequence) { free(g); return NULL; }
    for (i = 0; i < len; ++i, ++pos) {
        if (pos == 10) { g->sequence[ins++] = repr; repr = 0; pos = 0; }
        repr |= base_to_short(seq[i]) << (3 * pos);
    }
    if (pos != 0) g->sequence[ins] = repr;
    return g;
}
static void genome_free(genome_t *g) { if (g) { free(g->sequence); free(g); } }

/* gene.h:79-92 getSequence(start,end) inclusive; "" when end<start or end>length.
 * Returns length written (0 = empty string). */
static int genome_get(const genome_t *g, long start, long end, char *out) {
    long i; int n = 0;
    if (end < start) { out[0] = 0; return 0; }
    if (end > g->length) { out[0] = 0; return 0; }
    for (i = start; i <= end; ++i)
        out[n++] = k_int_to_base[(g->sequence[i / 10] >> (3 * (i % 10))) & 7];
    out[n] = 0;
    return n;
}

/* ------------------------------------------ per-contig tables (std::map stand-ins) */
typedef struct { char *name; unsigned long start, lineCount; } genepos_t;   /* call_vC.cpp:48-52 */
typedef struct { char *name; genome_t *g; } genomeent_t;
typedef struct { long start, end; char *name; char strand; } gene_t;         /* gene.h:108-122 */

struct orc_snpcall {
    orc_snpcall_opts opt;
    FILE *pop_out, *indiv_out, *genes;
    int   indiv_is_devnull;
    int   first_seen;
    int   nrSamples;
    long *cnt[10];                 /* counters for ".,actgACTG" each nrSamples+1 (call_vC.cpp:435-444) */
    genepos_t   *mapGenes;   size_t nGenes;
    genomeent_t *mapGenomes; size_t nGenomes;
    gene_t      *intervals;  size_t nIntervals;    /* current contig, file order (call_vC.cpp:83,276-278) */
    char *name;                    /* current contig name (call_vC.cpp:460) */
    int   genomeLoaded;            /* call_vC.cpp:465 */
    int   hasGenes;                /* call_vC.cpp:554 (keeps its last value) */
    char *linebuf; size_t linecap;
    char *tok;
    sbuf s, indiv, internal, covstr;
    uint64_t n_pop_lines, n_indiv_lines;
};

static int sym_index(char c) {
    switch (c) {
        case '.': return 0; case ',': return 1;
        case 'a': return 2; case 'c': return 3; case 't': return 4; case 'g': return 5;
        case 'A': return 6; case 'C': return 7; case 'T': return 8; case 'G': return 9;
        default: return -1;
    }
}

/* call_vC.cpp:287-293 getSum */
static int get_sum(const orc_snpcall *sc, const char *set, int sample) {
    int sum = 0; size_t i;
    for (i = 0; set[i]; ++i) sum += (int)sc->cnt[sym_index(set[i])][sample];
    return sum;
}

/* call_vC.cpp:316-325 getCoverageString */
static void coverage_string(orc_snpcall *sc, const char *set, sbuf *out) {
    int i;
    sb_clear(out);
    for (i = 1; i <= sc->nrSamples; ++i) {
        sb_putl(out, get_sum(sc, set, i));
        sb_putc(out, '|');
    }
    if (out->n) { out->n--; out->p[out->n] = 0; }   /* substr(0,size-1) */
}

/* call_vC.cpp:299-314 revComplement (drops every non-ACGT char) */
static void rev_complement(const char *in, char *out) {
    int n = (int)strlen(in), i, m = 0;
    for (i = n - 1; i >= 0; --i) {
        if (in[i] == 'A') out[m++] = 'T';
        else if (in[i] == 'T') out[m++] = 'A';
        else if (in[i] == 'C') out[m++] = 'G';
        else if (in[i] == 'G') out[m++] = 'C';
    }
    out[m] = 0;
}

static genepos_t *find_genes(orc_snpcall *sc, const char *name) {
    size_t i;
    for (i = 0; i < sc->nGenes; ++i) if (strcmp(sc->mapGenes[i].name, name) == 0) return &sc->mapGenes[i];
    return NULL;
}
static void set_genes(orc_snpcall *sc, const char *name, unsigned long start, unsigned long lineCount) {
    genepos_t *g = find_genes(sc, name);                 /* map::operator[] = overwrite */
    if (!g) {
        sc->mapGenes = (genepos_t *)realloc(sc->mapGenes, (sc->nGenes + 1) * sizeof(genepos_t));
        g = &sc->mapGenes[sc->nGenes++];
        g->name = strdup(name);
    }
    g->start = start; g->lineCount = lineCount;
}
static genomeent_t *find_genome(orc_snpcall *sc, const char *name) {
    size_t i;
    for (i = 0; i < sc->nGenomes; ++i) if (strcmp(sc->mapGenomes[i].name, name) == 0) return &sc->mapGenomes[i];
    return NULL;
}
static void set_genome(orc_snpcall *sc, const char *name, genome_t *g) {
    genomeent_t *e = find_genome(sc, name);
    if (!e) {
        sc->mapGenomes = (genomeent_t *)realloc(sc->mapGenomes, (sc->nGenomes + 1) * sizeof(genomeent_t));
        e = &sc->mapGenomes[sc->nGenomes++];
        e->name = strdup(name); e->g = NULL;
    }
    genome_free(e->g);          /* the reference leaks the old one */
    e->g = g;
}

/* call_vC.cpp:116-199 indexGenomeAndGenes */
static int index_genome_and_genes(orc_snpcall *sc, FILE *refGenome, FILE *refGenes) {
    char line[LINE_CAP];
    char *tok = sc->tok;
    unsigned long filePosStart = 0, fileConsumed = 0, lineCount = 0;
    char *name = NULL;
    sbuf genome = {0};
    int skip = 0;

    /* :129-130 header line */
    if (!fgets(line, LINE_CAP, refGenes)) line[0] = 0;
    filePosStart = strlen(line);
    while (fgets(line, LINE_CAP, refGenes)) {                          /* :132-156 */
        int pos = 0;
        const char *rest = toksplit(line, '\t', tok, TOK_CAP);
        while (*rest) {
            if (pos == 2) {
                if (name == NULL) {
                    name = strdup(tok);
                } else if (strcmp(name, tok) != 0) {
                    set_genes(sc, name, filePosStart, lineCount);     /* :140-145 */
                    lineCount = 0;
                    filePosStart += fileConsumed;
                    fileConsumed = 0;
                    free(name); name = strdup(tok);
                }
                break;
            }
            ++pos;
            rest = toksplit(rest, '\t', tok, TOK_CAP);
        }
        fileConsumed += strlen(line);
        lineCount += 1;
    }
    set_genes(sc, name ? name : "", filePosStart, lineCount);          /* :158-160 */
    free(name); name = strdup("");

    /* :165-193 FASTA: every fgets chunk loses its last char */
    while (fgets(line, LINE_CAP, refGenome)) {
        size_t l = strlen(line);
        if (l) line[l - 1] = '\0';
        if (line[0] == '>') {
            if (genome.n > 0 && !skip) {
                set_genome(sc, name, genome_new(genome.p, genome.n));
                sb_clear(&genome);
            }
            free(name); name = strdup(line + 1);
            skip = find_genes(sc, name) == NULL;
        } else {
            if (skip) continue;
            sb_puts(&genome, line);
        }
    }
    set_genome(sc, name, genome_new(genome.p ? genome.p : "", genome.n));   /* :193 */
    CHAT( "Genomes loaded!\n");
    free(name); sb_free(&genome);
    return ORC_OK;
}

/* call_vC.cpp:205-284 loadGenome: (re)builds the interval list of the current contig */
static int load_genome(orc_snpcall *sc, const char *gName) {
    size_t i;
    genepos_t *p;
    char line[LINE_CAP];
    char *tok = sc->tok;
    unsigned long lc = 0;
    long start = 0, end = 0; char strand = 'x';
    char geneName[TOK_CAP + 1]; geneName[0] = 0;

    for (i = 0; i < sc->nIntervals; ++i) free(sc->intervals[i].name);   /* :207 clear() */
    sc->nIntervals = 0;

    p = find_genes(sc, gName);
    if (!p) { sc->hasGenes = 0; return ORC_OK; }                        /* :209-213 */
    sc->hasGenes = 1;
    if (!find_genome(sc, gName)) CHAT( "Weird...%s\n", gName); /* :219-221 */
    if (fseek(sc->genes, (long)p->start, SEEK_SET) != 0) {               /* :231-234 */
        CHAT( "File seek failed %ld \n", (long)p->start);
        return ORC_OK;
    }
    while (lc < p->lineCount) {                                          /* :237-280 */
        const char *rest; int pos = 0;
        if (!fgets(line, LINE_CAP, sc->genes)) {
            CHAT( "Read failed. Seeking file to %ld\n", (long)p->start);
            line[0] = 0;
        }
        ++lc;
        rest = toksplit(line, '\t', tok, TOK_CAP);
        for (;;) {                                                       /* while (tok) */
            if (pos == 1) { strncpy(geneName, tok, TOK_CAP); geneName[TOK_CAP] = 0; }
            if (pos == 2) {
                if (strcmp(gName, tok) != 0) {
                    CHAT( "Reading wrong gene defintion for %s\n. Scafold supposed to be %s, but is %s\n.", geneName, tok, gName);
                    break;
                }
            } else if (pos == 6) {
                start = atol(tok) - 1;
            } else if (pos == 7) {
                end = atol(tok) - 1;
            } else if (pos == 8) {
                strand = tok[0];
                break;
            }
            ++pos;
            rest = toksplit(rest, '\t', tok, TOK_CAP);
        }
        if (start > end) {                                               /* :273-275 */
            CHAT( "This gene goes around :(.\nPretending we didn't see it.\n");
        } else {                                                         /* :276-278 */
            gene_t *g;
            sc->intervals = (gene_t *)realloc(sc->intervals, (sc->nIntervals + 1) * sizeof(gene_t));
            g = &sc->intervals[sc->nIntervals++];
            g->start = start; g->end = end; g->strand = strand; g->name = strdup(geneName);
        }
    }
    return ORC_OK;
}

/* call_vC.cpp:567-574 geneIntervals(lP): split_interval_map<long,GeneDef> whose codomain
 * combine is GeneDef::operator+= (gene.h:143-146, appends) and whose getGene() returns
 * front() (gene.h:139-141): the first gene added (= file order) whose closed interval
 * [start,end] contains the point. */
static const gene_t *lookup_gene(const orc_snpcall *sc, long lP) {
    size_t i;
    for (i = 0; i < sc->nIntervals; ++i)
        if (sc->intervals[i].start <= lP && lP <= sc->intervals[i].end) return &sc->intervals[i];
    return NULL;
}

/* ------------------------------------------------------------------ API */
void orc_snpcall_default_opts(orc_snpcall_opts *o) {
    o->min_coverage = 4; o->calling_threshold = 4; o->calling_min_fraction = 0.01;   /* :26-36 */
    o->fasta_path = NULL; o->genes_path = NULL; o->token_cap = 0;
}

int orc_snpcall_begin(orc_snpcall **out, const orc_snpcall_opts *opts, FILE *pop_out, FILE *indiv_out) {
    orc_snpcall *sc = (orc_snpcall *)calloc(1, sizeof *sc);
    if (!sc) return ORC_ERR_NOMEM;
    sc->opt = *opts;
    sc->pop_out = pop_out; sc->indiv_out = indiv_out;
    sc->tok = (char *)malloc(TOK_CAP + 1);
    sc->name = strdup("");
    *out = sc;
    return ORC_OK;
}

/* call_vC.cpp:423-452: the first line only defines nrSamples and is dropped */
static int first_line(orc_snpcall *sc, const char *line) {
    unsigned int number_of_tabs = 0; size_t i, n = strlen(line);
    int k;
    for (i = 0; i < n; ++i) if (line[i] == '\t') ++number_of_tabs;
    sc->nrSamples = (int)(number_of_tabs + 1 - 3) / 3;                   /* :431 */
    if (sc->nrSamples < 0) { orc_set_error("first mpileup line has fewer than 3 fields (reference: UB)"); return ORC_ERR_DOMAIN; }
    CHAT( "Identified %d samples\n", sc->nrSamples);
    for (k = 0; k < 10; ++k) {
        sc->cnt[k] = (long *)calloc((size_t)sc->nrSamples + 1, sizeof(long));
        if (!sc->cnt[k]) return ORC_ERR_NOMEM;
    }
    if (sc->opt.fasta_path && sc->opt.genes_path) {                      /* :448-452 */
        FILE *fa = fopen(sc->opt.fasta_path, "r");
        sc->genes = fopen(sc->opt.genes_path, "r");
        if (!fa || !sc->genes) { if (fa) fclose(fa); orc_set_error("cannot open -f/-g file"); return ORC_ERR_IO; }
        CHAT( "Found reference genomes and annotation file.\nLoading Genomes...\n");
        index_genome_and_genes(sc, fa, sc->genes);
        fclose(fa);
    } else if (sc->opt.genes_path) {
        sc->genes = fopen(sc->opt.genes_path, "r");     /* opened by getopt (:368) but never indexed */
    }
    return ORC_OK;
}

/* call_vC.cpp:466-668: one iteration of the main loop */
static int process_line(orc_snpcall *sc, char *line) {
    int lLen = (int)strlen(line);
    int pos = 0, k, i;
    int lP = 0; char base = 0;
    char *tok = sc->tok;
    const char *rest;
    int cov;
    const gene_t *g; int isInGene = 0; const char *geneName = "-";
    int write = 0;
    static const char snps[] = "actg";                                  /* :561 */
    char oldCodon[8], newCodon[8], tmp[8];

    if (lLen > 0) line[--lLen] = '\0';                                  /* :475 */
    for (k = 0; k < 10; ++k) memset(sc->cnt[k], 0, ((size_t)sc->nrSamples + 1) * sizeof(long));   /* :477-479 */

    const int tok_cap = (sc->opt.token_cap > 0 && sc->opt.token_cap < TOK_CAP) ? sc->opt.token_cap : TOK_CAP;   /* (tests only: orc.h) */
    rest = toksplit(line, '\t', tok, tok_cap);                          /* :483 */
    while (*rest) {                                                      /* :490 */
        if (pos == 0) {
            if (strcmp(sc->name, tok) != 0) { free(sc->name); sc->name = strdup(tok); sc->genomeLoaded = 0; }
        } else if (pos == 1) {
            lP = (int)(atol(tok) - 1);                                   /* :499 */
        } else if (pos == 2) {
            base = tok[0];                                               /* :502 */
        } else if (pos > 3 && pos % 3 == 1) {                            /* :503 */
            int len = (int)strlen(tok);
            int sample = pos / 3;
            if (sample > sc->nrSamples) { orc_set_error("line has more samples than the first line (reference: out-of-bounds write)"); return ORC_ERR_DOMAIN; }
            i = 0;
            while (i < len) {
                switch (tok[i]) {
                    case '^': ++i; break;                                /* :511-514 */
                    case '+': case '-': {                                /* :515-522 */
                        int skip = 0, any = 0;
                        while (isdigit((unsigned char)tok[++i])) { skip = skip * 10 + (tok[i] - '0'); any = 1; }
                        (void)any;
                        i += skip - 1;
                        break;
                    }
                    case '*': case '$': case 'N': case 'n': break;       /* :523-527 */
                    default: {                                           /* :528-531 */
                        int si = sym_index(tok[i]);
                        if (si < 0) {
                            snprintf(g_err, sizeof g_err, "pileup symbol '%c' (0x%02x): the reference dereferences an empty vector (SIGSEGV)", tok[i], (unsigned char)tok[i]);
                            return ORC_ERR_DOMAIN;
                        }
                        ++sc->cnt[si][0];
                        ++sc->cnt[si][sample];
                        break;
                    }
                }
                ++i;
            }
        }
        ++pos;
        rest = toksplit(rest, '\t', tok, tok_cap);                       /* :540 */
    }

    cov = get_sum(sc, "actgACTG,.", 0);                                  /* :545 */
    if (cov < sc->opt.min_coverage) return ORC_OK;                       /* :547 */
    if (get_sum(sc, "actgACTG", 0) < sc->opt.calling_threshold) return ORC_OK;   /* :550 */

    if (!sc->genomeLoaded) {                                             /* :556-559 */
        if (sc->genes && sc->nGenes) load_genome(sc, sc->name);
        else { size_t q; for (q = 0; q < sc->nIntervals; ++q) free(sc->intervals[q].name); sc->nIntervals = 0; sc->hasGenes = 0; }
        sc->genomeLoaded = 1;
    }

    g = lookup_gene(sc, lP);                                             /* :567-574 */
    if (g) { geneName = g->name; isInGene = 1; }
    sb_clear(&sc->s); sb_clear(&sc->indiv);

    for (i = 0; i < 4; ++i) {                                            /* :577 */
        char check[3];
        long snpCount;
        int writeThis = 0; sbuf *sToWrite = NULL;
        if (snps[i] == base) continue;                                   /* :580 case-sensitive */
        check[0] = snps[i]; check[1] = (char)toupper(snps[i]); check[2] = 0;
        snpCount = get_sum(sc, check, 0);                                /* :584 */
        if ((snpCount >= sc->opt.calling_threshold) &&
            ((double)snpCount >= cov * sc->opt.calling_min_fraction)) {  /* :588 */
            write = 1; writeThis = 1; sToWrite = &sc->s;
        } else {                                                         /* :592-601 */
            int smp;
            for (smp = 1; smp <= sc->nrSamples; ++smp) {
                if (get_sum(sc, check, smp) >= sc->opt.calling_threshold) { writeThis = 1; sToWrite = &sc->indiv; break; }
            }
        }
        if (!writeThis) continue;
        sb_clear(&sc->internal);
        if (sc->hasGenes && isInGene) {                                  /* :604-633 */
            long codonStart; int codonPosition;
            genomeent_t *ge;
            if (g->start < g->end) {
                codonPosition = (int)((lP - g->start) % 3);              /* :611 */
                codonStart = lP - codonPosition;
                ge = find_genome(sc, sc->name);
                if (!ge) { orc_set_error("contig has genes but no FASTA sequence (reference: deref of map::end())"); return ORC_ERR_DOMAIN; }
                if (genome_get(ge->g, codonStart, codonStart + 2, oldCodon) == 0) {
                    orc_set_error("codon runs past the contig end (reference: write into empty std::string)"); return ORC_ERR_DOMAIN;
                }
            } else {                                                     /* :614-617 */
                CHAT( "Will not handle circular genes\n");
                continue;
            }
            strcpy(newCodon, oldCodon);
            newCodon[codonPosition] = (char)toupper(snps[i]);            /* :619 */
            if (g->strand == '-') {                                      /* :621-624 */
                rev_complement(oldCodon, tmp); strcpy(oldCodon, tmp);
                rev_complement(newCodon, tmp); strcpy(newCodon, tmp);
            }
            sb_putl(&sc->internal, snpCount); sb_putc(&sc->internal, '|'); sb_putc(&sc->internal, check[1]); sb_putc(&sc->internal, '|');
            sb_putc(&sc->internal, codon_aa(newCodon) == codon_aa(oldCodon) ? 'S' : 'N');   /* :627-631 */
            sb_putc(&sc->internal, '['); sb_puts(&sc->internal, oldCodon); sb_putc(&sc->internal, '-');
            sb_puts(&sc->internal, newCodon); sb_puts(&sc->internal, "]|");
        } else {                                                         /* :634-637 */
            sb_putl(&sc->internal, snpCount); sb_putc(&sc->internal, '|'); sb_putc(&sc->internal, check[1]); sb_puts(&sc->internal, "|.|");
        }
        coverage_string(sc, check, &sc->covstr);
        sb_putc(sToWrite, ','); sb_puts(sToWrite, sc->internal.p); sb_puts(sToWrite, sc->covstr.p ? sc->covstr.p : "");
    }

    if (write) {                                                         /* :641-652 */
        coverage_string(sc, "actgACTG,.", &sc->covstr);
        fprintf(sc->pop_out, "%s\t%s\t%ld\t%c\t%s\t%s\n", sc->name, geneName, (long)(lP + 1), base,
                sc->covstr.p ? sc->covstr.p : "", sc->s.n ? sc->s.p + 1 : "");
        sc->n_pop_lines++;
    }
    if (sc->indiv.n != 0) {                                              /* :653-667 */
        if (sc->indiv_out == NULL) {
            if (!sc->indiv_is_devnull) {
                CHAT( "Individual SNPs detected, but no individual output file specified (-i option).\n");
                sc->indiv_is_devnull = 1;
            }
        } else {
            coverage_string(sc, "actgACTG,.", &sc->covstr);
            fprintf(sc->indiv_out, "%s\t%s\t%ld\t%c\t%s\t%s\n", sc->name, geneName, (long)(lP + 1), base,
                    sc->covstr.p ? sc->covstr.p : "", sc->indiv.p + 1);
            sc->n_indiv_lines++;
        }
    }
    return ORC_OK;
}

int orc_snpcall_line(orc_snpcall *sc, const char *line, size_t len) {
    if (len + 2 > sc->linecap) {
        sc->linecap = (len + 2) * 2;
        sc->linebuf = (char *)realloc(sc->linebuf, sc->linecap);
        if (!sc->linebuf) return ORC_ERR_NOMEM;
    }
    memcpy(sc->linebuf, line, len); sc->linebuf[len] = 0;
    if (!sc->first_seen) { sc->first_seen = 1; return first_line(sc, sc->linebuf); }
    return process_line(sc, sc->linebuf);
}

int orc_snpcall_end(orc_snpcall *sc) {
    size_t i; int k;
    if (!sc) return ORC_OK;
    for (k = 0; k < 10; ++k) free(sc->cnt[k]);
    for (i = 0; i < sc->nGenes; ++i) free(sc->mapGenes[i].name);
    free(sc->mapGenes);
    for (i = 0; i < sc->nGenomes; ++i) { free(sc->mapGenomes[i].name); genome_free(sc->mapGenomes[i].g); }
    free(sc->mapGenomes);
    for (i = 0; i < sc->nIntervals; ++i) free(sc->intervals[i].name);
    free(sc->intervals);
    if (sc->genes) fclose(sc->genes);
    free(sc->name); free(sc->linebuf); free(sc->tok);
    sb_free(&sc->s); sb_free(&sc->indiv); sb_free(&sc->internal); sb_free(&sc->covstr);
    free(sc);
    return ORC_OK;
}

/* `snpCall ... < mpileup > called_SNPs`: fgets with a 10 MB buffer (call_vC.cpp:38,423,466) */
int orc_snpcall_stream(const orc_snpcall_opts *opts, FILE *in, FILE *pop_out, FILE *indiv_out) {
    orc_snpcall *sc; int rc;
    size_t cap = 10000000; char *buf = (char *)malloc(cap);
    if (!buf) return ORC_ERR_NOMEM;
    rc = orc_snpcall_begin(&sc, opts, pop_out, indiv_out);
    if (rc) { free(buf); return rc; }
    while (fgets(buf, (int)cap, in)) {
        rc = orc_snpcall_line(sc, buf, strlen(buf));
        if (rc) break;
    }
    orc_snpcall_end(sc);
    free(buf);
    return rc;
}
