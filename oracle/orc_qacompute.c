/*
 * oracle/orc_qacompute.c -- TEST INFRASTRUCTURE (see orc.h).  PARITY UNPINNED.
 *
 * CPU restatement of /root/reference/src/qaTools/qaCompute.cpp for the single invocation
 * metaSNV makes:  qaCompute -c 10 -d -i BAM OUT   (metaSNV.py:63-65).
 * The reference needs htslib (absent here), so it cannot be built; its in-repo arithmetic
 * (read loop :441-593, CIGAR walk :530-552, compute_print_cov :125-221, printSkipped
 * :226-263, trailer :623-657) is fully specified and restated below.
 * Options -m -p -s -x -a -h are never used by metaSNV and are not restated.
 */
#include "orc.h"

#include <stdlib.h>
#include <string.h>

void orc_set_error(const char *msg);

static int32_t  rd_i32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }

#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP       4
#define BAM_FDUP      1024

/* qaCompute.cpp:125-221 compute_print_cov (detailed, silent, no median/profile/specific) */
static void compute_print_cov(FILE *out, FILE *detail, int max_cov, int *data, const char *name,
                              uint32_t chrSize, uint64_t *coverageHist) {
    int32_t covVal = 0; uint64_t covSum = 0; uint32_t i; int k, x;
    uint64_t *local = (uint64_t *)calloc((size_t)max_cov + 1, sizeof(uint64_t));
    for (i = 0; i < chrSize; ++i) {                                     /* :142-165 */
        covVal += data[i];
        data[i] = covVal;
        covSum += (uint64_t)(int64_t)covVal;
        if (covVal > max_cov) { ++coverageHist[max_cov]; ++local[max_cov]; }
        else if (covVal >= 0) { ++coverageHist[covVal];  ++local[covVal]; }
        /* covVal == -1 (last position, below a read whose cursor reached the contig end): the reference increments
           coverageHist[-1] and localCoverageHist[-1], out of bounds -- undefined; restated as "lands in no bin" */
    }
    fprintf(detail, "%s\t%d\t", name, (int)chrSize);                    /* :192-201 */
    for (k = 1; k <= max_cov; ++k) {
        uint64_t coverage = 0;
        for (x = k; x <= max_cov; ++x) coverage += local[x];
        fprintf(detail, "%d\t", (int)coverage);
    }
    fprintf(detail, "\n");
    free(local);
    fprintf(out, "%s\t%d\t%3.5f\n", name, (int)chrSize, (double)covSum / chrSize);   /* :217 */
}

/* qaCompute.cpp:226-263 printSkipped */
static void print_skipped(FILE *out, FILE *detail, int max_cov, const orc_ref *h, int start, int end) {
    int i, k;
    for (i = start; i < end; ++i) {
        fprintf(out, "%s\t%d\t%3.5f\n", h->names[i], (int)h->lengths[i], 0.0);
        fprintf(detail, "%s\t%d\t", h->names[i], (int)h->lengths[i]);
        for (k = 1; k <= max_cov; ++k) fprintf(detail, "%d\t", 0);
        fprintf(detail, "\n");
    }
}

int orc_qacompute(const orc_ref *head, const orc_sample *sample, int max_cov, int min_mapq,
                  const char *cov_path, const char *detail_path) {
    FILE *out = fopen(cov_path, "wt"), *detail = fopen(detail_path, "wt");
    uint64_t totalGenomeLength = 0, off = 0;
    uint32_t unmappedReads = 0, zeroQualityReads = 0, totalNumberOfReads = 0, totalProperPaires = 0, chrSize = 0, duplicates = 0;
    int *entireChr = NULL; int32_t currentTid = -1; int i;
    uint64_t *coverageHist = (uint64_t *)calloc((size_t)max_cov + 1, sizeof(uint64_t));
    if (!out || !detail) { if (out) fclose(out); if (detail) fclose(detail); orc_set_error("cannot create .cov/.detail"); return ORC_ERR_IO; }
    for (i = 0; i < head->n_contigs; ++i) totalGenomeLength += (uint64_t)head->lengths[i];     /* :425-427 */
    fprintf(out, "Chromosome\tSeq_lem\tAvg_Cov\n");                                             /* :439 */

    while (off < sample->n_bytes) {                                                              /* :441 */
        const uint8_t *p = sample->records + off;
        int32_t bs, tid, pos; uint32_t l_name; unsigned n_cigar; uint16_t flag; uint8_t mapq; const uint8_t *cigar;
        if (sample->n_bytes - off < 36) { orc_set_error("truncated BAM record"); free(entireChr); free(coverageHist); fclose(out); fclose(detail); return ORC_ERR_FORMAT; }
        bs = rd_i32(p); tid = rd_i32(p + 4); pos = rd_i32(p + 8);
        l_name = p[12]; mapq = p[13]; n_cigar = rd_u16(p + 16); flag = rd_u16(p + 18);
        cigar = p + 36 + l_name;
        {   /* sam_read1 -> bam_read1 puts a CIGAR that lives in the CG field back (orc.h: orc_resolve_cg) */
            const int32_t l_seq = rd_i32(p + 20);
            const uint8_t *aux = cigar + 4 * (size_t)n_cigar + ((size_t)(l_seq > 0 ? l_seq : 0) + 1) / 2 + (size_t)(l_seq > 0 ? l_seq : 0);
            if (bs >= 32 && aux <= p + 4 + (size_t)bs) orc_resolve_cg(aux, p + 4 + (size_t)bs, tid, pos, l_seq, &cigar, &n_cigar);
        }
        off += (uint64_t)bs + 4;

        if (flag & BAM_FUNMAP) {                                                                 /* :461-462 */
            ++unmappedReads;
        } else {
            if (tid != currentTid) {                                                             /* :465-515 */
                if (tid == -1) { ++unmappedReads; ++totalNumberOfReads; continue; }              /* :467-473 */
                if (currentTid != -1)
                    compute_print_cov(out, detail, max_cov, entireChr, head->names[currentTid], chrSize, coverageHist);
                chrSize = (uint32_t)head->lengths[tid];
                entireChr = (int *)realloc(entireChr, ((size_t)chrSize + 1) * sizeof(int));
                memset(entireChr, 0, ((size_t)chrSize + 1) * sizeof(int));
                if ((currentTid + 1 != tid) && (currentTid != -1)) {                             /* :500-504 */
                    print_skipped(out, detail, max_cov, head, currentTid + 1, tid);
                    coverageHist[0] += (uint64_t)head->lengths[tid];
                }
                if (currentTid == -1) { currentTid = tid; print_skipped(out, detail, max_cov, head, 0, currentTid); }
                else currentTid = tid;
            }
            if (mapq >= min_mapq) {                                                              /* :518 */
                if (flag & BAM_FPROPER_PAIR) ++totalProperPaires;
                if (flag & BAM_FDUP) {                                                           /* :524-526 */
                    ++duplicates;
                } else {                                                                         /* :530-552 */
                    uint32_t pp = (uint32_t)pos + 1; int k = 0;
                    const uint8_t *c = cigar;
                    if (!entireChr) { orc_set_error("mapped read with tid -1 before any contig (reference: NULL deref)"); free(coverageHist); fclose(out); fclose(detail); return ORC_ERR_DOMAIN; }
                    /* the reference reads *cigar even when n_cigar == 0 (then sees sequence bytes);
                       the loop below never runs in that case, so the peek has no effect */
                    if (n_cigar > 0 && (((rd_u32(c) & 15) == 4) || ((rd_u32(c) & 15) == 5))) { c += 4; ++k; }
                    while (k < (int)n_cigar) {
                        uint32_t op = rd_u32(c);
                        ++k;
                        if ((op & 15) != 0) {
                            pp = pp + (op >> 4);
                        } else {
                            /* the array has chrSize + 1 slots (:490): pp == chrSize is in bounds and lands outside the scanned range;
                               pp > chrSize writes out of bounds in the reference: that increment is left out here (what the
                               heap's slack makes of it in practice), the decrement below is the visible effect either way */
                            if (pp <= chrSize) ++entireChr[pp];
                            pp = pp + (op >> 4);
                            if (pp >= chrSize) --entireChr[chrSize - 1];
                            else --entireChr[pp];
                        }
                        c += 4;
                    }
                }
            } else {
                ++zeroQualityReads;                                                              /* :585-588 */
            }
        }
        ++totalNumberOfReads;                                                                    /* :591 */
    }

    if (currentTid == -1) {                                                                      /* :596 target_name[-1] */
        orc_set_error("BAM has no mapped reads (reference: reads target_name[-1])");
        free(entireChr); free(coverageHist); fclose(out); fclose(detail);
        return ORC_ERR_DOMAIN;
    }
    compute_print_cov(out, detail, max_cov, entireChr, head->names[currentTid], chrSize, coverageHist);
    if (currentTid != head->n_contigs) print_skipped(out, detail, max_cov, head, currentTid + 1, head->n_contigs);   /* :600-602 */
    free(entireChr);

    fprintf(out, "\nCov*X\tPercentage\tNr. of bases\n");                                          /* :623 */
    for (i = 1; i <= max_cov; ++i) {                                                             /* :628-640 */
        uint64_t coverage = 0; int x;
        for (x = i; x <= max_cov; ++x) coverage += coverageHist[x];
        fprintf(out, "%d\t%3.5f\t%lu\n", i, (double)(coverage) / totalGenomeLength * 100, (unsigned long)coverage);
    }
    fprintf(out, "\nOther\n");                                                                    /* :642-654 */
    {
        double procentageOfUnmapped = 100 * ((double)unmappedReads / totalNumberOfReads);
        double procentageOfZeroQuality = 100 * ((double)zeroQualityReads / totalNumberOfReads);
        int32_t nrOfPaires = (int32_t)(totalNumberOfReads / 2);
        double procOfProperPaires = (double)(100 * (double)totalProperPaires / 2) / nrOfPaires;
        fprintf(out, "Total number of reads: %u\n", totalNumberOfReads);
        fprintf(out, "Total number of duplicates found and ignored: %u\n", duplicates);
        fprintf(out, "Percentage of unmapped reads: %3.5f\n", procentageOfUnmapped);
        fprintf(out, "Percentage of sub-par quality mappings: %3.5f\n", procentageOfZeroQuality);
        fprintf(out, "Number of proper paired reads: %u\n", totalProperPaires);
        fprintf(out, "Percentage of proper pairs: %3.5f\n", procOfProperPaires);
    }
    free(coverageHist);
    fclose(out); fclose(detail);
    return ORC_OK;
}
