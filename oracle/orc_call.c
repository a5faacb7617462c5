/*
 * oracle/orc_call.c -- TEST INFRASTRUCTURE (see orc.h).  PARITY UNPINNED.
 *
 * `samtools mpileup ... | snpCall ...` (metaSNV.py:153-176) as one in-memory pipeline:
 * every line produced by the mpileup restatement is handed to the snpCall restatement,
 * which is what the POSIX pipe at metaSNV.py:173-175 does.
 */
#include "orc.h"

#include <stdlib.h>
#include <string.h>

uint64_t orc_mpileup_last_lines(void);
uint64_t orc_mpileup_last_bases(void);
void orc_set_error(const char *msg);

typedef struct { orc_snpcall *sc; int rc; } pipe_t;

static int pipe_cb(void *user, const char *line, size_t len) {
    pipe_t *p = (pipe_t *)user;
    p->rc = orc_snpcall_line(p->sc, line, len);
    return p->rc;
}

int orc_call(const orc_ref *ref, const orc_sample *samples, int n_samples,
             const orc_mpileup_opts *mopts, const orc_snpcall_opts *sopts,
             const char *pop_path, const char *indiv_path,
             uint64_t *n_lines_out, uint64_t *n_pileup_bases_out) {
    FILE *pop = fopen(pop_path, "wt");
    FILE *ind = indiv_path ? fopen(indiv_path, "wt") : NULL;
    pipe_t p; int rc;
    if (!pop || (indiv_path && !ind)) { if (pop) fclose(pop); if (ind) fclose(ind); orc_set_error("cannot create output file"); return ORC_ERR_IO; }
    p.rc = 0;
    rc = orc_snpcall_begin(&p.sc, sopts, pop, ind);
    if (rc == ORC_OK) {
        rc = orc_mpileup(ref, samples, n_samples, mopts, pipe_cb, &p);
        if (p.rc) rc = p.rc;
        orc_snpcall_end(p.sc);
    }
    fclose(pop); if (ind) fclose(ind);
    if (n_lines_out) *n_lines_out = orc_mpileup_last_lines();
    if (n_pileup_bases_out) *n_pileup_bases_out = orc_mpileup_last_bases();
    return rc;
}

/* mpileup text to a file (for fixtures / debugging) */
static int file_cb(void *user, const char *line, size_t len) { return fwrite(line, 1, len, (FILE *)user) != len; }

int orc_mpileup_to_file(const orc_ref *ref, const orc_sample *samples, int n_samples,
                        const orc_mpileup_opts *opts, const char *path) {
    FILE *f = fopen(path, "wt"); int rc;
    if (!f) { orc_set_error("cannot create mpileup text file"); return ORC_ERR_IO; }
    rc = orc_mpileup(ref, samples, n_samples, opts, file_cb, f);
    fclose(f);
    return rc;
}
