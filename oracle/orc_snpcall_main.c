/*
 * oracle/orc_snpcall_main.c -- TEST INFRASTRUCTURE (see orc.h).
 * Command-line front end of the snpCall restatement with the reference's argv surface
 * (call_vC.cpp:346-416):  orc_snpcall -f REF [-g ANN] -i INDIV -c INT -t INT [-p FLOAT] < mpileup > called_SNPs
 */
#include "orc.h"
#include <stdlib.h>
#include <unistd.h>

int main(int argc, char **argv) {
    orc_snpcall_opts o; FILE *indiv = NULL; int c, rc;
    orc_snpcall_default_opts(&o);
    opterr = 0;
    while ((c = getopt(argc, argv, "hdab:f:g:i:c:p:t:")) != -1) {
        switch (c) {
            case 'h': fprintf(stderr, "Usage: orc_snpcall [options] < stdin.mpileup\n"); return -1;
            case 'a': case 'd': case 'b': break;
            case 'f': o.fasta_path = optarg; break;
            case 'g': o.genes_path = optarg; break;
            case 'i': indiv = fopen(optarg, "w"); if (!indiv) { fprintf(stderr, "Cannot open %s\n", optarg); return -1; } break;
            case 'c': o.min_coverage = (int)atol(optarg); break;
            case 'p': o.calling_min_fraction = atof(optarg); break;
            case 't': o.calling_threshold = (int)atol(optarg); break;
            default: return 1;
        }
    }
    if (optind < argc) { printf("Non-option argument %s\n", argv[optind]); return 0; }
    rc = orc_snpcall_stream(&o, stdin, stdout, indiv);
    if (indiv) fclose(indiv);
    if (rc) fprintf(stderr, "orc_snpcall: error %d: %s\n", rc, orc_last_error());
    return rc;
}
