// msnv_snpcall -- process-level replacement of the pipe metaSNV.py:160-176 runs:
//   samtools mpileup -f REF [-l SPLIT] -B -b LIST | snpCall -f REF [-g ANN] -i INDIV -c C -t T [-p P] > CALLED
// One process, same options (mpileup's -f/-l/-b, snpCall's -f/-g/-i/-c/-t/-p, call_vC.cpp:346-410), population
// lines on stdout like snpCall.  The mpileup text never exists: BAMs are decoded on the host and counted on the GPU.
// Without -b it IS snpCall: the same argv, mpileup text on stdin (msnv_call_from_mpileup, parsed and called on the GPU):
//   samtools mpileup -f REF -B -b LIST | msnv_snpcall -f REF [-g ANN] -i INDIV -c C -t T > CALLED
// Exit status: 0 ok, > 0 failure (the driver treats v > 0 as fatal, metaSNV.py:212-221).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <unistd.h>
#include <vector>

#include "../../../include/msnv.h"

static void usage() {
    fprintf(stderr, "Usage: msnv_snpcall [-f REF.fa] [-g ANNOTATION] [-i INDIV_OUT] [-c 4] [-t 4] [-p 0.01] < mpileup > called_SNPs     (snpCall)\n"
                    "       msnv_snpcall -f REF.fa -b BAM_LIST [-l SPLIT.bed] [-g ANNOTATION] [-i INDIV_OUT]\n"
                    "                    [-c MIN_COVERAGE=4] [-t MIN_SNV_READS=4] [-p MIN_FRACTION=0.01] [-@ HOST_THREADS] > called_SNPs\n");
}

int main(int argc, char **argv) {
    std::string ref, list, bed, ann, indiv;
    msnv_params p;
    msnv_params_default(&p);
    int threads = 0, arg;
    while ((arg = getopt(argc, argv, "f:b:l:g:i:c:t:p:@:Badh")) >= 0) {
        switch (arg) {
        case 'f': ref = optarg; break;
        case 'b': list = optarg; break;
        case 'l': bed = optarg; break;
        case 'g': ann = optarg; break;
        case 'i': indiv = optarg; break;
        case 'c': p.min_coverage = atoi(optarg); break;
        case 't': p.calling_threshold = atoi(optarg); break;
        case 'p': p.min_fraction = atof(optarg); break;
        case '@': threads = atoi(optarg); break;
        case 'B': case 'a': case 'd': break;             // mpileup -B is implied (no BAQ); snpCall's -a / -d are flags WITHOUT an argument (getopt string "hdab:f:g:i:c:p:t:", call_vC.cpp:346), accepted and ignored (:351-357)
        default: usage(); return 1;                      // snpCall -h prints the usage and fails (:381-384)
        }
    }
    if (optind != argc || argc == 1) { usage(); return 1; }      // (no option at all: the usage text, not a silent wait on stdin)
    if (list.empty()) {                                  // snpCall's own interface: text on stdin; -f is only needed with -g (call_vC.cpp:448)
        if (!bed.empty()) { fprintf(stderr, "msnv_snpcall: -l selects regions of BAM files (-b); the text on stdin is what it is\n"); return 1; }
        msnv_ctx *ctx = nullptr;
        if (msnv_ctx_create(0, &ctx)) { fprintf(stderr, "msnv_snpcall: %s\n", msnv_last_error()); return 1; }
        msnv_mpileup_args m{};
        m.mpileup_path = "-";
        m.ref_fasta = ref.empty() ? nullptr : ref.c_str();
        m.ann_path = ann.empty() ? nullptr : ann.c_str();
        m.out_called_path = "/dev/stdout";
        m.out_indiv_path = indiv.empty() ? nullptr : indiv.c_str();
        m.params = p;
        const int rc = msnv_call_from_mpileup(ctx, &m, nullptr);
        if (rc) fprintf(stderr, "msnv_snpcall: %s\n", msnv_last_error());
        msnv_ctx_destroy(ctx);
        return rc;
    }
    if (ref.empty()) { usage(); return 1; }
    std::vector<std::string> bams;
    {
        std::ifstream in(list);
        if (!in) { fprintf(stderr, "msnv_snpcall: cannot open %s\n", list.c_str()); return 1; }
        for (std::string l; std::getline(in, l);) { while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.pop_back(); if (!l.empty()) bams.push_back(l); }
    }
    if (bams.empty()) { fprintf(stderr, "msnv_snpcall: %s lists no BAM files\n", list.c_str()); return 1; }
    std::vector<const char *> paths;
    for (const std::string &b : bams) paths.push_back(b.c_str());
    msnv_ctx *ctx = nullptr;
    if (msnv_ctx_create(0, &ctx)) { fprintf(stderr, "msnv_snpcall: %s\n", msnv_last_error()); return 1; }
    msnv_call_args a{};
    a.bam_paths = paths.data(); a.n_bams = (int32_t)paths.size();
    a.ref_fasta = ref.c_str();
    a.ann_path = ann.empty() ? nullptr : ann.c_str();
    a.bed_split_path = bed.empty() ? nullptr : bed.c_str();
    a.out_called_path = "/dev/stdout";
    a.out_indiv_path = indiv.empty() ? nullptr : indiv.c_str();
    a.host_threads = threads;
    a.params = p;
    const int rc = msnv_call(ctx, &a);
    if (rc) fprintf(stderr, "msnv_snpcall: %s\n", msnv_last_error());
    msnv_ctx_destroy(ctx);
    return rc;
}
