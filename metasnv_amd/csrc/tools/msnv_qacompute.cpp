// msnv_qacompute -- process-level drop-in for `qaCompute [-c INT] [-q INT] -d -i <in.bam> <out>` as metaSNV.py:63-65
// invokes it (argv: src/qaTools/qaCompute.cpp:312-359; outputs OUT and OUT.detail; "Printing details in ..." on
// stdout :387; exit status 1 for usage / unopenable files :356-359,376-379, 0 on success :680).
// A thin main over the C ABI (include/msnv.h): all arithmetic runs on the GPU.  Options of qaCompute that
// metaSNV never passes (-m -p -s -x -a -h) are rejected instead of being silently ignored.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>

#include "../../../include/msnv.h"

static void usage() {
    fprintf(stderr, "Usage: msnv_qacompute [-c INT] [-q INT] -d [-i] <in.bam> <output.out>\n"
                    "  -c INT  maximum coverage of the breadth histogram (1..15, default 10 as metaSNV passes it)\n"
                    "  -q INT  minimum mapping quality (default 1)\n"
                    "  -d      write <output.out>.detail (always written; metaSNV.py always passes -d)\n"
                    "  -i      silent\n");
}

int main(int argc, char **argv) {
    int max_cov = 10, min_mapq = 1, arg;
    while ((arg = getopt(argc, argv, "mdip:s:q:c:h:x:a:")) >= 0) {
        switch (arg) {
        case 'd': case 'i': break;
        case 'q': min_mapq = atoi(optarg); break;
        case 'c': max_cov = atoi(optarg); break;
        default:
            fprintf(stderr, "msnv_qacompute: option -%c of qaCompute is not supported (metaSNV.py never passes it)\n", arg);
            return 1;
        }
    }
    if (argc - optind != 2) { usage(); return 1; }
    const std::string out = argv[optind + 1], detail = out + ".detail";
    msnv_ctx *ctx = nullptr;
    if (msnv_ctx_create(0, &ctx)) { fprintf(stderr, "msnv_qacompute: %s\n", msnv_last_error()); return 1; }
    msnv_cov_args a{};
    a.bam_path = argv[optind]; a.max_cov = max_cov; a.min_mapq = min_mapq;
    a.out_cov_path = out.c_str(); a.out_detail_path = detail.c_str();
    fprintf(stdout, "Printing details in %s!\n", detail.c_str());
    const int rc = msnv_coverage(ctx, &a);
    if (rc) fprintf(stderr, "msnv_qacompute: %s\n", msnv_last_error());
    msnv_ctx_destroy(ctx);
    return rc ? 1 : 0;
}
