// metasnv_amd/csrc/inflate_bmi2.cpp -- the DEFLATE decoder of inflate.cpp once more, compiled with -mbmi2 (Makefile) under another
// name; hostio.cpp calls it when the CPU has BMI2 (__builtin_cpu_supports).
#define MSNV_INFLATE_NAME inflate_raw_bmi2
#include "inflate.cpp"
