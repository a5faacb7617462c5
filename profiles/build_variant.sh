#!/bin/bash
# profiles/build_variant.sh NAME "EXTRA_FLAGS" -- builds a variant of libmsnv.so (extra -D flags for both compilers) in a scratch
# copy of csrc/ and leaves it as ab/NAME.so (ab/ is git-ignored; it travels to the GPU box for same-box A/B runs: profiles/ab.sh)
set -e
NAME=$1; FLAGS=$2
B=/tmp/msnv_build_$NAME
rm -rf $B; mkdir -p $B/metasnv_amd $B/include
cp -r metasnv_amd/csrc $B/metasnv_amd/; cp include/msnv.h $B/include/
cd $B/metasnv_amd/csrc; rm -f *.o libmsnv.so
make -j8 libmsnv.so CXXFLAGS="-O2 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -pthread $FLAGS" HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter $FLAGS" > build.log 2>&1 || { tail -20 build.log; exit 1; }
cd - > /dev/null; mkdir -p ab; cp $B/metasnv_amd/csrc/libmsnv.so ab/$NAME.so; echo "ab/$NAME.so"
