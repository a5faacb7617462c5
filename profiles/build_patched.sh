#!/bin/bash
# profiles/build_patched.sh NAME 'sed-expression' [FLAGS] -- ablation build: kernels.hip of a scratch copy is edited with the sed
# expression (e.g. a store switched off), built, and left as ab/NAME.so.  Results of such builds are wrong by design; they tell what
# a part of the kernel costs (profiles/ab.sh prints the kernel times side by side).
set -e
NAME=$1; EXPR=$2; FLAGS=$3
B=/tmp/msnv_build_$NAME
rm -rf $B; mkdir -p $B/metasnv_amd $B/include
cp -r metasnv_amd/csrc $B/metasnv_amd/; cp include/msnv.h $B/include/
cd $B/metasnv_amd/csrc; rm -f libmsnv.so kernels.o
sed -i -e "$EXPR" kernels.hip
if cmp -s kernels.hip $OLDPWD/metasnv_amd/csrc/kernels.hip; then echo "sed expression changed nothing"; exit 1; fi
make -j8 libmsnv.so CXXFLAGS="-O2 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -pthread $FLAGS" HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter $FLAGS" > build.log 2>&1 || { tail -20 build.log; exit 1; }
cd - > /dev/null; mkdir -p ab; cp $B/metasnv_amd/csrc/libmsnv.so ab/$NAME.so; echo "ab/$NAME.so"
