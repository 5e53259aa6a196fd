#!/bin/bash
# tools/ffn32_variants.sh name1:"-DFLAG ..." name2:"..." -- builds variants of convffn32.hip into tools/bin/libfv_<name>.so (they
# travel with gpurun).  On the GPU box: tools/ffn32_run.sh name1 name2 ...   (one process per variant: compare against the
# 16x16x32 column printed beside each, which is the same code in every variant)
set -e
cd "$(dirname "$0")/../vla-from-fastvlm_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/bin
rm -f ../../tools/bin/libfv_*.so
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000"
build() {
  hipcc $FLAGS $2 -c convffn32.hip -o /tmp/cf32_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v convffn32) /tmp/cf32_$1.o -o ../../tools/bin/libfv_$1.so
}
for v in "$@"; do build "${v%%:*}" "${v#*:}" & done
wait
ls ../../tools/bin/ | grep libfv_
