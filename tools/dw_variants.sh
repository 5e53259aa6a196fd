#!/bin/bash
# tools/dw_variants.sh name1:"-DFLAG ..." name2:"..." -- builds variants of tower_kernels.hip into tools/bin/libdw_<name>.so (they
# travel with gpurun).  On the GPU box: for v in ...; FASTVLA_HIP_LIB=tools/bin/libdw_$v.so python tools/dwpair_bench.py
set -e -o pipefail
cd "$(dirname "$0")/../vla-from-fastvlm_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/bin
rm -f ../../tools/bin/libdw_*.so
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000"
build() {
  hipcc $FLAGS $2 -c tower_kernels.hip -o /tmp/dwv_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v tower_kernels) /tmp/dwv_$1.o -o ../../tools/bin/libdw_$1.so
}
for v in "$@"; do build "${v%%:*}" "${v#*:}" & done
wait
ls ../../tools/bin/ | grep libdw_
