#!/bin/bash
# tools/variants.sh <source.hip> name1:"-DFLAG ..." name2:"..." -- builds variants of ONE source file of csrc/ into
# tools/bin/libv_<name>.so (they travel with gpurun); run with FASTVLA_HIP_LIB=tools/bin/libv_<name>.so
set -e -o pipefail
src="$1"; shift
cd "$(dirname "$0")/../vla-from-fastvlm_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/bin
rm -f ../../tools/bin/libv_*.so
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000"
base="${src%.hip}"
build() {
  hipcc $FLAGS $2 -c $src -o /tmp/var_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "build/$base.o") /tmp/var_$1.o -o ../../tools/bin/libv_$1.so
}
for v in "$@"; do build "${v%%:*}" "${v#*:}" & done
wait
ls ../../tools/bin/ | grep libv_
