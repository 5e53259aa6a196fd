#!/bin/bash
# builds the micro-benchmarks under tools/ into tools/bin/ (git-ignored): variants with ablations for A/B timing
set -e
cd "$(dirname "$0")"
mkdir -p bin
for v in base:"" nogelu:"-DFFN_ABLATE_GELU"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=4000000 $flags ffn_micro.hip -o bin/ffn_micro_$name &
done
wait
ls -la bin
