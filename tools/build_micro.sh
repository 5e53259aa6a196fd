#!/bin/bash
# builds the micro-benchmarks under tools/ into tools/bin/ (git-ignored): variants with ablations for A/B timing
set -e
cd "$(dirname "$0")"
mkdir -p bin
for v in base:"" nogelu:"-DFFN_ABLATE_GELU" nostage:"-DFFN_ABLATE_STAGE" noboth:"-DFFN_ABLATE_GELU -DFFN_ABLATE_STAGE" noboth_pd10:"-DFFN_ABLATE_GELU -DFFN_ABLATE_STAGE -DFFN_PD=10" noboth_pd3:"-DFFN_ABLATE_GELU -DFFN_ABLATE_STAGE -DFFN_PD=3" noboth_nobar:"-DFFN_ABLATE_GELU -DFFN_ABLATE_STAGE -DFFN_ABLATE_BARRIER" noreads:"-DFFN_ABLATE_GELU -DFFN_ABLATE_STAGE -DFFN_ABLATE_BARRIER -DFFN_ABLATE_READS"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -std=c++17 --offload-arch=gfx950 $flags ffn_micro.hip -o bin/ffn_micro_$name &
done
wait
ls -la bin
