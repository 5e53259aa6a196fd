#!/bin/bash
# on the GPU box: time every variant built by tools/ffn32_variants.sh (arguments: variant names; default: all)
cd "$(dirname "$0")/.."
names="$@"; [ -z "$names" ] && names=$(ls tools/bin | grep '^libfv_' | sed 's/libfv_//; s/.so//')
for r in 1 2; do for v in $names; do
  echo "== $v (round $r)"; FASTVLA_HIP_LIB=tools/bin/libfv_$v.so timeout -k 10 100 python tools/ffn_bench.py 3 2>/dev/null | sed 's/16x16x32: med \([0-9]*\) us min \([0-9]*\) us[^3]*32x32x16/ref16 \2 |/'
done; done
