#!/bin/bash
# per-kernel time of the split-bf16 attention kernels with parts of the dq kernel removed (FASTVLA_ATTN_ABL bits: 1 = stage the first chunk only,
# 2 = no products, 4 = no score arithmetic); needs `make -C vla-from-fastvlm_amd/csrc AB=1`
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FASTVLA_HIP_LIB=$R/tools/bin/libfastvla_hip_ab.so
cd /tmp && export TMPDIR=/tmp
for abl in ${ABLS:-0 1 2 4}; do
  export FASTVLA_ATTN_ABL=$abl
  rm -rf $R/gpurun_out/attn_abl_$abl
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/attn_abl_$abl -- python3 $R/tools/attn_bench.py > $R/gpurun_out/attn_abl_$abl.log 2>&1
  f=$(find $R/gpurun_out/attn_abl_$abl -name "*kernel_stats.csv" | head -1)
  echo "== ABL=$abl"; grep -E "attn|attention" $f | sed 's/(float const.*)",/ /; s/(fv::.*)",/ /' | cut -d, -f1-3 | cut -c30-140
  find $R/gpurun_out/attn_abl_$abl -name "*kernel_trace.csv" -delete
done
