# same-box A/B: the fused ConvFFN's hidden ranges at 128 row tiles (B = 4, C = 384) -- tools build of the library (make AB=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export FASTVLA_HIP_LIB=$PWD/tools/bin/libfastvla_hip_ab.so
for r in 1 2; do for v in 64 128; do
  echo "== range forms up to $v row tiles (round $r)"
  FASTVLA_FFN32_RANGE_TILES=$v python tools/latency_small_batch.py 2>&1 | grep "literal B="
done; done
