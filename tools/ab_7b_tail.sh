# same-box A/B of launch_gemm's tail sub-launch on the 7B, B = 16 step (tools build of the library: make AB=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export FASTVLA_HIP_LIB=$PWD/tools/bin/libfastvla_hip_ab.so
for r in 1 2 3; do for v in 0 1; do
  if [ $v = 1 ]; then export FASTVLA_NO_GEMM_TAIL=1; else unset FASTVLA_NO_GEMM_TAIL; fi
  python bench.py --model fastvlm-7b --batch 16 --llm-precision 1 --steps 10 --warmup 3 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('7b no_tail=$v', d['ms_per_step'])"
done; done
