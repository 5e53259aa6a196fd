for i in 1 2; do
for v in 0 1; do
FASTVLA_FUSED_LETTERBOX=$v python bench.py --steps 12 --warmup 3 --no-train --no-train-unfrozen --no-cpu-baseline --no-surface --no-alt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$v', d['ms_per_step'])"
done
done
