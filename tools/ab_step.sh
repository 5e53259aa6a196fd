# usage: ab.sh name:lib ...  -> in-step A/B (two rounds) into gpurun_out/ab.log
cd /root/repo
L=gpurun_out/ab.log; : > $L
for r in 1 2; do for v in "$@"; do n=${v%%:*}; l=${v#*:}; echo "== $n (round $r)" >> $L; FASTVLA_HIP_LIB=$l timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --no-surface --no-alt --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['families'] if 'families' in d else {}
print('ms_per_step', d['ms_per_step'], 'power', d['power']['board_w_mean'], {k:v['ms_per_step'] for k,v in f.items()})" >> $L || exit 1; done; done
cat $L
