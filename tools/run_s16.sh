set -e
cd /root/repo
L=gpurun_out/s16.log; : > $L
for v in s3:3 s1:1; do n=${v%%:*}; m=${v#*:}; echo "== tests $n" >> $L; FFN32_S16_MASK=$m FASTVLA_HIP_LIB=tools/bin/libfv_$n.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -k convffn32 >> $L 2>&1; done
for r in 1 2; do for v in base:0 s1:1 s3:3; do n=${v%%:*}; m=${v#*:}; echo "== $n (round $r)" >> $L; FFN32_S16_MASK=$m FASTVLA_HIP_LIB=tools/bin/libfv_$n.so timeout -k 10 100 python tools/ffn_bench.py 3 2>/dev/null | sed 's/16x16x32: med \([0-9]*\) us min \([0-9]*\) us[^3]*32x32x16/ref16 \2 |/' >> $L; done; done
