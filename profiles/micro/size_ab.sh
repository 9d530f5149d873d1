#!/bin/bash
# one box: product library against build/exp/liblsf_old.so (an earlier commit's sources) at the sizes / arithmetics in ARGS
IFS=';' read -ra CASES <<< "${ARGS:---size 1024 --steps 16 --warmup 16;--size 512 --arith strict --steps 16 --warmup 8;--size 768 --steps 16 --warmup 16}"
for a in "${CASES[@]}"; do
  for L in ${LIBS:-prod old prod old}; do
    P=$PWD/levelsetfortran_amd/liblsf_hip.so; [ $L != prod ] && P=$PWD/build/exp/liblsf_$L.so
    LSF_LIB_PATH=$P python3 bench.py $a --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a', '$L', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms/sweep %.4f' % (d['roofline']['avg_launch_us'] * d['roofline']['launches_per_sweep'] / 1e3))"
  done
done
