#!/bin/bash
# one box: the single launch with its flag loads (ldsys) resp. flag loads and stores (ldstsys) at system scope against the
# product library (build/exp/liblsf_<name>.so), 256^3 and 512^3
for N in 256 512; do
  for L in ${LIBS:-prod ldsys ldstsys prod ldsys ldstsys}; do
    P=$PWD/levelsetfortran_amd/liblsf_hip.so; [ $L != prod ] && P=$PWD/build/exp/liblsf_$L.so
    LSF_LIB_PATH=$P python3 bench.py --size $N --steps 64 --warmup 64 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$N $L', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms/sweep %.4f' % (d['roofline']['avg_launch_us'] / 64e3))"
  done
done
