#!/bin/bash
# A/B of library builds on ONE box, Jacobi path only:  bash profiles/micro/ab_jacobi.sh lib1.so lib2.so ...
for rep in 1 2; do
for lib in "$@"; do
  LSF_LIB_PATH=$PWD/levelsetfortran_amd/$lib timeout -k 10 300 python bench.py --mode jacobi --no-secondary --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib', 'jacobi ms/step', round(d['ms_per_step'],4), 'kernel us', round(d['roofline']['avg_launch_us'],1))"
done; done
