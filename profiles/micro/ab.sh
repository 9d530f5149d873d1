#!/bin/bash
# A/B of two builds of the library ON THE SAME BOX (boxes differ by +-1.5 %, so numbers from different gpurun calls do
# not compare):  bash profiles/micro/ab.sh liblsf_hip_base.so liblsf_hip.so   (files under levelsetfortran_amd/)
for rep in 1 2; do
for lib in "$@"; do
  LSF_LIB_PATH=$PWD/levelsetfortran_amd/$lib timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib', 'gs ms/step', round(d['ms_per_step'],4), 'jacobi ms/step', round(d['jacobi']['ms_per_step'],4), 'jacobi kernel us', round(d['jacobi']['roofline']['avg_launch_us'],1))"
done; done
