#!/bin/bash
# one box: the slab launch against build/exp/liblsf_w4.so (-DLSF_SLAB_WAVES=4: 128 registers, four tiles per CU)
for v in "prod" "w4" "prod LSF_SLAB_GRID=1152" "w4 LSF_SLAB_GRID=1280"; do
  set -- $v
  echo "== $v"
  L=$PWD/levelsetfortran_amd/liblsf_hip.so; [ $1 = w4 ] && L=$PWD/build/exp/liblsf_w4.so
  env LSF_LIB_PATH=$L $2 python3 profiles/micro/slab_bench.py 512 64 2>&1 | grep -E "fast: (single|1 slab)"
done
