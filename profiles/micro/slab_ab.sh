#!/bin/bash
# one box, experiment builds: k_reinit_gs_persist around skew_tile<PUSH> (LSF_SLAB_USE_PERSIST=2) at 256^3 with one part of the
# PUSH write back switched off per library (xWB2: the copies of the cell results, xWB3: the copies of the wall points)
for L in exp xWB2 xWB3 exp xWB2 xWB3; do
  echo "== $L"
  env LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so LSF_SLAB_USE_PERSIST=2 python3 profiles/micro/slab_bench.py ${N:-256} 64 2>&1 | grep -E "fast: (1 slab)"
done
