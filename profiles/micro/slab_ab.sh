#!/bin/bash
# one box: the slab launch (profiles/micro/slab_bench.py) with experiment builds build/exp/liblsf_<name>.so against the product
# library.  How the write back of the wall tiles was found (DESIGN_HISTORY.md section 4.1 item 6): with -DLSF_EXPERIMENTS builds that
# launched k_reinit_gs_persist around skew_tile<PUSH> on the one-slab launch's buffers and switched the parts of PUSH off one at
# a time (commits 6d97f0c..fadb6da carry those switches; they are gone from the sources now).
for L in ${LIBS:-prod}; do
  P=$PWD/levelsetfortran_amd/liblsf_hip.so; [ $L != prod ] && P=$PWD/build/exp/liblsf_$L.so
  echo "== $L"
  LSF_LIB_PATH=$P python3 profiles/micro/slab_bench.py ${N:-256} 64 2>&1 | grep -E "fast: (single|1 slab|2 slab\(s\) fine=0)"
done
