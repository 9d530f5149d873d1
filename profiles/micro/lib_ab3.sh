#!/bin/bash
# A/B of experiment libs on one box
for L in tab fsl tsl tab fsl tsl; do
  echo "== lib $L"
  LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so python3 profiles/micro/stream_ab.py time sizes=${1:-512,256} sweeps=64 shapes=${2:-default,c1x4} 2>&1 | grep persist
done
