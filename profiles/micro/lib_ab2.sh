#!/bin/bash
# A/B/A/B on one box: the product library against an experiment build (build/exp/liblsf_$1.so); kernel ms per sweep of the
# one-block-per-tile launch, 64 after 64.   usage: lib_ab2.sh name [sizes] [shapes]
for L in "" "$PWD/build/exp/liblsf_$1.so" "" "$PWD/build/exp/liblsf_$1.so"; do
  echo "== lib ${L:-product}"
  LSF_LIB_PATH=$L python3 profiles/micro/stream_ab.py time sizes=${2:-512,256} sweeps=64 shapes=${3:-default,c1x4} 2>&1 | grep persist
done
