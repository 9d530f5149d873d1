#!/bin/bash
# A/B of experiment builds of the library: bash profiles/micro/lib_ab.sh "a b" [shape] ; libs are build/exp/liblsf_<name>.so
for L in $1; do
  export LSF_LIB_PATH=$PWD/build/exp/liblsf_$L.so
  echo "== lib $L"; bash profiles/micro/gs_quick.sh ${2:-2x2}
  for A in fast strict; do echo -n "nodeps $A: "; LSF_GS_SKEW_W=${2:-2x2} LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew python3 bench.py --steps 16 --warmup 8 --arith $A --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4))"; done
done
