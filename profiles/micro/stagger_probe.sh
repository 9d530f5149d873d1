#!/bin/bash
# Round 5: is the one-lane-per-cell work term bound by PHASE LOCKING (both tiles of a CU load together, then march together)?
# Work-term launch (every tile of a sweep in one launch, dependencies ignored) with the second block of every CU delayed by S
# microseconds (LSF_PROBE_STAGGER, experiment build x).  Run ON THE GPU BOX: bash profiles/micro/stagger_probe.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LSF_LIB_PATH=$PWD/build/exp/liblsf_x.so
B="python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["ms_per_step"],4), "ms/step; avg launch us", round(d["roofline"]["avg_launch_us"],1))'
for A in fast strict; do
  for S in 0 5 10 15 20 30; do
    echo -n "c1x4 $A nodeps stagger $S us: "; LSF_PROBE_STAGGER=$S LSF_GS_SKEW_W=c1x4 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B --arith $A 2>/dev/null | python3 -c "$J"
  done
done
for S in 0 15; do
echo -n "c1x4 fast phases nodeps stagger $S: "; LSF_PROBE_STAGGER=$S LSF_GS_SKEW_W=c1x4 LSF_TRACE_TILES=1 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B 2>&1 | grep "tile phases" | tail -n 1
done
# did the stagger reach every CU?  (stderr line "stagger probe: N CU entries used")
LSF_PROBE_STAGGER=15 LSF_GS_SKEW_W=c1x4 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B 2>&1 | grep "stagger probe" | tail -n 1
# one tile per CU (experiment build solo: 8 KB of LDS padding): the phases of a tile that has the CU to itself
export LSF_LIB_PATH=$PWD/build/exp/liblsf_solo.so
for A in fast strict; do
  echo -n "solo c1x4 $A nodeps: "; LSF_GS_SKEW_W=c1x4 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B --arith $A 2>/dev/null | python3 -c "$J"
  echo -n "solo c1x4 $A phases: "; LSF_GS_SKEW_W=c1x4 LSF_TRACE_TILES=1 LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=skew timeout -k 10 120 $B --arith $A 2>&1 | grep "tile phases" | tail -n 1
done
