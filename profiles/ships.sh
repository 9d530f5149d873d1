#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/ships.sh r04 ["512 fast" "512 strict" ...]
# Counters of the kernels that SHIP: for every configuration (size, arithmetic [, LSF_GS_SKEW_W]) of the exact
# ordering -- the instances the library picks by itself -- three separate rocprofv3 PMC passes with --kernel-trace only
# (FETCH_SIZE | WRITE_SIZE | SQ issue counters | read requests by size | write requests by size), 32 sweeps after 32, and the Jacobi kernels once.  profiles/ships_summarize.py
# writes profiles/<tag>_ships.json and profiles/traffic.json (keyed by kernel INSTANCE and size: what bench.py looks up).
set -u
TAG=${1:-r04}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ships_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
if [ $# -eq 0 ]; then set -- "512 fast gs" "512 strict gs" "256 fast gs" "256 strict gs" "1024 fast gs" "1024 strict gs" "512 fast jacobi" "512 strict jacobi" "256 fast jacobi" "1024 fast jacobi"; fi
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
for CFG in "$@"; do
  set -- $CFG; N=$1; A=$2; M=$3; K=32; [ "$N" -ge 1024 ] && K=16
  D="$OUT/${N}_${A}_${M}"
  for P in FETCH_SIZE WRITE_SIZE SQ RDREQ WRREQ; do
    C=$P; [ $P = SQ ] && C="$SQ"
    # requests of the L2s to the fabric BY SIZE (FETCH_SIZE counts a request as 64 bytes whatever its size)
    [ $P = RDREQ ] && C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
    [ $P = WRREQ ] && C="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D/$P" -- python3 bench.py --size $N --steps $K --warmup $K --arith $A --mode $M --no-cpu-baseline --no-secondary > "$D.$P.json" 2> "$D.$P.log"
    echo "$CFG $P done" >&2
  done
  echo "$N $A $M $((2 * K))" >> "$OUT/configs.txt"
done
python3 profiles/ships_summarize.py "$OUT" "$TAG"
mkdir -p "gpurun_out/profiles_$TAG"
cp profiles/${TAG}_ships.json profiles/traffic.json "gpurun_out/profiles_$TAG/"
