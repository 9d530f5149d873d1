#!/bin/bash
# Run ON THE GPU BOX from the repo root, AFTER collect.sh (traffic.json must be current):  bash profiles/bench_set.sh r02
# The bench.py lines committed as profiles/<tag>_bench_*.json, all on one box.
set -u
TAG=${1:-r02}
D=gpurun_out/profiles_$TAG
mkdir -p "$D"
run() { local name=$1; shift; python3 bench.py "$@" 2> "$D/${TAG}_bench_$name.err" | grep '^{' | tail -1 > "$D/${TAG}_bench_$name.json"; echo "$name: $(cut -c1-160 "$D/${TAG}_bench_$name.json")"; }
run default
run steps20 --steps 20 --warmup 5
run 256 --size 256 --steps 16 --warmup 16 --no-cpu-baseline
run 1024 --size 1024 --steps 16 --warmup 16 --no-cpu-baseline
run strict --arith strict --steps 16 --warmup 8 --no-cpu-baseline
run jacobi_strict --mode jacobi --arith strict --steps 16 --warmup 8 --no-cpu-baseline --no-secondary
run f32 --dtype f32 --no-cpu-baseline
run f32_768 --dtype f32 --size 768 --no-cpu-baseline
run f32_1536 --dtype f32 --size 1536 --steps 8 --warmup 4 --no-cpu-baseline
python3 profiles/micro/decomp_step.py > "$D/${TAG}_decomp_step.txt" 2>&1; cat "$D/${TAG}_decomp_step.txt"
find "$D" -name '*.err' -size 0 -delete
