#!/bin/bash
# Run ON THE GPU BOX: shader clock and socket power while a sweep kernel runs (rocm-smi polled beside a long bench.py run).
# usage: bash profiles/micro/clocks.sh "--mode jacobi" | "--mode gs" | "--dtype f32"
ARGS=${1:---mode jacobi}
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
python3 bench.py $ARGS --steps ${STEPS:-6000} --warmup 64 --no-cpu-baseline --no-secondary > /tmp/clocks_bench.json 2>/dev/null &
PID=$!
sleep 4
for i in 1 2 3 4; do
  echo "under load ($ARGS), sample $i:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
  sleep 1
done
wait $PID
cut -c1-220 /tmp/clocks_bench.json
