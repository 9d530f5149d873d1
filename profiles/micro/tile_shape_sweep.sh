for S in 128 256 384; do for W in 1x1 2x1 1x2 2x2 4x1 4x2 2x4; do
  echo -n "size $S W $W: "; LSF_GS_SKEW_W=$W python3 bench.py --size $S --steps 32 --warmup 16 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), 'ms', '%.3g'%d['value'])"
done; done
