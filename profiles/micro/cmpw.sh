# usage: bash profiles/micro/cmpw.sh "512 1024" "1 2 4 4x2"   (skewed tiles, wavefronts per tile WY or WYxWZ)
for N in $1; do for W in $2; do
LSF_GS_SKEW_W=$W LSF_GS_SCHEDULE=skew python bench.py --size $N --steps 8 --warmup 8 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($N, 'W=$W', round(d['ms_per_step'],3), d['roofline']['launches_per_sweep'], round(d['roofline']['avg_launch_us'],1))"
done; done
