# usage: bash profiles/micro/cmp.sh "256 512 1024" "slots skew"
for N in $1; do for S in $2; do
LSF_GS_SCHEDULE=$S python bench.py --size $N --steps 8 --warmup 8 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($N, '$S', round(d['ms_per_step'],3), d['roofline']['launches_per_sweep'], round(d['roofline']['avg_launch_us'],1))"
done; done
