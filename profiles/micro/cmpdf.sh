# usage: bash profiles/micro/cmpdf.sh "256 512" "2x2 4x2"   (dataflow schedule, wavefronts per tile)
for N in $1; do for W in $2; do
LSF_GS_SKEW_W=$W python bench.py --size $N --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($N, 'W=$W', round(d['ms_per_step'],3), d['roofline']['kernel'])"
done; done
