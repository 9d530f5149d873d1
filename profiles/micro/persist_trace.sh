cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ptrace
LSF_GS_SCHEDULE=dataflow rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ptrace -- python3 bench.py --size ${1:-512} --steps 16 --warmup 16 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ptrace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gs_' in r['Name']: print(r['Name'][:50], r['Calls'], 'avg ms', float(r['AverageNs'])/1e6, 'per sweep ms', float(r['AverageNs'])/1e6/16)
PY
