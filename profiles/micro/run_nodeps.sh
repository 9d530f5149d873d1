for S in slots skew; do
 echo "== $S nodeps"; LSF_GS_NODEPS_EXPERIMENT=1 LSF_GS_SCHEDULE=$S python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launches_per_sweep'], d['roofline']['avg_launch_us'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LSF_GS_SCHEDULE=skew rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/skewtrace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/skewtrace/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gs_' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
f=glob.glob('gpurun_out/skewtrace/**/*kernel_trace.csv',recursive=True)[0]
import numpy as np
d=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),int(r['Grid_Size_X'])//64 if 'Grid_Size_X' in r else 0) for r in csv.DictReader(open(f)) if 'gs_skew' in r['Kernel_Name']]
d.sort()
dur=np.array([e-s for s,e,_ in d]); gap=np.array([d[i+1][0]-d[i][1] for i in range(len(d)-1)]); grid=np.array([g for _,_,g in d])
print('n',len(d),'dur mean',dur.mean(),'gap mean',np.median(gap))
# duration vs grid size bins
for lo,hi in ((0,256),(256,1024),(1024,2048),(2048,3072),(3072,6144),(6144,10**9)):
    m=(grid>=lo)&(grid<hi)
    if m.sum(): print(lo,hi,m.sum(),'dur',dur[m].mean(), 'per-tile ns', (dur[m]/np.maximum(grid[m],1)).mean())
PY
