# per-launch duration of the exact-GS tile kernel against its grid size (default schedule): bash profiles/micro/slot_hist.sh [N]
N=${1:-512}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/slothist
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/slothist -- python3 bench.py --size $N --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
import numpy as np
f = glob.glob('gpurun_out/slothist/**/*kernel_trace.csv', recursive=True)[0]
d = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
           for r in csv.DictReader(open(f)) if 'k_reinit_gs_' in r['Kernel_Name'])
dur = np.array([e - s for s, e, _ in d]); grid = np.array([g for _, _, g in d])
gap = np.array([d[i + 1][0] - d[i][1] for i in range(len(d) - 1)])
print('launches', len(d), 'mean us', dur.mean() / 1e3, 'min', dur.min() / 1e3, 'median gap us', np.median(gap) / 1e3, 'sum ms', dur.sum() / 1e6)
for lo, hi in ((0, 64), (64, 256), (256, 512), (512, 768), (768, 1024), (1024, 2048), (2048, 10 ** 9)):
    m = (grid >= lo) & (grid < hi)
    if m.sum():
        print(f'tiles {lo:5d}-{hi:<10d} launches {m.sum():5d}  mean {dur[m].mean() / 1e3:7.1f} us  share of time {dur[m].sum() / dur.sum():.2f}')
PY
