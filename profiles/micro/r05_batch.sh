set -o pipefail
python -m pytest tests -x -q -m gpu -k "minmax or config3_cubic or stage_by_stage or own_decomposition or jacobi or multi" > gpurun_out/b_tests.txt 2>&1; tail -n 8 gpurun_out/b_tests.txt
LSF_TRACE=1 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-sizes > gpurun_out/b_bench.txt 2>&1; grep -E "min/max on the band" gpurun_out/b_bench.txt | tail -n 4
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/b_bench.txt") if l.startswith("{")][-1])
print(json.dumps(d["minmax"]["gs"])); print(json.dumps(d["minmax"]["jacobi"]))
print("strict jacobi", d["strict_arithmetic"]["jacobi"]["ms_per_step"], d["strict_arithmetic"]["jacobi"]["roofline"]["avg_launch_us"], "fast jacobi", d["jacobi"]["ms_per_step"], "gs", d["ms_per_step"], "strict gs", d["ms_per_step_strict"])
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/counters_avail.txt 2>&1; cd $GRAFT_REPO_ROOT; grep -c . gpurun_out/counters_avail.txt; grep -i -o -E "TCC_EA0?_[A-Z0-9_]+|TCC_(HIT|MISS|REQ|READ|WRITE)[A-Z0-9_]*|MALL[A-Z0-9_]*" gpurun_out/counters_avail.txt | sort -u | tr '\n' ' '
