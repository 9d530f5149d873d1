#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/collect.sh r01
# 1) rocprofv3 --kernel-trace --stats of the default bench.py command
# 2) separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a short run, per MI355X_MICROARCH.md "HBM"
# Raw CSVs stay under gpurun_out/ (scratch); profiles/summarize.py writes the committed summaries.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
# the fp32 Jacobi path (BASELINE configuration 5), same command shape
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_f32" -- python3 bench.py --dtype f32 > "$OUT/bench_f32_under_rocprof.json" 2> "$OUT/trace_f32.log"
# the driver's own command (--steps 20 --warmup 5) under the kernel trace: its persist-kernel average must agree with the
# roofline.avg_launch_us of the BENCH line
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_s20" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/bench_s20_under_rocprof.json" 2> "$OUT/trace_s20.log"
# min/max flow, exact ordering: kernel trace of 1 + 16 iterations (k_minmax_fp<0/1/2> per iteration)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mm" -- python3 profiles/micro/mm_pmc.py 512 gs 16 > "$OUT/mm_under_rocprof.txt" 2> "$OUT/trace_mm.log"
# PMC passes at STEADY STATE (round 1 used 2-sweep launches, i.e. fill and drain): PMC_STEPS sweeps after as many warm-up
# sweeps; the counters are summed over both launches and divided by the sweeps they cover (summarize.py)
PMC_STEPS=${PMC_STEPS:-32}
for C in FETCH_SIZE WRITE_SIZE; do
  for M in gs jacobi; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_$M" -- python3 bench.py --steps $PMC_STEPS --warmup $PMC_STEPS --mode $M --no-cpu-baseline --no-secondary > /dev/null 2> "$OUT/pmc_${C}_$M.log"
  done
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_f32" -- python3 bench.py --steps $PMC_STEPS --warmup $PMC_STEPS --dtype f32 > /dev/null 2> "$OUT/pmc_${C}_f32.log"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_mm" -- python3 profiles/micro/mm_pmc.py 512 gs 16 > /dev/null 2> "$OUT/pmc_${C}_mm.log"
done
cat > "$OUT/cal.py" <<'PY'
import sys, torch
sys.path.insert(0, '.')
import levelsetfortran_amd as L
n = 511
phi = torch.rand((n + 1) ** 3, dtype=torch.float64, device='cuda')
nb = torch.zeros(phi.numel(), dtype=torch.int32, device='cuda'); sb = torch.zeros_like(nb)
for _ in range(2):
    L.narrowBand(n, n, n, 0.01, phi, nb, sb)
# 4 B per lane: k_pack<float> over the whole field reads n floats and writes n floats
import ctypes
from levelsetfortran_amd import _lib
f = torch.rand((n + 1) ** 3, dtype=torch.float32, device='cuda'); g = torch.empty_like(f)
box = _lib.LsfBox(n + 1, n + 1, n + 1, 0, 0, 0, n, n, n)
for _ in range(2):
    _lib.check(_lib.load().lsf_pack_box_f32(f.data_ptr(), ctypes.byref(box), _lib.int3((0, 0, 0)), _lib.int3((n + 1,) * 3), g.data_ptr(), None))
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_cal" -- python3 "$OUT/cal.py" > /dev/null 2> "$OUT/pmc_${C}_cal.log"
done
python3 profiles/summarize.py "$OUT" "$TAG" $((2 * PMC_STEPS))
bash profiles/sq_pass.sh "$TAG"
# gpurun only carries gpurun_out/ back: stage the committed summaries there (copy them into profiles/ afterwards)
mkdir -p "gpurun_out/profiles_$TAG"
cp profiles/${TAG}_* "gpurun_out/profiles_$TAG/"
