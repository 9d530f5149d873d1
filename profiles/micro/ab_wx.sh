for V in "" 4x1 2x2 4x2 8x1; do  # LSF_JAC_SH unset = the plan's own choice
  for DT in f64 f32; do
    if [ $DT = f64 ]; then A="--mode jacobi"; else A="--dtype f32"; fi
    echo -n "single-domain $DT LSF_JAC_SH='$V': "; LSF_JAC_SH=$V python3 bench.py $A --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))"
  done
done
for V in "" 4x1 2x2; do LSF_JAC_SH=$V python3 profiles/micro/core_probe.py coreonly 2>/dev/null; done
