cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_skew; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
for V in normal nodeps; do
  if [ $V = nodeps ]; then export LSF_GS_NODEPS_EXPERIMENT=1; fi
  LSF_GS_SCHEDULE=skew rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1_$V -- python3 bench.py --steps 2 --warmup 0 --mode gs --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/p1_$V.log
  LSF_GS_SCHEDULE=skew rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/p2_$V -- python3 bench.py --steps 2 --warmup 0 --mode gs --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/p2_$V.log
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out=sys.argv[1]
for V in ("normal","nodeps"):
    tot={}
    for p in ("p1","p2"):
        for f in glob.glob(os.path.join(out,f"{p}_{V}","**","*counter_collection.csv"),recursive=True):
            for r in csv.DictReader(open(f)):
                if "gs_skew" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
    w=tot.get("SQ_WAVES",1)
    print(V, {k:round(v/w,1) for k,v in tot.items()}, "waves",w)
PY
