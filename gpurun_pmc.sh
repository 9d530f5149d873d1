cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+\b)" | sort -u | tr '\n' ' ' | head -c 3000; echo
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -- python3 gpurun_prof1.py 512 gs 1 > /dev/null 2> gpurun_out/pmc_sq1.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- python3 gpurun_prof1.py 512 gs 1 > /dev/null 2> gpurun_out/pmc_sq2.log
python3 - <<'PY'
import csv,glob,collections
for d in ("pmc_sq1","pmc_sq2"):
    f=glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not f: print(d,"no csv"); print(open(f"gpurun_out/{d}.log").read()[-1500:]); continue
    tot=collections.Counter(); n=0
    rows=list(csv.DictReader(open(f[0])))
    for r in rows:
        if "gs_quad" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"])>=64*4000:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n+=1
    print(d, "big-plane dispatch rows", n)
    for k,v in tot.items(): print("  ",k,v)
PY
