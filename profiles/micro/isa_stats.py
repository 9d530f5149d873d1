"""Instruction statistics of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only): per kernel the number of
vector / scalar / LDS / vector-memory instructions, waits and barriers, and the same between consecutive s_barrier's
(the marching steps of the exact-ordering kernels are the long barrier-to-barrier stretches).
usage: isa_stats.py file.s [kernel-name-substring] [--segments]"""
import re
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
segs = "--segments" in sys.argv
kern, cur = {}, None
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        kern[cur] = []
        continue
    if cur is None:
        continue
    t = line.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kern[cur].append(op)
    if op == "s_endpgm":
        cur = None


def klass(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


for k, ops in kern.items():
    if flt not in k:
        continue
    c = {}
    for op in ops:
        c[klass(op)] = c.get(klass(op), 0) + 1
    f64 = sum(1 for op in ops if op.endswith("_f64") or "_f64_" in op)
    print(f"{k}: {len(ops)} instructions, " + ", ".join(f"{a} {b}" for a, b in sorted(c.items())) + f", fp64 {f64}")
    if segs:
        seg, out = {}, []
        for op in ops:
            if op == "s_barrier":
                out.append(seg)
                seg = {}
            else:
                seg[klass(op)] = seg.get(klass(op), 0) + 1
        out.append(seg)
        for i, sg in enumerate(out):
            print(f"   segment {i}: " + ", ".join(f"{a} {b}" for a, b in sorted(sg.items())))
