"""Print a window of a rocprofv3 kernel trace as a timeline: start (us), duration (us), kernel, grid.
usage: timeline.py DIR ANCHOR_SUBSTRING NTH [BEFORE AFTER]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i0 = idx[int(sys.argv[3])]
before, after = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (8, 16)
t0 = int(rows[max(i0 - before, 0)]["Start_Timestamp"])
for r in rows[max(i0 - before, 0): i0 + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}us  {r['Kernel_Name'][:64]}  grid {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
