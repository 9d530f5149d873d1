"""Print the per-kernel rows of a rocprofv3 --kernel-trace --stats --output-format csv directory: name, calls, total, average."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
