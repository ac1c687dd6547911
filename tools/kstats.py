#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel summary (…kernel_stats.csv) as a compact table."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{n[:64]:64s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {float(r["Percentage"]):6.2f}%')
