#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats output directory into a small CSV (per kernel: calls, avg/min/max us,
share) — the file that gets committed under profiles/.  usage: prof_summary.py <dir> <out.csv> [min_calls]"""
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
min_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv under", d)
    sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
with open(out, "w") as o:
    o.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for r in rows:
        if int(r["Calls"]) < min_calls:
            continue
        name = r["Name"].split("(")[0]
        o.write(f"\"{name[:90]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f},"
                f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{r['Percentage']}\n")
print(open(out).read()[:6000])
