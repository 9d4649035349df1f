#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average) from a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(float)
cnt = defaultdict(int)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        tot[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        cnt[k] += 1
allt = sum(tot.values())
print("%-60s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%-60s %8d %14.1f %12.1f %7.2f" % (k[:60], cnt[k], tot[k], tot[k] / cnt[k], 100 * tot[k] / allt))
