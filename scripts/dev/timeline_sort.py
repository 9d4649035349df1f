#!/usr/bin/env python3
"""The kernels of ONE subject's suffix sort (between the 3rd and 4th k_sa_keys0) from a rocprofv3 --kernel-trace CSV."""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:110], r.get("Queue_Id", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_sa_keys0" in r[2]]
i0, i1 = starts[3], starts[4]
t0 = rows[i0][0]
busy = 0
for s, e, k, q in rows[i0:i1]:
    busy += e - s
    print("%9.3f ms  %9.1f us  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, k))
print("span %.3f ms, busy %.3f ms" % ((rows[i1][0] - t0) / 1e6, busy / 1e6))
