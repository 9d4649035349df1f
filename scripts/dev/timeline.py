#!/usr/bin/env python3
"""The kernels of the LAST bench step in launch order (start offset, duration, queue) from a rocprofv3 --kernel-trace CSV."""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r.get("Queue_Id", "")))
rows.sort()
# the last step: from the last k_probe_table_batch on
starts = [i for i, r in enumerate(rows) if "k_probe_table" in r[2]]
i0 = starts[-1] if starts else 0
t0 = rows[i0][0]
for s, e, k, q in rows[i0:]:
    print("%9.3f ms  %9.1f us  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, k[:70]))
