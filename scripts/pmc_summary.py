#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output per kernel and counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    if not any(s in k for s in ("k_scan", "k_coop", "k_pool", "k_lane", "k_rounds", "k_pair", "k_pack", "k_suffix", "k_probe", "k_child", "k_plcp", "k_lcp", "k_phi", "k_kmer", "k_min")):
        continue
    print("==", k)
    for c in sorted(acc[k]):
        print("   %-40s %18.0f  (dispatches %d)" % (c, acc[k][c], calls[k][c]))
