#!/bin/bash
# Calibration of FETCH_SIZE / TCC_EA0_RDREQ for scattered accesses: the line_fetch
# micro-benchmark touches a KNOWN number of distinct lines per launch (chains x iterations, random over 1 GiB),
# so bytes per request and requests per line follow.  One rocprofv3 --pmc pass per counter set.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/line_fetch_pmc
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p1 -o p1 -- ./scripts/micro/line_fetch > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $out/p2 -o p2 -- ./scripts/micro/line_fetch > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/line_fetch_pmc/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_lines" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["16-B piece of a 64-B line", "64-B line, 4 loads of one lane", "128-B line, 8 loads of one lane", "64-B line, LDS-DMA",
         "128-B line, LDS-DMA", "64-B line shared by 4 lanes", "128-B line shared by 8 lanes"]
chains = [1, 1, 1, 1, 1, 4, 8]
line_b = [64, 64, 128, 64, 128, 64, 128]
print("# footprint 1 GiB, 8 blocks per CU; timed launch of every mode (400 dependent steps x 8192 x 256 / lanes-per-chain chains)")
print("%-34s %12s %12s %12s %12s %9s %9s %9s" % ("mode", "lines", "EA_RDREQ", "TCC_MISS", "FETCH_SIZE_B", "B/RDREQ", "RDREQ/line", "FETCH/line_bytes"))
timed = [d for d in sorted(rows) if True]
# dispatches come in pairs (warm-up of 10 steps, timed 400 steps); the first 14 are footprint 1 GiB at occupancy 8
for m in range(7):
    d = timed[2 * m + 1]
    v = rows[d]
    lines = 8192 * 256 / chains[m] * 400
    fetch = v.get("FETCH_SIZE", 0) * 1024  # FETCH_SIZE is reported in KiB
    rd = v.get("TCC_EA0_RDREQ_sum", 0)
    print("%-34s %12.0f %12.0f %12.0f %12.0f %9.1f %9.2f %9.2f" % (names[m], lines, rd, v.get("TCC_MISS_sum", 0), fetch,
          fetch / rd if rd else 0, rd / lines, fetch / (lines * line_b[m])))
PY
