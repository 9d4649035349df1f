#!/usr/bin/env python3
"""profiles/traffic.json from the counters of an end state: scripts/traffic.py <pmc summary> <bench line> <name the summary is kept under>.
HBM-side bytes per launch of pass A = FETCH_SIZE + WRITE_SIZE (KiB, separate --pmc passes), corrected as
MI355X_MICROARCH.md prescribes for gfx950: wide coalesced streaming reads are counted at half, so half of the window
stream's bytes (0.75 byte per scanned query nucleotide: two bit-sliced texts) are added when a wavefront kernel ran; the
scattered 8-64 byte probe loads are counted exactly (profiles/micro/r02_fetch_size_calibration.txt)."""
import json
import os
import re
import sys

summary, bench, kept_as = sys.argv[1:4]
line = json.loads(open(bench).read().strip().splitlines()[-1])
kernel = line["roofline"]["kernel"]
# a routed call's pass A ("k_coop_cold+lanes") is the wavefront kernel beside the lane scan's: their counters are added
parts = ["k_coop_cold", "k_pool_cold", "k_lane_cold", "k_lane_quad"] if "+lanes" in kernel else [kernel]
fetch = write = 0.0
seen = []
name = None
for ln in open(summary):
    if ln.startswith("=="):
        name = ln[2:].strip()
        continue
    if name and any(p in name for p in parts):
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", ln)
        if m:
            if m.group(1) == "FETCH_SIZE":
                fetch += float(m.group(2))
            else:
                write += float(m.group(2))
            if name not in seen:
                seen.append(name)
if not seen:
    sys.exit("traffic.py: no FETCH_SIZE / WRITE_SIZE of %s in %s" % (kernel, summary))
raw = (fetch + write) * 1024.0
# the wavefront kernels stream both texts bit-sliced (12 bytes per 32 symbols): 0.75 byte per scanned query nucleotide (until round 6 k_coop_cold streamed 4-bit symbols: 1 byte)
stream = 0.75 * line["roofline"]["algorithmic_bytes_per_launch"] / 2.0 if ("coop" in kernel or "pool" in kernel) else 0.0
cfg = line["config"]
key = "G%d_L%d_seg0" % (cfg["genomes"], cfg["length"])
print(json.dumps({
    "_note": "HBM-side bytes per launch of pass A from rocprofv3 --pmc (separate passes for FETCH_SIZE and WRITE_SIZE; KiB * 1024), "
             "collected with scripts/pmc.sh on MI355X by scripts/evidence.sh; correction per MI355X_MICROARCH.md (see scripts/traffic.py)",
    key: {"hbm_bytes_per_launch": raw + stream / 2.0, "raw_bytes": raw, "kernel": kernel,
          "source": "%s: FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB (%s)" % (kept_as, fetch, write, " + ".join(seen)),
          "commit": os.environ.get("ANDI_COMMIT", "commit not recorded"),  # (the GPU box has no .git: the caller passes `git rev-parse --short HEAD`)
          "correction": "+ half of the coalesced window stream (%.2f GB per launch)" % (stream / 1e9) if stream else "none"}}, indent=1))
