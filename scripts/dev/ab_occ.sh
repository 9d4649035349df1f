#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "29 4900000 0.0004 0.03" "64 2100000 0.001 0.015"; do set -- $cfg
for occ in 6 7; do
ANDI_LANE_OCC=$occ timeout 120 python3 bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-28s occ %s index %.3f  pass A %.3f  B/C %.3f  step %.3f' % ('$cfg', '$occ', b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done; done
