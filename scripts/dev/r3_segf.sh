#!/bin/bash
# development aid: the bench step against the segment-length factor (segments of so many mean match lengths) and the shortest class
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "" "--genomes 64 --length 2100000 --dlo 0.001 --dhi 0.015"; do
for f in 8 12 16 24 32; do
ANDI_SEG_FACTOR=$f timeout 120 python3 bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-52s factor %2d  pass A %.3f  B/C %.3f  step %.3f' % ('$cfg', $f, b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done; done
