#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for k in 12 13; do
ANDI_DEEP_K=$k timeout 300 python3 bench.py --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('c4shape K $k index %.3f  pass A %.3f  B/C %.3f  step %.3f' % (b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done
