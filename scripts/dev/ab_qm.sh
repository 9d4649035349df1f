#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "64 2100000 0.001 0.015" "32 5100000 0.0001 0.005"; do set -- $cfg
for qm in 128 96 64; do
ANDI_QUAD_MATCH=$qm timeout 120 python3 bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-28s quad_match %3s  pass A %.3f  B/C %.3f  step %.3f frac %.3f' % ('$cfg', '$qm', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], r['roofline']['frac']))"
done; done
