#!/bin/bash
# genomes a few substitutions apart (the regime of one sequence type, BASELINE's config 3): pass A by divergence, star sets and sets with structure
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "0.0001 0.0005" "0.00001 0.00005" "0.000001 0.000005"; do set -- $cfg
timeout 300 python3 bench.py --genomes 32 --length 5100000 --dlo $1 --dhi $2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('star      32 x 5.1 Mbp d %-20s pass A %.3f  B/C %.3f  step %.3f frac %.3f' % ('$cfg', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], r['roofline']['frac']))"
done
bash scripts/dev/st131.sh
