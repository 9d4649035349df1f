#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "32 5100000 0.000001 0.00005" "32 5100000 0.0000001 0.000005"; do set -- $cfg
for lib in $LIBS; do
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 120 python3 bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-28s %-20s index %.3f  pass A %.3f  B/C %.3f  step %.3f frac %.3f' % ('$cfg', '$lib', b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], r['roofline']['frac']))"
done; done
