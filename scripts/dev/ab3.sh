#!/bin/bash
# usage: scripts/dev/ab3.sh lib1.so lib2.so ...: pass A on the C2, C4-like and C3-like sets with each build (development aid)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "29 4900000 0.0004 0.03" "64 2100000 0.001 0.015" "32 5100000 0.0001 0.005"; do set -- $cfg
for lib in $LIBS; do
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 120 python3 bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-28s %-20s index %.3f  pass A %.3f  B/C %.3f  step %.3f' % ('$cfg', '$lib', b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done; done
