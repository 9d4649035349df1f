#!/bin/bash
# development aid: up to which mean match length unrelated stretches keep a pair away from k_lane_quad (builds with -DANDI_ISLAND_MEAN_MAX)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "29 4900000 0.0004 0.03" "32 5100000 0.00001 0.0001" "32 5100000 0.0005 0.005"; do set -- $cfg
for lib in $LIBS; do
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 300 python3 bench.py --set realistic --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('realistic %-28s %-20s pass A %.3f  B/C %.3f  step %.3f frac %.3f fixups %s' % ('$cfg', '$lib', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], r['roofline']['frac'], b['fixups']))"
done; done
