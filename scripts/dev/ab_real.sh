#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in $LIBS; do
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('C2        %-20s pass A %.3f  B/C %.3f  step %.3f fixups %s' % ('$lib', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], b['fixups']))"
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 120 python3 bench.py --set realistic --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('realistic %-20s pass A %.3f  B/C %.3f  step %.3f fixups %s' % ('$lib', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], b['fixups']))"
done
