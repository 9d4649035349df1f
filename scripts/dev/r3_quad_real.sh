#!/bin/bash
# development aid: the realistic set with k_lane_quad's threshold varied
cd "$GRAFT_REPO_ROOT" || exit 1
for q in "" 0 64 256 1000 -1; do
ANDI_QUAD_MATCH=$q timeout 120 python3 bench.py --set realistic --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('realistic quad_match=%-6s pass A %.3f  B/C %.3f  step %.3f fixups %s' % ('$q', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], b['fixups']))"
done
