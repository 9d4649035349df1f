#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "32 5100000 0.00001 0.0001" "32 5100000 0.0005 0.005"; do set -- $cfg
timeout 300 python3 bench.py --set realistic --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('realistic %-28s index %.3f  pass A %.3f  B/C %.3f  step %.3f frac %.3f fixups %s' % ('$cfg', b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step'], r['roofline']['frac'], b['fixups']))"
done
