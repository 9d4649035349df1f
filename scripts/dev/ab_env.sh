#!/bin/bash
# usage: VAR=ANDI_NO_TAIL_FOLD scripts/dev/ab_env.sh : the three sets with the variable set / unset (development aid)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "29 4900000 0.0004 0.03" "64 2100000 0.001 0.015" "32 5100000 0.0001 0.005"; do set -- $cfg
for v in 1 0; do
if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
timeout 120 python3 bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-28s $VAR=%s index %.3f  pass A %.3f  B/C %.3f  step %.3f' % ('$cfg', '$v', b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done; done
