#!/bin/bash
# usage: VAR=... scripts/dev/ab_env2.sh : C2, C4-like, realistic with the variable set / unset
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "--genomes 29 --length 4900000 --dlo 0.0004 --dhi 0.03" "--genomes 64 --length 2100000 --dlo 0.001 --dhi 0.015" "--set realistic" "--set realistic --genomes 32 --length 5100000 --dlo 0.00001 --dhi 0.0001"; do
for v in 1 0; do
if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
timeout 120 python3 bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-72s $VAR=%s pass A %.3f  B/C %.3f  step %.3f' % ('$cfg', '$v', b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done; done
