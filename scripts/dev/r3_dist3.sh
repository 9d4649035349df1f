#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
p() { python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('$1', round(r['ms_per_step'],3), {k:round(v,3) for k,v in r['breakdown_ms_per_step'].items() if k.startswith(('index','scan_cold','scan_st'))})"; }
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | p plain
ANDI_BENCH_EARLY_INIT=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | p dist_early
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | p dist_late
