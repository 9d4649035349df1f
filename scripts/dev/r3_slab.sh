#!/bin/bash
# development aid: the bench's device allocations carved out of slabs of several sizes (experimental build), against separate hipMallocs
cd "$GRAFT_REPO_ROOT" || exit 1
p() { python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('$1', round(r['ms_per_step'],3), {k:round(v,3) for k,v in r['breakdown_ms_per_step'].items() if k.startswith(('index','scan_cold','scan_st'))})"; }
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | p plain
for mb in 256 1024 4096 16384; do
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_slab.so ANDI_SLAB_MB=$mb python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | p slab_$mb
done
for sk in 16 64 512; do
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_skew.so ANDI_SKEW=$sk python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | p skew_$sk
done
