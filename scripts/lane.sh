#!/bin/bash
# lane scan: parity first, then access statistics and the bench step for a few settings
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_LANE_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep lane_stats
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']), round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()}, round(d['roofline']['frac'],3))"; }
for occ in ${OCCS:-6 4}; do
TAG="lane occ=$occ adaptive" ANDI_LANE_OCC=$occ run
TAG="lane occ=$occ uniform 4096" ANDI_LANE_OCC=$occ ANDI_UNIFORM_SEGMENTS=1 run
done
