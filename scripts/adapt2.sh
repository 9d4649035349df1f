#!/bin/bash
cd "$(dirname "$0")/.."
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']), round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()}, round(d['roofline']['frac'],3))"; }
for s0 in ${SEG0S:-1536 2048}; do for f in ${FACTORS:-8 12 16 24}; do TAG="seg0=$s0 factor=$f" ANDI_SEG0=$s0 ANDI_SEG_FACTOR=$f run; done; done
