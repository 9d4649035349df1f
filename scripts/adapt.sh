#!/bin/bash
# per-pair segment lengths: parity, then the bench step against the factor (segment >= factor x mean match length)
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']), round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()}, round(d['roofline']['frac'],3))"; }
TAG="uniform 4096" ANDI_UNIFORM_SEGMENTS=1 run
for f in ${FACTORS:-2 4 8 16 32}; do TAG="adaptive factor=$f" ANDI_SEG_FACTOR=$f run; done
