#!/bin/bash
# pass A against the segment length: powers of two put the 64 lanes of a wavefront on few L2 channels
cd "$(dirname "$0")/.."
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']), round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()}, round(d['roofline']['frac'],3))"; }
for seg in ${SEGS:-4096 4160 4224 4352 4608 5000 3968 3840 3072 6144}; do TAG="seg=$seg" run --segment $seg; done
