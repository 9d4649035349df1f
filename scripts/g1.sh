#!/bin/bash
# one lane per chain experiment: parity, then the bench step for G = 1 (two occupancies) and G = 4
cd "$(dirname "$0")/.."
ANDI_SCAN_G=1 timeout 900 python -m pytest tests/test_scan_gpu.py -x -q -m gpu 2>&1 | tail -3
for cfg in "4 8" "1 8" "1 4"; do set -- $cfg
ANDI_SCAN_G=$1 ANDI_SCAN_OCC=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('G=$1 occ=$2', round(d['value']), round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()})"
done
