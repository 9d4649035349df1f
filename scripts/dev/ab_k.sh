#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for k in 12 ""; do
if [ -n "$k" ]; then export ANDI_DEEP_K=$k; else unset ANDI_DEEP_K; fi
timeout 300 python3 bench.py --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('c4shape ANDI_DEEP_K=$k index %.3f  pass A %.3f  B/C %.3f  step %.3f' % (b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done
timeout 600 python3 -m pytest tests/test_configs_gpu.py tests/test_multi_gpu.py -x -q -m gpu 2>&1 | tail -3
