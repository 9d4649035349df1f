#!/bin/bash
# usage: scripts/dev/ab.sh lib1.so lib2.so ... : the bench step's parts with each build of the library (development aid)
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
for lib in "$@"; do
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-24s index %.3f  pass A %.3f  B/C %.3f  step %.3f' % ('$lib', b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['ms_per_step']))"
done; done
