#!/bin/bash
# the bench step's parts with a list of builds of the library (andi_amd/<lib>): scripts/dev/ablib.sh libandihip.so libandihip_x.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "$@"; do
  ANDI_HIP_LIB=$PWD/andi_amd/$lib python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); b=d['breakdown_ms_per_step']; print('%-36s step %.2f ms  build %.2f  passA %.3f  B/C %.3f  frac %.3f fixups %d %s' % ('$lib', d['ms_per_step'], b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], d['roofline']['frac'], b['fixups'], d['roofline']['kernel']))"
done
