#!/bin/bash
# pass A time of the bench step under a list of environment settings: scripts/dev/ab.sh "ANDI_COOP=4" "ANDI_COOP=8 ANDI_COOP_SEG=65536" ...
cd "$GRAFT_REPO_ROOT" || exit 1
export ANDI_HIP_LIB=${ANDI_HIP_LIB:-$PWD/andi_amd/libandihip_test.so} # the build with the experiment switches (andi_amd/csrc/knobs.h)
for cfg in "$@"; do
  env $cfg python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); b=d['breakdown_ms_per_step']; print('%-50s step %.2f ms  build %.2f  passA %.3f  B/C %.3f  frac %.3f fixups %d' % ('$cfg', d['ms_per_step'], b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], d['roofline']['frac'], b['fixups']))"
done
