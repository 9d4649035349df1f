#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_LANE_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" 2>&1 >/dev/null | grep "stitch_\|lane_stats" | tail -80
