#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o k -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > gpurun_out/prof_tl.log 2>&1
python3 scripts/dev/timeline_sort.py gpurun_out/prof_tl > gpurun_out/timeline_sort.txt
rm -rf gpurun_out/prof_tl
