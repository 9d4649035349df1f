#!/bin/bash
# kernel-trace statistics of the default bench step -> gpurun_out/<tag>_kstats.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-x}; shift
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o k -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "$@" > gpurun_out/prof_$tag.log 2>&1
python3 scripts/kstats.py gpurun_out/prof_$tag > gpurun_out/${tag}_kstats.txt
head -12 gpurun_out/${tag}_kstats.txt
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
