#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for shape in "--set realistic" "" "--set tree" "--genomes 24 --length 10000000 --dlo 0.001 --dhi 0.05" "--genomes 12 --length 1000000 --set realistic" "--genomes 5 --length 1300000 --set realistic" "--genomes 3 --length 1000000 --set realistic"; do
  echo "# $shape"; BENCH_ARGS="$shape" bash scripts/dev/ab.sh "X=1"
done
timeout 1200 python3 -m pytest tests/test_scan_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3
