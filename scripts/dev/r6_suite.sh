#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for shape in "--set tree" "--genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "--set realistic" "--genomes 24 --length 10000000 --dlo 0.001 --dhi 0.05" "--genomes 3 --length 1000000" "--genomes 29 --length 500000" "--genomes 32 --length 5100000 --dlo 0.00002 --dhi 0.00003"; do
  echo "# $shape"; BENCH_ARGS="$shape" bash scripts/dev/ab.sh "ANDI_POOL=0" "ANDI_POOL=1"
done
ANDI_BENCH_C3_STRONG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6_bench_c3.json 2> gpurun_out/r6_bench_c3.err; tail -c 1500 gpurun_out/r6_bench_c3.json; tail -3 gpurun_out/r6_bench_c3.err
