#!/bin/bash
# the evidence of an end state, as kept under profiles/: -m gpu suite, kernel-trace statistics and counters of the default
# bench step, the bench line itself (CPU baseline, secondary workloads), the bench lines of the other shapes.
# usage: scripts/r4_final.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${tag}_pytest.txt
bash scripts/kst.sh $tag > /dev/null 2>&1
bash scripts/pmc.sh $tag > /dev/null 2>&1
cp gpurun_out/pmc_$tag/summary.txt gpurun_out/${tag}_pmc.txt; rm -rf gpurun_out/pmc_$tag gpurun_out/prof_$tag
timeout 900 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for shape in "c4shape_8x3085x2.1Mbp --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015" "c3like --genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "realistic --set realistic" "tree --set tree" "close --genomes 32 --length 5100000 --dlo 0.00002 --dhi 0.00003" "far --genomes 24 --length 10000000 --dlo 0.001 --dhi 0.05"; do
  set -- $shape; name=$1; shift
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$name.json
done
cat gpurun_out/${tag}_pytest.txt; head -14 gpurun_out/${tag}_kstats.txt; cut -c1-700 gpurun_out/${tag}_bench.json
