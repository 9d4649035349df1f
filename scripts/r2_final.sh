#!/bin/bash
# the evidence of an end state, as kept under profiles/: -m gpu suite, kernel-trace statistics and counters of the default bench
# step, the bench line itself (with the CPU baseline), the bench lines of the other shapes.  usage: scripts/r2_final.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r02zz}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${tag}_pytest.txt
bash scripts/kst.sh $tag > /dev/null 2>&1
bash scripts/pmc.sh $tag > /dev/null 2>&1
cp gpurun_out/pmc_$tag/summary.txt gpurun_out/${tag}_pmc.txt; rm -rf gpurun_out/pmc_$tag gpurun_out/prof_$tag
timeout 900 python3 bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
bash scripts/r2_endstate.sh $tag > gpurun_out/${tag}_shapes.txt 2>&1
cat gpurun_out/${tag}_pytest.txt; head -14 gpurun_out/${tag}_kstats.txt; cut -c1-600 gpurun_out/bench_${tag}.json; tail -4 gpurun_out/${tag}_shapes.txt
