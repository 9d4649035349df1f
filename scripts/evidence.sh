#!/bin/bash
# The evidence of an end state, as kept under profiles/ (named by the round it belongs to): the -m gpu suite; for the
# headline AND the C4 shape, the tree-structured and the structured set: rocprofv3 kernel-trace statistics and --pmc
# counters (the program directly after `--`, separate passes per counter set); the bench line itself (CPU baseline,
# extra.*); the bench lines of the other shapes; profiles/traffic.json regenerated from this run's counters.
# usage: scripts/evidence.sh <tag>          (then copy gpurun_out/<tag>_* into profiles/)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r07}
mkdir -p gpurun_out
timeout 1800 python3 -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${tag}_pytest.txt
shapes=("headline" "c4shape --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015" "tree --set tree" "realistic --set realistic")
for shape in "${shapes[@]}"; do
  set -- $shape; name=$1; shift
  sfx=_$name; [ $name = headline ] && sfx=
  bash scripts/kst.sh ${tag}$sfx "$@" > /dev/null 2>&1
  bash scripts/pmc.sh ${tag}$sfx "$@" > /dev/null 2>&1
  cp gpurun_out/pmc_${tag}$sfx/summary.txt gpurun_out/${tag}${sfx}_pmc.txt
  rm -rf gpurun_out/pmc_${tag}$sfx gpurun_out/prof_${tag}$sfx gpurun_out/prof_${tag}$sfx.log
done
timeout 900 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for shape in "c4shape_8x3085x2.1Mbp --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015" "c3like --genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "realistic --set realistic" "tree --set tree" "close --genomes 32 --length 5100000 --dlo 0.00002 --dhi 0.00003" "far --genomes 24 --length 10000000 --dlo 0.001 --dhi 0.05"; do
  set -- $shape; name=$1; shift
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$name.json
done
# small calls: the step as it comes, by lanes only, and with the wavefront kernel forced
{
  echo "# bench.py --steps 5 --warmup 1 on small shapes: default (tiny calls: wavefront kernel for every pair; others routed per pair) / ANDI_COOP=0 (lane scan only) / ANDI_COOP=4 (wavefront kernel forced)"
  for shape in "3 --length 1000000" "3 --length 1000000 --dlo 0.05 --dhi 0.05" "3 --length 200000" "10 --length 500000" "10 --length 1000000 --dlo 0.03 --dhi 0.03" "29 --length 100000" "29 --length 500000" "100 --length 30000" "300 --length 10000" "1000 --length 5000" "8 --length 4900000" "3 --length 1000000 --set realistic" "5 --length 1300000 --set realistic" "12 --length 1000000 --set realistic" "12 --length 1000000 --set tree"; do
    BENCH_ARGS="--genomes $shape" bash scripts/dev/ab.sh "X=1" "ANDI_COOP=0" "ANDI_COOP=4" | sed "s/^/--genomes $shape: /"
  done
} > gpurun_out/${tag}_small_calls.txt 2>&1
python3 scripts/traffic.py gpurun_out/${tag}_pmc.txt gpurun_out/${tag}_bench.json profiles/${tag}_pmc.txt > gpurun_out/${tag}_traffic.json
cat gpurun_out/${tag}_pytest.txt; head -14 gpurun_out/${tag}_kstats.txt; cut -c1-700 gpurun_out/${tag}_bench.json
