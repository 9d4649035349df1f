#!/bin/bash
cd $GRAFT_REPO_ROOT
for shape in "--set tree" "--genomes 24 --length 10000000 --dlo 0.001 --dhi 0.05" "--genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "--set realistic" "--contigs 60" "--genomes 32 --length 5100000 --dlo 0.00002 --dhi 0.00003"; do
  echo "== $shape"
  BENCH_ARGS="$shape" scripts/dev/ablib.sh libandihip.so
done
