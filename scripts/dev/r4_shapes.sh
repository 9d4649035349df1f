#!/bin/bash
# the bench step's parts on four shapes under a list of environment settings
cd "$GRAFT_REPO_ROOT" || exit 1
for shape in "--genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015" "--genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "--set realistic" "--genomes 32 --length 5100000 --dlo 0.00002 --dhi 0.00003"; do
  echo "== $shape"
  BENCH_ARGS="$shape" bash scripts/dev/r4_ab.sh "$@"
done
