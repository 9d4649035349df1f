#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
echo "# realistic"; BENCH_ARGS="--set realistic" bash scripts/dev/ab.sh "ANDI_SEG0=4096" "ANDI_SEG0=8192" "ANDI_SEG0=16384" "ANDI_SEG0=4096 ANDI_SEG_FACTOR=8" "ANDI_SEG0=8192 ANDI_SEG_FACTOR=8" "ANDI_SEG0=8192 ANDI_SEG_FACTOR=4"
for shape in "" "--set tree" "--genomes 24 --length 10000000 --dlo 0.001 --dhi 0.05" "--genomes 32 --length 5100000 --dlo 0.00002 --dhi 0.00003" "--genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "--genomes 12 --length 1000000 --set realistic" "--genomes 29 --length 500000"; do
  echo "# $shape"; BENCH_ARGS="$shape" bash scripts/dev/ab.sh "X=1" "ANDI_SEG0=4096" "ANDI_SEG0=8192"
done
