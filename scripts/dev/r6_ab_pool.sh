#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
C4="--genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015"
echo "# bench set"; bash scripts/dev/ab.sh "ANDI_POOL=0" "ANDI_POOL=1" "$@"
echo "# C4 shape"; BENCH_ARGS="$C4" bash scripts/dev/ab.sh "ANDI_POOL=0" "ANDI_POOL=1" "$@"
echo "# tree"; BENCH_ARGS="--set tree" bash scripts/dev/ab.sh "ANDI_POOL=0" "ANDI_POOL=1"
echo "# realistic"; BENCH_ARGS="--set realistic" bash scripts/dev/ab.sh "ANDI_POOL=0" "ANDI_POOL=1"
