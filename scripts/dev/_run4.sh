cd "$GRAFT_REPO_ROOT"
BENCH_ARGS="" bash scripts/dev/ab.sh "X=1" "ANDI_COOP=4" "X=1"
BENCH_ARGS="--set tree" bash scripts/dev/ab.sh "X=1"
BENCH_ARGS="--genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015" bash scripts/dev/ab.sh "X=1"
