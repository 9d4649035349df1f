#!/bin/bash
# round 6: k_pool_cold with the stretches or-ed into the bits (default) against the second bitmap (libandihip_ebits.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
python -m pytest tests/test_coop_gpu.py -m gpu -x -q 2>&1 | tail -3
python scripts/fuzz_large.py 90 911 2>&1 | tail -3
for rep in 1 2; do
BENCH_ARGS="--genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015" bash scripts/dev/ablib.sh libandihip.so libandihip_ebits.so
done
BENCH_ARGS="--genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" bash scripts/dev/ablib.sh libandihip.so libandihip_ebits.so
BENCH_ARGS="--set tree" bash scripts/dev/ablib.sh libandihip.so libandihip_ebits.so
} > gpurun_out/r07c_ab.txt 2>&1
cat gpurun_out/r07c_ab.txt
