cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5_pytest_a.txt
BENCH_ARGS="" bash scripts/dev/ablib.sh libandihip.so libandihip_occ8.so > gpurun_out/r5_occ8.txt 2>&1
cat gpurun_out/r5_pytest_a.txt gpurun_out/r5_occ8.txt
