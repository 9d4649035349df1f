#!/bin/bash
# round 6: CLI tests, the seam's trace by sort width, C4 at full size through the seam and through the command line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_cli.py tests/test_multi_gpu.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r07b_pytest.txt
for w in 1 2 4 8; do
  echo "== ANDI_SORT_WIDTH=$w"
  ANDI_HIP_LIB=$PWD/andi_amd/libandihip_test.so ANDI_SORT_WIDTH=$w bash scripts/dev/r6_e2e.sh 2>&1 | grep -v "^andi_hip_dist_matrix trace: slots"
done > gpurun_out/r07b_sort_width.txt 2>&1
timeout 900 python scripts/full_size.py c4 --check-rows 3 --cpu-rows 1 --out gpurun_out/r07_c4_full.json > gpurun_out/r07b_c4.txt 2>&1
timeout 1200 python scripts/full_size.py c4 --cli --out gpurun_out/r07_cli_c4.json > gpurun_out/r07b_cli_c4.txt 2>&1
cat gpurun_out/r07b_pytest.txt gpurun_out/r07b_sort_width.txt; tail -5 gpurun_out/r07b_c4.txt; tail -3 gpurun_out/r07b_cli_c4.txt
