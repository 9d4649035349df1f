#!/bin/bash
# development aid: parity of the scan with the current build, then pass A old / new (ANDI_LANE_OCC 6 and 7) on three sets
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_scan_gpu.py tests/test_esa_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3_ab.txt
LIBS="libandihip_old.so libandihip.so" bash scripts/dev/ab3.sh >> gpurun_out/r3_ab.txt 2>&1
echo "--- ANDI_LANE_OCC=7" >> gpurun_out/r3_ab.txt
ANDI_LANE_OCC=7 LIBS="libandihip.so" bash scripts/dev/ab3.sh >> gpurun_out/r3_ab.txt 2>&1
LIBS="libandihip_old.so libandihip.so" bash scripts/dev/ab3.sh >> gpurun_out/r3_ab.txt 2>&1
cat gpurun_out/r3_ab.txt
