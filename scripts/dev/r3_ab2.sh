#!/bin/bash
# development aid: parity, then pass A of several builds at ANDI_LANE_OCC 7 and 8 on three sets
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r3_ab2.txt
timeout 900 python3 -m pytest tests/test_scan_gpu.py tests/test_esa_gpu.py -x -q -m gpu 2>&1 | tail -3 > $out
for occ in 7 8; do
echo "--- ANDI_LANE_OCC=$occ" >> $out
ANDI_LANE_OCC=$occ LIBS="$LIBS" bash scripts/dev/ab3.sh >> $out 2>&1
done
cat $out
