#!/bin/bash
# development aid: parity of the scan with the current build, then the bench step's parts of the builds in $LIBS on three sets, twice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r3_ab3.txt
timeout 900 python3 -m pytest tests/test_scan_gpu.py tests/test_esa_gpu.py -x -q -m gpu 2>&1 | tail -3 > $out
for i in 1 2; do LIBS="$LIBS" bash scripts/dev/ab3.sh >> $out 2>&1; done
cat $out
