#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python3 scripts/fuzz_parity.py 400 777 > gpurun_out/r06b_fuzz.txt 2>&1; echo "fuzz_parity rc $?"; tail -1 gpurun_out/r06b_fuzz.txt
timeout 500 python3 scripts/fuzz_large.py 360 23 > gpurun_out/r06b_fuzz_large.txt 2>&1; echo "fuzz_large rc $?"; tail -1 gpurun_out/r06b_fuzz_large.txt
timeout 400 python3 scripts/fuzz_seam.py 240 5 > gpurun_out/r06b_fuzz_seam.txt 2>&1; echo "fuzz_seam rc $?"; tail -1 gpurun_out/r06b_fuzz_seam.txt
