#!/bin/bash
# a long campaign of the three fuzzers with seeds of its own (scripts/dev/r6_fuzz_long.sh)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1300 python3 scripts/fuzz_parity.py 1200 31337 > gpurun_out/r06c_fuzz.txt 2>&1; echo "fuzz_parity rc $?"; tail -1 gpurun_out/r06c_fuzz.txt
timeout 800 python3 scripts/fuzz_large.py 700 4711 > gpurun_out/r06c_fuzz_large.txt 2>&1; echo "fuzz_large rc $?"; tail -1 gpurun_out/r06c_fuzz_large.txt
timeout 500 python3 scripts/fuzz_seam.py 400 271828 > gpurun_out/r06c_fuzz_seam.txt 2>&1; echo "fuzz_seam rc $?"; tail -1 gpurun_out/r06c_fuzz_seam.txt
