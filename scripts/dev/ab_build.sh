#!/bin/bash
# usage: scripts/dev/ab_build.sh lib1.so lib2.so ...: the index builds alone with each build of the library
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1; do for lib in "$@"; do ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 120 python3 scripts/dev/build_time.py 8 2>&1 | tail -1; done; done
