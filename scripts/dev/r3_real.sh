#!/bin/bash
# development aid: the realistic set with the builds in $LIBS, then kernel statistics of the last one
cd "$GRAFT_REPO_ROOT" || exit 1
LIBS="$LIBS" bash scripts/dev/ab_real.sh 2>&1 | grep realistic
last=$(echo $LIBS | awk '{print $NF}')
export ANDI_HIP_LIB=$PWD/andi_amd/$last
bash scripts/kst.sh r3real --set realistic > /dev/null 2>&1
head -14 gpurun_out/r3real_kstats.txt
