cd "$GRAFT_REPO_ROOT"
bash scripts/kst.sh r5route > /dev/null 2>&1
head -30 gpurun_out/r5route_kstats.txt
