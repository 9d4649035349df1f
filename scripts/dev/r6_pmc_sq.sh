#!/bin/bash
# SQ counters of the bench step (two passes): usage scripts/dev/r6_pmc_sq.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/pmcsq_$tag
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > $out/p$i.log 2>&1
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE
SETS
python3 scripts/pmc_summary.py $out > $out/summary.txt
grep -A22 "k_pool_cold\|k_coop_cold" $out/summary.txt | head -60
rm -rf $out/p1 $out/p2 $out/p3
