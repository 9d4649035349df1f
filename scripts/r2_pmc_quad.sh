#!/bin/bash
# counters of pass A on the C4-like set (64 x 2.1 Mbp, d <= 0.015): every pair through k_lane_cold, every pair through k_lane_quad
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for qm in -1 0; do
  export ANDI_QUAD_MATCH=$qm
  out=gpurun_out/pmc_quad_$qm
  rm -rf $out; mkdir -p $out
  i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python3 bench.py --genomes 64 --length 2100000 --dlo 0.001 --dhi 0.015 --steps 1 --warmup 0 --no-cpu-baseline > $out/p$i.log 2>&1
  done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_TOTAL_WAVEFRONTS_sum
SETS
  echo "== ANDI_QUAD_MATCH=$qm"; python3 scripts/pmc_summary.py $out | grep -A12 "k_lane_cold\|k_lane_quad"
done 2>&1 | tee gpurun_out/pmc_quad.txt
