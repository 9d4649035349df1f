#!/bin/bash
# the cooperative pass A on the bench set: how its work splits (stats build), kernel times, SQ counters
# usage: scripts/dev/r4_coop_prof.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/coop_$tag
mkdir -p $out
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_COOP_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" 2> $out/stats.txt | tail -1 | cut -c1-300
grep coop_stats $out/stats.txt | tail -18
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 > $out/bench.json
python3 -c "
import json; d=json.load(open('$out/bench.json')); print('ms/step', d['ms_per_step'], d['breakdown_ms_per_step'], 'frac', d['roofline']['frac'])"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > $out/p$i.log 2>&1
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE
SETS
python3 scripts/pmc_summary.py $out > $out/summary.txt
grep -A30 "k_coop_cold" $out/summary.txt | head -34
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
