#!/bin/bash
# the pooled kernel with parts switched off (libandihip_pk.so, ANDI_KNOCK=bits): VALU/SALU instructions and time of k_pool_cold
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export ANDI_HIP_LIB=$PWD/andi_amd/libandihip_pk.so
for k in "$@"; do
  out=gpurun_out/knock_$k
  mkdir -p $out
  ANDI_KNOCK=$k rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o p1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $BENCH_ARGS > $out/p1.log 2>&1
  python3 scripts/pmc_summary.py $out | grep -A5 "k_pool_cold" | awk -v k=$k '/GRBM|VALU|SALU|VMEM|LDS/{printf "%s=%.3fG ", $1, $2/1e9} END{print " knock=" k}'
  rm -rf $out
done
