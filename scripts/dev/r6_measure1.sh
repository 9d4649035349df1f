#!/bin/bash
# round 5, first GPU call: where k_coop_cold's cycles go on BOTH shapes (stats build), and the probe table's depth
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_coop
mkdir -p $out
C4="--genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015"
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_COOP_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra 2> $out/stats_bench.err | tail -1 | cut -c1-200
grep "coop_" $out/stats_bench.err > $out/work_split_bench.txt
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_COOP_STATS=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $C4 2> $out/stats_c4.err | tail -1 | cut -c1-200
grep "coop_" $out/stats_c4.err > $out/work_split_c4shape.txt
{
echo "# bench set, probe-table depth"
bash scripts/dev/ab.sh "ANDI_DEEP_K=11" "ANDI_DEEP_K=12" "ANDI_DEEP_K=13" "ANDI_COOP_SEG=32768" "ANDI_COOP_SEG=65536" "ANDI_DEEP_K=11 ANDI_COOP_SEG=32768"
echo "# C4 shape"
BENCH_ARGS="$C4" bash scripts/dev/ab.sh "ANDI_DEEP_K=11" "ANDI_DEEP_K=12" "ANDI_DEEP_K=13"
} > $out/k_sweep.txt 2>&1
cat $out/work_split_bench.txt $out/work_split_c4shape.txt $out/k_sweep.txt
