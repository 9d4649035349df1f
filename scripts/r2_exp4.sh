#!/bin/bash
# round 2, experiment 4: pass A in rounds -- trips of the compute loop per round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for lines in 4,2 4,4 8,8; do
for passes in 1 2 3 5 100; do
  ANDI_ROUNDS_PASSES=$passes ANDI_ROUNDS=$lines timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/exp4.json 2> gpurun_out/exp4.err
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/exp4.json'))
print('rounds', '$lines', 'passes', $passes, 'scan ms %.3f' % r['roofline']['avg_launch_ms'], r['sample_distances'][0])" || tail -3 gpurun_out/exp4.err
done
done
for passes in 1 3 100; do
ANDI_ROUNDS_PASSES=$passes ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_LANE_STATS=1 ANDI_ROUNDS=4,4 timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/exp4_stats44_$passes.txt
echo passes $passes; grep round_stats gpurun_out/exp4_stats44_$passes.txt | grep -E "rounds|passes|line_fills"
done
