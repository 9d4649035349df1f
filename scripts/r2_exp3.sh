#!/bin/bash
# round 2, experiment 3: pass A in rounds -- parity, then the bench step per line-buffer size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_scan_gpu.py -x -q -m gpu > gpurun_out/exp3_pytest.txt 2>&1
tail -5 gpurun_out/exp3_pytest.txt
for lines in 0 2,2 4,2 4,4 8,4 8,8; do
  ANDI_ROUNDS=$lines timeout 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/exp3_$lines.json 2> gpurun_out/exp3_$lines.err
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/exp3_$lines.json'))
print('rounds', '$lines', 'ms/step %.3f' % r['ms_per_step'], 'scan ms %.3f' % r['roofline']['avg_launch_ms'], 'frac %.3f' % r['roofline']['frac'], r['breakdown_ms_per_step'], r['sample_distances'])" || tail -3 gpurun_out/exp3_$lines.err
done
ANDI_HIP_LIB=$PWD/andi_amd/libandihip_stats.so ANDI_LANE_STATS=1 ANDI_ROUNDS=8,8 timeout 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/exp3_stats88.txt
grep round_stats gpurun_out/exp3_stats88.txt | head -20
