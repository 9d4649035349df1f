#!/bin/bash
# round 2, experiment 2: pass A through LDS line buffers -- parity, then the bench step per buffer size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_scan_gpu.py -x -q -m gpu > gpurun_out/exp2_pytest.txt 2>&1
tail -5 gpurun_out/exp2_pytest.txt
for lines in 0 4,4 8,4 4,8 8,8; do
  ANDI_LANE_LINES=$lines timeout 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/exp2_$lines.json 2> gpurun_out/exp2_$lines.err
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/exp2_$lines.json'))
print('lines', '$lines', 'ms/step %.3f' % r['ms_per_step'], 'scan ms %.3f' % r['roofline']['avg_launch_ms'], 'frac %.3f' % r['roofline']['frac'], r['breakdown_ms_per_step'], r['sample_distances'])" || tail -3 gpurun_out/exp2_$lines.err
done
