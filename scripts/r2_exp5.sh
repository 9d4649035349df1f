#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_scan_gpu.py tests/test_esa_gpu.py -x -q -m gpu > gpurun_out/exp5_pytest.txt 2>&1
tail -3 gpurun_out/exp5_pytest.txt
timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/exp5.json 2> gpurun_out/exp5.err
python3 -c "
import json,sys
r=json.load(open('gpurun_out/exp5.json'))
print('ms/step %.3f' % r['ms_per_step'], 'scan ms %.3f' % r['roofline']['avg_launch_ms'], 'frac %.4f' % r['roofline']['frac'], r['breakdown_ms_per_step'], r['sample_distances'])" || tail -3 gpurun_out/exp5.err
