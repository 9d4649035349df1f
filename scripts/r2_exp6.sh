#!/bin/bash
# round 2, experiment 6: pass A as a two-phase loop (k_lane_stream) against the straight-line step (k_lane_cold)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_scan_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/exp6_pytest.txt 2>&1
tail -3 gpurun_out/exp6_pytest.txt
for cfg in "1 6" "1 4" "0 6"; do
  set -- $cfg
  for s in "" "--dlo 0.001 --dhi 0.015 --length 2100000 --genomes 64" "--dlo 0.0001 --dhi 0.005 --length 5100000 --genomes 32" "--set realistic"; do
  ANDI_LANE_STREAM=$1 ANDI_LANE_OCC=$2 timeout 300 python3 bench.py $s --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/exp6.json 2> gpurun_out/exp6.err
  python3 -c "
import json
r=json.load(open('gpurun_out/exp6.json'))
print('stream $1 occ $2 | $s | step %.2f ms  pass A %.3f ms  frac %.3f' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']), r['breakdown_ms_per_step']['scan_stitch_reduce'], r['sample_distances'][0])" || tail -3 gpurun_out/exp6.err
  done
done
