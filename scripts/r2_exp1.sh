#!/bin/bash
# round 2, experiment 1: line-fetch micro-benchmark; pass A of round 1 at limited occupancy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
./scripts/micro/line_fetch > gpurun_out/line_fetch.txt 2>&1
for pad in 0 11000 17000 25000 38000; do
  ANDI_LANE_LDS_PAD=$pad python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/occ_pad$pad.json 2> gpurun_out/occ_pad$pad.err
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/occ_pad$pad.json'))
print('pad', $pad, 'ms/step', r['ms_per_step'], 'scan ms', r['roofline']['avg_launch_ms'], r['breakdown_ms_per_step'])"
done
