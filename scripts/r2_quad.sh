#!/bin/bash
# pass A with the pairs of long matches through k_lane_quad (ANDI_QUAD_MATCH = mean sampled match length from which; -1: none) on four sets
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "29 4900000 0.0004 0.03" "64 2100000 0.001 0.015" "32 5100000 0.0001 0.005" "32 5100000 0.00001 0.0005"; do set -- $cfg
for qm in -1 128 64 256; do
  ANDI_QUAD_MATCH=$qm timeout 300 python3 bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/quad.json 2> gpurun_out/quad.err
  python3 -c "
import json
r=json.load(open('gpurun_out/quad.json'))
print('$cfg quad_match %4d  pass A %.3f ms  step %.3f ms  parity %s' % ($qm, r['roofline']['avg_launch_ms'], r['ms_per_step'], r.get('parity_vs_cpu_baseline')))" || tail -3 gpurun_out/quad.err
done; done 2>&1 | tee gpurun_out/quad_timings.txt
