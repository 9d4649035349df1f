#!/bin/bash
# knock-out timing of pass A (diagnostic build; results are wrong on purpose): which part of a chain step costs what
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for k in 0 1 2 4 8 16 3 7 23; do
  ANDI_KNOCK=$k ANDI_HIP_LIB=$PWD/andi_amd/libandihip_knock.so timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/knock.json 2> gpurun_out/knock.err
  python3 -c "
import json
r=json.load(open('gpurun_out/knock.json'))
print('knock %2d  pass A %.3f ms   (1 = repeated K-mers, 2 = slides, 4 = gap counts, 8 = table + what follows, 16 = the whole probe)' % ($k, r['roofline']['avg_launch_ms']))" || tail -3 gpurun_out/knock.err
done
