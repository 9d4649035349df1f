#!/bin/bash
# the bench lines of the other shapes (C4 shape, C3-like, realistic), as kept under profiles/
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r02zz}
mkdir -p gpurun_out
python3 bench.py --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${tag}_c4shape.json 2>gpurun_out/b.err
python3 bench.py --genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${tag}_c3like.json 2>gpurun_out/b.err
python3 bench.py --set realistic --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${tag}_realistic.json 2>gpurun_out/b.err
for f in c4shape c3like realistic; do python3 -c "
import json
r=json.load(open('gpurun_out/bench_${tag}_$f.json'))
print('$f', round(r['value']), 'pairs/s  step %.2f ms  pass A %.2f ms  frac %.4f' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']), r['breakdown_ms_per_step'])"; done
