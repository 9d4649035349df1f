#!/bin/bash
# development aid: pass A against resident wavefronts per SIMD (unused LDS per block limits them; the kernel compiled for 8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r3_occ.txt
: > $out
for pad in 0 4000 7000 11000 17000 25000; do
ANDI_LANE_LDS_PAD=$pad timeout 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; w=min(8,(160*1024)//(16384+$pad)); print('lds_pad %5d  waves/SIMD %d  pass A %.3f ms  B/C %.3f' % ($pad, w, b['scan_cold_pass'], b['scan_stitch_reduce']))" >> $out
done
cat $out
