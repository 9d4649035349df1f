#!/bin/bash
# development aid: calls with one segment length (small genomes): pass A compiled for 8 (seven spilled registers) / 7 (none) wavefronts per SIMD
cd "$GRAFT_REPO_ROOT" || exit 1
for occ in 8 7; do
for cfg in "--genomes 2000 --subjects 64 --length 16500 --dlo 0.001 --dhi 0.02" "--genomes 500 --subjects 64 --length 150000 --dlo 0.001 --dhi 0.02" "--genomes 29 --length 4900000 --segment 4096"; do
ANDI_LANE_OCC=$occ timeout 200 python3 bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('occ $occ %-80s step %.3f ms: index %.3f pass A %.3f B/C %.3f uniform %s' % ('$cfg', r['ms_per_step'], b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], b['scan_calls_with_one_segment_length']))"
done; done
