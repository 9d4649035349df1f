#!/bin/bash
# small genomes (one segment length per call): 2000 x 16.5 kbp (mitochondria-like) and 500 x 150 kbp (phage / plasmid-like)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "--genomes 2000 --subjects 64 --length 16500 --dlo 0.001 --dhi 0.02" "--genomes 500 --subjects 64 --length 150000 --dlo 0.001 --dhi 0.02" "--genomes 2000 --subjects 64 --length 16500 --dlo 0.00001 --dhi 0.0005"; do
timeout 200 python3 bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-80s %8.0f pairs/s step %.3f ms: index %.3f pass A %.3f B/C %.3f frac %.3f adaptive %s uniform %s' % ('$cfg', r['value'], r['ms_per_step'], b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['roofline']['frac'], b['scan_calls_with_per_pair_segments'], b['scan_calls_with_one_segment_length']))"
done
