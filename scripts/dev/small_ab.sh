#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "--genomes 29 --length 4900000 --dlo 0.0004 --dhi 0.03" "--genomes 2000 --subjects 64 --length 16500 --dlo 0.001 --dhi 0.02" "--genomes 2000 --subjects 64 --length 16500 --dlo 0.00001 --dhi 0.0005" "--genomes 500 --subjects 64 --length 150000 --dlo 0.00001 --dhi 0.0005"; do
for lib in $LIBS; do
ANDI_HIP_LIB=$PWD/andi_amd/$lib timeout 200 python3 bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['breakdown_ms_per_step']; print('%-78s %-18s step %.3f ms: pass A %.3f B/C %.3f frac %.3f' % ('$cfg', '$lib', r['ms_per_step'], b['scan_cold_pass'], b['scan_stitch_reduce'], r['roofline']['frac']))"
done; done
