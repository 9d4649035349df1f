#!/bin/bash
# development aid: knock-out timing of the lane scan's pass A on the structured set (results are wrong by design): what
# repeated K-mers (1), long lucky matches (2), gap counting (4), the table (8), probes altogether (16) cost there
cd "$GRAFT_REPO_ROOT" || exit 1
export ANDI_HIP_LIB=$PWD/andi_amd/libandihip_knock.so
for k in 0 1 2 4 8 16; do
  ANDI_KNOCK=$k timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --set realistic 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); b=d['breakdown_ms_per_step']; print('knock $k: passA %.3f  B/C %.3f' % (b['scan_cold_pass'], b['scan_stitch_reduce']))"
done
