#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python3 -m pytest tests/test_esa_gpu.py tests/test_scan_gpu.py -x -q -m gpu 2>&1 | tail -3
for lib in base test; do
  [ -f andi_amd/libandihip_$lib.so ] || continue
  export ANDI_HIP_LIB=$PWD/andi_amd/libandihip_$lib.so
  echo "## $lib"
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); e=d['end_to_end']; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if not isinstance(v,(dict,list,str))})"
done
