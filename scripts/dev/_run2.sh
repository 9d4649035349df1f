cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=gpurun_out/r5_route.txt
timeout 1500 python3 -m pytest tests/test_coop_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -12 > $out
for shape in "headline" "tree --set tree" "realistic --set realistic" "c3like --genomes 32 --length 5100000 --dlo 0.0001 --dhi 0.005" "c4shape --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015"; do
  set -- $shape; name=$1; shift
  for cfg in "X=1" "ANDI_COOP=0"; do
  env $cfg timeout 600 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); b=d['breakdown_ms_per_step']; print('%-10s %-12s step %.2f ms  build %.2f  passA %.3f  B/C %.3f  frac %.3f fixups %d %s %s' % ('$name', '$cfg', d['ms_per_step'], b['index_build'], b['scan_cold_pass'], b['scan_stitch_reduce'], d['roofline']['frac'], b['fixups'], d['roofline']['kernel'], b.get('pass_a_query_nt_fraction')))" >> $out 2>&1
  done
done
cat $out
