cd "$GRAFT_REPO_ROOT"
for shape in "headline" "tree --set tree" "realistic --set realistic" "c4shape --genomes 3085 --subjects 8 --length 2100000 --dlo 0.001 --dhi 0.015"; do
  set -- $shape; name=$1; shift
  echo "== $name"
  ANDI_DEBUG_STITCH=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" 2>&1 | grep "^route:" | head -2
done
