# usage: bash scripts/sweep.sh "<G list>" "<segment list>" "<deepK list>"
for k in ${3:-12}; do for g in ${1:-8}; do for seg in ${2:-4096}; do
ANDI_DEEP_K=$k ANDI_SCAN_G=$g python bench.py --steps 2 --warmup 1 --no-cpu-baseline --segment $seg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$k G=$g seg=$seg', round(d['value']), round(d['ms_per_step'],1), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()})"
done; done; done
