for g in 4 8 16 64; do for seg in 4096 16384; do
ANDI_SCAN_G=$g python bench.py --steps 2 --warmup 1 --no-cpu-baseline --segment $seg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('G=$g seg=$seg', round(d['value']), round(d['ms_per_step'],1), d['breakdown_ms_per_step'])"
done; done
