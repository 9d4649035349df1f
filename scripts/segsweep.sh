# usage: segsweep.sh "<genomes length dlo dhi>" "<segments>"
set -- $1 "$2"
for seg in $5; do
python bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --segment $seg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seg=$seg', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()})"
done
