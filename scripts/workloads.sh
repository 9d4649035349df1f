# three synthetic workloads (C4-like, C3-like, C2 = the bench default), one line each
for cfg in "64 2100000 0.001 0.015" "32 5100000 0.0001 0.005" "29 4900000 0.0004 0.03"; do set -- $cfg
python bench.py --genomes $1 --length $2 --dlo $3 --dhi $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4), {a:round(b,2) for a,b in d['breakdown_ms_per_step'].items()})"
done
