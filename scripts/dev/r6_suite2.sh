#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/r6_bench2.json 2> gpurun_out/r6_bench2.err; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench2.json").read().strip().splitlines()[-1])
print("step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
e = d["end_to_end"]
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if not isinstance(v, (dict, list, str))})
print(e.get("traced", e.get("dist_matrix_traced")))
PY
tail -3 gpurun_out/r6_bench2.err
