#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_esa_gpu.py -x -q -m gpu -k "suffix or device_built" --durations=5 2>&1 | tail -25
python3 - <<'PY'
import time, numpy as np, andi_amd
from andi_amd import synth
ctx = andi_amd.Context(0)
for L in (1_000_000, 2_100_000, 4_900_000, 50_000_000):
    seq = synth.to_bytes(synth.base_codes(L, 5))
    E = andi_amd.Esa(ctx, seq, sa="device", build=False); E.close()  # warm (workspace allocation)
    ctx.timings_reset()
    t0 = time.time(); E = andi_amd.Esa(ctx, seq, sa="device", build=False); dt = time.time() - t0
    t = ctx.timings()
    print("device SA  L=%9d  n=%10d  sort %.2f ms (%d rounds)   stage+sort wall %.1f ms" % (L, E.n, t["sa_ms"], t["sa_rounds"], dt * 1e3), flush=True)
    if L <= 4_900_000:
        t0 = time.time(); want = andi_amd.suffix_array(E.RS); th = time.time() - t0
        print("   host SA-IS %.0f ms; equal: %s" % (th * 1e3, bool((E.SA == want).all())), flush=True)
    E.close()
PY
