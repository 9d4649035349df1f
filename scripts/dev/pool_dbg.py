#!/usr/bin/env python3
"""pooled pass A (ANDI_POOL=1) against coop_window (ANDI_POOL=0) and the lane scan: which pairs / cells differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANDI_HIP_LIB", os.path.join(ROOT, "andi_amd", "libandihip_test.so"))  # the build with the test hooks (andi_amd/csrc/knobs.h)
import numpy as np
import andi_amd
from andi_amd import lib, synth

def rows(seqs, env, model=1):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env); lib.reload_knobs()
    try:
        ctx = andi_amd.Context(0)
        Q = andi_amd.Queries(ctx, seqs)
        esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
        got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q, model=model)
        t = ctx.timings()
        for e in esas: e.close()
        Q.close(); ctx.close()
        return got, t
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        lib.reload_knobs()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
length = int(sys.argv[2]) if len(sys.argv) > 2 else 4_900_000
dhi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
seqs, _ = synth.genome_set(n, length, 0.004, dhi, seed=5)
extra = dict(kv.split("=") for kv in sys.argv[4:])
ref, _ = rows(seqs, dict(ANDI_POOL="0", **extra))
got, t = rows(seqs, dict(ANDI_POOL="1", **extra))
bad = np.argwhere((got != ref).any(axis=2))
print("pairs that differ:", len(bad), "of", n * n - n, t.get("coop_query_nt"), t.get("fixups"))
for i, j in bad[:12]:
    d = got[i, j].astype(np.int64) - ref[i, j].astype(np.int64)
    print(i, j, "diff", d[:16].reshape(4, 4).tolist(), "sum", int(d[:16].sum()), "total", int(ref[i, j][:16].sum()))
