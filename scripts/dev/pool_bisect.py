#!/usr/bin/env python3
"""bisect the query prefix at which pooled pass A (ANDI_POOL=1) and coop_window (ANDI_POOL=0) start to differ for one pair"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANDI_HIP_LIB", os.path.join(ROOT, "andi_amd", "libandihip_test.so"))  # the build with the test hooks (andi_amd/csrc/knobs.h)
import numpy as np
import andi_amd
from andi_amd import lib, synth

si, qi = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
length = int(sys.argv[4]) if len(sys.argv) > 4 else 4_900_000
seqs, _ = synth.genome_set(n, length, 0.004, 0.03, seed=5)
S, Qs = bytes(seqs[si]), bytes(seqs[qi])
env = dict(kv.split("=") for kv in sys.argv[5:])
os.environ.update(env)
os.environ["ANDI_COOP"] = "4"

def scan(pool, q, ctx_esa=None):
    os.environ["ANDI_POOL"] = pool; lib.reload_knobs()
    ctx = andi_amd.Context(0)
    Q = andi_amd.Queries(ctx, [q])
    E = andi_amd.Esa(ctx, S, sa="device")
    got = andi_amd.scan_rows(ctx, [E], [-1], Q, model=1)
    E.close(); Q.close(); ctx.close()
    return got[0, 0]

def differs(qlen):
    a, b = scan("0", Qs[:qlen]), scan("1", Qs[:qlen])
    return (a != b).any(), (b.astype(np.int64) - a.astype(np.int64))[:16].reshape(4, 4).tolist()

print("full:", differs(len(Qs)))
lo, hi = 1000, len(Qs)
while hi - lo > 1:
    mid = (lo + hi) // 2
    if differs(mid)[0]: hi = mid
    else: lo = mid
print("first differing prefix length:", hi, differs(hi))
for d in (0, 1, 2, 5, 50, 500):
    print(hi + d, differs(hi + d))
