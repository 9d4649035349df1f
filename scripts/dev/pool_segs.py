#!/usr/bin/env python3
"""which 32768-segment of one pair makes pooled pass A differ from coop_window: slices of three segments, repeated"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANDI_HIP_LIB", os.path.join(ROOT, "andi_amd", "libandihip_test.so"))  # the build with the test hooks (andi_amd/csrc/knobs.h)
import numpy as np
import andi_amd
from andi_amd import lib, synth

si, qi = int(sys.argv[1]), int(sys.argv[2])
seqs, _ = synth.genome_set(12, 4_900_000, 0.004, 0.03, seed=5)
S, Qs = bytes(seqs[si]), bytes(seqs[qi])
SEG = 32768
os.environ["ANDI_COOP"] = "4"; os.environ["ANDI_COOP_SEG"] = str(SEG)
lo_s = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi_s = int(sys.argv[4]) if len(sys.argv) > 4 else len(Qs) // SEG + 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = andi_amd.Context(0)
E = andi_amd.Esa(ctx, S, sa="device")
def scan(pool, q):
    os.environ["ANDI_POOL"] = pool; lib.reload_knobs()
    Q = andi_amd.Queries(ctx, [q])
    got = andi_amd.scan_rows(ctx, [E], [-1], Q, model=1)
    Q.close()
    return got[0, 0].astype(np.int64)
for s in range(lo_s, hi_s):
    q = Qs[max(0, s - 1) * SEG:(s + 2) * SEG]
    if len(q) < 100: break
    ref = scan("0", q)
    for r in range(reps):
        got = scan("1", q)
        if (got != ref).any():
            print("segment", s, "rep", r, "diff", (got - ref)[:16].reshape(4, 4).tolist(), flush=True)
print("done")
