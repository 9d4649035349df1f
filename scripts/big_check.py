"""One-off validation at config-5 genome size (50 Mbp): GPU counts vs the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import andi_amd
from andi_amd import synth
from oracle import orc

L = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
base = synth.base_codes(L, 11)
a = synth.to_bytes(synth.mutate_codes(base, 0.002, 1))
b = synth.to_bytes(synth.mutate_codes(base, 0.03, 2))
del base
t = time.time()
ctx = andi_amd.Context(0)
Q = andi_amd.Queries(ctx, [a, b])
E = andi_amd.Esa(ctx, a)
print("stage+SA %.1fs, index bytes %.2f GB, K=%s" % (time.time() - t, E.nbytes() / 1e9, "auto"), flush=True)
t = time.time()
got = andi_amd.scan_rows(ctx, [E], [0], Q)
print("scan %.3fs" % (time.time() - t), ctx.timings(), flush=True)
t = time.time()
O = orc.OracleEsa(a, sa=E.SA)  # reuse the product's suffix array; the rest is the oracle's own
want = O.dist_anchor(b)
print("oracle %.1fs" % (time.time() - t), flush=True)
print("equal:", bool((got[0, 1] == want).all()), got[0, 1][:4], want[:4], "JC", andi_amd.estimate(got[0, 1]))
assert (got[0, 1] == want).all()
