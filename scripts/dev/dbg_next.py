import os, sys
sys.path.insert(0, ".")
import numpy as np
import andi_amd
from andi_amd import synth
sys.path.insert(0, "tests")
seqs, ds = synth.realistic_set(6, 250000, 0.001, 0.06, seed=77, novel_fraction=0.07)
ctx = andi_amd.Context()
def rows(seg):
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, s) for s in seqs]
    out = andi_amd.scan_rows(ctx, esas, list(range(6)), Q, 1, seg)
    t = ctx.timings()
    for e in esas: e.close()
    Q.close()
    return out, t
for seg in (100, 2048):
    os.environ["ANDI_KNOCK"] = "64"
    andi_amd.lib.reload_knobs()
    ref, _ = rows(seg)
    os.environ.pop("ANDI_KNOCK")
    andi_amd.lib.reload_knobs()
    got, t = rows(seg)
    bad = np.argwhere((got != ref).any(axis=2))
    print("segment", seg, "differing pairs", len(bad), bad[:10].tolist(), "fixups", t["fixups"])
    if len(bad):
        i, j = bad[0]
        print(got[i, j], ref[i, j])
