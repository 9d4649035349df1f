"""development aid: the probe tables of two builds of the library entry by entry (run twice with ANDI_HIP_LIB set,
the first run with an output file, the second compares): scripts/dev/table_diff.py save|cmp <file.npz>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import andi_amd
from andi_amd import synth
from conftest import rand_dna

rng = np.random.default_rng(91)
base = synth.base_codes(30000, 3)
seqs = [synth.to_bytes(base), synth.to_bytes(synth.mutate_codes(base, 0.06, 4)),
        synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.02, 5)), 9, seed=4),
        rand_dna(rng, 9000, b"AC"), rand_dna(rng, 500) * 30]
ctx = andi_amd.Context()
out = {}
for K in ("9", "11", "13"):
    os.environ["ANDI_DEEP_K"] = K
    andi_amd.lib.reload_knobs()
    for i, s in enumerate(seqs):
        e = andi_amd.Esa(ctx, s)
        k, t = e.download_index()
        out["K%s_s%d" % (K, i)] = t
        e.close()
if sys.argv[1] == "save":
    np.savez(sys.argv[2], **out)
else:
    old = np.load(sys.argv[2])
    for name in out:
        a, b = old[name], out[name]
        a = a.reshape(-1, 2) if a.ndim == 1 else a
        b = b.reshape(-1, 2) if b.ndim == 1 else b
        d = np.nonzero((a != b).any(axis=1))[0]
        print(name, "entries", len(a), "differ", len(d), "first", d[:5].tolist())
        for c in d[:3]:
            print("   code", int(c), "old", [hex(int(x)) for x in a[c]], "new", [hex(int(x)) for x in b[c]])
