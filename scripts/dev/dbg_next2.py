import os, sys
sys.path.insert(0, ".")
import numpy as np
import andi_amd
from andi_amd import synth
seqs, ds = synth.realistic_set(6, 250000, 0.001, 0.06, seed=77, novel_fraction=0.07)
ctx = andi_amd.Context()
Q = andi_amd.Queries(ctx, seqs)
esas = [andi_amd.Esa(ctx, s) for s in seqs]
out = andi_amd.scan_rows(ctx, esas, list(range(6)), Q, 1, 100)
out = andi_amd.scan_rows(ctx, esas, list(range(6)), Q, 1, 100)
