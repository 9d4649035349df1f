import os, sys
sys.path.insert(0, ".")
import numpy as np
import andi_amd
from andi_amd import synth
from oracle import orc
base = synth.base_codes(300000, 5)
seqs = [synth.to_bytes(synth.mutate_codes(base, d, 10 + k)) for k, d in enumerate((0.0, 0.0005, 0.005, 0.02, 0.05, 0.15))]
want = orc.dist_matrix(seqs, model=orc.M_JC, threads=4)
for coop in (4, 2):
    for seg in (0, 4096, 1000, 300000):
        os.environ["ANDI_COOP"] = str(coop)
        andi_amd.lib.reload_knobs()
        got = andi_amd.dist_matrix(seqs, model=andi_amd.M_JC, segment=seg)
        bad = np.argwhere((got != want).any(axis=2))
        print("coop", coop, "seg", seg, "bad pairs", bad.tolist())
        for i, j in bad[:3]:
            print("  ", i, j, (got[i, j].astype(np.int64) - want[i, j]).tolist())
