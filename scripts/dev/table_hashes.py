"""Hashes of the probe tables of a fixed list of subjects (development aid: run once per library build,
ANDI_HIP_LIB=..., and compare the outputs)."""
import hashlib
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import andi_amd
from andi_amd import synth


def subjects():
    rng = np.random.default_rng(7)
    yield "tiny", b"ACGTTGCA"
    yield "homopolymer", b"A" * 5000
    yield "two-letter", bytes(rng.choice(list(b"AC"), 20000).astype(np.uint8))
    yield "poly-then-random", b"A" * 3000 + synth.to_bytes(synth.base_codes(50000, 3)) + b"T" * 2000
    yield "repeats", synth.to_bytes(synth.base_codes(3000, 5)) * 40
    yield "contigs", synth.join_contigs(synth.to_bytes(synth.base_codes(200000, 9)), 12, seed=4)
    yield "short-contigs", b"!".join([b"ACG", b"ACGT", b"AC", b"ACGTA", b"ACG"] * 50)
    yield "realistic", synth.realistic_set(1, 300000, 0.001, 0.01, seed=11)[0][0]
    yield "random-2M", synth.to_bytes(synth.base_codes(2000000, 1))
    yield "random-4.9M", synth.to_bytes(synth.base_codes(4900000, 2))


ctx = andi_amd.Context()
if True:
    for name, seq in subjects():
        for K in ([None] if len(seq) > 1000000 else [None, "5", "8", "11", "13"]):
            import os
            if K is None:
                os.environ.pop("ANDI_DEEP_K", None)
            else:
                os.environ["ANDI_DEEP_K"] = K
            andi_amd.lib.reload_knobs()  # (the library reads its switches once)
            e = andi_amd.Esa(ctx, seq, sa="device")
            k, t = e.download_index()
            print(name, "K", k, "flags", e.flags().tolist(), hashlib.sha256(t.tobytes()).hexdigest()[:20])
            e.close()
