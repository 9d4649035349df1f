"""Seeded synthetic genome sets.

Same model as the reference's generator (test/test_fasta.cxx:73-118): a uniform
ACGT base sequence and, per genome, an exact number of substitutions at
uniformly random positions, never to the same base; divergence d is converted
with p = 0.75 - 0.75*exp(-4d/3) unless raw.  The PRNG is numpy's PCG64, so the
bytes are identical on every machine (the reference's generator depends on the
C++ standard library it was built with).
"""
import math

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def base_codes(length, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 4, size=length, dtype=np.uint8)


def mutate_codes(base, d, seed, raw=False):
    length = len(base)
    p = d if raw else 0.75 - 0.75 * math.exp(-4.0 * d / 3.0)
    k = int(round(p * length))
    rng = np.random.Generator(np.random.PCG64(seed))
    out = base.copy()
    if k > 0:
        pos = rng.choice(length, size=k, replace=False)
        out[pos] = (out[pos] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    return out


def to_bytes(codes):
    return _ACGT[codes].tobytes()


def pair(length, d, seed=42):
    """Base genome and one mutated copy (the test_random.sh shape)."""
    b = base_codes(length, seed)
    return to_bytes(b), to_bytes(mutate_codes(b, d, seed + 1))


def genome_set(n, length, d_lo, d_hi, seed=1729):
    """n genomes derived from one base; genome k diverges from the base by d_k
    drawn uniformly from [d_lo, d_hi].  Returns (list of bytes, list of d_k)."""
    b = base_codes(length, seed)
    drng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    ds = drng.uniform(d_lo, d_hi, size=n)
    return [to_bytes(mutate_codes(b, float(ds[k]), seed + 1 + k)) for k in range(n)], [float(x) for x in ds]


def unrelated(length, seed):
    return to_bytes(base_codes(length, seed))


def join_contigs(seq: bytes, n_contigs, seed=7):
    """Cut a genome into contigs and join them with '!' (dsa_join,
    src/sequence.c:78-125)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cuts = sorted(int(c) for c in rng.choice(np.arange(1, len(seq)), size=n_contigs - 1, replace=False))
    parts, last = [], 0
    for c in cuts + [len(seq)]:
        parts.append(seq[last:c])
        last = c
    return b"!".join(parts)
