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


def genome_set_fast(n, length, d_lo, d_hi, seed=1729, threads=8):
    """As genome_set, for sets of hundreds of genomes: the substitutions of a genome are drawn position by position
    (geometric gaps: their number is binomial instead of exact) and the genomes are made by a pool of threads -- two orders
    of magnitude faster; same base, same model (test/test_fasta.cxx:73-118).  Returns (list of bytes, list of d_k)."""
    from concurrent.futures import ThreadPoolExecutor
    base = base_codes(length, seed)
    ds = np.random.Generator(np.random.PCG64(seed ^ 0x5EED)).uniform(d_lo, d_hi, size=n)

    def one(k):
        rng = np.random.Generator(np.random.PCG64(seed + 1 + k))
        p = 0.75 - 0.75 * math.exp(-4.0 * float(ds[k]) / 3.0)
        out = base.copy()
        if p > 0:
            m = int(p * length * 1.1) + 1000
            pos = np.cumsum(rng.geometric(p, size=m)) - 1
            pos = pos[pos < length]
            out[pos] = (out[pos] + rng.integers(1, 4, size=len(pos), dtype=np.uint8)) & 3
        return to_bytes(out)

    with ThreadPoolExecutor(max(1, threads)) as pool:
        return list(pool.map(one, range(n))), [float(x) for x in ds]


def tree_set(n, length, d_max=2.6e-2, d_min=4.4e-4, seed=1729):
    """n genomes at the tips of a random ultrametric tree instead of a star: every pair is as far apart as twice the height
    of its last common ancestor, so pairwise distances range from d_min (sister tips) to d_max (pairs that meet at the
    root) -- the range the manual reports for the 29 E. coli/Shigella genomes (docs/manual/andi-manual.tex:316-320: 4.4e-4
    ... 2.6e-2), where a star from a common base makes every pair about d_i + d_j.  Substitutions only, as genome_set.
    Returns (list of bytes, n x n matrix of the expected pairwise distances)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x7EE))
    root = base_codes(length, seed)
    seqs = [None] * n
    height_of_lca = np.zeros((n, n))
    counter = [0]

    def grow(codes, h, tips):  # a node of height h (expected distance to each of its tips) carrying `codes`
        for a in tips:
            for b in tips:
                if a != b and height_of_lca[a, b] == 0:
                    height_of_lca[a, b] = h  # (overwritten by every lower common ancestor below)
        if len(tips) == 1:
            counter[0] += 1
            seqs[tips[0]] = to_bytes(mutate_codes(codes, h, seed + 17 * counter[0]))
            return
        k = int(rng.integers(1, len(tips)))
        for part in (tips[:k], tips[k:]):
            # the child: a tip's branch is h long; an inner node lies lower, never below d_min / 2 (sister tips are d_min apart)
            hc = 0.0 if len(part) == 1 else max(d_min / 2, h * float(rng.uniform(0.25, 0.85)))
            counter[0] += 1
            child = mutate_codes(codes, h - hc, seed + 17 * counter[0]) if len(part) > 1 else codes
            if len(part) == 1:
                grow(codes, h, part)
            else:
                for a in part:
                    for b in part:
                        height_of_lca[a, b] = 0.0
                grow(child, hc, part)

    order = [int(x) for x in rng.permutation(n)]
    grow(root, d_max / 2, order)
    np.fill_diagonal(height_of_lca, 0.0)
    return seqs, 2.0 * height_of_lca


def unrelated(length, seed):
    return to_bytes(base_codes(length, seed))


def join_contigs(seq: bytes, n_contigs, seed=7):
    """Cut a genome into contigs and join them with '!' (dsa_join,
    src/sequence.c:78-125)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cuts = sorted(1 + int(c) for c in rng.choice(len(seq) - 1, size=n_contigs - 1, replace=False))  # (O(contigs), not O(length))
    parts, last = [], 0
    for c in cuts + [len(seq)]:
        parts.append(seq[last:c])
        last = c
    return b"!".join(parts)


# ---------------------------------------------------------------- realistic structure
def _revcomp_codes(c):
    return (3 - c)[::-1]


def realistic_base(length, seed, families=((1300, 10), (5000, 7), (300, 25))):
    """A base genome with multi-copy repeats on both strands: per family (unit length, copies) one random unit is
    pasted at random places, every other copy reverse-complemented and every copy lightly mutated (IS elements,
    rRNA operons, REP-like short repeats).  andi's manual lists exactly these as what real data sets have and the
    test generator has not (docs/manual/andi-manual.tex:303-320)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.integers(0, 4, size=length, dtype=np.uint8)
    for unit_len, copies in families:
        if unit_len * copies * 2 > length:
            continue
        unit = rng.integers(0, 4, size=unit_len, dtype=np.uint8)
        for k in range(copies):
            u = unit.copy()
            m = rng.random(unit_len) < 0.004 * k  # older copies have drifted
            u[m] = (u[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            if k & 1:
                u = _revcomp_codes(u)
            at = int(rng.integers(0, length - unit_len))
            g[at:at + unit_len] = u
    return g


def evolve_codes(base, d, seed, indel_rate=0.1, inversions=2, novel_fraction=0.1):
    """A descendant of `base`: substitutions at divergence d (as mutate_codes), indels (indel_rate per substitution,
    lengths geometric with mean 6, a few up to 1 kbp), `inversions` inverted segments of 5-60 kbp (reverse
    complement in place) and about novel_fraction of the genome replaced by islands of unrelated sequence
    (5-40 kbp): the non-homologous stretches in which the scan probes at every step."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = mutate_codes(base, d, seed + 7)
    L = len(g)
    for _ in range(inversions):
        ln = int(rng.integers(5000, max(5001, min(60000, L // 4))))
        at = int(rng.integers(0, L - ln))
        g[at:at + ln] = _revcomp_codes(g[at:at + ln])
    # islands of novel sequence, then indels, applied left to right
    edits = []  # (position, deleted length, inserted codes)
    target, covered = int(novel_fraction * L), 0
    while covered < target:
        ln = int(rng.integers(5000, max(5001, min(40000, L // 5))))
        edits.append((int(rng.integers(0, L - ln)), ln, rng.integers(0, 4, size=int(ln * rng.uniform(0.5, 1.5)), dtype=np.uint8)))
        covered += ln
    p = d if d < 0.75 else 0.7
    n_indel = int(indel_rate * p * L)
    for _ in range(n_indel):
        ln = min(int(rng.geometric(1 / 6.0)), 1000) if rng.random() > 0.01 else int(rng.integers(50, 1000))
        at = int(rng.integers(0, L - ln))
        if rng.random() < 0.5:
            edits.append((at, ln, np.empty(0, np.uint8)))
        else:
            edits.append((at, 0, rng.integers(0, 4, size=ln, dtype=np.uint8)))
    edits.sort(key=lambda e: e[0])
    parts, cursor = [], 0
    for at, dele, ins in edits:
        if at < cursor:
            continue  # overlaps the previous edit
        parts.append(g[cursor:at])
        parts.append(ins)
        cursor = at + dele
    parts.append(g[cursor:])
    return np.concatenate(parts)


def realistic_set(n, length, d_lo, d_hi, seed=1729, contigs=0, **kw):
    """n genomes descending from one repeat-carrying base, each with substitutions, indels, inversions and novel
    islands of its own (star phylogeny); contigs > 0 cuts every genome into that many contigs joined by '!'
    (andi --join, src/sequence.c:78-125).  Returns (list of bytes, list of d_k)."""
    base = realistic_base(length, seed)
    drng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    ds = drng.uniform(d_lo, d_hi, size=n)
    out = []
    for k in range(n):
        s = to_bytes(evolve_codes(base, float(ds[k]), seed + 100 + k, **kw))
        out.append(join_contigs(s, contigs, seed=seed + k) if contigs > 1 else s)
    return out, [float(x) for x in ds]
