#!/usr/bin/env python3
"""Calls of the size at which pass A is routed per pair (scan.h: ScanArgs.route): random sets of 8-12 genomes of
3-5 Mbp (star, tree, structured, close, mixed), every model the kernel takes, through andi_hip_scan_rows --
the call as it comes (routed), the lane scan (ANDI_COOP=0) and the wavefront kernel forced (ANDI_COOP=4 or 5) must agree
bit for bit, and one sampled subject row must equal the oracle's.  scripts/fuzz_large.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANDI_HIP_LIB", os.path.join(ROOT, "andi_amd", "libandihip_test.so"))  # the build with the test hooks (andi_amd/csrc/knobs.h)
import numpy as np

import andi_amd
from andi_amd import lib, synth
from oracle import orc


def rows(seqs, model, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    lib.reload_knobs()
    try:
        ctx = andi_amd.Context(0)
        Q = andi_amd.Queries(ctx, seqs)
        esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
        ctx.timings_reset()
        got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q, model=model)
        t = ctx.timings()
        for e in esas:
            e.close()
        Q.close()
        ctx.close()
        return got, t
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.reload_knobs()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    t_end = time.time() + budget
    case = 0
    while time.time() < t_end:
        kind = str(rng.choice(["star", "tree", "realistic", "close", "mixed", "joined", "strands", "sparse"]))
        n = int(rng.integers(10, 15)) if kind != "sparse" else 12
        length = int(rng.integers(3_000_000, 5_000_000))
        if rng.random() < 0.8:  # most cases of the size from which a call is routed: n^2 * length >= 2^14 * 32768 symbols
            length = max(length, (1 << 29) // (n * n) + 100_000)
        seed = int(rng.integers(1, 1 << 30))
        if kind == "star":
            seqs, _ = synth.genome_set(n, length, 0.0004, float(rng.choice([0.01, 0.03, 0.06])), seed=seed)
        elif kind == "tree":
            seqs, _ = synth.tree_set(n, length, seed=seed)
        elif kind == "realistic":
            seqs, _ = synth.realistic_set(n, length, 0.001, 0.03, seed=seed)
        elif kind == "close":
            base = synth.base_codes(length, seed)
            seqs = [synth.to_bytes(synth.mutate_codes(base, float(rng.choice([2e-5, 2e-4])), seed + 1 + k)) for k in range(n)]
        elif kind == "joined":  # clean genomes cut into contigs joined by '!' (andi --join): windows with separators
            a, _ = synth.genome_set(n, length, 0.002, 0.03, seed=seed)
            seqs = [synth.join_contigs(bytes(s), int(rng.integers(2, 40)), seed=seed + k) for k, s in enumerate(a)]
        elif kind == "strands":  # every other genome as its reverse complement: the chains run on the other strand of RS
            a, _ = synth.genome_set(n, length, 0.002, 0.03, seed=seed)
            rc = bytes.maketrans(b"ACGT", b"TGCA")
            seqs = [bytes(s) if k % 2 == 0 else bytes(s)[::-1].translate(rc) for k, s in enumerate(a)]
        elif kind == "sparse":  # a few clean pairs among pairs 8-12 % apart: the far ones are the lane scan's (many of them)
            seqs, _ = synth.genome_set(n, length, 0.002, 0.06, seed=seed)
        else:
            a, _ = synth.genome_set(n - 2, length, 0.002, 0.03, seed=seed)
            b, _ = synth.realistic_set(2, length, 0.002, 0.03, seed=seed + 3)
            seqs = a + b
        seqs = [bytes(s) for s in seqs]
        model = int(rng.choice([0, 1, 1, 2, 3, 4]))
        lane, t0 = rows(seqs, model, {"ANDI_COOP": "0"})
        got, t1 = rows(seqs, model, {})
        forced, t2 = rows(seqs, model, {"ANDI_COOP": str(rng.choice([4, 5]))}) if kind in ("star", "tree", "joined", "strands") else (lane, t0)
        # ... and with every routed pair taken for one that suits k_pool_cold (coop_pool.h): that kernel where the call is large enough for
        # the host to look at the layout, whatever the pairs are like
        pooled, t3 = rows(seqs, model, {"ANDI_POOL_MATCH": "0", "ANDI_ROUTE_SMALL": "20"}) if model < 3 else (lane, t0)
        k = int(rng.integers(0, n))
        want = orc.scan_row(orc.OracleEsa(seqs[k]), seqs, k, model, threads=0)
        ok = bool((got == lane).all() and (forced == lane).all() and (pooled == lane).all() and (lane[k] == want).all())
        case += 1
        print("case %3d %-9s n=%2d len=%d model=%d  routed calls %d (k_pool_cold: %d, forced: %d), pairs handed back %d/%d; fix-ups lane/routed %d/%d  %s" % (
            case, kind, n, length, model, t1["routed_calls"], t1["pool_calls"], t3["pool_calls"], t1["coop_fallbacks"], t3["coop_fallbacks"], t0["fixups"], t1["fixups"], "ok" if ok else "DIFFERENT"), flush=True)
        if not ok:
            sys.exit(1)
    print("fuzz_large: %d cases, all equal (routed = lane scan = forced kernel = pooled kernel forced, sampled rows = oracle)" % case)


if __name__ == "__main__":
    main()
