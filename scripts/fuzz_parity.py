#!/usr/bin/env python3
"""Random sets, random segment lengths, models and scan variants through the C-ABI against the oracle, for as long as
asked: scripts/fuzz_parity.py [seconds] [seed].  Prints every case; stops at the first difference (exit code 1).
(Checker code from oracle/ is used as the checker only.)"""
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


def make_case(rng):
    kind = rng.choice(["star", "realistic", "tree", "joined", "repeat", "mixed"])
    n = int(rng.integers(2, 7))
    length = int(rng.choice([3000, 20000, 60000, 150000, 400000]))
    seed = int(rng.integers(1, 1 << 30))
    dlo = float(rng.choice([1e-5, 1e-3, 5e-3]))
    dhi = dlo + float(rng.choice([1e-4, 5e-3, 3e-2, 8e-2]))
    if kind == "star":
        seqs, _ = synth.genome_set(n, length, dlo, dhi, seed=seed)
    elif kind == "tree":
        seqs, _ = synth.tree_set(max(n, 3), length, seed=seed)
    elif kind == "realistic":
        seqs, _ = synth.realistic_set(n, max(length, 20000), dlo, dhi, seed=seed, novel_fraction=float(rng.choice([0.0, 0.05, 0.2])))
    elif kind == "joined":
        seqs, _ = synth.realistic_set(n, max(length, 20000), dlo, dhi, seed=seed, contigs=int(rng.integers(2, 12)))
    elif kind == "repeat":
        unit = synth.to_bytes(synth.base_codes(int(rng.integers(200, 4000)), seed))
        base = unit * int(rng.integers(3, 12)) + synth.to_bytes(synth.base_codes(length // 2, seed + 1)) + unit * 2
        codes = np.frombuffer(base.translate(bytes.maketrans(b"ACGT", bytes(range(4)))), np.uint8)
        seqs = [base] + [synth.to_bytes(synth.mutate_codes(codes, float(rng.uniform(dlo, dhi)), seed + 2 + k)) for k in range(n - 1)]
    else:
        a, _ = synth.genome_set(2, length, dlo, dhi, seed=seed)
        seqs = a + [synth.unrelated(max(length // 3, 500), seed + 5), a[0][: max(length // 7, 50)], a[1][length // 3:]]
    return kind, [bytes(s) for s in seqs]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    t_end = time.time() + budget
    cases = 0
    ctx = andi_amd.Context(0)
    while time.time() < t_end:
        kind, seqs = make_case(rng)
        model = int(rng.choice([0, 1, 1, 2, 3, 4]))
        want = orc.dist_matrix(seqs, model=model, threads=0)
        for _ in range(3):
            segment = int(rng.choice([0, 0, 0, 0, 64, 100, 300, 700, 2048, 5000, 40000]))  # 0: the engine's choice -- pass A routed per pair from 2^18 symbols
            env = {}
            if rng.random() < 0.3:
                env["ANDI_COOP"] = str(rng.choice([2, 4, 5, 8]))
            else:
                if rng.random() < 0.6:  # no call is tiny: calls of this size are routed per pair (else: the wavefront kernel for every pair)
                    env["ANDI_ROUTE_TINY"] = "1"
                    if rng.random() < 0.4:  # routed calls: wavefronts hand their pairs back early (the second lane layout runs)
                        env["ANDI_COOP_GIVEUP"] = str(rng.choice([4, 48]))
            if rng.random() < 0.15:
                env["ANDI_UNIFORM_SEGMENTS"] = "1"
            if rng.random() < 0.15:
                env["ANDI_FORCE_ADAPTIVE"] = "1"
            sa = str(rng.choice(["device", "host"]))
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            lib.reload_knobs()
            try:
                Q = andi_amd.Queries(ctx, seqs)
                esas = [andi_amd.Esa(ctx, s, sa=("device" if sa == "device" else None)) for s in seqs]
                ctx.timings_reset()
                got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q, model=model, segment=segment)
                t = ctx.timings()
                for e in esas:
                    e.close()
                Q.close()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                lib.reload_knobs()
            ok = bool((got == want).all())
            cases += 1
            print("case %4d %-9s n=%d len=%-7d model=%d segment=%-5d sa=%-6s %-40s fixups=%-5d %s %s" % (
                cases, kind, len(seqs), len(seqs[0]), model, segment, sa, " ".join("%s=%s" % kv for kv in env.items()), t["fixups"],
                "routed (%s layout, %d%% by wavefronts, %d handed back)" % ("per-pair" if t["adaptive_calls"] else "uniform", 100 * t["coop_query_nt"] // max(t["coop_query_nt"] + t["lane_query_nt"], 1), t["coop_fallbacks"]) if t["routed_calls"] else "by wavefronts" if t["coop_calls"] else "",
                "ok" if ok else "DIFFERENT"), flush=True)
            if not ok:
                bad = np.argwhere((got != want).any(axis=2))
                print("pairs that differ:", bad[:10].tolist())
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "fuzz_fail_seqs.npy"), np.array(seqs, dtype=object), allow_pickle=True)
                sys.exit(1)
    print("fuzz: %d cases, all equal to the oracle" % cases)


if __name__ == "__main__":
    main()
