#!/usr/bin/env python3
"""The one-call seam (andi_hip_dist_matrix) on random sets against the oracle: one context or several on device 0 (several:
the queries are packed once on the host and every context uploads the 4-bit pool), low_memory or not, every model,
host- and device-made suffix arrays.  scripts/fuzz_seam.py [seconds] [seed]   (checker code from oracle/ as the checker only)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANDI_HIP_LIB", os.path.join(ROOT, "andi_amd", "libandihip_test.so"))  # the build with the test hooks (andi_amd/csrc/knobs.h)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np

import andi_amd
from fuzz_parity import make_case
from oracle import orc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    t_end = time.time() + budget
    cases = 0
    while time.time() < t_end:
        kind, seqs = make_case(rng)
        if rng.random() < 0.3:  # odd lengths, a one-symbol stub
            seqs = [s[: len(s) - int(rng.integers(0, 2))] for s in seqs] + [seqs[0][:1] if rng.random() < 0.3 else seqs[0][:33]]
        model = int(rng.choice([0, 1, 2, 3, 4]))
        want = orc.dist_matrix(seqs, model=model, threads=0)
        for _ in range(2):
            ndev = int(rng.choice([1, 2, 3, 5]))
            kw = {"model": model, "host_threads": int(rng.choice([1, 3, 8])), "low_memory": bool(rng.random() < 0.3),
                  "sa_on_host": bool(rng.random() < 0.3)}
            if ndev > 1:
                kw["devices"] = [0] * ndev
            got = andi_amd.dist_matrix(seqs, **kw)
            ok = bool((got == want).all())
            cases += 1
            print("case %4d %-9s n=%d len=%-7d %s  %s" % (cases, kind, len(seqs), len(seqs[0]), kw, "ok" if ok else "DIFFERENT"), flush=True)
            if not ok:
                sys.exit(1)
    print("fuzz_seam: %d calls, all equal to the oracle" % cases)


if __name__ == "__main__":
    main()
