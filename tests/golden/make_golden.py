#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz and reference_recorded.json.

Inputs come from the REFERENCE's own generator, test/test_fasta.cxx, which is
the one piece of the upstream tree that compiles from its own source alone; it
is built by `make -C oracle ref` into oracle/_ref/test_fasta (needs
/root/reference, so this script only runs in the build container).  Its output
depends on the C++ standard library, hence the sequences are committed here,
2-bit packed.

Expected values are of two kinds and kept apart:
  * reference_recorded.json — numbers the UNMODIFIED reference produced on
    exactly these inputs in this image, as recorded in SURVEY.md §6.2 and
    BASELINE.md §2 (instrumented dist_anchor counters; 4-decimal PHYLIP
    output).  They pin the oracle.
  * the `counts_*` arrays inside the .npz — the oracle's own 17 x u32 output
    for the same inputs, regression vectors for the oracle and the golden
    answers the GPU path is compared with on the GPU box.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

GEN = os.path.join(ROOT, "oracle", "_ref", "test_fasta")
CODE = np.zeros(256, np.uint8)
for k, c in enumerate(b"ACGT"):
    CODE[c] = k


def test_fasta(args):
    out = subprocess.check_output([GEN] + args).decode()
    seqs = []
    for rec in out.split(">")[1:]:
        seqs.append("".join(rec.split("\n")[1:]).encode())
    return seqs


def pack(seq: bytes):
    c = CODE[np.frombuffer(seq, np.uint8)]
    pad = (-len(c)) % 4
    c = np.concatenate([c, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    recorded = {
        "source": "SURVEY.md §6.2 and BASELINE.md §2 (unmodified reference, this image)",
        "s42": {
            # test_fasta -s 42 -l 1000000 -d D ; subject S0, query S1, threshold 14
            "0.1": {"iterations": 67241, "lucky_tries": 29449, "lucky_hits": 7347,
                    "anchor_pairs": 16567, "gap_chars": 568112, "cache_hit_probes": 59893},
            "0.01": {"iterations": 9890, "esa_probes": 1528, "lucky_tries": 9633, "lucky_hits": 8362,
                     "anchor_pairs": 8586, "gap_chars": 23435},
            "0.001": {"iterations": 1001, "esa_probes": 19, "lucky_tries": 1001, "lucky_hits": 982,
                      "anchor_pairs": 982, "gap_chars": 1187},
            "threshold": 14,
        },
        "s1729": {
            # test_fasta -s 1729 -l 1000000 -d 0.1 -d 0.1 ; pairs (0,1) (0,2) (1,2)
            "JC": ["0.0982", "0.0984", "0.1955"],
            "RAW": ["0.0921", "0.0922", "0.1721"],
            "KIMURA": ["0.0982", "0.0984", "0.1955"],
            "JC_vv": {"01": "0.0982", "10": "0.0983", "12": "0.1957", "21": "0.1953"},
        },
    }
    with open(os.path.join(HERE, "reference_recorded.json"), "w") as f:
        json.dump(recorded, f, indent=1)

    # -s 42 ladder
    arrays = {}
    for d in ("0.1", "0.01", "0.001"):
        s0, s1 = test_fasta(["-s", "42", "-l", "1000000", "-d", d])
        arrays["s0"] = pack(s0)
        arrays["s1_" + d] = pack(s1)
        E = orc.OracleEsa(s0)
        counts, st = E.dist_anchor(s1, stats=True)
        arrays["counts_" + d] = counts
        arrays["stats_" + d] = np.array([st[k] for k in sorted(st)], np.uint64)
        E.close()
    arrays["length"] = np.array([1000000])
    np.savez(os.path.join(HERE, "testfasta_s42.npz"), **arrays)

    # -s 1729 triple
    seqs = test_fasta(["-s", "1729", "-l", "1000000", "-d", "0.1", "-d", "0.1"])
    arrays = {"length": np.array([1000000])}
    for k, s in enumerate(seqs):
        arrays["s%d" % k] = pack(s)
    arrays["counts_jc"] = orc.dist_matrix(seqs, model=orc.M_JC)
    np.savez(os.path.join(HERE, "testfasta_s1729.npz"), **arrays)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
