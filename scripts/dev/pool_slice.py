#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANDI_HIP_LIB", os.path.join(ROOT, "andi_amd", "libandihip_test.so"))  # the build with the test hooks (andi_amd/csrc/knobs.h)
import numpy as np
import andi_amd
from andi_amd import lib, synth
seqs, _ = synth.genome_set(12, 4_900_000, 0.004, 0.03, seed=5)
S, Qs = bytes(seqs[6]), bytes(seqs[0])
os.environ["ANDI_COOP"] = "4"
ctx = andi_amd.Context(0)
E = andi_amd.Esa(ctx, S, sa="device")
def scan(pool, q, seg):
    os.environ["ANDI_POOL"] = pool; os.environ["ANDI_COOP_SEG"] = str(seg); lib.reload_knobs()
    Q = andi_amd.Queries(ctx, [q])
    got = andi_amd.scan_rows(ctx, [E], [-1], Q, model=1)
    Q.close()
    return got[0, 0].astype(np.int64)
a0, a1 = 122 * 32768, 125 * 32768
for seg in (32768, 16384, 8192, 4096, 2048):
    q = Qs[a0:a1]
    ref = scan("0", q, seg)
    d = [int((scan("1", q, seg) - ref)[:16].sum()) for _ in range(3)]
    print("seg", seg, "diffs", d, flush=True)
# sub-slices at seg 32768: start offsets in steps of 2048 within the failing segment (alignment of 32 kept), always one segment of lead-in
seg = 32768
for start in range(a0, a0 + 32768 + 1, 4096):
    for ln in (65536,):
        q = Qs[start:start + ln]
        ref = scan("0", q, seg)
        d = [int((scan("1", q, seg) - ref)[:16].sum()) for _ in range(2)]
        print("start", start - a0, "len", ln, "diffs", d, flush=True)
