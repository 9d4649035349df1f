"""Time of the index builds alone: 29 subjects of the bench set's size, built as one batch, five times
(development aid; ANDI_HIP_LIB selects the build of the library)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import andi_amd
from andi_amd import lib, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 29
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4900000
ctx = andi_amd.Context()
seqs, _ = synth.genome_set(G, L, 0.0004, 0.03)
esas = [andi_amd.Esa(ctx, s, sa="device", build=None) for s in seqs]
lib.build_indexes(ctx, esas)
ctx.timings_reset()
for _ in range(5):
    lib.build_indexes(ctx, esas)
t = ctx.timings()
print("%s: index builds of %d subjects x %d nt: %.3f ms per batch" % (os.environ.get("ANDI_HIP_LIB", "libandihip.so").split("/")[-1], G, L, t["build_ms"] / 5))
