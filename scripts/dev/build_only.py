"""development aid: the index builds of the bench set alone (no scan): ms per batch of 29 subjects"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import andi_amd
from andi_amd import lib, synth
seqs, _ = synth.genome_set(29, 4_900_000, 0.0004, 0.03, seed=1729)
ctx = andi_amd.Context(0)
esas = [andi_amd.Esa(ctx, s, 0.025, build=False, sa="device") for s in seqs]
for rep in range(3):
    ctx.timings_reset()
    for _ in range(5):
        lib.build_indexes(ctx, esas)
    ctx.sync()
    print("build ms per batch: %.3f" % (ctx.timings()["build_ms"] / 5))
