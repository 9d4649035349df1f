"""Which ingredient of the realistic sets costs pass B what: the scan's timings for variants of synth.realistic_set."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import andi_amd
from andi_amd import synth

G, L = 16, 2_000_000
ctx = andi_amd.Context(0)


def run(name, seqs):
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
    sel = list(range(len(seqs)))
    andi_amd.scan_rows(ctx, esas, sel, Q)
    ctx.timings_reset()
    andi_amd.scan_rows(ctx, esas, sel, Q)
    t = ctx.timings()
    print("%-34s pass A %7.2f ms   passes B+C %7.2f ms   fixups %6d" % (name, t["scan_ms"], t["stitch_ms"], t["fixups"]), flush=True)
    for e in esas:
        e.close()
    Q.close()


base_plain = synth.base_codes(L, 3)
base_rep = synth.realistic_base(L, 3)
drng = np.random.default_rng(9)
ds = drng.uniform(0.0004, 0.03, G)
mk = lambda base, **kw: [synth.to_bytes(synth.evolve_codes(base, float(ds[k]), 50 + k, **kw)) for k in range(G)]
only = sys.argv[1] if len(sys.argv) > 1 else None
if only == "islands":
    run("+ 10 % unrelated islands", mk(base_plain, indel_rate=0, inversions=0, novel_fraction=0.1))
    sys.exit(0)
run("substitutions only", [synth.to_bytes(synth.mutate_codes(base_plain, float(ds[k]), 50 + k)) for k in range(G)])
run("+ repeats in the base", mk(base_rep, indel_rate=0, inversions=0, novel_fraction=0))
run("+ indels", mk(base_plain, indel_rate=0.1, inversions=0, novel_fraction=0))
run("+ inversions", mk(base_plain, indel_rate=0, inversions=2, novel_fraction=0))
run("+ 10 % unrelated islands", mk(base_plain, indel_rate=0, inversions=0, novel_fraction=0.1))
run("everything", mk(base_rep, indel_rate=0.1, inversions=2, novel_fraction=0.1))
