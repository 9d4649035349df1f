"""Config-3-like end-to-end check: andi_hip_dist_matrix on G genomes against the oracle's OpenMP matrix.
Usage: e2e_check.py [genomes length dlo dhi]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import andi_amd
from andi_amd import synth
from oracle import orc
a = sys.argv[1:]
G, L = int(a[0]) if a else 109, int(a[1]) if len(a) > 1 else 2_000_000
dlo, dhi = float(a[2]) if len(a) > 2 else 0.0001, float(a[3]) if len(a) > 3 else 0.005
seqs, _ = synth.genome_set(G, L, dlo, dhi, seed=4242)
andi_amd.dist_matrix(seqs[:2], host_threads=2)
t = time.time()
M = andi_amd.dist_matrix(seqs, model=andi_amd.M_KIMURA)
dt = time.time() - t
print("dist_matrix: %d x %d nt, Kimura: %.2f s end to end, %.0f pairs/s" % (G, L, dt, (G * G - G) / dt), flush=True)
t = time.time()
want = orc.dist_matrix(seqs, model=orc.M_KIMURA, threads=64)
print("oracle: %.1f s; equal: %s" % (time.time() - t, bool((M == want).all())), flush=True)
assert (M == want).all()
