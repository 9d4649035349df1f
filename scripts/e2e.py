"""End-to-end timing of the one-call seam (andi_hip_dist_matrix): host suffix arrays in a thread pool,
uploads, device index builds, scans, row copies.  Usage: e2e.py [genomes length dlo dhi threads]"""
import sys, time
sys.path.insert(0, ".")
import andi_amd
from andi_amd import synth
a = sys.argv[1:]
G, L, dlo, dhi = int(a[0]) if a else 29, int(a[1]) if len(a) > 1 else 4_900_000, float(a[2]) if len(a) > 2 else 0.0004, float(a[3]) if len(a) > 3 else 0.03
thr = int(a[4]) if len(a) > 4 else 0
seqs, _ = synth.genome_set(G, L, dlo, dhi, seed=1729)
andi_amd.dist_matrix(seqs[:2], host_threads=2)  # warm up the runtime
t = time.time()
M = andi_amd.dist_matrix(seqs, host_threads=thr)
dt = time.time() - t
print("dist_matrix: %d genomes x %d nt: %.2f s end to end -> %.0f pairs/s (host threads %s)" % (G, L, dt, (G * G - G) / dt, thr or "all"))
