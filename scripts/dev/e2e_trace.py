import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["ANDI_E2E_TRACE"] = "1"
import andi_amd
from andi_amd import synth
seqs, _ = synth.genome_set(29, 4_900_000, 0.0004, 0.03, seed=1729)
for rep in range(3):
    t0 = time.perf_counter()
    M = andi_amd.dist_matrix(seqs, host_threads=29)
    print("rep", rep, "wall %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
