"""The seam's time in a process that has just run the OpenMP port (as bench.py does): ANDI_E2E_TRACE split, before and after."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["ANDI_E2E_TRACE"] = "1"
import andi_amd
from andi_amd import synth
from oracle import orc
seqs, _ = synth.genome_set(29, 4_900_000, 0.0004, 0.03, seed=1729)
def run(tag):
    for rep in range(2):
        t0 = time.perf_counter()
        andi_amd.dist_matrix(seqs)
        print(tag, "rep", rep, "wall %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
run("before the OpenMP port:")
t0 = time.perf_counter()
orc.dist_matrix(seqs[:8], threads=8)
print("OpenMP port on 8 genomes: %.1f s" % (time.perf_counter() - t0), flush=True)
run("after the OpenMP port:")
print("load average", os.getloadavg(), "cpus", os.cpu_count(), flush=True)
