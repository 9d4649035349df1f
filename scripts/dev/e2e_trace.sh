#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
import sys, time, os
sys.path.insert(0, ".")
import andi_amd
from andi_amd import synth
seqs = synth.genome_set_fast(29, 4_900_000, 0.0004, 0.03, seed=1729)[0] if hasattr(synth, "genome_set_fast") else synth.genome_set(29, 4_900_000, 0.0004, 0.03, seed=1729)[0]
andi_amd.dist_matrix(seqs[:3], host_threads=2)
for k in range(3):
    t = time.time(); andi_amd.dist_matrix(seqs, host_threads=0); print("call %d: %.4f s" % (k, time.time() - t), flush=True)
os.environ["ANDI_E2E_TRACE"] = "1"
andi_amd.lib.reload_knobs() if hasattr(andi_amd.lib, "reload_knobs") else None
t = time.time(); andi_amd.dist_matrix(seqs, host_threads=0); print("traced: %.4f s" % (time.time() - t), flush=True)
PY
