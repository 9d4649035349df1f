#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
ANDI_E2E_TRACE=1 python3 - <<'PY'
import sys, time, os
sys.path.insert(0, ".")
import andi_amd
from andi_amd import synth
seqs = synth.genome_set_fast(29, 4_900_000, 0.0004, 0.03, seed=1729)[0]
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
t = time.time(); andi_amd.dist_matrix(seqs, host_threads=0); print("cold: %.4f s" % (time.time() - t), flush=True)
t = time.time(); andi_amd.dist_matrix(seqs, host_threads=0); print("warm: %.4f s" % (time.time() - t), flush=True)
PY
