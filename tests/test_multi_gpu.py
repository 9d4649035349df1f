"""The N x N loop tiled over several devices BEHIND the C-ABI (andi_hip_dist_matrix with opts.num_gpus /
opts.devices; src/dist_hack.h:46-47 is the loop that is being tiled).  A box with one GPU can still run every
piece of it: several driver threads and contexts on device 0 (rows go to the host matrix directly), and the
RCCL gather forced for one device (communicator, grouped exchange, the copy of the gathered matrix)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _set():
    from andi_amd import synth
    base = synth.base_codes(60000, 77)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 90 + k)) for k, d in
            enumerate((0.0, 0.001, 0.004, 0.01, 0.02, 0.03, 0.05, 0.08, 0.1, 0.002, 0.015))]
    seqs.append(synth.unrelated(20000, 5))
    return seqs


def test_row_blocks_on_several_contexts_equal_one_call(orc):
    import andi_amd
    seqs = _set()
    want = orc.dist_matrix(seqs, threads=0)
    one = andi_amd.dist_matrix(seqs, host_threads=4)
    assert (one == want).all() and andi_amd.lib.last_gather() == "direct"
    for devices in ([0, 0], [0, 0, 0], [0] * 5, [0] * 12, [0] * 40):  # up to one row per context; more contexts than rows
        got = andi_amd.dist_matrix(seqs, host_threads=4, devices=devices)
        assert (got == want).all(), devices
        assert andi_amd.lib.last_gather() == "direct"
    got = andi_amd.dist_matrix(seqs, host_threads=2, devices=[0, 0, 0], low_memory=True, model=andi_amd.M_KIMURA)
    assert (got == orc.dist_matrix(seqs, model=orc.M_KIMURA, threads=0)).all()


_RCCL_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import andi_amd
from oracle import orc
from test_multi_gpu import _set
seqs = _set()
want = orc.dist_matrix(seqs, threads=0)
os.environ["ANDI_GATHER"] = "rccl"
andi_amd.lib.reload_knobs()
got = andi_amd.dist_matrix(seqs, host_threads=4, num_gpus=-1)
assert (got == want).all()
assert andi_amd.lib.last_gather() == "rccl", andi_amd.lib.last_gather()
# contexts that share a device: the rows stay in HBM as on the RCCL route, the communicators cannot be made (one device
# twice), and the route's fallback copies every block to the host matrix directly -- the error is kept in the gather string
got = andi_amd.dist_matrix(seqs, host_threads=4, devices=[0, 0, 0])
assert (got == want).all()
assert andi_amd.lib.last_gather().startswith("direct (rccl:"), andi_amd.lib.last_gather()
del os.environ["ANDI_GATHER"]
andi_amd.lib.reload_knobs()
if andi_amd.lib.device_count() > 1:  # several GPUs visible: the same call spans all of them
    got = andi_amd.dist_matrix(seqs, host_threads=4, num_gpus=-1)
    assert (got == want).all() and andi_amd.lib.last_gather() == "rccl", andi_amd.lib.last_gather()
print("rccl gather ok on", andi_amd.lib.device_count(), "device(s)")
"""


def test_rccl_gather_path_on_one_device():
    """ANDI_GATHER=rccl takes the multi-device route with the devices there are: rows stay in HBM, librccl is
    loaded, a communicator per device is created, the (here empty) grouped send/recv runs, the gathered matrix is
    copied once.  With several GPUs visible the same call spans all of them.  In a process of its own: the
    library loads ROCm's librccl, which must not meet the copy of the ROCm runtime that PyTorch carries."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl gather ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_errors_come_back_through_errbuf():
    import andi_amd
    seqs = _set()[:3]
    with pytest.raises(andi_amd.AndiHipError, match="out of range"):
        andi_amd.dist_matrix(seqs, devices=[0, 99])
    with pytest.raises(andi_amd.AndiHipError):
        andi_amd.dist_matrix(seqs[:2] + [b""], devices=[0, 0])
    # a byte outside the alphabet: refused whether the device packs the queries (one device) or the host did (several)
    bad = seqs[:2] + [seqs[2][:5000] + b"N" + seqs[2][5000:]]
    for kw in ({}, {"devices": [0, 0]}):
        with pytest.raises(andi_amd.AndiHipError, match="outside"):
            andi_amd.dist_matrix(bad, **kw)


def test_queries_packed_on_the_host_equal_the_device_packed_ones(orc):
    """Several devices: the queries are packed once on the host and every device uploads the 4-bit pool (and unpacks the
    bytes); one device: the bytes, packed on the device.  ANDI_QUERIES_PACKED / ANDI_QUERIES_BYTES force either; ragged
    lengths (odd ones), joined contigs, every model's counts."""
    import andi_amd
    from andi_amd import synth
    from conftest import knobs
    seqs = _set()
    seqs[3] = seqs[3][:-1]  # an odd length
    seqs[5] = synth.join_contigs(seqs[5], 7, seed=3)
    seqs.append(seqs[1][:777])
    for model in (andi_amd.M_JC, andi_amd.M_LOGDET):
        want = orc.dist_matrix(seqs, model=model, threads=0)
        with knobs(QUERIES_PACKED=1):
            assert (andi_amd.dist_matrix(seqs, model=model, host_threads=4) == want).all()
        with knobs(QUERIES_BYTES=1):
            assert (andi_amd.dist_matrix(seqs, model=model, host_threads=4, devices=[0, 0]) == want).all()
        assert (andi_amd.dist_matrix(seqs, model=model, host_threads=4, devices=[0, 0, 0]) == want).all()


_TRIM_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import andi_amd
from andi_amd import lib
from oracle import orc
from test_multi_gpu import _set
seqs = _set()[:5]
want = orc.dist_matrix(seqs, threads=0)
assert (andi_amd.dist_matrix(seqs, host_threads=2) == want).all()
kept = lib.trim()
assert kept > 0, "no chunk was kept after the call"
assert lib.trim() == 0
assert (andi_amd.dist_matrix(seqs, host_threads=2) == want).all()
os.environ["ANDI_ARENA_KEEP"] = "0"
lib.reload_knobs()
lib.trim()
assert (andi_amd.dist_matrix(seqs, host_threads=2) == want).all()
assert lib.trim() == 0  # (the call's last context released them)
print("trim ok", kept)
"""


def test_chunks_outlive_the_contexts_and_trim_gives_them_back():
    """The device-memory chunks stay with the process between calls of the seam (a fresh hipMalloc of a job's 6.5 GB costs
    0.3 s per call on some hosts); andi_hip_trim() returns them to the driver, after which a call still works; with
    ANDI_ARENA_KEEP=0 the last context releases them as before.  (In a process of its own: no other context about.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TRIM_SCRIPT], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "trim ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_large_sets_are_uploaded_by_threads(orc, knob):
    """Sets of 2 GiB and more are staged by four host threads through pinned buffers (api.hip: andi_hip_queries_stage).  The test hook
    ANDI_UPLOAD_MIN_MB lowers the limit so that the path runs here: sequences longer than a chunk (8 MiB), shorter ones, an odd count;
    the one-call seam's matrix equals the oracle's rows and the matrix staged the ordinary way."""
    import andi_amd
    from andi_amd import synth
    base = synth.base_codes(9_000_000, 77)
    seqs = [synth.to_bytes(synth.mutate_codes(base, 0.004 * (k + 1), 80 + k))[: 9_000_000 - 1_234_567 * k] for k in range(5)]
    seqs += [synth.to_bytes(synth.mutate_codes(base[:70_001], 0.01, 90)), synth.join_contigs(synth.to_bytes(base[:300_000]), 4)]
    plain = andi_amd.dist_matrix(seqs, model=andi_amd.M_JC)
    knob("ANDI_UPLOAD_MIN_MB", "1")
    threaded = andi_amd.dist_matrix(seqs, model=andi_amd.M_JC)
    assert (threaded == plain).all()
    for i in (1, 5, 6):
        O = orc.OracleEsa(seqs[i])
        want = orc.scan_row(O, seqs, i, orc.M_JC, threads=os.cpu_count() or 1)
        O.close()
        assert (threaded[i] == want).all(), i
