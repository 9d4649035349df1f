"""The decomposition the cooperative pass A rests on (tests/coop_model.py) against the plain loop of dist_anchor,
on the CPU: every segment's exit state, counts, anchor count, first anchor and mark must be identical."""
import numpy as np
import pytest

from andi_amd import synth
from tests import coop_model as cm


def _revcomp(b: bytes) -> bytes:
    return bytes(b[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))


def _check(subject, query, seg, W, label):
    P = cm.Pair(subject, query)
    nseg = (P.qlen + seg - 1) // seg
    tot = cm.Stats()
    for k in range(nseg):
        st0 = cm.State() if k == 0 else cm.cold_state(k * seg, P.n)
        end = min((k + 1) * seg, P.qlen)
        want = cm.plain_segment(P, st0, end)
        got, stats = cm.coop_segment(P, st0, end, W, tot)
        assert got.key() == want.key(), "%s: segment %d of %d (seg %d, W %d)" % (label, k, nseg, seg, W)
    return tot


@pytest.mark.parametrize("d", [0.0, 0.001, 0.01, 0.03, 0.06, 0.1, 0.3])
def test_star_pairs(d):
    base = synth.base_codes(120000, 5)
    s = synth.to_bytes(synth.mutate_codes(base, d / 2, 6))
    q = synth.to_bytes(synth.mutate_codes(base, d / 2, 7))
    for seg, W in ((8192, 2048), (32768, 8192), (3000, 500)):
        st = _check(s, q, seg, W, "d=%g" % d)
    if 0.005 < d < 0.2:
        assert st.windows > 0 and st.heads_on_path > 0


def test_realistic_structure():
    seqs, _ = synth.realistic_set(3, 150000, 0.005, 0.04, seed=11)
    for i in range(3):
        for j in range(3):
            if i != j:
                _check(seqs[i], seqs[j], 16384, 4096, "realistic %d/%d" % (i, j))


def test_reverse_strand_and_border():
    base = synth.base_codes(60000, 9)
    s = synth.to_bytes(base)
    q = _revcomp(synth.to_bytes(synth.mutate_codes(base, 0.02, 10)))
    _check(s, q, 8192, 2048, "reverse strand")
    # a query that runs across the '#' between the strands of RS: revcomp(S)'s tail followed by S's head
    rs_like = _revcomp(s)[-20000:] + s[:20000]
    _check(s, rs_like, 8192, 4096, "across the border")


def test_identical_unrelated_short_joined():
    base = synth.base_codes(50000, 3)
    s = synth.to_bytes(base)
    _check(s, s, 8192, 2048, "identical")
    _check(s, synth.unrelated(40000, 77), 8192, 2048, "unrelated")
    _check(s[:900], synth.to_bytes(synth.mutate_codes(base[:900], 0.05, 4)), 256, 128, "short")
    j = synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.02, 5)), 7)
    _check(synth.join_contigs(s, 5), j, 8192, 2048, "joined contigs")


def test_window_edges_tiny_windows():
    base = synth.base_codes(30000, 21)
    s = synth.to_bytes(synth.mutate_codes(base, 0.02, 22))
    q = synth.to_bytes(synth.mutate_codes(base, 0.02, 23))
    for W in (64, 100, 333):
        _check(s, q, 4096, W, "W=%d" % W)
