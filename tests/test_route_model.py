"""The statistic pass A's routing rests on (andi_amd/csrc/scan_lane.hip: k_pair_estimate), restated on the CPU on the
oracle's matcher: runs of five short samples tell a pair with unrelated stretches from a clean one whatever the pair's
divergence.  TEST INFRASTRUCTURE: the oracle is the checker of the model's inputs only; the device code is checked
against the lane scan and the oracle in tests/test_coop_gpu.py (which pairs go where never changes a result)."""
import numpy as np
import pytest

from andi_amd import synth
from oracle import orc


def _suspected(subject: bytes, query: bytes, nsamples=512):
    """k_pair_estimate's verdict for one ordered pair: (shorts, runs, suspected)"""
    E = orc.OracleEsa(subject)
    T = E.threshold + 3
    qlen = len(query)
    shorts = runs = 0
    for i in range(nsamples):
        p = (2 * i + 1) * qlen // (2 * nsamples)
        all_short = True
        for j in range(5):
            pj = p + 128 * j
            if not (pj + T < qlen and E.get_match(query[pj:pj + T + 1])[0] < T):
                all_short = False
                break
            if j == 0:
                shorts += 1
        runs += all_short
    E.close()
    g = min((shorts - runs) / nsamples, 0.53)
    f0 = g
    for _ in range(5):
        f4 = f0 ** 4
        f0 -= (f0 - f0 * f4 - g) / (1 - 5 * f4)
    f5 = f0 ** 5
    expect = nsamples * f5
    return shorts, runs, runs > expect + 3 * np.sqrt(expect * (1 - f5)) + 3


def _mean_match(subject: bytes, query: bytes):
    """the mean of the 64 sampled longest matches (k_pair_estimate's first samples)"""
    E = orc.OracleEsa(subject)
    qlen = len(query)
    total = 0
    for i in range(64):
        p = (2 * i + 1) * qlen // 128
        total += E.get_match(query[p:p + 1100])[0]
    E.close()
    return total / 64


def _small_call_verdict(shorts, runs, suspected, mean, nsamples=512):
    """The same pair in a SMALL call (k_pair_estimate with route_all_few): where the mean sampled match is below 19, or the
    rate of non-run short samples is 0.45 and more -- pairs some 4 % and more apart, which the first test cannot judge --
    the pair is suspected only if its runs exceed what the rate of ALL short samples explains (an upper bound of f0);
    returns (suspected, guessed)."""
    if not (mean < 19 or (shorts - runs) >= 0.45 * nsamples):
        return suspected, False
    fa = shorts / nsamples
    most = nsamples * fa ** 5
    bad = runs > most + 3 * np.sqrt(most * (1 - fa ** 5)) + 3
    return bad, not bad


@pytest.mark.parametrize("d", [0.005, 0.02, 0.04, 0.05])
def test_clean_pairs_are_not_suspected(d):
    """(up to some 5.5 % apart; beyond, f0 - f0^5 is flat, f0 comes out low and the pair is taken for suspicious: such pairs
    are the lane scan's anyway where they are many)"""
    a, b = synth.pair(1_000_000, d, seed=int(d * 1e4))
    shorts, runs, bad = _suspected(a, b)
    assert not bad, (d, shorts, runs)


@pytest.mark.parametrize("d", [0.004, 0.03, 0.05])
def test_pairs_with_unrelated_stretches_are(d):
    seqs, _ = synth.realistic_set(2, 1_000_000, d / 2, d / 2 + 1e-9, seed=7 + int(d * 1e3))
    shorts, runs, bad = _suspected(seqs[0], seqs[1])
    assert bad, (d, shorts, runs)


@pytest.mark.parametrize("d", [0.06, 0.08, 0.1])
def test_small_calls_take_clean_pairs_far_apart_for_the_wavefront_kernel(d):
    """Beyond 5.5 % the first test suspects every pair; a small call asks the second question, and clean pairs pass it
    (marked as a guess: dropped again where the call has pairs with stretches clearly seen)."""
    a, b = synth.pair(1_000_000, d, seed=int(d * 1e4))
    shorts, runs, bad = _suspected(a, b)
    small_bad, guessed = _small_call_verdict(shorts, runs, bad, _mean_match(a, b))
    assert not small_bad and guessed, (d, shorts, runs, bad)


def test_small_calls_still_see_stretches_in_pairs_moderately_far_apart():
    seqs, _ = synth.realistic_set(2, 1_000_000, 0.03, 0.03 + 1e-9, seed=19)  # 6 % apart, 10 % unrelated sequence
    shorts, runs, bad = _suspected(seqs[0], seqs[1])
    small_bad, _ = _small_call_verdict(shorts, runs, bad, _mean_match(seqs[0], seqs[1]))
    assert bad and small_bad, (shorts, runs)


def _far_clean(shorts, runs, suspected, nsamples=512):
    """Any call, a suspected pair (round 5): it is MERELY FAR APART if its runs of five are at most twice as many as the rate of
    all its short samples gives by itself (k_pair_estimate: far_clean).  Such a pair is a candidate of the wavefront kernel
    again (soft, and a guess), and it does not count towards a call of structured genomes (k_pair_route's longer segments)."""
    fa = shorts / nsamples
    return suspected and runs <= 2 * nsamples * fa ** 5 + 10


@pytest.mark.parametrize("d", [0.06, 0.08, 0.1])
def test_suspected_clean_pairs_are_told_from_structured_ones(d):
    a, b = synth.pair(1_000_000, d, seed=int(d * 1e4))
    shorts, runs, bad = _suspected(a, b)
    assert not bad or _far_clean(shorts, runs, bad), (d, shorts, runs, bad)


@pytest.mark.parametrize("d", [0.004, 0.03])
def test_structured_pairs_are_not_taken_for_far_ones(d):
    """(up to some 4 % apart; a structured pair 5 % and more apart passes for a far one -- in a call of structured genomes it
    gets the others' segment length all the same, k_pair_route)"""
    seqs, _ = synth.realistic_set(2, 1_000_000, d / 2, d / 2 + 1e-9, seed=7 + int(d * 1e3))
    shorts, runs, bad = _suspected(seqs[0], seqs[1])
    assert bad and not _far_clean(shorts, runs, bad), (d, shorts, runs)
