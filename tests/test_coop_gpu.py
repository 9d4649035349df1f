"""Pass A with one wavefront per chain (andi_amd/csrc/scan_coop.hip) against the oracle and against the lane scan
(ANDI_COOP=0), through the C-ABI: 17 x u32 per ordered pair, bit-exact, for every window length."""
import os

import numpy as np
import pytest

import andi_amd
from andi_amd import synth
from conftest import knobs
from oracle import orc

pytestmark = pytest.mark.gpu


def _matrix(seqs, model, coop, segment=0):
    with knobs(COOP=coop):
        return andi_amd.dist_matrix(seqs, model=model, segment=segment)


def _revcomp(b: bytes) -> bytes:
    return bytes(b[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))


@pytest.mark.parametrize("coop", [2, 4, 5, 8])
def test_divergence_ladder(coop):
    base = synth.base_codes(300000, 5)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 10 + k)) for k, d in enumerate((0.0, 0.0005, 0.005, 0.02, 0.05, 0.15))]
    want = orc.dist_matrix(seqs, model=orc.M_JC, threads=4)
    for segment in (0, 4096, 1000):
        got = _matrix(seqs, andi_amd.M_JC, coop, segment)
        assert (got == want).all(), "window of %d chunks, segment %d" % (coop, segment)


def test_structured_genomes_equal_lane_scan_and_oracle():
    seqs, _ = synth.realistic_set(5, 400000, 0.002, 0.04, seed=23)
    want = orc.dist_matrix(seqs, model=orc.M_KIMURA, threads=5)
    lane = _matrix(seqs, andi_amd.M_KIMURA, 0)
    assert (lane == want).all()
    for segment in (0, 8192, 700):
        assert (_matrix(seqs, andi_amd.M_KIMURA, 4, segment) == want).all(), "segment %d" % segment
    assert (_matrix(seqs, andi_amd.M_KIMURA, 5, 0) == want).all() and (_matrix(seqs, andi_amd.M_KIMURA, 5, 12000) == want).all()


def test_strands_contigs_edges():
    base = synth.base_codes(120000, 9)
    s = synth.to_bytes(base)
    seqs = [s, _revcomp(synth.to_bytes(synth.mutate_codes(base, 0.02, 3))),  # reverse strand
            _revcomp(s)[-40000:] + s[:40000],  # across the '#' of RS
            synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.01, 4)), 9),  # '!' separators
            s, synth.unrelated(90000, 77), s[:700]]  # identical, unrelated, short
    want = orc.dist_matrix(seqs, model=orc.M_RAW, threads=4)
    for coop in (2, 4, 5):
        for segment in (0, 2048, 300):
            got = _matrix(seqs, andi_amd.M_RAW, coop, segment)
            assert (got == want).all(), "window of %d chunks, segment %d: pairs %s" % (
                coop, segment, np.argwhere((got != want).any(axis=2))[:8].tolist())


@pytest.mark.parametrize("model", [3, 4])
def test_logdet_and_ani_count_every_anchor(model):
    """LogDet and ANI count the nucleotides of every anchor (src/model.c:256-278): the wavefront kernel's EXACT form on a
    divergence ladder, structured genomes (walks with anchors off the diagonal, breaks, generic steps), both strands and
    joined contigs, at the default and at tiny segments -- 16 counts per pair against the oracle."""
    base = synth.base_codes(200000, 15)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 30 + k)) for k, d in enumerate((0.0, 0.002, 0.02, 0.06))]
    seqs.append(_revcomp(seqs[2]))
    seqs.append(synth.join_contigs(seqs[1], 6))
    want = orc.dist_matrix(seqs, model=model, threads=4)
    for coop in (2, 5):
        for segment in (0, 3000):
            got = _matrix(seqs, model, coop, segment)
            assert (got == want).all(), (coop, segment, np.argwhere((got != want).any(axis=2))[:6].tolist())
    real, _ = synth.realistic_set(4, 300000, 0.002, 0.05, seed=41)
    want = orc.dist_matrix(real, model=model, threads=4)
    assert (_matrix(real, model, 4, 0) == want).all()
    assert (_matrix(real, model, 4, 5000) == want).all()


def test_headline_pair_full_length():
    """one pair at BASELINE's genome length, both directions, against the lane scan (itself pinned to the oracle)"""
    seqs, _ = synth.genome_set(3, 4_900_000, 0.0004, 0.03, seed=1729)
    lane = _matrix(seqs, andi_amd.M_JC, 0)
    assert (_matrix(seqs, andi_amd.M_JC, 4) == lane).all()
    assert (_matrix(seqs, andi_amd.M_JC, 5) == lane).all()


def _rows(seqs, env):
    """rows of all subjects through andi_hip_scan_rows on a context of its own; returns (counts, timings)"""
    with knobs(**{k[len("ANDI_"):]: v for k, v in env.items()}):
        ctx = andi_amd.Context(0)
        Q = andi_amd.Queries(ctx, seqs)
        esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
        ctx.timings_reset()
        got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q)
        t = ctx.timings()
        for e in esas:
            e.close()
        Q.close()
        ctx.close()
        return got, t


def test_large_calls_are_routed_per_pair():
    """Unset, ANDI_COOP means: a large call's pass A is ROUTED PER PAIR.  On a star set every pair takes the wavefront
    kernel; genomes a few substitutions apart are k_lane_quad's class throughout (none takes it); on genomes with unrelated
    stretches the sampling keeps most pairs away from it and it hands others back.  Same counts every way."""
    os.environ.pop("ANDI_COOP", None)
    star, _ = synth.genome_set(12, 4_900_000, 0.004, 0.03, seed=5)
    lane, t0 = _rows(star, {"ANDI_COOP": "0"})
    assert t0["coop_calls"] == 0 and t0["routed_calls"] == 0
    got, t = _rows(star, {})
    assert t["routed_calls"] == 1 and t["coop_fallbacks"] == 0 and t["coop_query_nt"] + t["lane_query_nt"] == 132 * 4_900_000, t
    assert t["coop_query_nt"] >= 0.9 * 132 * 4_900_000, t  # (pairs more than some 6 % apart are the lane scan's)
    assert (got == lane).all()
    # genomes a few substitutions apart: matches longer than a segment (and k_lane_quad's class throughout)
    base = synth.base_codes(4_900_000, 5)
    close = [synth.to_bytes(synth.mutate_codes(base, 0.00002, 90 + k)) for k in range(12)]
    lane, _ = _rows(close, {"ANDI_COOP": "0"})
    got, t = _rows(close, {})
    assert t["routed_calls"] == 1 and t["coop_query_nt"] == 0, t
    assert (got == lane).all()
    real, _ = synth.realistic_set(12, 4_900_000, 0.004, 0.03, seed=9)
    lane, _ = _rows(real, {"ANDI_COOP": "0"})
    got, t = _rows(real, {})
    assert t["routed_calls"] == 1, t
    assert (got == lane).all()


def test_pairs_handed_back_take_the_second_lane_layout():
    """A wavefront that meets long stretches without homology hands its PAIR back to the lane scan, which gives such
    pairs a second layout of their own.  With the default limit that is rare (the sampling keeps pairs with unrelated
    stretches away from the kernel), so the genomes here carry only 0.5 % of unrelated sequence -- less than the sampling
    notices -- and the limit is lowered to 48 generic steps per segment (ANDI_COOP_GIVEUP).  Same counts as the lane
    scan; pairs were handed back and their layout's passes ran."""
    os.environ.pop("ANDI_COOP", None)
    real, _ = synth.realistic_set(12, 4_900_000, 0.004, 0.03, seed=9, novel_fraction=0.005)
    lane, _ = _rows(real, {"ANDI_COOP": "0"})
    got, t = _rows(real, {"ANDI_COOP_GIVEUP": "48"})
    assert t["routed_calls"] == 1 and t["coop_fallbacks"] > 0 and t["lane_query_nt"] > 0, t
    assert (got == lane).all()


def test_small_calls_are_routed_too():
    """Small calls are routed as well (from 2^25 query symbols x subjects; TINY ones, from 2^18, take the wavefront kernel
    for every pair without sampling), on shorter segments of the wavefront kernel:
    BASELINE's configs[0] shape (3 x 1 Mbp: per-pair segments for the lanes), a call of many short genomes (one segment
    length for the lanes: the routing marks alone decide what a lane of that layout does), the same with structured
    genomes and an early hand-back (the second lane layout in its one-segment-length form), ragged lengths around the
    shortest query the wavefront kernel takes; calls of queries below that length keep the lane scan.  All against the
    oracle, every model the kernel counts differently for."""
    os.environ.pop("ANDI_COOP", None)
    star, _ = synth.genome_set(3, 1_000_000, 0.0004, 0.03, seed=3)
    want = orc.dist_matrix(star, model=orc.M_JC, threads=3)
    got, t = _rows(star, {})  # a TINY call (below 2^25 symbols): the wavefront kernel for every pair, no sampling
    assert t["routed_calls"] == 0 and t["coop_calls"] == 1, t
    assert (got == want).all()
    got, t = _rows(star, {"ANDI_ROUTE_TINY": "1"})  # the same call routed
    assert t["routed_calls"] == 1 and t["adaptive_calls"] == 1 and t["coop_query_nt"] == 6_000_000, t
    assert (got == want).all()
    far, _ = synth.genome_set(3, 1_000_000, 0.05, 0.05, seed=13)  # configs[0] itself: pairs 10 % apart -- the sampling cannot judge them
    want = orc.dist_matrix(far, model=orc.M_JC, threads=3)
    got, t = _rows(far, {})
    assert t["routed_calls"] == 0 and t["coop_calls"] == 1 and (got == want).all(), t
    got, t = _rows(far, {"ANDI_ROUTE_TINY": "1"})
    assert t["routed_calls"] == 1 and t["coop_query_nt"] == 6_000_000 and t["coop_fallbacks"] == 0, t
    assert (got == want).all()
    mixed = [star[0], star[1], synth.unrelated(700_000, 5), star[0][:90_000], star[2][400_000:]]  # tiny, with an unrelated genome and fragments
    got, t = _rows(mixed, {})
    assert t["routed_calls"] == 0 and t["coop_calls"] == 1, t
    assert (got == orc.dist_matrix(mixed, model=orc.M_JC, threads=4)).all()
    many, _ = synth.genome_set(60, 20_000, 0.001, 0.04, seed=4)
    want = orc.dist_matrix(many, model=orc.M_JC, threads=4)
    got, t = _rows(many, {})
    assert t["routed_calls"] == 1 and t["uniform_calls"] == 1 and t["coop_query_nt"] > 0, t
    assert (got == want).all()
    lane, t0 = _rows(many, {"ANDI_COOP": "0"})
    assert t0["routed_calls"] == 0 and (lane == want).all()
    real, _ = synth.realistic_set(40, 30_000, 0.002, 0.03, seed=6, novel_fraction=0.02)
    want = orc.dist_matrix(real, model=orc.M_JC, threads=4)
    got, t = _rows(real, {})
    assert t["routed_calls"] == 1 and (got == want).all(), t
    got, t = _rows(real, {"ANDI_COOP_GIVEUP": "4"})
    assert t["routed_calls"] == 1 and t["uniform_calls"] == 1 and t["coop_fallbacks"] > 0, t
    assert (got == want).all()
    # more than 4096 pairs: the sampling kernel takes four pairs per block; structured genomes, hand-backs forced
    lots, _ = synth.realistic_set(80, 20_000, 0.002, 0.04, seed=16, novel_fraction=0.05)
    want = orc.dist_matrix(lots, model=orc.M_KIMURA, threads=4)
    for env in ({}, {"ANDI_COOP_GIVEUP": "4"}):
        with knobs(**{k[len("ANDI_"):]: v for k, v in env.items()}):
            ctx = andi_amd.Context(0)
            Q = andi_amd.Queries(ctx, lots)
            esas = [andi_amd.Esa(ctx, x, sa="device") for x in lots]
            ctx.timings_reset()
            got = andi_amd.scan_rows(ctx, esas, list(range(len(lots))), Q, model=andi_amd.M_KIMURA)
            t = ctx.timings()
            for e in esas:
                e.close()
            Q.close()
            ctx.close()
        assert t["routed_calls"] == 1 and t["uniform_calls"] == 1 and (got == want).all(), (env, t)
        assert not env or t["coop_fallbacks"] > 0, t
    rng = np.random.default_rng(8)
    base = synth.base_codes(40_000, 9)
    ragged = []
    for k in range(50):  # 2 ... 40 kbp: some below the shortest query the wavefront kernel takes, joined contigs among them
        codes = synth.mutate_codes(base, float(rng.uniform(0.001, 0.05)), 200 + k)
        b = synth.to_bytes(codes[: int(rng.integers(2_000, 40_000))])
        ragged.append(synth.join_contigs(b, 3, seed=k) if k % 7 == 0 else b)
    for model in (orc.M_RAW, orc.M_LOGDET):
        want = orc.dist_matrix(ragged, model=model, threads=4)
        with knobs(COOP=None):
            got = andi_amd.dist_matrix(ragged, model=model)
        assert (got == want).all(), model
    short, _ = synth.genome_set(150, 3_000, 0.001, 0.04, seed=5)
    got, t = _rows(short, {})
    assert t["routed_calls"] == 0, t
    assert (got == orc.dist_matrix(short, model=orc.M_JC, threads=4)).all()


def test_mixed_call_clean_close_and_structured_pairs_against_the_oracle():
    """ONE call whose pairs are of every kind -- clean pairs a few percent apart (the wavefront kernel's), pairs a few
    substitutions apart (k_lane_quad's), pairs with unrelated stretches, repeats and indels (k_lane_cold's, some of them
    handed back by the wavefront kernel) -- routed per pair, against the oracle: 16 sampled ordered pairs."""
    from tests.conftest import verify_suffix_array
    from oracle import orc
    os.environ.pop("ANDI_COOP", None)
    n = 3_400_000
    base = synth.base_codes(n, 31)
    clean = [synth.to_bytes(synth.mutate_codes(base, d, 40 + k)) for k, d in enumerate((0.002, 0.01, 0.02, 0.03))]
    close = [synth.to_bytes(synth.mutate_codes(base, 0.00003, 50 + k)) for k in range(3)]
    real, _ = synth.realistic_set(5, n, 0.002, 0.03, seed=31)
    seqs = clean + close + real  # (the structured genomes come from another base: unrelated to the others but for chance)
    seqs.append(synth.to_bytes(synth.mutate_codes(synth.realistic_base(n, 31), 0.015, 77)))  # a clean relative of the structured ones
    with knobs(COOP=None):
        ctx = andi_amd.Context(0)
        ctx.expect_queries(len(seqs) - 1)
        Q = andi_amd.Queries(ctx, seqs)
        esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
        ctx.timings_reset()
        got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q)
        t = ctx.timings()
        assert t["routed_calls"] == 1 and t["coop_query_nt"] > 0 and t["lane_query_nt"] > 0, t
        for i in (1, 5, 8, 12):
            RS = seqs[i][::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")) + b"#" + seqs[i]
            verify_suffix_array(RS, esas[i].SA)
            O = orc.OracleEsa(seqs[i], sa=esas[i].SA)
            for j in (0, 3, 4, 6, 7, 9, 11, 12):
                if j != i:
                    assert (got[i, j] == O.dist_anchor(seqs[j])).all(), (i, j)
            O.close()
        for e in esas:
            e.close()
        Q.close()
        ctx.close()


def test_ragged_rows_thresholds_and_windows_full_of_heads():
    """The wavefront kernel forced on what the trial would not pick: a C4-shaped call of ragged short genomes (hundreds
    of queries per subject: extended probe-table entries), anchor thresholds from other significance levels, and pairs
    so far apart that a window lists more heads than it has room for (the window then ends at the first one dropped)."""
    rng = np.random.default_rng(7)
    base = synth.base_codes(30000, 3)
    seqs = []
    for k in range(300):
        codes = synth.mutate_codes(base, float(rng.uniform(0.001, 0.02)), 100 + k)
        seqs.append(synth.to_bytes(codes[: int(30000 * (1.0 - 0.3 * rng.random()))]))
    subjects = [0, 7, 150, 299]
    want = np.stack([orc.scan_row(orc.OracleEsa(seqs[i]), seqs, i, orc.M_JC, threads=0) for i in subjects])
    with knobs(COOP=4):
        ctx = andi_amd.Context(0)
        ctx.expect_queries(len(seqs) - 1)
        Q = andi_amd.Queries(ctx, seqs)
        esas = [andi_amd.Esa(ctx, seqs[i]) for i in subjects]
        got = andi_amd.scan_rows(ctx, esas, subjects, Q)
        assert ctx.timings()["coop_calls"] == 1
        for e in esas:
            e.close()
        Q.close()
        ctx.close()
    assert (got == want).all(), np.argwhere((got != want).any(axis=2))[:5]
    a, b = synth.pair(200000, 0.08, seed=21)
    c = synth.to_bytes(synth.mutate_codes(synth.base_codes(200000, 21), 0.25, 5))
    for p_value in (0.025, 0.3, 1e-6):
        want = orc.dist_matrix([a, b, c], p_value=p_value, model=orc.M_RAW, threads=3)
        for coop in (2, 8):
            with knobs(COOP=coop):
                got = andi_amd.dist_matrix([a, b, c], p_value=p_value, model=andi_amd.M_RAW)
            assert (got == want).all(), (p_value, coop)


# ------------------------------------------------------------------ k_pool_cold (coop_pool.h): the windows' walks pooled through global memory
def _pooled_rows(seqs, model, segment, **kn):
    """every pair by the forced wavefront kernel on segments long enough for its pooled form; checks that k_pool_cold ran"""
    with knobs(COOP=4, POOL=None, **kn):
        ctx = andi_amd.Context(0)
        Q = andi_amd.Queries(ctx, seqs)
        esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
        ctx.timings_reset()
        got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q, model=model, segment=segment)
        t = ctx.timings()
        for e in esas:
            e.close()
        Q.close()
        ctx.close()
    assert t["pool_calls"] == 1 and t["coop_calls"] == 1 and t["fixups"] == 0, t
    return got


def test_pooled_walks_divergence_ladder():
    base = synth.base_codes(300000, 5)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 10 + k)) for k, d in enumerate((0.0, 0.0005, 0.005, 0.02, 0.05, 0.15))]
    want = orc.dist_matrix(seqs, model=orc.M_JC, threads=4)
    for segment in (32768, 65536, 262144):
        for first in (None, 1, 3):  # (the first window's length: whole segments, 2048 and 6144 positions)
            got = _pooled_rows(seqs, andi_amd.M_JC, segment, POOL_FIRST=first)
            assert (got == want).all(), "segment %d, first window %s: pairs %s" % (segment, first, np.argwhere((got != want).any(axis=2))[:8].tolist())


def test_pooled_walks_structured_strands_contigs_edges():
    seqs, _ = synth.realistic_set(5, 400000, 0.002, 0.04, seed=23)
    want = orc.dist_matrix(seqs, model=orc.M_KIMURA, threads=5)
    for segment in (32768, 131072):
        assert (_pooled_rows(seqs, andi_amd.M_KIMURA, segment) == want).all(), "structured, segment %d" % segment
    base = synth.base_codes(300000, 9)
    s = synth.to_bytes(base)
    seqs = [s, _revcomp(synth.to_bytes(synth.mutate_codes(base, 0.02, 3))),  # reverse strand
            _revcomp(s)[-100000:] + s[:100000],  # across the '#' of RS
            synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.01, 4)), 9),  # '!' separators: windows that are not clean
            s, synth.unrelated(200000, 77), s[:700]]  # identical, unrelated, short
    want = orc.dist_matrix(seqs, model=orc.M_RAW, threads=4)
    for segment in (32768, 100000):
        got = _pooled_rows(seqs, andi_amd.M_RAW, segment)
        assert (got == want).all(), "segment %d: pairs %s" % (segment, np.argwhere((got != want).any(axis=2))[:8].tolist())


def test_routed_calls_choose_one_wavefront_kernel():
    """A routed call takes ONE wavefront kernel (scan.h: pool_match): k_pool_cold where the pairs whose sampled matches are long
    hold half of its segments -- a set 1 % apart --, k_coop_cold on the bench set's kind (3 % apart on average) and on genomes
    1e-5 apart; ANDI_POOL_MATCH=0 (a test hook) makes every pair suit k_pool_cold.  Same counts every way."""
    for d_hi, expect_pool in ((0.006, 1), (0.03, 0)):
        seqs, _ = synth.genome_set(16, 4_900_000, 0.0004, d_hi, seed=5)  # (2^30 query symbols x subjects and more: the host looks at the layout)
        lane, t0 = _rows(seqs, {"ANDI_COOP": "0"})
        got, t = _rows(seqs, {})
        assert t["routed_calls"] == 1 and t["pool_calls"] == expect_pool, (d_hi, t)
        assert (got == lane).all(), d_hi
        forced, t = _rows(seqs, {"ANDI_POOL_MATCH": "0"})
        assert t["pool_calls"] == 1 and (forced == lane).all(), (d_hi, t)
    base = synth.base_codes(4_900_000, 3)
    close = [synth.to_bytes(synth.mutate_codes(base, 2e-5, 40 + k)) for k in range(16)]
    lane, _ = _rows(close, {"ANDI_COOP": "0"})
    got, t = _rows(close, {})
    assert t["pool_calls"] == 0 and (got == lane).all(), t
    forced, t = _rows(close, {"ANDI_POOL_MATCH": "0", "ANDI_QUAD_MATCH": "-1"})  # (... and keeps them from k_lane_quad: the wavefront kernel's)
    assert (forced == lane).all(), t


def test_joined_contigs_are_routed_by_the_distance_between_their_ends():
    """andi --join (src/sequence.c:78-125): every genome a set of contigs joined by '!', cut at different places in every genome, so each
    separator of the query or the subject ends the pair's diagonal.  A large call routes such pairs by the mean distance between the
    ends (k_pair_estimate, from the separators counted per sequence on the device): many contigs -- the lane scan; few -- the wavefront
    kernels.  Same counts every way, rows against the oracle."""
    os.environ.pop("ANDI_COOP", None)
    base = synth.base_codes(3_000_000, 61)
    whole = [synth.to_bytes(synth.mutate_codes(base, d, 70 + k)) for k, d in enumerate((0.001, 0.004, 0.008, 0.012, 0.003, 0.006, 0.01))]
    many = [synth.join_contigs(s, 150, seed=5 + k) for k, s in enumerate(whole)]   # an end every 10 000 positions
    few = [synth.join_contigs(s, 4, seed=9 + k) for k, s in enumerate(whole)]      # ... every 430 000
    for seqs, expect_wavefronts in ((many, False), (few, True)):
        got, t = _rows(seqs, {})
        assert t["routed_calls"] == 1 and t["fixups"] == 0, t
        assert (t["coop_query_nt"] > 0) == expect_wavefronts and (expect_wavefronts or t["lane_query_nt"] > 0), t
        lane, _ = _rows(seqs, {"ANDI_COOP": "0"})
        assert (got == lane).all()
        for i in (0, 3):
            O = orc.OracleEsa(seqs[i])
            want = orc.scan_row(O, seqs, i, orc.M_JC, threads=os.cpu_count() or 1)
            O.close()
            assert (got[i] == want).all(), i
