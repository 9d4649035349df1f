"""Pins the CPU oracle to the reference: every known answer the reference's own
tests hold for this path, plus the numbers the unmodified reference produced in
this image as recorded in SURVEY.md / BASELINE.md.  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

FIX200 = (b"TACGAGCACTGGTGGAATTGATGTC" b"CAGTCTTATATGGCGCACCAGGCTG" b"ATAGTAGTAGCAGTTTGCTTATCTC"
          b"ATCGCGTGTTTCCGGATGACAGAGA" b"TACGTGCACTGGTGGGATTGATGTC" b"TAGTATTATATGGCGCACCAGGATG"
          b"ATAGTAGTAGCAGTTTGCTTATCCC" b"ATCGCGTGTTTGCGGATGACCGAGA")  # test/test_esa.c:53-62
FIX200_SEP = FIX200[:100] + b"!" + FIX200[100:]  # test/test_esa.c:78-88


def test_rs_known_answers(orc):
    # test/test_seq.c:34-36
    E = orc.OracleEsa(b"ACGTTGCA")
    assert E.RS == b"TGCAACGT#ACGTTGCA" and E.n == 17 and E.gc == 0.5
    # test/test_seq.c:69-70
    assert orc.OracleEsa(b"ACGT!TGCA").RS == b"TGCA;ACGT#ACGT!TGCA"


def test_normalize(orc):
    # test/test_seq.c:55-66: strip non-ACGT, keep '!', flag it
    import ctypes as C
    for raw, want in ((b"11ACGTNN7682394689NNTGCA11", b"ACGTTGCA"), (b"@ACGT_!0TGCA        ", b"ACGT!TGCA"),
                      (b"acgt", b"ACGT")):
        buf = C.create_string_buffer(raw)
        flag = C.c_int(0)
        n = orc.lib().orc_normalize(buf, C.byref(flag))
        assert buf.value == want and n == len(want)
        assert bool(flag.value) == (raw.upper() != want if raw != b"acgt" else False)


def test_threshold_minimal(orc):
    # test/test_process.c:16-29
    L = orc.lib()
    length, gc, p = 100000, 0.5, 0.025
    thr = L.orc_min_anchor_length(p, gc, length)
    assert 1 - p < L.orc_shustring_cum_prob(thr + 1, gc / 2, length)
    assert 1 - p <= L.orc_shustring_cum_prob(thr, gc / 2, length)
    assert 1 - p > L.orc_shustring_cum_prob(thr - 1, gc / 2, length)


def test_worked_example(orc):
    # arrays of the reference for RS = TGCAACGT#ACGTTGCA
    E = orc.OracleEsa(b"ACGTTGCA")
    assert E.threshold == 5
    assert list(E.SA) == [8, 16, 3, 4, 9, 15, 2, 5, 10, 14, 1, 6, 11, 7, 13, 0, 12]
    assert list(E.LCP) == [-1, 0, 1, 1, 4, 0, 2, 1, 3, 0, 3, 1, 2, 0, 1, 4, 1, -1]
    assert bytes(E.FVC) == b"TAACTCAGTGATTTGAT"
    assert E.get_match(b"GTTGA", cached=False) == (4, 12, 12) and E.SA[12] == 11
    assert E.get_match(b"TTTT", cached=False) == (2, 16, 16)
    assert E.get_match(b"CAAC", cached=False) == (4, 6, 6) and E.SA[6] == 2


def _check_match(E, q):
    # assert_equal_cache_nocache, test/test_esa.c:38-44
    a, b = E.get_match(q, True), E.get_match(q, False)
    assert a == b, q
    l, i, _ = a
    rs, pos = E.RS, int(E.SA[i])
    assert rs[pos:pos + l] == q[:l]
    assert l == len(q) or pos + l >= len(rs) or rs[pos + l] != q[l]


@pytest.mark.parametrize("fix", [FIX200, FIX200_SEP])
def test_esa_samples(orc, fix):
    # test/test_esa.c:107-170
    E = orc.OracleEsa(fix)
    for q in (b"A", b"C", b"CT", b"AAGACTGG", b"AATTAAAA", b"ACCGAGAA", b"AAAAAAAAAAAA", b"!AAAAAAAAAAA"):
        assert E.get_match(q, True) == E.get_match(q, False), q
    for q in (b"AAGACTGG", b"AATTAAAA", b"ACCGAGAA", b"AAAAAAAAAAAA"):
        _check_match(E, q)


@pytest.mark.parametrize("fix", [FIX200, FIX200_SEP])
def test_esa_all_11mers_sampled(orc, fix):
    # test/test_esa.c:172-192 walks all 4^11 11-mers; here every 11-mer of the
    # fixture itself (all present) plus 60k seeded random ones on the CPU; the
    # exhaustive walk is done on the device (tests/test_esa_gpu.py).
    E = orc.OracleEsa(fix)
    rng = np.random.default_rng(11)
    codes = rng.integers(0, 4 ** 11, 60000)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for c in codes:
        q = acgt[[(int(c) >> (2 * (10 - t))) & 3 for t in range(11)]].tobytes()
        _check_match(E, q)
    clean = fix.replace(b"!", b"A")
    for p in range(len(clean) - 11):
        _check_match(E, clean[p:p + 11])


def test_reference_recorded_scan_statistics(orc, golden_s42):
    """dist_anchor of the unmodified reference, instrumented."""
    rec = json.load(open(os.path.join(GOLDEN, "reference_recorded.json")))["s42"]
    E = orc.OracleEsa(golden_s42["s0"])
    assert E.threshold == rec["threshold"]
    for d in ("0.1", "0.01", "0.001"):
        counts, st = E.dist_anchor(golden_s42["s1_" + d], stats=True)
        for key, want in rec[d].items():
            if key == "cache_hit_probes":
                # the survey counted table hits; one probe near the query end
                # (remaining length <= 10) takes the uncached path
                assert st["esa_probes"] - want in (0, 1)
            else:
                assert st[key] == want, (d, key)
        assert st["iterations"] == st["esa_probes"] + st["lucky_hits"]
        assert (counts == golden_s42["counts_" + d]).all()


def test_reference_recorded_phylip(orc, golden_s1729):
    """4-decimal distances andi printed for test_fasta -s 1729."""
    rec = json.load(open(os.path.join(GOLDEN, "reference_recorded.json")))["s1729"]
    seqs = golden_s1729["seqs"]
    pairs = [(0, 1), (0, 2), (1, 2)]
    for name, model in (("JC", orc.M_JC), ("RAW", orc.M_RAW), ("KIMURA", orc.M_KIMURA)):
        M = orc.dist_matrix(seqs, model=model, threads=3)
        got = ["%.4f" % orc.estimate(M[i, j].astype(np.uint64) + M[j, i], model) for i, j in pairs]
        assert got == rec[name], name
        if name == "JC":
            assert (M == golden_s1729["counts_jc"]).all()
            vv = rec["JC_vv"]
            assert "%.4f" % orc.estimate(M[0, 1], model) == vv["01"]
            assert "%.4f" % orc.estimate(M[1, 0], model) == vv["10"]
            assert "%.4f" % orc.estimate(M[1, 2], model) == vv["12"]
            assert "%.4f" % orc.estimate(M[2, 1], model) == vv["21"]
            cov = [orc.coverage(M[i, j]) for i, j in ((0, 1), (1, 0), (1, 2), (2, 1))]
            assert 0.96 < cov[0] < 0.975 and 0.81 < cov[2] < 0.84


def test_statistical_tolerance(orc):
    """test/test_random.sh:19-69: |est-d| <= 0.055 and <= 5.5 % of d."""
    from andi_amd import synth
    for d in (0.0, 0.001, 0.01, 0.02, 0.05, 0.1, 0.2, 0.3):
        a, b = synth.pair(100000, d, seed=100 + int(d * 1000))
        M = orc.dist_matrix([a, b], model=orc.M_JC, threads=2)
        est = orc.estimate(M[0, 1].astype(np.uint64) + M[1, 0], orc.M_JC)
        assert abs(est - d) <= 0.055 and abs(est - d) <= 0.055 * d + 1e-12, (d, est)


def test_estimators_edge_cases(orc):
    z = np.zeros(17, np.uint32)
    assert np.isnan(orc.estimate(z, orc.M_RAW)) and np.isnan(orc.estimate(z, orc.M_JC))
    m = z.copy()
    m[0] = m[5] = m[10] = m[15] = 25
    m[16] = 100
    assert orc.estimate(m, orc.M_RAW) == 0.0 and orc.estimate(m, orc.M_JC) == 0.0
    assert orc.estimate(m, orc.M_KIMURA) == 0.0 and orc.estimate(m, orc.M_ANI) == 100.0
    assert orc.coverage(m) == 1.0
    m[1] = 10  # A->C
    raw = 10 / 110
    assert orc.estimate(m, orc.M_RAW) == raw
    assert orc.estimate(m, orc.M_JC) == -0.75 * np.log(1.0 - (4.0 / 3.0) * raw)


# ---- the reference's shell tests that hold an expectation for dist_anchor + the estimators: pinned on
# the ORACLE here (tests/test_cli.py holds the product's CLI to the same expectations on the GPU box).  The reference's
# scripts draw random seeds (RANDOM_SEED unset => test_fasta -s 0 = a seed from the clock), so the expectation holds
# for ANY seed; three seeds each, sequences from andi_amd/synth.py (the generator's model, test/test_fasta.cxx:73-118).
def _tf(length, seed, d=0.1):
    """`test_fasta -s seed -l length -d d`: S0 (the base) and S1 (d substitutions per site from it)."""
    from andi_amd import synth
    return synth.pair(length, d, seed=seed)


def _raw(orc, a, b):
    M = orc.dist_matrix([a, b], model=orc.M_RAW, threads=2)
    return orc.estimate(M[0, 1].astype(np.uint64) + M[1, 0], orc.M_RAW), M


@pytest.mark.parametrize("seed", [1, 23, 777])
def test_join_three_and_two_contigs(orc, seed):
    # test/test_join.sh:15-38: three contigs (1000, 1000, 10000 nt) per genome, joined by -j ('!'): RAW within 0.03 of 0.1
    p1, p2, p3 = _tf(1000, seed), _tf(1000, seed + 2), _tf(10000, seed + 3)
    d, _ = _raw(orc, b"!".join(p[0] for p in (p1, p2, p3)), b"!".join(p[1] for p in (p1, p2, p3)))
    assert abs(d - 0.1) < 0.03, d
    # test/test_join.sh:46-67: unbalanced -- one contig against two, the extra contig (1000 nt) has no partner
    p2, p3 = _tf(1000, seed + 5), _tf(10000, seed + 6)
    d, _ = _raw(orc, p3[0], b"!".join((p2[1], p3[1])))
    assert abs(d - 0.1) < 0.03, d
    # test/test_join.sh:69-96: unbalanced 2 -- two contigs against three
    p1, p2, p3 = _tf(1000, seed + 11), _tf(1000, seed + 12), _tf(10000, seed + 13)
    d, _ = _raw(orc, b"!".join((p1[0], p3[0])), b"!".join((p1[1], p2[1], p3[1])))
    assert abs(d - 0.1) < 0.03, d


@pytest.mark.parametrize("seed", [1, 23, 777])
def test_unrelated_sequences_give_nan(orc, seed):
    # test/nan.sh:11-25: two files of unrelated 10 kbp sequences, joined per file: "reported as nan" (src/io.c:282-289)
    a, b = _tf(10000, seed), _tf(10000, seed + 1)
    M = orc.dist_matrix([b"!".join(a), b"!".join(b)], model=orc.M_JC, threads=2)
    assert np.isnan(orc.estimate(M[0, 1].astype(np.uint64) + M[1, 0], orc.M_JC))


@pytest.mark.parametrize("seed", [1, 23, 777])
def test_low_homology_is_below_the_warning_line(orc, seed):
    # test/low_homo.sh:12-29: 100 homologous nt in front of 100 kbp of unrelated sequence per genome: the distance is a
    # number or nan, and where it is a number a coverage is below 0.2 (the warning's condition, src/io.c:291-303)
    a, b, both = _tf(100000, seed), _tf(100000, seed + 1), _tf(100, seed + 2)
    s0, s1 = both[0] + b"!" + a[0], both[1] + b"!" + b[1]
    M = orc.dist_matrix([s0, s1], model=orc.M_JC, threads=2)
    d = orc.estimate(M[0, 1].astype(np.uint64) + M[1, 0], orc.M_JC)
    assert np.isnan(d) or orc.coverage(M[0, 1]) < 0.2 or orc.coverage(M[1, 0]) < 0.2
    assert orc.coverage(M[0, 1]) < 0.2 and orc.coverage(M[1, 0]) < 0.2


# ---------------------------------------------------------------- bootstrap: GSL's published multinomial, restated
def test_oracle_multinomial_follows_the_published_law(orc):
    """model_bootstrap (src/model.c:222-232) = gsl_ran_multinomial(RNG, 16, N, counts / N): the oracle restates GSL's
    published conditional-binomial construction (oracle/andi_oracle.c).  The reference seeds GSL from the clock
    (src/andi.c:272-279) and holds no vector for it -- parity unpinned --, so the LAW is pinned instead: N is preserved,
    empty cells stay empty, every cell's marginal is Binomial(N, p_c) and so are sums of cells (which a wrong
    conditioning would break), tiny counts and counts of 10^5..10^6; the binomial sampler itself against the exact law."""
    import ctypes as C
    from conftest import binomial_gof_pvalue
    L = orc.lib()
    rng = C.c_uint64(2024)
    for n, p in ((5, 0.3), (40, 0.02), (1000, 0.5), (123456, 0.25), (3_000_000, 1e-5), (2_000_000, 0.9993)):
        x = np.array([L.orc_ran_binomial(C.byref(rng), p, n) for _ in range(20000)])
        assert x.min() >= 0 and x.max() <= n
        assert binomial_gof_pvalue(x, n, p) > 1e-4, (n, p)
    reps = 10000
    M = np.zeros((3, 3, 17), np.uint32)
    M[0, 1, :16] = [30, 0, 1, 0, 0, 7, 0, 0, 0, 2, 2, 0, 1, 0, 0, 1]          # tiny counts
    M[1, 0, :16] = [0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    M[0, 2, :16] = [300000, 50, 700, 30, 60, 250000, 20, 10, 5, 3, 280000, 40, 9, 1, 0, 270000]
    M[2, 0, :16] = [290000, 45, 650, 33, 70, 255000, 25, 12, 4, 2, 275000, 35, 11, 0, 0, 265000]
    M[:, :, 16] = 1234
    B = orc.bootstrap(M, reps, seed=11)
    assert (B[:, 1, 2, :16] == 0).all() and (B[:, 0, 0, 0] == 1).all() and (B[:, 0, 0, 16] == 1).all()
    for i, j in ((0, 1), (0, 2)):
        c = M[i, j, :16].astype(np.int64) + M[j, i, :16]
        N = int(c.sum())
        x = B[:, i, j, :16].astype(np.int64)
        assert (B[:, i, j] == B[:, j, i]).all() and (x.sum(axis=1) == N).all() and (x[:, c == 0] == 0).all()
        assert (B[:, i, j, 16] == 2468).all()  # seq_len summed, src/model.c:44
        for cell in np.nonzero(c)[0]:
            assert binomial_gof_pvalue(x[:, cell], N, c[cell] / N) > 1e-5, (i, j, cell)
        for cells in ((0, 5), (0, 5, 10, 15), (1, 2, 3, 4), (10, 15)):
            assert binomial_gof_pvalue(x[:, list(cells)].sum(axis=1), N, c[list(cells)].sum() / N) > 1e-5, (i, j, cells)
