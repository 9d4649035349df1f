"""Device index build (K1-K4) and longest-match queries (K5) against the oracle.
Bit-exact; calls go through the C-ABI."""
import numpy as np
import pytest

from conftest import rand_dna
from test_oracle_pins import FIX200, FIX200_SEP

pytestmark = pytest.mark.gpu


def _subjects():
    from andi_amd import synth
    rng = np.random.default_rng(21)
    yield "tiny", b"ACGTTGCA"
    yield "one-char", b"A"
    yield "homopolymer", b"A" * 500
    yield "fix200", FIX200
    yield "fix200-sep", FIX200_SEP
    yield "random-3k", rand_dna(rng, 3000)
    yield "two-letter", rand_dna(rng, 5000, b"AT")
    yield "repeats", rand_dna(rng, 700) * 9 + rand_dna(rng, 300)
    yield "joined", synth.join_contigs(rand_dna(rng, 40000), 12)
    yield "random-300k", rand_dna(rng, 300000)
    yield "palindromic", (lambda s: s + s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))(rand_dna(rng, 20000))


@pytest.mark.parametrize("name,seq", list(_subjects()), ids=[n for n, _ in _subjects()])
def test_index_arrays_bit_exact(ctx, orc, name, seq):
    import andi_amd
    O = orc.OracleEsa(seq)
    E = andi_amd.Esa(ctx, seq)
    assert E.RS == O.RS and E.threshold == O.threshold
    LCP, CLD, FVC, cache = E.download()
    assert (LCP == O.LCP).all()
    # FVC[0] = S[SA[0]-1] is never consulted; both sides compute it the same way
    assert (FVC == O.FVC).all()
    # child table: identical on every slot the sweep writes; unwritten = -1 on both sides
    assert (CLD == O.CLD).all()
    # 10-mer table: the fields callers use (src/test/test_esa.c:32-36) plus m
    assert (cache == O.cache).all()
    E.close()
    O.close()


@pytest.mark.parametrize("fix", [FIX200, FIX200_SEP], ids=["fix200", "fix200-sep"])
def test_all_11mers_cached_equals_uncached(ctx, orc, fix):
    """test/test_esa.c:172-192 (prefix_dfs over ALL 4^11 11-mers): cached ==
    uncached on (l,i,j), the match is a true and maximal prefix match; on the
    device, and a 1/61 sample of them also against the oracle.

    The 11-mers are laid out in one query as XXXXXXXXXXXN: 'N' never occurs in
    RS, so the match of the suffix starting at an 11-mer ends exactly where a
    query of length 11 would end."""
    import andi_amd
    E = andi_amd.Esa(ctx, fix)
    O = orc.OracleEsa(fix)
    rs = np.frombuffer(O.RS + b"\0", np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    shifts = 2 * (10 - np.arange(11))
    block = 4 ** 9
    for start in range(0, 4 ** 11, block):
        codes = np.arange(start, start + block, dtype=np.int64)
        kmers = acgt[(codes[:, None] >> shifts[None, :]) & 3]
        text = np.concatenate([kmers, np.full((block, 1), ord("N"), np.uint8)], axis=1)
        Q = andi_amd.Queries(ctx, [text.tobytes()])
        a = andi_amd.match_positions(E, Q, 0, 0, 12 * block, cached=True)[::12]
        b = andi_amd.match_positions(E, Q, 0, 0, 12 * block, cached=False)[::12]
        Q.close()
        assert (a == b).all()
        l, pos = a[:, 0].astype(np.int64), a[:, 3].astype(np.int64)
        assert (l >= 0).all() and (l <= 11).all()
        # true prefix match ...
        for t in range(11):
            sel = l > t
            assert (rs[pos[sel] + t] == kmers[sel, t]).all()
        # ... and maximal
        sel = l < 11
        assert (rs[pos[sel] + l[sel]] != kmers[sel, l[sel]]).all()
        for k in range(start % 61, block, 61):
            assert tuple(a[k][:3]) == O.get_match(kmers[k].tobytes(), True)
    E.close()
    O.close()


def _match_all_positions(ctx, orc, subject, query):
    import andi_amd
    E = andi_amd.Esa(ctx, subject)
    O = orc.OracleEsa(subject)
    Q = andi_amd.Queries(ctx, [query])
    n = len(query)
    for cached in (True, False):
        got = andi_amd.match_positions(E, Q, 0, 0, n, cached=cached)
        step = 1 if n <= 4000 else max(1, n // 4000)
        for p in list(range(0, n, step)) + list(range(max(0, n - 15), n)):
            l, i, j = O.get_match(query[p:], cached)
            assert tuple(got[p][:3]) == (l, i, j), (cached, p)
            assert got[p][3] == O.SA[i]
    Q.close()
    E.close()
    O.close()


def test_match_every_position_small(ctx, orc):
    from andi_amd import synth
    a, b = synth.pair(3000, 0.05, seed=5)
    _match_all_positions(ctx, orc, a, b)
    _match_all_positions(ctx, orc, FIX200_SEP, FIX200)
    _match_all_positions(ctx, orc, FIX200, FIX200_SEP)  # '!' inside the query


def test_match_positions_joined_and_unrelated(ctx, orc):
    from andi_amd import synth
    rng = np.random.default_rng(8)
    g = rand_dna(rng, 120000)
    a = synth.join_contigs(g, 7, seed=1)
    b = synth.join_contigs(synth.to_bytes(synth.mutate_codes(
        np.frombuffer(g.translate(bytes.maketrans(b"ACGT", bytes(range(4)))), np.uint8), 0.02, 3)), 5, seed=2)
    _match_all_positions(ctx, orc, a, b)
    _match_all_positions(ctx, orc, rand_dna(rng, 50000), rand_dna(rng, 20000))


def _sa_texts():
    from andi_amd import synth
    rng = np.random.default_rng(303)
    yield from _subjects()
    yield "homopolymer-100k", b"A" * 100000  # every round keeps every suffix: 13 rounds of doubling
    yield "period-3", b"ACG" * 30000
    yield "repeat-5k-x7", b"".join(rand_dna(rng, 30000) + u for u in [rand_dna(rng, 5000)] * 7)  # rRNA-operon-like
    yield "two-copies", (lambda s: s + b"!" + s)(rand_dna(rng, 60000))
    yield "length-21", rand_dna(rng, 21)
    yield "length-22", rand_dna(rng, 22)
    yield "random-2M", synth.to_bytes(synth.base_codes(2_000_000, 9))
    # suffixes with a separator among their first 16 symbols (round 0 keys them with zeros from there on: they tie with
    # "...AAAA" suffixes and with each other, and round 1 has to put them in their true order)
    yield "short-contigs", b"!".join([b"ACG", b"ACGT", b"AC", b"ACGTA", b"ACG", b"A", b"AAAA", b"AAAAAAAAAAAAAAAAAAAA"] * 40)
    yield "poly-a-at-separators", b";".join([b"GATTACA" + b"A" * k for k in range(0, 40)] * 3)
    yield "same-contig-ends", b"!".join([rand_dna(rng, 50 + 7 * k) + b"ACGTTGCAACGTAC" for k in range(60)] + [b"TTTT" + rand_dna(rng, 300)] * 5)
    yield "contigs-in-a-repeat", b"!".join([(lambda u: u[:900 + 13 * k])(rand_dna(np.random.default_rng(5), 4000)) for k in range(40)])


@pytest.mark.parametrize("name,seq", list(_sa_texts()), ids=[n for n, _ in _sa_texts()])
def test_device_suffix_array_equals_host(ctx, name, seq):
    """esa_init_SA (src/esa.c:294-304) on the device: the suffix array of RS is unique, so the device sorter
    (sa_device.hip) must reproduce the host sorter's (host_sais.cpp, itself pinned to the oracle's independent
    sorter in tests/test_host.py) entry for entry."""
    import andi_amd
    ctx.timings_reset()
    E = andi_amd.Esa(ctx, seq, sa="device", build=False)
    t = ctx.timings()
    assert t["sa_builds"] == 1 and t["sa_rounds"] >= 1
    want = andi_amd.suffix_array(E.RS)
    got = E.SA
    assert got.shape == want.shape and (got == want).all(), (name, np.argwhere(got != want)[:5].ravel())
    E.close()


def test_scan_on_device_built_suffix_arrays(ctx, orc):
    """The whole device path of a subject: text up, suffix array, scan index, scan -- and the one-call seam with
    the suffix arrays on the device (the default) and on the host."""
    import andi_amd
    from andi_amd import synth
    base = synth.base_codes(150000, 31)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 60 + k)) for k, d in enumerate((0.0, 0.003, 0.03, 0.1))]
    seqs.append(synth.join_contigs(seqs[1], 7, seed=2))
    want = orc.dist_matrix(seqs, threads=0)
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, s, sa="device") for s in seqs]
    got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q)
    assert (got == want).all()
    for e in esas:
        e.close()
    Q.close()
    assert (andi_amd.dist_matrix(seqs, host_threads=2) == want).all()
    assert (andi_amd.dist_matrix(seqs, host_threads=2, sa_on_host=True) == want).all()
    with pytest.raises(andi_amd.AndiHipError, match="outside"):
        andi_amd.Esa(ctx, seqs[0][:500] + b"N" + seqs[0][500:1000], sa="device")


def test_index_builds_in_one_batch(ctx, orc):
    """andi_hip_esa_build_index_batch: the scan indexes of several subjects of different lengths (and one that
    needs the reference walk) in one pair of launches; the scan on them equals the oracle."""
    import andi_amd
    from andi_amd import synth
    rng = np.random.default_rng(8)
    base = synth.base_codes(90000, 41)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 70 + k))[: 90000 - 7000 * k] for k, d in enumerate((0.0, 0.01, 0.05))]
    seqs += [rand_dna(rng, 300), synth.join_contigs(seqs[1], 5, seed=3), b"ACGT" * 2000]
    want = orc.dist_matrix(seqs, threads=0)
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, s, sa="device", build=False) for s in seqs]
    andi_amd.lib.build_indexes(ctx, esas)
    got = andi_amd.scan_rows(ctx, esas, list(range(len(seqs))), Q)
    assert (got == want).all()
    for e in esas:
        e.close()
    Q.close()


def _table_subjects():
    from andi_amd import synth
    rng = np.random.default_rng(33)
    yield "tiny", b"ACGTTGCA"
    yield "homopolymer", b"A" * 700
    yield "two-letter", rand_dna(rng, 2500, b"AC")
    yield "random-1.5k", rand_dna(rng, 1500)
    yield "repeats", rand_dna(rng, 300) * 6 + rand_dna(rng, 200)
    yield "joined", synth.join_contigs(rand_dna(rng, 3000), 9)
    yield "short-contigs", b"!".join([b"ACG", b"ACGT", b"AC", b"ACGTA", b"ACG"] * 30)
    yield "poly-a-at-separators", b";".join([b"GATTACA" + b"A" * k for k in range(0, 30)] * 2)
    yield "same-contig-ends", b"!".join([rand_dna(rng, 40 + 5 * k) + b"ACGTTGCAACGTAC" for k in range(30)])
    yield "random-40k", rand_dna(rng, 40000)


@pytest.mark.parametrize("name,seq", list(_table_subjects()), ids=[n for n, _ in _table_subjects()])
def test_probe_table_entries_against_brute_force(ctx, name, seq):
    """The scan index (esa_build.hip: k_probe_table) entry by entry: for EVERY K-mer w the table must say what
    get_match would find for a query that starts with w (src/esa.c:615-631) -- where w occurs (once: its position;
    several times: its suffix-array interval), or, if it does not occur, the length of its longest prefix that
    does, whether exactly one suffix starts with that prefix, and which.  The expectation is made from the text
    alone by counting, per length, how many positions start with each prefix."""
    import andi_amd
    E = andi_amd.Esa(ctx, seq, sa="device")
    K, table = E.download_index()
    rs = np.frombuffer(E.RS, np.uint8)
    n = len(rs)
    SA = E.SA
    rank = np.empty(n, np.int64)
    rank[SA] = np.arange(n)
    code = np.full(256, 4, np.int64)
    code[list(b"ACGT")] = [0, 1, 2, 3]
    t = np.concatenate([code[rs], np.full(K, 4, np.int64)])
    # count[l][c], where[l][c]: positions that start with the l-mer c (ACGT only), and one of them
    count, where = [None], [None]
    pref = np.zeros(n, np.int64)
    ok = np.ones(n, bool)
    for l in range(1, K + 1):
        ok &= t[l - 1:l - 1 + n] < 4
        pref = pref * 4 + np.minimum(t[l - 1:l - 1 + n], 3)
        count.append(np.bincount(pref[ok], minlength=4 ** l))
        w = np.zeros(4 ** l, np.int64)
        w[pref[ok]] = np.nonzero(ok)[0]
        where.append(w)
    codes = np.arange(4 ** K)
    x, y = table[:, 0].astype(np.int64), table[:, 1].astype(np.int64)
    kind = y & 3
    occurs = count[K]
    assert ((kind == 1) == (occurs == 1)).all() and ((kind == 2) == (occurs > 1)).all() and (kind != 3).all()
    once = occurs == 1
    assert (x[once] == where[K][once]).all()
    form = E.single_form()
    if form:  # the nucleotides behind the one occurrence, as many as the entry's form holds (andi_dev.h: DEEP_SINGLE)
        room = 13 if form == 1 else min(4, 16 - K)
        pos = x[once]
        nval, ext = (y[once] >> 2) & 15, y[once] >> 6
        want_n = np.zeros(len(pos), np.int64)
        want_e = np.zeros(len(pos), np.int64)
        alive = np.ones(len(pos), bool)
        tt = np.concatenate([t, np.full(16, 4, np.int64)])
        for j in range(room):
            sym = tt[pos + K + j]
            alive &= sym < 4
            want_n += alive
            want_e |= np.where(alive, sym, 0) << (2 * j)
        assert (nval == want_n).all() and (ext == want_e).all(), name
    many = occurs > 1
    assert ((y[many] >> 8) + 1 == occurs[many]).all()
    first = np.full(4 ** K, n, np.int64)  # smallest suffix-array index among the positions of each K-mer
    full = np.nonzero(ok)[0]
    np.minimum.at(first, pref[full], rank[full])
    assert (x[many] == first[many]).all()
    absent = occurs == 0
    best = np.zeros(4 ** K, np.int64)
    cnt = np.full(4 ** K, n, np.int64)  # (the empty prefix: every suffix shares it)
    one = np.zeros(4 ** K, np.int64)
    for l in range(1, K):
        c = codes >> (2 * (K - l))
        hit = count[l][c] > 0
        best[hit], cnt[hit], one[hit] = l, count[l][c][hit], where[l][c][hit]
    assert ((y[absent] >> 8) == best[absent]).all()
    uniq = (cnt == 1) & absent
    assert ((((y >> 2) & 1) == 1)[absent] == uniq[absent]).all()
    assert (x[uniq] == rank[one[uniq]]).all()
    E.close()
