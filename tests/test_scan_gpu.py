"""Anchor scan (dist_anchor, K5-K7) on the device against the oracle and the
golden vectors: bit-exact 17 x u32 per ordered pair, through the C-ABI."""
import numpy as np
import pytest

from conftest import rand_dna

pytestmark = pytest.mark.gpu


_SA_MODE = None  # None: the subjects' suffix arrays by the host sorter; "device": built on the device (the seam's default)


def _gpu_rows(ctx, seqs, subjects=None, model=1, segment=0, p_value=0.025):
    import andi_amd
    subjects = list(range(len(seqs))) if subjects is None else subjects
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, seqs[i], p_value, sa=_SA_MODE) for i in subjects]
    out = andi_amd.scan_rows(ctx, esas, subjects, Q, model, segment)
    t = ctx.timings()
    for e in esas:
        e.close()
    Q.close()
    return out, t


def _check_set(ctx, orc, seqs, segments=(0,), model=1):
    want = orc.dist_matrix(seqs, model=model, threads=4)
    for seg in segments:
        got, _ = _gpu_rows(ctx, seqs, model=model, segment=seg)
        assert got.shape == want.shape
        bad = np.argwhere((got != want).any(axis=2))
        assert len(bad) == 0, (seg, bad[:5], got[tuple(bad[0])] if len(bad) else None,
                               want[tuple(bad[0])] if len(bad) else None)


@pytest.fixture(autouse=True)
def _lane_scan(monkeypatch):
    """This module is about the lane scan (scan_lane.hip, scan.hip): ANDI_COOP=0 keeps the calls of its small sets away
    from the wavefront kernel, which the engine would otherwise choose for them (tests/test_coop_gpu.py, test_configs_gpu.py
    and the fuzz test run the engine's own choice)."""
    import os
    from andi_amd.lib import reload_knobs
    if not os.environ.get("ANDI_TESTS_ENGINE_CHOICE"):  # (set: the module's sets through the engine's own choice instead -- parity holds either way)
        monkeypatch.setenv("ANDI_COOP", "0")
        reload_knobs()
    yield


def test_pair_ladder(ctx, orc):
    """test/test_random.sh's divergence ladder, counts bit-exact instead of +-5.5 %."""
    from andi_amd import synth
    for d in (0.0, 0.001, 0.01, 0.02, 0.05, 0.1, 0.2, 0.3):
        a, b = synth.pair(100000, d, seed=100 + int(d * 1000))
        _check_set(ctx, orc, [a, b], segments=(0, 4096))


def test_tiny_segments_stress_stitching(ctx, orc):
    """Segments far shorter than the resynchronisation distance force the
    fix-up path of pass C; the counts must not change."""
    from andi_amd import synth
    a, b = synth.pair(60000, 0.03, seed=77)
    c = synth.to_bytes(synth.mutate_codes(synth.base_codes(60000, 77), 0.002, 9))
    want = orc.dist_matrix([a, b, c], threads=3)
    for seg in (64, 300, 1000, 1 << 20):
        got, t = _gpu_rows(ctx, [a, b, c], segment=seg)
        assert (got == want).all(), seg


def test_identical_unrelated_short(ctx, orc):
    rng = np.random.default_rng(31)
    g = rand_dna(rng, 50000)
    # identical sequences (src/process.c:199-203), unrelated (test/nan.sh), low homology (test/low_homo.sh)
    h = rand_dna(rng, 50000)
    low = g[:100] + rand_dna(rng, 49900)
    _check_set(ctx, orc, [g, g, h, low], segments=(0, 1024))
    # sequences shorter than 1000 nt only warn in the reference (src/andi.c:306-316)
    _check_set(ctx, orc, [rand_dna(rng, 40), rand_dna(rng, 11), rand_dna(rng, 10), rand_dna(rng, 300), b"ACGT"],
               segments=(0, 16))


def test_join_mode_and_revcomp(ctx, orc):
    """'!' separators (test/test_join.sh) and a query on the reverse strand."""
    from andi_amd import synth
    base = synth.base_codes(150000, 5)
    a = synth.join_contigs(synth.to_bytes(base), 3, seed=1)
    b = synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.1, 6, raw=True)), 2, seed=2)
    c = synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.1, 7, raw=True)), 40, seed=3)
    rc = synth.to_bytes(synth.mutate_codes(base, 0.05, 8))[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    _check_set(ctx, orc, [a, b, c, rc], segments=(0, 2048))


def _separator_spanning_subject(rng):
    """3 contigs x 3 kbp whose junctions make a 10-mer table entry span a
    separator: a 7-mer that occurs only right before
    two '!T' junctions."""
    while True:
        c = [bytearray(rand_dna(rng, 3000)) for _ in range(3)]
        w = b"GACCGGA"
        c[0][-7:] = w
        c[1][-7:] = w
        c[1][0:2] = b"TA"
        c[2][0:2] = b"TC"
        seq = b"!".join(bytes(x) for x in c)
        rc = seq[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        if seq.count(w) == 2 and rc.count(w) == 0:
            return seq


def test_separator_spanning_table_entry(ctx, orc):
    """The reference AS RUN uses the cached lookup (src/process.c:117), which on
    such a subject is not the true longest match; the device must follow it
    (the probe table is switched off by the build kernel's flag)."""
    rng = np.random.default_rng(57)
    subj = _separator_spanning_subject(rng)
    O = orc.OracleEsa(subj)
    q = b"GACCGGAATGCGTCAGATGA"
    assert O.get_match(q, True) != O.get_match(q, False)  # the quirk is really triggered
    queries = [rand_dna(rng, 2000) + q + rand_dna(rng, 3000) + b"GACCGGATTTT" + rand_dna(rng, 500),
               subj.replace(b"!", b"")[:7000]]
    import andi_amd
    Q = andi_amd.Queries(ctx, queries)
    E = andi_amd.Esa(ctx, subj)
    assert E.flags()[0] == 1  # the index build noticed the closed "w!" run
    ctx.timings_reset()
    got = andi_amd.scan_rows(ctx, [E], [-1], Q, andi_amd.M_JC, 512)
    assert ctx.timings()["reference_subjects"] == 1  # ... and the scan followed the reference walk
    assert E.flags()[2] == 1  # the 10-mer table kernel really hit the separator branch
    for k, qq in enumerate(queries):
        assert (got[0, k] == O.dist_anchor(qq)).all(), k
    E.close()
    Q.close()


def test_mixed_modes_in_one_scan_call(ctx, orc):
    """One andi_hip_scan_rows call over subjects that use the probe table and one
    that needs the reference walk (two kernel variants write disjoint rows)."""
    import andi_amd
    from andi_amd import synth
    rng = np.random.default_rng(58)
    flagged = _separator_spanning_subject(rng)
    base = synth.base_codes(20000, 2)
    seqs = [synth.to_bytes(base), flagged, synth.to_bytes(synth.mutate_codes(base, 0.04, 3)),
            flagged.replace(b"!", b"")[:8000]]
    want = orc.dist_matrix(seqs, threads=4)
    ctx.timings_reset()
    got, t = _gpu_rows(ctx, seqs, segment=900)
    assert t["reference_subjects"] == 1
    assert (got == want).all()


def test_probe_table_depths_and_reference_walk_agree(ctx, orc, knob):
    """Every probe-table depth K and the reference walk give the same counts."""
    import andi_amd
    from andi_amd import synth
    rng = np.random.default_rng(91)
    base = synth.base_codes(30000, 3)
    seqs = [synth.to_bytes(base), synth.to_bytes(synth.mutate_codes(base, 0.06, 4)),
            synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.02, 5)), 9, seed=4),
            rand_dna(rng, 9000, b"AC"), rand_dna(rng, 500) * 30]
    want = orc.dist_matrix(seqs, threads=4)
    for K in ("4", "5", "7", "9", "11", "13"):
        knob("ANDI_DEEP_K", K)
        got, t = _gpu_rows(ctx, seqs, segment=700)
        assert (got == want).all(), K
    knob("ANDI_DEEP_K", None)
    knob("ANDI_FORCE_REFERENCE", "1")
    ctx.timings_reset()
    got, t = _gpu_rows(ctx, seqs, segment=700)
    assert (got == want).all() and t["reference_subjects"] == len(seqs)


def test_repeats_and_models(ctx, orc):
    rng = np.random.default_rng(41)
    unit = rand_dna(rng, 3000)
    from andi_amd import synth
    g = rand_dna(rng, 30000) + unit + rand_dna(rng, 20000) + unit + rand_dna(rng, 10000) + unit
    codes = np.frombuffer(g.translate(bytes.maketrans(b"ACGT", bytes(range(4)))), np.uint8)
    h = synth.to_bytes(synth.mutate_codes(codes, 0.02, 4))
    for model in (0, 1, 2, 3, 4):  # Raw, JC, Kimura; LogDet and ANI count anchors per nucleotide
        _check_set(ctx, orc, [g, h], segments=(0, 512), model=model)


def test_edge_cases_on_device_built_suffix_arrays(ctx, orc, monkeypatch):
    """The edge-case suite once more with the suffix arrays built on the device (sa_device.hip), as the one-call
    seam stages its subjects: identical / unrelated / short sequences, joined contigs and reverse strands, repeats
    under every model."""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "_SA_MODE", "device")
    test_identical_unrelated_short(ctx, orc)
    test_join_mode_and_revcomp(ctx, orc)
    test_repeats_and_models(ctx, orc)


def test_tree_structured_set(ctx, orc):
    """Genomes at the tips of a tree (andi_amd/synth.py: tree_set): pairwise distances from 4.4e-4 to 2.6e-2 in one
    matrix, as the manual reports for real E. coli sets (docs/manual/andi-manual.tex:316-320) -- pairs of very different
    divergence share a call, every segment-length class and both pass A kernels are in use."""
    import andi_amd
    from andi_amd import synth
    seqs, D = synth.tree_set(10, 150000, seed=77)
    _check_set(ctx, orc, seqs, segments=(0, 2048))
    got, _ = _gpu_rows(ctx, seqs)
    for i, j in ((0, 1), (2, 7), (4, 9)):
        d = andi_amd.estimate(np.minimum(got[i, j].astype(np.uint64) + got[j, i], 0xFFFFFFFF).astype(np.uint32), andi_amd.M_JC)
        assert abs(d - D[i, j]) < 0.25 * D[i, j] + 2e-4, (i, j, d, D[i, j])


def test_anchor_significance_parameter(ctx, orc):
    from andi_amd import synth
    a, b = synth.pair(80000, 0.05, seed=12)
    for p in (0.5, 1e-4):
        want = orc.dist_matrix([a, b], p_value=p, threads=2)
        import andi_amd
        Q = andi_amd.Queries(ctx, [a, b])
        esas = [andi_amd.Esa(ctx, s, p) for s in (a, b)]
        got = andi_amd.scan_rows(ctx, esas, [0, 1], Q)
        assert (got == want).all(), p
        for e in esas:
            e.close()
        Q.close()


def test_logdet_ani_equal_runs(ctx, orc):
    """model_count_equal's per-character path (src/model.c:256-278): identical
    sequences, join separators inside anchors, skewed composition."""
    from andi_amd import synth
    rng = np.random.default_rng(17)
    base = synth.base_codes(70000, 21)
    a = synth.to_bytes(base)
    b = synth.join_contigs(synth.to_bytes(synth.mutate_codes(base, 0.01, 22)), 6, seed=5)
    skew = rand_dna(rng, 30000, b"AAAAACGT")
    skew2 = synth.to_bytes(synth.mutate_codes(
        np.frombuffer(skew.translate(bytes.maketrans(b"ACGT", bytes(range(4)))), np.uint8), 0.03, 3))
    for model in (3, 4):
        _check_set(ctx, orc, [a, a, b], segments=(0, 1500), model=model)
        _check_set(ctx, orc, [skew, skew2], segments=(0,), model=model)
    import andi_amd
    with pytest.raises(andi_amd.AndiHipError):
        _gpu_rows(ctx, [a, b], model=7)


def test_golden_testfasta_s42(ctx, golden_s42):
    """Inputs from the reference's generator; expected counts committed in
    tests/golden (the oracle's, itself pinned to the reference's statistics)."""
    import andi_amd
    s0 = golden_s42["s0"]
    E = andi_amd.Esa(ctx, s0)
    assert E.threshold == 14
    for d in ("0.1", "0.01", "0.001"):
        q = golden_s42["s1_" + d]
        Q = andi_amd.Queries(ctx, [q])
        for seg in (0, 5000):
            got = andi_amd.scan_rows(ctx, [E], [-1], Q, andi_amd.M_JC, seg)
            assert (got[0, 0] == golden_s42["counts_" + d]).all(), (d, seg)
        Q.close()
    E.close()


def test_golden_testfasta_s1729_dist_matrix(golden_s1729):
    """The one-call seam (distMatrix) on the 3 x 1 Mbp plumbing case
    (BASELINE.json configs[0]); PHYLIP text equals what andi printed."""
    import andi_amd
    seqs = golden_s1729["seqs"]
    M = andi_amd.dist_matrix(seqs, model=andi_amd.M_JC, host_threads=3)
    assert (M == golden_s1729["counts_jc"]).all()
    text, warn, flags = andi_amd.format_distances(M, ["S0", "S1", "S2"], andi_amd.M_JC)
    rows = [l.split() for l in text.splitlines()[1:]]
    assert [rows[0][2], rows[0][3], rows[1][3]] == ["0.0982", "0.0984", "0.1955"]
    assert flags == 0 and warn == ""


def test_full_size_properties(ctx, orc):
    """BASELINE.json configs[1] genome length (4.9 Mbp): a pair at full size,
    checked through size-independent properties and against the oracle."""
    from andi_amd import synth
    import andi_amd
    n = 4_900_000
    base = synth.base_codes(n, 1729)
    a = synth.to_bytes(synth.mutate_codes(base, 0.004, 1))
    b = synth.to_bytes(synth.mutate_codes(base, 0.02, 2))
    got, t = _gpu_rows(ctx, [a, b], segment=0)
    # diagonal placeholder, seq_len, coverage <= 1, symmetric-ish distances
    assert got[0, 0, 0] == 9 and got[0, 0, 16] == 9 and got[1, 1, 16] == 9
    assert got[0, 1, 16] == n and got[1, 0, 16] == n
    assert got[0, 1, :16].sum() <= n and got[1, 0, :16].sum() <= n
    d01 = andi_amd.estimate(got[0, 1]), andi_amd.estimate(got[1, 0])
    assert abs(d01[0] - 0.024) < 0.0015 and abs(d01[1] - 0.024) < 0.0015
    # independent of the segmentation (idempotence of the stitching)
    got2, _ = _gpu_rows(ctx, [a, b], segment=100_000)
    assert (got == got2).all()
    # and equal to the sequential oracle
    want = orc.dist_matrix([a, b], threads=2)
    assert (got == want).all()


def test_contigs_models_and_segment_lengths(ctx, orc):
    """Passes A/B (one lane per chain on packed symbols, scan_lane.hip) on a set with joined contigs, at two
    segment lengths, and LogDet / ANI (per-nucleotide anchor counts)."""
    from andi_amd import synth
    rng = np.random.default_rng(5)
    a, b = synth.pair(80000, 0.04, seed=3)
    c = synth.to_bytes(synth.mutate_codes(synth.base_codes(80000, 3), 0.003, 11))
    joined = a[:30000] + b"!" + rand_dna(rng, 500) + b"!" + b[30000:60000]  # contigs, src/sequence.c:260-282
    _check_set(ctx, orc, [a, b, c, joined], segments=(0, 777))
    for model in (3, 4):  # LogDet, ANI: per-nucleotide anchor counts
        _check_set(ctx, orc, [a, b, joined], model=model)


def test_bytes_outside_the_alphabet_are_refused(ctx):
    """The engine's contract is the alphabet the reference's reader produces
    (src/sequence.c:260-282: A C G T and '!'); anything else fails loudly instead of
    being silently treated as a nucleotide."""
    import andi_amd
    from andi_amd import synth
    a, b = synth.pair(50000, 0.02, seed=8)
    a2 = bytearray(a)
    a2[7000] = ord("N")
    Qbad = andi_amd.Queries(ctx, [bytes(a2), b])
    Q = andi_amd.Queries(ctx, [a, b])
    Ebad = andi_amd.Esa(ctx, bytes(a2))  # subject text with a foreign byte: noticed by the index build
    E = andi_amd.Esa(ctx, a)
    with pytest.raises(andi_amd.AndiHipError, match="outside"):
        andi_amd.scan_rows(ctx, [E], [-1], Qbad, 1, 0)
    with pytest.raises(andi_amd.AndiHipError, match="outside"):
        andi_amd.scan_rows(ctx, [Ebad], [-1], Q, 1, 0)
    assert andi_amd.scan_rows(ctx, [E], [0], Q, 1, 0).shape == (1, 2, 17)
    for x in (E, Ebad, Q, Qbad):
        x.close()


def test_segment_length_per_pair_changes_nothing(ctx, orc, knob):
    """segment = 0 lets the engine choose a segment length per pair from sampled match
    lengths (scan_lane.hip: k_pair_estimate); pairs of very different divergence in one
    call, every factor, and the uniform layout must all give the oracle's counts."""
    from andi_amd import synth
    base = synth.base_codes(120000, 21)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 30 + k)) for k, d in enumerate((0.0, 0.0005, 0.004, 0.03, 0.12))]
    seqs.append(rand_dna(np.random.default_rng(2), 70000))  # unrelated, and a different length
    want = orc.dist_matrix(seqs, threads=4)
    for env in ({}, {"ANDI_SEG_FACTOR": "1"}, {"ANDI_SEG_FACTOR": "200"}, {"ANDI_SEG0": "128"},
                {"ANDI_UNIFORM_SEGMENTS": "1"}):
        for k in ("ANDI_SEG_FACTOR", "ANDI_SEG0", "ANDI_UNIFORM_SEGMENTS"):
            knob(k, None)
        for k, v in env.items():
            knob(k, v)
        got, t = _gpu_rows(ctx, seqs)
        assert (got == want).all(), env


def test_realistic_divergence_structure(ctx, orc):
    """Genomes as real data sets have them (docs/manual/andi-manual.tex:303-320) and the reference's generator has
    not: multi-copy repeats on both strands, indels, inversions, 5-10 % of unrelated sequence, optionally cut into
    contigs -- the regime in which the scan probes at nearly every step, walks repeated K-mers' occurrences and
    changes diagonal all the time (src/process.c:113-123, src/esa.c:531-601).  Counts bit-exact, every model."""
    from andi_amd import synth
    seqs, ds = synth.realistic_set(6, 250000, 0.001, 0.06, seed=77, novel_fraction=0.07)
    want = orc.dist_matrix(seqs, threads=0)
    cov = want[:, :, :16].sum(axis=2) / np.maximum(want[:, :, 16], 1)
    assert 0.3 < cov[0, 1] < 0.98  # homology is partial, as in real sets
    _check_set(ctx, orc, seqs, segments=(0, 2048, 100))
    for model in (0, 2, 4):
        _check_set(ctx, orc, seqs[:4], model=model)
    joined, _ = synth.realistic_set(4, 120000, 0.003, 0.04, seed=78, contigs=9)
    _check_set(ctx, orc, joined, segments=(0, 700))


def test_device_buffers_survive_churn(ctx, orc):
    """The engine's device buffers are carved out of large chunks (andi_amd/csrc/dev_arena.h: first fit, coalescing,
    shared by the contexts of a device).  Subjects and query sets of very different sizes are created and released
    in interleaved order -- blocks are split, joined and handed out again -- and what is scanned afterwards, and
    in between by a second context, is still bit-exact."""
    import andi_amd
    from andi_amd import synth
    rng = np.random.default_rng(11)
    seqs = [synth.to_bytes(synth.mutate_codes(synth.base_codes(60000, 3), 0.02, 50 + k)) for k in range(4)]
    want = orc.dist_matrix(seqs, threads=4)
    filler = [synth.to_bytes(synth.base_codes(int(n), 100 + i)) for i, n in enumerate(rng.integers(1000, 900000, 14))]
    held = []
    for round_ in range(3):
        for i, f in enumerate(filler):
            held.append(andi_amd.Esa(ctx, f))
            if i % 3 == round_ % 3 and held:
                held.pop(int(rng.integers(0, len(held)))).close()
        Qf = andi_amd.Queries(ctx, filler[: 3 + round_])
        got, _ = _gpu_rows(ctx, seqs)
        assert (got == want).all(), round_
        other = andi_amd.Context(0)  # a second context on the device shares the arena
        got2, _ = _gpu_rows(other, seqs, segment=512)
        other.close()
        assert (got2 == want).all(), round_
        Qf.close()
        for e in held[::2]:
            e.close()
        held = held[1::2]
    for e in held:
        e.close()


def test_fixups_in_pass_c(ctx, orc, knob):
    """Pass C stitches again, one after the other, the segments whose assumed entry state turned out wrong
    (k_scan_reduce): it looks for the failing checks 64 at a time and follows each stretch until the true chain
    enters a segment in the state that was assumed for it.  Without pass B's re-stitch rounds, on genomes with
    repeats and unrelated stretches and with short segments, there are hundreds of such stretches -- isolated ones,
    runs over several segments, the last segment of a query: the counts stay bit-exact and the fix-ups are counted."""
    from andi_amd import synth
    knob("ANDI_NO_RESTITCH", "1")
    seqs, _ = synth.realistic_set(5, 200000, 0.001, 0.06, seed=41, novel_fraction=0.1)
    seqs.append(seqs[1][:150017])  # ends inside a segment
    want = orc.dist_matrix(seqs, threads=4)
    total = 0
    for seg in (0, 64, 300, 2048):
        ctx.timings_reset()
        got, t = _gpu_rows(ctx, seqs, segment=seg)
        assert (got == want).all(), seg
        total += t["fixups"]
    assert total > 100, total


def test_bad_stretches_are_stitched_in_pass_b(ctx, orc):
    """Pass B stitches again, one lane per stretch, the runs of segments that were entered in a state their predecessor's
    true chain did not leave in (scan_lane.hip: k_stitch_heads, stitch_stretch) -- repeats the true chain crosses on
    lucky anchors (src/process.c:95-97), the edges of unrelated stretches.  With segments of 64 ... 2048 nucleotides a
    repeat of 5000 spans up to 80 of them: counts bit-exact, and next to nothing is left for pass C's fix-ups (the same
    sets leave it hundreds without these rounds: test_fixups_in_pass_c)."""
    from andi_amd import synth
    seqs, _ = synth.realistic_set(5, 200000, 0.001, 0.06, seed=41, novel_fraction=0.1)
    seqs.append(seqs[1][:150017])  # ends inside a segment
    want = orc.dist_matrix(seqs, threads=4)
    left = {}
    for seg in (0, 64, 300, 2048):
        ctx.timings_reset()
        got, t = _gpu_rows(ctx, seqs, segment=seg)
        assert (got == want).all(), seg
        left[seg] = t["fixups"]
    print("fix-ups left to pass C by segment length:", left)
    assert left[0] + left[300] + left[2048] <= 20, left  # (segments of 64: stretches run into each other, pass C takes what three rounds leave)


def test_pairs_with_long_matches_take_the_quad_kernel(ctx, orc, knob):
    """With per-pair segment lengths the pairs whose sampled matches are long (>= 128 symbols on average) go through
    k_lane_quad, the others through k_lane_cold, side by side on two streams: a set with both kinds, every threshold."""
    from andi_amd import synth
    base = synth.base_codes(400000, 77)
    seqs = [synth.to_bytes(synth.mutate_codes(base, d, 20 + k)) for k, d in enumerate((0.0, 0.0002, 0.001, 0.004, 0.02, 0.06))]
    seqs.append(synth.join_contigs(seqs[1], 4, seed=9))
    want = orc.dist_matrix(seqs, threads=0)
    for env in ({}, {"ANDI_QUAD_MATCH": "0"}, {"ANDI_QUAD_MATCH": "1000"}, {"ANDI_QUAD_MATCH": "-1"}, {"ANDI_NO_SIDE_STREAM": "1"},
                {"ANDI_QUAD_UNLISTED": "1"}, {"ANDI_QUAD_BLOCKS4": "1"}, {"ANDI_QUAD_MATCH": "0", "ANDI_QUAD_BLOCKS4": "1"}):
        for k in ("ANDI_QUAD_MATCH", "ANDI_NO_SIDE_STREAM", "ANDI_QUAD_UNLISTED", "ANDI_QUAD_BLOCKS4"):
            knob(k, None)
        for k, v in env.items():
            knob(k, v)
        got, t = _gpu_rows(ctx, seqs)
        assert t["adaptive_calls"] >= 1 and (got == want).all(), env
    for k in ("ANDI_QUAD_MATCH", "ANDI_NO_SIDE_STREAM", "ANDI_QUAD_UNLISTED", "ANDI_QUAD_BLOCKS4"):
        knob(k, None)
    for model in (3, 4):  # LogDet, ANI: equal runs counted per nucleotide, in k_lane_quad's single-wavefront blocks too
        want_m = orc.dist_matrix(seqs, model=model, threads=0)
        got, _ = _gpu_rows(ctx, seqs, model=model)
        assert (got == want_m).all(), model


def test_deeper_probe_table_for_many_queries(ctx, orc):
    """andi_hip_ctx_expect_queries: subjects that will meet a thousand queries or more get a probe table one level
    deeper (what andi_hip_dist_matrix asks for on a C4-shaped job).  The table's depth changes, the counts do not."""
    import andi_amd
    from andi_amd import synth
    seqs, _ = synth.realistic_set(4, 120000, 0.001, 0.05, seed=3, novel_fraction=0.05)
    seqs.append(synth.join_contigs(seqs[1], 5, seed=2))
    try:
        ctx.expect_queries(0)
        e = andi_amd.Esa(ctx, seqs[0], sa="device")
        k0, _ = e.download_index()
        e.close()
        ctx.expect_queries(3084)
        e = andi_amd.Esa(ctx, seqs[0], sa="device")
        k1, _ = e.download_index()
        e.close()
        assert k1 == k0 + 1
        _check_set(ctx, orc, seqs, segments=(0, 512))
    finally:
        ctx.expect_queries(0)


def test_matches_longer_than_many_segments(ctx, orc):
    """Genomes a handful of substitutions apart: one match covers hundreds of segments, several wavefronts, several
    blocks.  The cold chain of every covered segment finds it; in k_lane_quad a chain whose comparison has reached its
    segment's end takes the rest from the next segment's first anchor -- the lane to its right, or what the next
    wavefront's first lane has published (scan_lane.hip: first_pub) -- instead of following the match to its end.
    Same counts as the sequential loop, with the shortest segments (every boundary kind) and the usual ones."""
    from andi_amd import synth
    base = synth.base_codes(2500000, 41)
    rng = np.random.default_rng(5)

    def with_snps(k):
        c = base.copy()
        pos = rng.choice(len(c), size=k, replace=False)
        c[pos] = (c[pos] + 1 + rng.integers(0, 3, size=k)) % 4
        return synth.to_bytes(c)

    seqs = [synth.to_bytes(base), with_snps(3), with_snps(40), synth.to_bytes(base), synth.to_bytes(synth.mutate_codes(base, 0.001, 8))]
    want = orc.dist_matrix(seqs, threads=0)
    got, t = _gpu_rows(ctx, seqs)
    assert t["adaptive_calls"] >= 1 and (got == want).all()
    got, _ = _gpu_rows(ctx, seqs[:3], segment=1024)  # (one segment length for the call: k_lane_cold follows them itself)
    assert (got == want[:3, :3]).all()
