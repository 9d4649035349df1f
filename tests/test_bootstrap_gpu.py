"""K8, pairwise bootstrap (src/process.c:289-321, src/model.c:222-232).  The
reference draws from GSL seeded by the clock and has no test for it: parity is
unpinned, so the properties of the distribution are checked instead."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _matrix(rng, n):
    M = np.zeros((n, n, 17), np.uint32)
    for i in range(n):
        for j in range(n):
            if i == j:
                M[i, j, 0] = M[i, j, 16] = 9
                continue
            total = int(rng.integers(2000, 3_000_000))
            p = rng.dirichlet(np.r_[np.full(4, 40.0), np.full(12, 0.6)])
            c = rng.multinomial(total, p)
            M[i, j, [0, 5, 10, 15]] = c[:4]
            M[i, j, [1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13, 14]] = c[4:]
            M[i, j, 16] = total + int(rng.integers(0, 1000))
    return M


def test_bootstrap_structure_and_determinism(ctx):
    import andi_amd
    rng = np.random.default_rng(5)
    n, reps = 6, 40
    M = _matrix(rng, n)
    M[2, 3, :16] = M[3, 2, :16] = 0  # an empty pair stays empty
    B = andi_amd.bootstrap(ctx, M, reps, seed=123)
    assert B.shape == (reps, n, n, 17)
    summed = M.astype(np.uint64) + M.transpose(1, 0, 2)
    for i in range(n):
        assert (B[:, i, i, 0] == 1).all() and (B[:, i, i, 16] == 1).all() and (B[:, i, i, 1:16] == 0).all()
        for j in range(i + 1, n):
            assert (B[:, i, j] == B[:, j, i]).all()  # mirrored, src/process.c:314
            assert (B[:, i, j, :16].sum(axis=1) == summed[i, j, :16].sum()).all()  # N preserved
            assert (B[:, i, j, 16] == summed[i, j, 16]).all()  # seq_len is summed, src/model.c:44
            zero = summed[i, j, :16] == 0
            assert (B[:, i, j, :16][:, zero] == 0).all()  # empty cells stay empty
    assert (B == andi_amd.bootstrap(ctx, M, reps, seed=123)).all()
    assert (B != andi_amd.bootstrap(ctx, M, reps, seed=124)).any()
    # replicate r does not depend on how many replicates are drawn
    assert (andi_amd.bootstrap(ctx, M, 7, seed=123) == B[:7]).all()


def test_bootstrap_moments(ctx):
    """Per cell, mean = N p and variance = N p (1 - p) over many replicates; the
    JC distances of the replicates are centred on the point estimate."""
    import andi_amd
    rng = np.random.default_rng(8)
    n, reps = 3, 4000
    M = _matrix(rng, n)
    M[0, 1, :16] = M[1, 0, :16] = 0
    M[0, 1, [0, 5, 10, 15]] = [30, 7, 2, 1]  # tiny counts: the waiting-time branch
    M[1, 0, 1] = 3
    B = andi_amd.bootstrap(ctx, M, reps, seed=99).astype(np.float64)
    for i, j in ((0, 1), (0, 2), (1, 2)):
        c = (M[i, j, :16].astype(np.float64) + M[j, i, :16])
        N = c.sum()
        p = c / N
        mean, var = B[:, i, j, :16].mean(axis=0), B[:, i, j, :16].var(axis=0)
        se_mean = np.sqrt(N * p * (1 - p) / reps) + 1e-9
        assert (np.abs(mean - N * p) <= 5 * se_mean + 1e-6).all(), (i, j)
        big = N * p * (1 - p) > 5
        assert (np.abs(var[big] / (N * p * (1 - p))[big] - 1) < 0.15).all(), (i, j)
        # covariance sign: multinomial cells are negatively correlated
        cov = np.cov(B[:, i, j, 0], B[:, i, j, 5])[0, 1]
        assert cov < 0
    point = andi_amd.estimate(M[0, 2].astype(np.uint64) + M[2, 0], andi_amd.M_JC)
    reps_jc = np.array([andi_amd.estimate(B[r, 0, 2].astype(np.uint32), andi_amd.M_JC) for r in range(0, reps, 10)])
    assert abs(reps_jc.mean() - point) < 4 * reps_jc.std() / np.sqrt(len(reps_jc)) + 1e-6


def test_bootstrap_cells_chi_square_against_the_oracle(ctx, orc):
    """The device's draws against the oracle's restatement of gsl_ran_multinomial (oracle/andi_oracle.c, model.c:222-232)
    and against the exact law: 10^4 draws of one pair's sixteen cells each way, tiny counts (the waiting-time branch of the
    device's binomial) and counts of 10^5..10^6 (its BTRS branch).  Per cell and per sum of cells: chi-square goodness of fit
    against Binomial(N, p) and a two-sample chi-square device vs oracle.  The generators differ (Philox streams per pair
    and replicate / splitmix64; the reference's is clock-seeded GSL), so this is the strongest comparison there is:
    parity of the bootstrap stays "unpinned", its distribution is held."""
    import andi_amd
    from conftest import binomial_gof_pvalue, two_sample_pvalue
    reps = 10000
    M = np.zeros((3, 3, 17), np.uint32)
    M[0, 1, :16] = [30, 0, 1, 0, 0, 7, 0, 0, 0, 2, 2, 0, 1, 0, 0, 1]
    M[1, 0, :16] = [0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    M[0, 2, :16] = [300000, 50, 700, 30, 60, 250000, 20, 10, 5, 3, 280000, 40, 9, 1, 0, 270000]
    M[2, 0, :16] = [290000, 45, 650, 33, 70, 255000, 25, 12, 4, 2, 275000, 35, 11, 0, 0, 265000]
    M[1, 2, :16] = [900, 11, 12, 13, 14, 800, 15, 16, 17, 18, 700, 19, 20, 21, 22, 600]  # both branches in one pair
    M[:, :, 16] = 1234
    G = andi_amd.bootstrap(ctx, M, reps, seed=2026)
    O = orc.bootstrap(M, reps, seed=2026)
    worst = 1.0
    for i, j in ((0, 1), (0, 2), (1, 2)):
        c = M[i, j, :16].astype(np.int64) + M[j, i, :16]
        N = int(c.sum())
        g, o = G[:, i, j, :16].astype(np.int64), O[:, i, j, :16].astype(np.int64)
        assert (g.sum(axis=1) == N).all() and (o.sum(axis=1) == N).all()
        assert (g[:, c == 0] == 0).all() and (G[:, i, j, 16] == O[:, i, j, 16]).all()
        groups = [(k,) for k in np.nonzero(c)[0]] + [(0, 5), (0, 5, 10, 15), (1, 2, 3, 4), (10, 15), (6, 7, 8, 9, 11)]
        for cells in groups:
            xs, ys, pc = g[:, list(cells)].sum(axis=1), o[:, list(cells)].sum(axis=1), c[list(cells)].sum() / N
            if pc in (0.0, 1.0):
                continue
            p1, p2, p3 = binomial_gof_pvalue(xs, N, pc), binomial_gof_pvalue(ys, N, pc), two_sample_pvalue(xs, ys)
            worst = min(worst, p1, p3)
            assert p1 > 1e-5, ("device vs Binomial", i, j, cells, p1)
            assert p2 > 1e-5, ("oracle vs Binomial", i, j, cells, p2)
            assert p3 > 1e-5, ("device vs oracle", i, j, cells, p3)
        # the cells' covariance matrix: -N p_a p_b off the diagonal (multinomial), within sampling error of the oracle's
        big = np.nonzero(c * (N - c) / N > 50)[0]
        if len(big) >= 2:
            cg, co = np.cov(g[:, big].T), np.cov(o[:, big].T)
            theory = -np.outer(c[big], c[big]) / N + np.diag(c[big].astype(np.float64))
            scale = np.sqrt(np.outer(np.diag(theory), np.diag(theory)))
            assert np.abs((cg - theory) / scale).max() < 0.08 and np.abs((co - theory) / scale).max() < 0.08
    print("smallest p-value over all cells and groups: %.3g" % worst)
