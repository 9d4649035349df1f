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
