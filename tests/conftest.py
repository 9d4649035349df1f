import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The suite drives the engine through its experiment switches and test hooks (forced kernels, layouts, segment lengths:
# andi_amd/csrc/knobs.h); the shipped library does not have them.  So the suite loads libandihip_test.so -- the same
# sources with -DANDI_TEST_HOOKS -- unless told otherwise (ANDI_HIP_LIB; ANDI_TESTS_SHIPPED_LIB=1: the shipped library --
# tests/test_configs_gpu.py::test_whole_suite_against_the_shipped_library runs EVERY -m gpu test that way, and a test is
# skipped only where it asks for a hook the shipped library lacks: needs_hooks() below).
if not os.environ.get("ANDI_HIP_LIB") and not os.environ.get("ANDI_TESTS_SHIPPED_LIB"):
    os.environ["ANDI_HIP_LIB"] = os.path.join(ROOT, "andi_amd", "libandihip_test.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is built in-tree and git-ignored: build it if this checkout has none
    (hipcc cross-compiles without a GPU; nothing here ever falls back to a CPU path)."""
    so = os.path.join(ROOT, "andi_amd", "libandihip.so")
    if not os.path.exists(so) or not os.path.exists(os.path.join(ROOT, "andi_amd", "libandihip_test.so")):
        import __graft_entry__
        __graft_entry__.build()


def unpack(packed, length):
    """2-bit packed golden sequence -> ACGT bytes."""
    p = np.asarray(packed, np.uint8)
    codes = np.stack([(p >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)[:length]
    return np.frombuffer(b"ACGT", np.uint8)[codes].tobytes()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def golden_s42():
    z = np.load(os.path.join(GOLDEN, "testfasta_s42.npz"))
    n = int(z["length"][0])
    out = {"s0": unpack(z["s0"], n)}
    for d in ("0.1", "0.01", "0.001"):
        out["s1_" + d] = unpack(z["s1_" + d], n)
        out["counts_" + d] = z["counts_" + d]
        out["stats_" + d] = z["stats_" + d]
    return out


@pytest.fixture(scope="session")
def golden_s1729():
    z = np.load(os.path.join(GOLDEN, "testfasta_s1729.npz"))
    n = int(z["length"][0])
    return {"seqs": [unpack(z["s%d" % k], n) for k in range(3)], "counts_jc": z["counts_jc"]}


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session (fails loudly without a GPU)."""
    import andi_amd
    c = andi_amd.Context(0)
    yield c
    c.close()


def verify_suffix_array(text: bytes, sa) -> None:
    """O(n) proof that `sa` is THE suffix array of `text` (unsigned byte order, the terminator excluded, src/esa.c:294-304):
    a permutation of 0..n-1, and every suffix smaller than its successor -- by the first byte, or, the first bytes being
    equal, by the ranks of the two suffixes one position on (the empty suffix ranks lowest).  With this the oracle may be
    fed the product's suffix array and still be an independent check of everything built on it."""
    n = len(text)
    sa = np.asarray(sa, dtype=np.int64)
    assert sa.shape == (n,)
    seen = np.zeros(n, dtype=bool)
    assert sa.min() >= 0 and sa.max() < n
    seen[sa] = True
    assert seen.all(), "not a permutation"
    rank = np.empty(n + 1, dtype=np.int64)
    rank[sa] = np.arange(n)
    rank[n] = -1  # the empty suffix
    t = np.frombuffer(text, dtype=np.uint8)
    a, b = sa[:-1], sa[1:]
    ta, tb = t[a], t[b]
    ok = (ta < tb) | ((ta == tb) & (rank[a + 1] < rank[b + 1]))
    assert ok.all(), "suffixes out of order at ranks %s" % np.nonzero(~ok)[0][:5].tolist()


def rand_dna(rng, n, alphabet=b"ACGT"):
    return rng.choice(np.frombuffer(alphabet, np.uint8), n).tobytes()


# ---------------------------------------------------------------- the library's environment switches
# libandihip.so reads its ANDI_* switches once (andi_amd/csrc/knobs.h); a test that changes one tells it to look again.
SHIPPED_LIB = bool(os.environ.get("ANDI_TESTS_SHIPPED_LIB")) and not os.environ.get("ANDI_HIP_LIB")


def _shipped_knob_names():
    import re
    text = open(os.path.join(ROOT, "andi_amd", "csrc", "knobs.h")).read()
    line = text.split("#define ANDI_KNOB_LIST_SHIPPED(X)")[1].split("\n")[0]
    return set(re.findall(r"X\((\w+)\)", line))


_SHIPPED_KNOBS = _shipped_knob_names()


def needs_hooks(name, value):
    """With ANDI_TESTS_SHIPPED_LIB=1 the suite runs against libandihip.so itself, which knows only the shipped switches:
    a test that asks for one of the test hooks is skipped at that point (what it ran before that point ran on the shipped
    library); a test that sets none, or only shipped switches, runs whole."""
    name = name[len("ANDI_"):] if name.startswith("ANDI_") else name
    if SHIPPED_LIB and value is not None and name not in _SHIPPED_KNOBS:
        pytest.skip("ANDI_%s is a test hook: not in the shipped library" % name)


def reload_knobs():
    from andi_amd import lib
    if lib._lib is not None:
        lib.reload_knobs()


import contextlib


@contextlib.contextmanager
def knobs(**kv):
    """ANDI_<NAME>=value for the block (None: unset), e.g. knobs(COOP=4); the library looks again on entry and exit."""
    for k, v in kv.items():
        needs_hooks(k, v)
    old = {k: os.environ.get("ANDI_" + k) for k in kv}
    try:
        for k, v in kv.items():
            if v is None:
                os.environ.pop("ANDI_" + k, None)
            else:
                os.environ["ANDI_" + k] = str(v)
        reload_knobs()
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop("ANDI_" + k, None)
            else:
                os.environ["ANDI_" + k] = v
        reload_knobs()


@pytest.fixture
def knob(monkeypatch):
    """knob("ANDI_X", "1") / knob("ANDI_X", None): monkeypatch.setenv / delenv and the library looks again (and once more
    when the test is over, after monkeypatch has put the environment back: _knobs_restored below)."""
    def set_(name, value):
        needs_hooks(name, value)
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
        reload_knobs()
    return set_


@pytest.fixture(autouse=True)
def _knobs_restored():
    yield
    reload_knobs()  # (set up before any fixture the test asks for, so torn down after monkeypatch's undo)


# ---------------------------------------------------------------- distribution checks of the bootstrap (parity unpinned)
def binomial_gof_pvalue(x, N, p, bins=24):
    """chi-square goodness of fit of integer draws x against Binomial(N, p): cells = quantile bins of the exact law with
    expected counts >= 8 (fewer bins for narrow laws)."""
    from scipy import stats
    x = np.asarray(x, dtype=np.int64)
    law = stats.binom(int(N), float(p))
    qs = np.unique(law.ppf(np.linspace(0, 1, bins + 1)[1:-1]).astype(np.int64))
    edges = np.concatenate(([-1], qs, [int(N)]))
    edges = np.unique(edges)
    obs = np.histogram(x, bins=edges + 0.5)[0].astype(np.float64)
    exp = np.diff(law.cdf(edges)) * len(x)
    keep = exp > 0
    obs, exp = obs[keep], exp[keep]
    while len(exp) > 2 and exp.min() < 8:  # merge the thinnest cell into a neighbour
        k = int(exp.argmin())
        j = k - 1 if k > 0 else 1
        exp[j] += exp[k]; obs[j] += obs[k]
        exp, obs = np.delete(exp, k), np.delete(obs, k)
    if len(exp) < 2:
        return 1.0 if obs.sum() == len(x) else 0.0
    return float(stats.chisquare(obs, exp * obs.sum() / exp.sum()).pvalue)


def two_sample_pvalue(x, y, bins=24):
    """chi-square test that integer samples x and y come from one law (cells = pooled quantiles)."""
    from scipy import stats
    x, y = np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64)
    edges = np.unique(np.quantile(np.concatenate([x, y]), np.linspace(0, 1, bins + 1)[1:-1]).astype(np.int64))
    edges = np.concatenate(([min(x.min(), y.min()) - 1], edges, [max(x.max(), y.max())])) + 0.5
    edges = np.unique(edges)
    a, b = np.histogram(x, edges)[0], np.histogram(y, edges)[0]
    keep = (a + b) > 0
    if keep.sum() < 2:
        return 1.0
    return float(stats.chi2_contingency(np.stack([a[keep], b[keep]]))[1])
