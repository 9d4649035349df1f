"""Host side of the product (libandihip.so) against the oracle; C-ABI surface.
No GPU needed: nothing here launches a kernel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, rand_dna


def test_library_exports_every_declared_symbol():
    from andi_amd import lib
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "andi_hip.h")).read()
    declared = set(re.findall(r"\b(andi_hip_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    # the suite's library carries the test hooks (tests/conftest.py); the SHIPPED one exports the same symbols -- and knows none
    # of the hooks' names (andi_amd/csrc/knobs.h: seven switches)
    shipped_path = os.path.join(ROOT, "andi_amd", "libandihip.so")
    shipped = C.CDLL(shipped_path)
    for name in declared:
        assert getattr(shipped, name) is not None
    blob = open(shipped_path, "rb").read()
    names = set(m.decode() for m in re.findall(rb"ANDI_[A-Z0-9_]{3,}", blob))
    knobs_h = open(os.path.join(ROOT, "andi_amd", "csrc", "knobs.h")).read()
    listed = set("ANDI_" + n for n in re.findall(r"X\((\w+)\)", knobs_h.split("#define ANDI_KNOB_LIST_HOOKS")[0].split("#define ANDI_KNOB_LIST_SHIPPED(X)")[1]))
    hooks = set("ANDI_" + n for n in re.findall(r"X\((\w+)\)", knobs_h.split("#define ANDI_KNOB_LIST_HOOKS(X)")[1].split("#define ANDI_KNOB_LIST(X)")[0]))
    assert len(listed) <= 8 and listed <= names, (listed, listed - names)
    assert hooks and not (hooks & names), hooks & names
    assert L.andi_hip_abi_version() == 5  # 2: opts.num_gpus, opts.devices; timings.adaptive_calls.  3: timings.routed_calls ...; andi_hip_esa_single_form.  4: andi_hip_trim (chunks outlive contexts).  5: timings.pool_calls
    assert C.sizeof(lib.Model) == 68 and C.sizeof(lib.Interval) == 16


def test_no_gpu_means_loud_failure():
    import torch
    from andi_amd import lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.AndiHipError):
        lib.Context(0)
    with pytest.raises(lib.AndiHipError):
        lib.dist_matrix([b"ACGTACGTACGT", b"ACGTACGAACGT"])


def test_subject_prepare_matches_oracle(orc):
    from andi_amd import lib
    rng = np.random.default_rng(3)
    assert lib.subject_prepare(b"ACGTTGCA")[0] == b"TGCAACGT#ACGTTGCA"  # test/test_seq.c:34
    assert lib.subject_prepare(b"ACGT!TGCA")[0] == b"TGCA;ACGT#ACGT!TGCA"  # test/test_seq.c:69
    for trial in range(60):
        n = int(rng.integers(1, 5000))
        s = rand_dna(rng, n, [b"ACGT", b"AC", b"GGC", b"ACGT!"][trial % 4])
        if not s.replace(b"!", b""):
            continue
        RS, gc, thr = lib.subject_prepare(s, 0.025)
        E = orc.OracleEsa(s)
        assert RS == E.RS and gc == E.gc and thr == E.threshold
    for p in (0.5, 0.1, 0.025, 1e-3, 1e-6):
        for g in (0.2, 0.5, 0.7):
            for l in (100, 20001, 9800001):
                assert lib.min_anchor_length(p, g, l) == orc.lib().orc_min_anchor_length(p, g, l)
                x = lib.min_anchor_length(p, g, l)
                assert lib.shustring_cum_prob(x, g / 2, l) == orc.lib().orc_shustring_cum_prob(x, g / 2, l)


def test_threshold_minimal():
    # test/test_process.c:16-29 against the product's host code
    from andi_amd import lib
    thr = lib.min_anchor_length(0.025, 0.5, 100000)
    assert 0.975 < lib.shustring_cum_prob(thr + 1, 0.25, 100000)
    assert 0.975 <= lib.shustring_cum_prob(thr, 0.25, 100000)
    assert 0.975 > lib.shustring_cum_prob(thr - 1, 0.25, 100000)


def test_suffix_array_matches_oracle_sorter(orc):
    from andi_amd import lib
    rng = np.random.default_rng(5)
    cases = [b"A", b"AA", b"AAAAAAAAAAAAAAAA", b"ACGT", b"TGCA", b"ABABABABAB", b"banana", b"mississippi"]
    for trial in range(80):
        n = int(rng.integers(1, 3000))
        cases.append(rand_dna(rng, n, [b"ACGT", b"AC", b"A", b"ACGT!#;"][trial % 4]))
    cases.append(rand_dna(rng, 50, b"ACGT") * 40)  # long repeats
    cases.append(lib.subject_prepare(rand_dna(rng, 200000, b"ACGT"))[0])
    for t in cases:
        assert (lib.suffix_array(t) == orc.suffix_array(t)).all()


def test_estimators_and_printer_match_oracle(orc):
    from andi_amd import lib
    rng = np.random.default_rng(9)
    n = 5
    M = rng.integers(0, 2000, size=(n, n, 17), dtype=np.uint32)
    M[:, :, [0, 5, 10, 15]] += 40000
    M[:, :, 16] = 200000
    M[1, 2, :16] = 0  # NaN pair in one direction only
    M[3, 4, :16] //= 100  # low coverage
    for model in (lib.M_RAW, lib.M_JC, lib.M_KIMURA, lib.M_LOGDET, lib.M_ANI):
        for i in range(n):
            for j in range(n):
                a, b = lib.estimate(M[i, j], model), orc.estimate(M[i, j], model)
                assert (np.isnan(a) and np.isnan(b)) or a == b
    assert lib.coverage(M[0, 1]) == orc.coverage(M[0, 1])
    names = ["S%d" % k for k in range(n - 1)] + ["a_rather_long_name"]
    text, warn, flags = lib.format_distances(M, names, lib.M_JC)
    lines = text.splitlines()
    assert lines[0] == str(n) and len(lines) == n + 1
    assert lines[1].startswith("S0         0.0000 ")
    assert lines[n].startswith("a_rather_long_name ")
    avg = M[0, 1].astype(np.uint64) + M[1, 0]
    assert lines[1].split()[2] == "%1.4f" % orc.estimate(avg, orc.M_JC)
    assert "very little homology" in warn and flags & 2
    text_t, _, _ = lib.format_distances(M, names, lib.M_JC, truncate_names=True)
    assert text_t.splitlines()[n].startswith("a_rather_l ")
    # scientific notation as soon as one distance is in (0, 0.001)  (src/io.c:280-282)
    M2 = M.copy()
    M2[0, 1, :16] = M2[1, 0, :16] = 0
    M2[0, 1, [0, 5, 10, 15]] = M2[1, 0, [0, 5, 10, 15]] = 50000
    M2[0, 1, 1] = 3
    text2, _, _ = lib.format_distances(M2, names, lib.M_JC)
    assert "e-0" in text2.splitlines()[1]
    # all-zero counts -> nan + warning (src/io.c:284-291)
    M3 = M.copy()
    M3[1, 2, :16] = M3[2, 1, :16] = 0
    text3, warn3, flags3 = lib.format_distances(M3, names, lib.M_JC)
    assert "nan" in text3 and "reported as nan" in warn3 and flags3 & 1


def test_suffix_array_verifier():
    """tests/conftest.py: verify_suffix_array accepts the suffix array and nothing else (it is what lets the oracle run on the
    product's device-built array at 2-50 Mbp and stay an independent check)."""
    import numpy as np
    from conftest import verify_suffix_array
    from oracle import orc
    rng = np.random.default_rng(5)
    for text in (b"TGCAACGT#ACGTTGCA", rng.choice(np.frombuffer(b"ACGT", np.uint8), 5000).tobytes() + b"#" + b"A" * 300 + b";CCGT"):
        sa = orc.suffix_array(text)
        verify_suffix_array(text, sa)
        bad = sa.copy()
        bad[[3, 4]] = bad[[4, 3]]
        with pytest.raises(AssertionError):
            verify_suffix_array(text, bad)
        dup = sa.copy()
        dup[7] = dup[8]
        with pytest.raises(AssertionError):
            verify_suffix_array(text, dup)


def test_tree_structured_generator():
    """andi_amd/synth.py: tree_set -- the observed mismatch fractions follow the tree's pairwise distances, which span
    d_min ... d_max (docs/manual/andi-manual.tex:316-320)."""
    import math
    import numpy as np
    from andi_amd import synth
    seqs, D = synth.tree_set(12, 300000, seed=3)
    iu = np.triu_indices(12, 1)
    assert abs(D[iu].max() - 2.6e-2) < 1e-9 and 4.4e-4 - 1e-9 <= D[iu].min() < 2.6e-3 and (D == D.T).all()
    a = [np.frombuffer(x, np.uint8) for x in seqs]
    for i, j in zip(*iu):
        p = float((a[i] != a[j]).mean())
        want = 0.75 - 0.75 * math.exp(-4.0 * D[i, j] / 3.0)
        assert abs(p - want) < 0.15 * want + 1.5e-4, (i, j, p, want)


def _patch_text():
    with open(os.path.join(ROOT, "integration", "andi-hip.patch")) as f:
        return f.read()


def test_integration_snippet_compiles(tmp_path):
    """integration/andi-hip.patch: the binding a maintainer adds to src/process.c, compiled (syntax and types) against
    include/andi_hip.h with stand-ins for what the reference's headers give it (struct model, seq_t and the globals,
    src/model.h:52-57, src/sequence.h:18-29, src/global.h) -- the layout assertion of the snippet holds there too."""
    import subprocess
    added = []
    take = False
    for line in _patch_text().splitlines():
        if line.startswith("+++ ") and "process.c" in line:
            take = True
        elif line.startswith("+++ "):
            take = False
        elif take and line.startswith("+") and not line.startswith("+++"):
            added.append(line[1:])
    body = "\n".join(added)
    assert "distMatrixHIP" in body and "andi_hip_dist_matrix" in body
    src = r"""
#define HAVE_ANDI_HIP 1
#include <stdlib.h>
#include <stddef.h>
#include <err.h>
struct model { unsigned int counts[16]; unsigned int seq_len; };      /* src/model.h:52-57 */
typedef struct { char *S, *name; size_t len; double gc; } seq_t;      /* the fields the binding reads, src/sequence.h */
extern double ANCHOR_P_VALUE; extern int MODEL, THREADS, FLAGS;
enum { F_LOW_MEMORY = 8 };
#define CHECK_MALLOC(p) do { if (!(p)) err(1, "Out of memory"); } while (0)
static void distMatrix(struct model *M, const seq_t *s, size_t n) { (void)M; (void)s; (void)n; }
static void distMatrixLM(struct model *M, const seq_t *s, size_t n) { (void)M; (void)s; (void)n; }
void calculate_distances_stub(struct model *M, seq_t *sequences, size_t n);
""" + body.split("#ifdef HAVE_ANDI_HIP\n\tdistMatrixHIP(M, sequences, n);")[0] + r"""
void calculate_distances_stub(struct model *M, seq_t *sequences, size_t n) {
#ifdef HAVE_ANDI_HIP
	distMatrixHIP(M, sequences, n);
#else
	if (FLAGS & F_LOW_MEMORY) distMatrixLM(M, sequences, n); else distMatrix(M, sequences, n);
#endif
}
"""
    f = tmp_path / "snippet.c"
    f.write_text(src)
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Wno-unused-function", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(f)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr


def test_integration_patch_applies_to_the_reference(tmp_path):
    """the patch against the upstream tree, where that is mounted (this container; not the GPU box)"""
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "src", "process.c")) or not shutil.which("patch"):
        pytest.skip("upstream tree not mounted")
    os.makedirs(tmp_path / "src")
    shutil.copy(os.path.join(ref, "src", "process.c"), tmp_path / "src" / "process.c")
    shutil.copy(os.path.join(ref, "configure.ac"), tmp_path / "configure.ac")
    r = subprocess.run(["patch", "-p1", "-i", os.path.join(ROOT, "integration", "andi-hip.patch")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = (tmp_path / "src" / "process.c").read_text()
    assert "distMatrixHIP(M, sequences, n);" in text and "HAVE_ANDI_HIP" in (tmp_path / "configure.ac").read_text()


def test_host_packer_equals_the_alphabet_table():
    """andi_hip_pack_symbols (the seam packs its queries with it, eight nucleotides at a time) against the table the device's
    pack kernel uses: A C G T ! ; # NUL = 0 ... 7, two symbols per byte, NUL behind an odd length; bytes outside flagged."""
    from andi_amd import lib
    code = np.full(256, 7, np.uint8)
    for ch, v in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"!", 4), (b";", 5), (b"#", 6)):
        code[ch[0]] = v
    for c in range(0x41, 256):
        code[c] = ((c & 6) ^ ((c & 6) >> 1)) >> 1  # (scan_lane.hip: symbol_of -- what a byte outside the alphabet becomes)
    inside = set(b"ACGT!;#\0")
    rng = np.random.default_rng(3)
    cases = [b"A", b"AC", b"ACG", b"ACGTACGT", b"ACGTACGTA", rand_dna(np.random.default_rng(5), 1000), rand_dna(np.random.default_rng(6), 4097)]
    cases.append(b"!".join(rand_dna(np.random.default_rng(50 + k), int(rng.integers(1, 40))) for k in range(60)))           # separators at every alignment
    cases.append(bytes(rng.choice(list(b"ACGTN"), 3000).astype(np.uint8)))                             # N: outside
    cases.append(bytes(rng.integers(0, 256, 5000).astype(np.uint8)))                                   # any byte
    cases.append(rand_dna(np.random.default_rng(9), 777) + b"acgt" + rand_dna(np.random.default_rng(10), 100))                                       # lower case: outside
    cases.append(rand_dna(np.random.default_rng(11), 64) + b"#" + rand_dna(np.random.default_rng(12), 63) + b";" + rand_dna(np.random.default_rng(13), 5))
    for seq in cases:
        got, bad = lib.pack_symbols(seq)
        a = np.frombuffer(seq + (b"\0" if len(seq) % 2 else b""), np.uint8)
        want = (code[a[0::2]] & 7) | ((code[a[1::2]] & 7) << 4)
        assert (got == want).all(), seq[:40]
        assert bad == any(c not in inside for c in seq), seq[:40]


def test_format_distances_row_parallel_equals_the_sequential_loop(orc):
    """andi_hip_format_distances deals the rows of a large matrix to a pool of threads (host_model.c): for n = 300 -- several
    threads at work -- the bytes equal print_distances' sequential loop (src/io.c:246-322) restated here from the estimators:
    averaging rule, the scientific switch taken for the WHOLE matrix from one small distance, NaN and low-homology warnings in
    row order, -vv (no averaging), truncated names."""
    import numpy as np
    from andi_amd import lib
    rng = np.random.default_rng(300)
    n = 300
    M = rng.integers(0, 900, size=(n, n, 17), dtype=np.uint32)
    M[:, :, [0, 5, 10, 15]] += 30000
    M[:, :, 16] = 140000
    M[17, 230, :16] = M[230, 17, :16] = 0            # nan
    M[5, 9, :16] //= 200                             # low coverage in one direction
    M[250, 299, :16] //= 300
    names = ["g%03d_%s" % (k, "x" * (k % 14)) for k in range(n)]

    def sequential(M, model, vv, trunc, small):
        M = M.copy()
        if small:  # one distance in (0, 0.001): every number of the matrix in scientific notation (src/io.c:280-283)
            M[1, 2, :16] = M[2, 1, :16] = 0
            M[1, 2, [0, 5, 10, 15]] = M[2, 1, [0, 5, 10, 15]] = 50000
            M[1, 2, 1] = 3
        D = np.zeros((n, n))
        warn = []
        for i in range(n):
            for j in range(n):
                datum = M[i, j] if vv else (M[i, j].astype(np.uint64) + M[j, i]).astype(np.uint32)
                d = D[i, j] = 0.0 if i == j else orc.estimate(datum, model)
                if np.isnan(d):
                    warn.append("For the two sequences '%s' and '%s' the distance computation failed and is reported as nan. "
                                "Please refer to the documentation for further details.\n" % (names[i], names[j]))
                elif i < j:
                    c1, c2 = orc.coverage(M[i, j]), orc.coverage(M[j, i])
                    if c1 < 0.2 or c2 < 0.2:
                        warn.append("For the two sequences '%s' and '%s' very little homology was found (%f and %f, respectively).\n"
                                    % (names[i], names[j], c1, c2))
        sci = bool(((D > 0) & (D < 0.001)).any())
        import ctypes
        libc, buf = ctypes.CDLL(None), ctypes.create_string_buffer(64)

        def c_printf(fmt, d):  # the C library's conversion (a NaN prints with its sign: "-nan")
            libc.snprintf(buf, 64, fmt, ctypes.c_double(d))
            return buf.value.decode()
        rows = ["%d\n" % n]
        for i in range(n):
            rows.append(("%-10.10s" if trunc else "%-10s") % names[i] + "".join(c_printf(b" %1.4e" if sci else b" %1.4f", D[i, j]) for j in range(n)) + "\n")
        return M, "".join(rows), "".join(warn)

    for model, vv, trunc, small in ((lib.M_JC, False, False, False), (lib.M_KIMURA, True, True, True), (lib.M_RAW, False, False, True)):
        M1, text, warn = sequential(M, model, vv, trunc, small)
        got_text, got_warn, flags = lib.format_distances(M1, names, model, extra_verbose=vv, truncate_names=trunc)
        assert got_text == text and got_warn == warn
        assert flags == (1 if "reported as nan" in warn else 0) | (2 if "very little homology" in warn else 0)
        assert ("e-0" in text.splitlines()[1]) == small


def test_libdivsufsort_is_an_optional_link(tmp_path):
    """north_star: 'SA via libdivsufsort on host' (configure.ac:33-38, src/esa.c:303).  The image has no libdivsufsort, so the
    built-in SA-IS sorts (andi_hip_suffix_sorter says so); where the host has the library it is loaded at first use.  The
    plumbing is exercised with a test double: a libdivsufsort.so.3 made here (a plain comparison sort that also leaves a mark),
    found through LD_LIBRARY_PATH by a process of its own -- same suffix array, the double's mark proves who sorted."""
    import subprocess
    import sys
    from andi_amd import lib
    assert lib.load().andi_hip_suffix_sorter().decode().startswith("SA-IS")
    src = tmp_path / "fake_divsufsort.c"
    src.write_text(r"""
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static const unsigned char *g_t; static int32_t g_n;
static int cmp(const void *a, const void *b) {
	int32_t i = *(const int32_t *)a, j = *(const int32_t *)b;
	int32_t li = g_n - i, lj = g_n - j, m = li < lj ? li : lj;
	int c = memcmp(g_t + i, g_t + j, (size_t)m);
	return c ? c : (li < lj ? -1 : 1);
}
int32_t divsufsort(const unsigned char *T, int32_t *SA, int32_t n) {
	g_t = T, g_n = n;
	for (int32_t i = 0; i < n; i++) SA[i] = i;
	qsort(SA, (size_t)n, sizeof *SA, cmp);
	FILE *f = fopen(getenv("FAKE_DIVSUFSORT_MARK"), "w");
	if (f) { fprintf(f, "%d", n); fclose(f); }
	return 0;
}
""")
    so = tmp_path / "libdivsufsort.so.3"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(so), str(src)])
    mark = tmp_path / "mark"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from andi_amd import lib\n"
            "text = b'TGCAACGT#ACGTTGCA' * 50 + b'ACGT!GGTTAAC;'\n"
            "sa = lib.suffix_array(text)\n"
            "print(lib.load().andi_hip_suffix_sorter().decode()); print(' '.join(map(str, sa.tolist())))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FAKE_DIVSUFSORT_MARK=str(mark))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    name, sa_line = r.stdout.strip().splitlines()[-2:]
    assert name.startswith("libdivsufsort") and mark.read_text() == str(17 * 50 + 13)
    text = b"TGCAACGT#ACGTTGCA" * 50 + b"ACGT!GGTTAAC;"
    assert [int(x) for x in sa_line.split()] == lib.suffix_array(text).tolist()  # the built-in sorter, in this process
    env["ANDI_HIP_NO_DIVSUFSORT"] = "1"
    r2 = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and r2.stdout.strip().splitlines()[-2].startswith("SA-IS")
