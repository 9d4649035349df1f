"""The device suffix sorter's first two rounds (andi_amd/csrc/sa_device.hip), restated on the CPU and checked against
plain suffix comparison: round 0 keys a suffix by the 2-bit codes of its first 16 symbols with ZEROS from the first
symbol on that is no nucleotide (a separator, the end of the text); round 1 keys the suffixes that tie by (where that
symbol is, which it is, the symbols of the text behind it -- or behind the first 16).  The claim the kernels rest on:
sorting by (round-0 key, round-1 key) orders the suffixes as the text's byte order does, as far as those keys reach --
also where "prefix + separator" ties with "prefix + AAAA...".  TEST INFRASTRUCTURE (no product code is called; the device
sorter itself is compared with the host sorter entry for entry in tests/test_esa_gpu.py)."""
import numpy as np
import pytest

ORDER = {0: 0, ord("!"): 1, ord("#"): 2, ord(";"): 3, ord("A"): 4, ord("C"): 5, ord("G"): 6, ord("T"): 7}


def _codes(text: bytes, pad=64):
    return np.array([ORDER[b] for b in text] + [0] * pad, dtype=np.int64)  # (behind the text: NUL, below every symbol)


def _round0(c, i):
    """(key, jj, sep): 2-bit codes of 16 symbols, zeros from the first non-nucleotide on; where and which that one is"""
    key, jj, sep = 0, 16, 0
    for j in range(16):
        s = int(c[i + j])
        if jj == 16 and s < 4:
            jj, sep = j, s
        key = (key << 2) | ((s & 3) if jj == 16 else 0)
    return key, jj, sep


def _round1(c, i, jj, sep, syms):
    off = jj + 1 if jj < 16 else 16
    k = (jj << 2) | sep
    for j in range(syms):
        k = (k << 3) | int(c[i + off + j])
    return k


def _texts():
    rng = np.random.default_rng(5)
    dna = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    yield b"#".join([dna(300), dna(300)])
    yield b"!".join([b"ACG", b"ACGT", b"AC", b"ACGTA", b"ACG", b"A", b"AAAA", b"A" * 20] * 6)
    yield b";".join([b"GATTACA" + b"A" * k for k in range(0, 24)] * 2)
    yield b"!".join([dna(20 + 3 * k) + b"ACGTTGCAACGTAC" for k in range(12)] + [b"TTTT" + dna(40)] * 3) + b"#" + dna(100)
    yield b"A" * 200 + b"!" + b"A" * 37 + b"#" + b"A" * 90


@pytest.mark.parametrize("text", list(_texts()), ids=lambda t: "%d-bytes" % len(t))
def test_two_rounds_of_keys_order_suffixes_as_the_text_does(text):
    n, syms = len(text), 11
    c = _codes(text)
    keys = []
    for i in range(n):
        k0, jj, sep = _round0(c, i)
        keys.append((k0, _round1(c, i, jj, sep, syms)))
    # what the keys see of a suffix: its first 16 symbols, and the `syms` behind them (behind its first non-nucleotide)
    def seen(i):
        _, jj, _ = _round0(c, i)
        return 16 + syms if jj == 16 else jj + 1 + syms
    order = sorted(range(n), key=lambda i: keys[i])
    for a, b in zip(order, order[1:]):
        depth = min(seen(a), seen(b))
        sa, sb = tuple(c[a:a + depth]), tuple(c[b:b + depth])
        assert sa <= sb, (a, b, text[a:a + 20], text[b:b + 20])
        if keys[a] == keys[b]:  # a tie is a tie of everything the keys see: the doubling rounds may count on `depth` sorted symbols
            assert sa == sb, (a, b)
