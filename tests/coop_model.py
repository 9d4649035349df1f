"""CPU model of the wavefront-cooperative pass A (andi_amd/csrc/scan_coop.hip), TEST INFRASTRUCTURE ONLY.

dist_anchor (src/process.c:141-214) is a sequential chain.  The cooperative kernel computes the cold chain of a
segment -- exactly the states and counts the sequential loop would produce -- in two modes:

  G  generic steps, one after the other (lucky_anchor, src/process.c:82-100; anchor, 113-123; the accounting
     of 157-190), until a LUCKY anchor has been found: two anchors in a row on one diagonal;
  W  a window of the query along that diagonal: one bit per position (query symbol != subject symbol).  The
     chain through the window is resolved from the bits: behind a mismatch followed by >= threshold equal
     symbols the next step is a lucky anchor whose outcome the bits alone decide ("easy"); a mismatch followed
     by a shorter run but preceded by a long one is a HEAD: the chain arrives there in a known (canonical) state
     whatever happened before, so the walks from all heads of the window -- probe, step, probe ... until an
     anchor on the diagonal is found -- are independent of each other and are taken by the lanes in parallel,
     speculatively; the chain then hops from head to head.  Counts follow from the set G of gap positions (all
     mismatches + the stretches between a head and the anchor its walk lands on) and the anchors between them.

This file restates that decomposition in Python on top of the oracle's primitives and checks it against the plain
loop (tests/test_coop_model.py): what is proven here is the ALGORITHM; the kernel is then checked against
k_lane_cold slot by slot on the GPU (tests/test_coop_gpu.py).
"""
import ctypes as C

import numpy as np

from oracle import orc


class Pair:
    def __init__(self, subject: bytes, query: bytes, p_value=0.025):
        self.E = orc.OracleEsa(subject, p_value)
        self.q = query
        self.qlen = len(query)
        self.n = self.E.n
        self.thr = self.E.threshold
        self.border = self.n // 2
        self._qbuf = C.create_string_buffer(query, len(query) + 1)
        self._qaddr = C.addressof(self._qbuf)
        self.Q = np.frombuffer(query, dtype=np.uint8)
        self.S = np.frombuffer(self.E.RS + b"\0", dtype=np.uint8)
        self.SA = self.E.SA
        self.probes = 0

    def probe(self, p):
        """anchor()'s lookup (src/process.c:113-123): (length, unique, SA[i])"""
        self.probes += 1
        L = orc.lib()
        r = L.orc_get_match_cached(C.byref(self.E.esa), C.c_char_p(self._qaddr + p), self.qlen - p)
        return max(r.l, 0), r.i == r.j, int(self.SA[r.i])

    def lcp(self, p, s, maxlen):
        k = 0
        while k < maxlen:
            m = min(4096, maxlen - k)
            a, b = self.Q[p + k:p + k + m], self.S[s + k:s + k + m]
            m = min(len(a), len(b))
            if m == 0:
                return k
            x = np.nonzero(a[:m] != b[:m])[0]
            if len(x):
                return k + int(x[0])
            k += m
        return maxlen

    def count_gap(self, counts, q, s, ln):
        """model_count (src/model.c:309-337)"""
        if ln <= 0:
            return
        a, b = self.Q[q:q + ln], self.S[s:s + ln]
        ok = (a >= 65) & (b >= 65)
        code = lambda c: (((c & 6) ^ ((c & 6) >> 1)) >> 1).astype(np.int64)
        cells = 4 * code(b[ok]) + code(a[ok])
        counts += np.bincount(cells, minlength=16).astype(np.int64)


def count_equal(counts, ln):
    """model_count_equal for RAW/JC/Kimura (src/model.c:247-253)"""
    counts[0] += ln >> 2
    counts[5] += ln >> 2
    counts[10] += ln >> 2
    counts[15] += (ln >> 2) + (ln & 3)


class State:
    __slots__ = ("p", "lastS", "lastQ", "lastLen", "lwra")

    def __init__(self, p=0, lastS=0, lastQ=0, lastLen=0, lwra=0):
        self.p, self.lastS, self.lastQ, self.lastLen, self.lwra = p, lastS, lastQ, lastLen, lwra

    def tup(self):
        return (self.p, self.lastS, self.lastQ, self.lastLen, self.lwra)

    def copy(self):
        return State(*self.tup())


def cold_state(start, n):
    return State(start, n, 0, 0, 0)


class Record:
    """what pass A leaves per segment (scan.h: cold_exit, cold_counts, marks)"""

    def __init__(self):
        self.counts = np.zeros(16, np.int64)
        self.anchors = 0
        self.first = None
        self.mark = None
        self.exit = None

    def found(self, st):  # st: the state after the anchor's step
        self.anchors += 1
        if self.anchors == 1:
            self.first = (st.lastQ, st.lastS, st.lastLen)
        if self.anchors == 2:
            self.mark = (st.tup(), self.counts.copy())

    def key(self):
        return (self.exit, tuple(int(x) for x in self.counts), min(self.anchors, 255), self.first,
                None if self.mark is None else (self.mark[0], tuple(int(x) for x in self.mark[1])))


def lucky_applies(P, st):
    adv = st.p - st.lastQ
    return st.lastS + adv < P.n and adv - st.lastLen <= P.thr


def account(P, st, rec, curS):
    """src/process.c:157-190, before last_match = this_match"""
    endS, endQ = st.lastS + st.lastLen, st.lastQ + st.lastLen
    if curS > endS and st.p - endQ == curS - endS and (curS < P.border) == (st.lastS < P.border):
        count_equal(rec.counts, st.lastLen)
        P.count_gap(rec.counts, endQ, endS, st.p - endQ)
        st.lwra = 1
    else:
        if st.lwra or st.lastLen >= 2 * P.thr:
            count_equal(rec.counts, st.lastLen)
        st.lwra = 0


def plain_step(P, st, rec):
    """one trip of the loop; returns (found, lucky)"""
    found = lucky = False
    curS = curLen = 0
    if lucky_applies(P, st):
        curS = st.lastS + (st.p - st.lastQ)
        curLen = P.lcp(st.p, curS, P.qlen - st.p)
        found = lucky = curLen >= P.thr
    if not found:
        curLen, uniq, curS = P.probe(st.p)
        found = uniq and curLen >= P.thr
    if found:
        account(P, st, rec, curS)
        st.lastS, st.lastQ, st.lastLen = curS, st.p, curLen
    st.p += curLen + 1
    if found:
        rec.found(st)
    return found, lucky


def plain_segment(P, st0, end):
    st, rec = st0.copy(), Record()
    while st.p < end:
        plain_step(P, st, rec)
    rec.exit = st.tup()
    return rec


# ------------------------------------------------------------------ the cooperative decomposition
class Stats:
    def __init__(self):
        self.windows = self.heads = self.heads_on_path = self.walk_probes = self.wasted_probes = 0
        self.g_steps = self.w_nodes = self.breaks = self.opens = self.x_walks = self.heads_wasted = 0


MAX_X = 3  # anchors off the window's diagonal a walk follows before it gives up


def window_mode(P, st, rec, end, W, stats):
    """The chain stands at a canonical state of diagonal d: st.p = e0 + 1 behind the anchor [lastQ, e0) with
    lastS - lastQ = d.  Resolve as much of the next W positions as the bits decide; returns True if the chain
    moved, with st a genuine loop-top state either way."""
    thr, d = P.thr, st.lastS - st.lastQ
    e0 = st.lastQ + st.lastLen
    wb = e0
    wlen = min(W, P.qlen - wb)  # valid bit positions wb .. wb + wlen - 1
    qs = P.Q[wb:wb + wlen]
    ss = P.S[wb + d:wb + d + wlen]
    if len(ss) < wlen:  # the subject ends inside the window: NUL padding, never equal to a query symbol
        ss = np.concatenate([ss, np.zeros(wlen - len(ss), np.uint8)])
    bits = qs != ss
    sentinel = wb + wlen == P.qlen  # lcp() stops at the query's end: a mismatch as far as runs are concerned
    pos = np.nonzero(bits)[0] + wb
    if sentinel:
        pos = np.concatenate([pos, [P.qlen]])
    assert len(pos) and pos[0] == e0, "the window starts at the mismatch behind the anchor"
    stats.windows += 1

    def next_mismatch(x):  # first mismatch at or after x, None if the window does not show one
        k = np.searchsorted(pos, x)
        return int(pos[k]) if k < len(pos) else None

    # ---- heads: short run behind, long run before.  All their walks are independent: one lane each.
    runs_after = np.diff(pos) - 1  # for pos[:-1]
    is_head = np.zeros(len(pos), bool)
    for k in range(len(pos) - 1):
        before = thr if k == 0 else runs_after[k - 1]
        is_head[k] = runs_after[k] < thr and before >= thr
    walks = {}
    for k in np.nonzero(is_head)[0]:
        e = int(pos[k])
        p, nprobes = e + 1, 0
        X = None  # the last anchor, once it is one off the diagonal: (pos_Q, pos_S, length)
        nX, extra = 0, []  # lengths of such anchors that get counted (>= 2 thr, src/process.c:182-186)
        while True:
            if p >= end:
                res = ("exit",) if X is None else ("break",)
                break
            if X is None:
                if p + d < P.n and p - e <= thr:  # lucky_anchor applies on the diagonal: the bits answer
                    nm = next_mismatch(p)
                    if nm is None:
                        res = ("open",)
                        break
                    if nm - p >= thr:
                        res = ("ok", p, nm - p, False, extra)
                        break
            else:
                adv = p - X[0]
                if X[1] + adv < P.n and adv - X[2] <= thr:  # lucky_anchor on the other anchor's diagonal: compare
                    if P.lcp(p, X[1] + adv, min(P.qlen - p, thr)) >= thr:
                        res = ("break",)  # the chain really changes its diagonal: not this window's business
                        break
            ln, uniq, s = P.probe(p)
            nprobes += 1
            if uniq and ln >= thr:
                if s == p + d:  # back on the diagonal (or never left it)
                    if X is not None and X[2] >= 2 * thr:
                        extra = extra + [X[2]]
                    res = ("ok", p, ln, X is not None, extra)
                    break
                if X is not None:
                    endS, endQ = X[1] + X[2], X[0] + X[2]
                    if s > endS and p - endQ == s - endS and (s < P.border) == (X[1] < P.border):
                        res = ("break",)  # a right anchor off the diagonal: its gap is not in the window
                        break
                    if X[2] >= 2 * thr:
                        extra = extra + [X[2]]
                nX += 1
                if nX > MAX_X:
                    res = ("break",)
                    break
                X = (p, s, ln)
            p += ln + 1
        walks[e] = (res, nprobes, nX)
        stats.heads += 1

    # ---- the chain hops from node to node
    cur, aQ, lw = e0, st.lastQ, st.lwra
    used = set()
    moved = False
    while True:
        if cur + 1 >= end:
            break
        if cur + d == P.border:  # '#': the next anchor lies on the other strand, no right anchor (src/process.c:162)
            break
        nm = next_mismatch(cur + 1)
        if nm is None:
            stats.opens += 1
            break
        r = nm - cur - 1
        if r >= thr:  # easy: lucky anchor behind a single mismatch
            count_equal(rec.counts, cur - aQ)
            P.count_gap(rec.counts, cur, cur + d, 1)
            aQ, cur, lw = cur + 1, nm, 1
            stats.w_nodes += 1
            rec.found(State(cur + 1, aQ + d, aQ, cur - aQ, 1))
            moved = True
            continue
        assert cur in walks, "every node the chain reaches with a short run behind it is a head"
        res, nprobes, nX = walks[cur]
        used.add(cur)
        if res[0] != "ok":
            if res[0] == "break":
                stats.breaks += 1
            if res[0] == "open":
                stats.opens += 1
            break
        stats.heads_on_path += 1
        stats.walk_probes += nprobes
        _, a, ln, hadX, extra = res
        if not hadX:  # the anchor the walk lands on is a right anchor of the one before the head
            count_equal(rec.counts, cur - aQ)
            P.count_gap(rec.counts, cur, cur + d, a - cur)
            lw = 1
        else:  # anchors off the diagonal in between: nothing pairs up (src/process.c:176-188)
            if lw or cur - aQ >= 2 * thr:
                count_equal(rec.counts, cur - aQ)
            for x in extra:
                count_equal(rec.counts, x)
            rec.anchors += nX  # (never the 1st or 2nd of a cold chain: the window is entered behind a lucky anchor)
            lw = 0
            stats.x_walks += 1
        aQ, cur = a, a + ln
        stats.w_nodes += 1
        rec.found(State(cur + 1, aQ + d, aQ, cur - aQ, lw))
        moved = True
    for e, (res, nprobes, nX) in walks.items():
        if e not in used:
            stats.wasted_probes += nprobes
            stats.heads_wasted += 1
    st.p, st.lastS, st.lastQ, st.lastLen, st.lwra = cur + 1, aQ + d, aQ, cur - aQ, lw
    return moved


def coop_segment(P, st0, end, W=8192, stats=None):
    stats = stats or Stats()
    st, rec = st0.copy(), Record()
    while st.p < end:
        found, lucky = plain_step(P, st, rec)
        stats.g_steps += 1
        if found and lucky:
            # windows one after the other while the chain stays canonical on the diagonal and moves
            while st.p < end and st.lastQ + st.lastLen < P.qlen and window_mode(P, st, rec, end, W, stats):
                pass
    rec.exit = st.tup()
    return rec, stats
