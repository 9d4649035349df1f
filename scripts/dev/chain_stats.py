"""Instrumented replay of dist_anchor (oracle primitives) on a star pair: how the chain's steps split into
easy lucky steps, cluster walks, chance anchors off the diagonal -- the numbers the cooperative pass A is sized by."""
import sys, collections
import numpy as np
sys.path.insert(0, ".")
from andi_amd import synth
from oracle import orc

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for d1, d2 in ((0.015, 0.015), (0.03, 0.03), (0.005, 0.005), (0.001, 0.002)):
    base = synth.base_codes(L, 7)
    s = synth.to_bytes(synth.mutate_codes(base, d1, 11))
    q = synth.to_bytes(synth.mutate_codes(base, d2, 12))
    E = orc.OracleEsa(s)
    thr, n = E.threshold, E.n
    RS = E.RS
    SA = E.SA
    delta = L + 1  # forward strand main diagonal
    p = 0; lastS = lastQ = lastLen = 0; lwra = False
    steps = probes = lucky_ok = lucky_try = 0
    off_anchor = on_probe_anchor = 0
    walk = 0; walks = collections.Counter(); in_walk = False
    qlen = len(q)
    qa = np.frombuffer(q, dtype=np.uint8); sa_ = np.frombuffer(RS, dtype=np.uint8)
    def lcp(a, b, lim):
        k = 0
        while k < lim:
            m = min(256, lim - k)
            x = np.nonzero(qa[a + k:a + k + m] != sa_[b + k:b + k + m])[0]
            if len(x):
                return k + int(x[0])
            k += m
        return lim
    while p < qlen:
        steps += 1
        found = False
        adv = p - lastQ; gap = adv - lastLen; tryS = lastS + adv
        if tryS < n and gap <= thr:
            lucky_try += 1
            ln = lcp(p, tryS, min(qlen - p, n - tryS))
            curS = tryS
            if ln >= thr:
                found = True; lucky_ok += 1
        if not found:
            probes += 1
            walk += 1
            l, i, j = E.get_match(q[p:p + 4000])
            ln = max(l, 0); curS = int(SA[i])
            found = (i == j and ln >= thr)
            if found:
                if curS - p == delta: on_probe_anchor += 1
                else: off_anchor += 1
        if found:
            if curS - p == delta and walk:
                walks[walk] += 1; walk = 0
            lastS, lastQ, lastLen = curS, p, ln
        p += ln + 1
    tot = sum(walks.values())
    print(f"d={d1}+{d2} thr={thr} steps={steps} ({qlen/steps:.1f} nt/step) probes={probes} lucky_try={lucky_try} lucky_ok={lucky_ok} "
          f"probe-anchors on diag={on_probe_anchor} off diag={off_anchor} ({off_anchor/max(1,probes)*100:.2f}% of probes) "
          f"walks={tot} mean probes/walk={sum(k*v for k,v in walks.items())/max(1,tot):.2f} max={max(walks) if walks else 0} "
          f"dist={[walks[k] for k in range(1,9)]}")
