"""How the cooperative pass A's work splits, from the CPU model: generic steps, window nodes, heads, probes wasted."""
import sys
sys.path.insert(0, ".")
from andi_amd import synth
from tests import coop_model as cm

L = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
W = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
base = synth.base_codes(L, 7)
for d1, d2 in ((0.001, 0.002), (0.005, 0.005), (0.015, 0.015), (0.03, 0.03)):
    s = synth.to_bytes(synth.mutate_codes(base, d1, 11))
    q = synth.to_bytes(synth.mutate_codes(base, d2, 12))
    P = cm.Pair(s, q)
    tot = cm.Stats()
    nseg = (P.qlen + seg - 1) // seg
    plain_probes = 0
    for k in range(nseg):
        st0 = cm.State() if k == 0 else cm.cold_state(k * seg, P.n)
        end = min((k + 1) * seg, P.qlen)
        P.probes = 0
        want = cm.plain_segment(P, st0, end)
        plain_probes += P.probes
        got, _ = cm.coop_segment(P, st0, end, W, tot)
        assert got.key() == want.key()
    print(f"d={d1}+{d2} seg={seg} W={W}: segments {nseg}, windows {tot.windows} ({L/max(1,tot.windows):.0f} nt each), G steps {tot.g_steps} "
          f"({tot.g_steps/nseg:.1f}/segment, {tot.g_steps/max(1,tot.windows):.2f}/window), W nodes {tot.w_nodes}, heads {tot.heads} on path {tot.heads_on_path}, "
          f"(wasted {tot.heads_wasted}, with off-diagonal anchors {tot.x_walks}), walk probes used {tot.walk_probes} wasted {tot.wasted_probes}, plain-loop probes {plain_probes}, breaks {tot.breaks}, opens {tot.opens}")
