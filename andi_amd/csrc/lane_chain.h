// lane_chain.h -- the chain step of dist_anchor (src/process.c:141-214) for ONE LANE on the 4-bit symbols: the
// window of 32 symbols, lcp(), gap counting (model_count, src/model.c:309-337), the probe through the probe table
// (anchor(), src/process.c:113-123) and one trip of the loop.  Shared by scan_lane.hip (a lane per chain: passes A
// and B) and scan_coop.hip (a wavefront per chain: its lanes walk the clusters of a window with these).
#pragma once
#include "esa_build.h"
#include "lane_dev.h"

namespace {

__device__ __forceinline__ void win_load(LWin &w, const PairCtx &c, uint32_t qa, int32_t dg) {
	w.q0 = qa, w.dg = dg;
	w.q = ld_query(c, qa);
	if (dg != NO_DIAG) {
		w.s = ld_subject(c, (int32_t)qa + dg);
	}
}

// lcp(Q + p, S + t, maxlen) (src/process.c:59-65), the part the window answers: lane_step has fetched it on the
// diagonal t - p, from `back` (<= 16) symbols before p on -- where the gap since the last anchor begins, so that the
// gap's substitutions can be counted from it too.  Returns the matching symbols inside the window and sets `open` if
// the match runs on past the window's end (lane_step then follows it in a window of its own).
__device__ __forceinline__ uint32_t lcp_window(const LWin &w, uint32_t p, bool &open) {
	const uint32_t o = p - w.q0;
	const uint32_t f = first_from(neq32(w.q, w.s), o);
	open = f >= WNT;
	return f - o;
}

// Common prefix of Q[q0 + from ..] and S[q0 + from + dg ..], at most lim, where the
// window holds the query symbols q0 .. q0 + 31 (the subject side is fetched).  One loop,
// one place where windows are compared: a wavefront executes every such place once per
// trip whenever one of its lanes gets there, and pass A is bound by instruction issue.
__device__ __forceinline__ uint32_t lane_extend(const LWin &w, const PairCtx &c, uint32_t from, int32_t dg,
												uint32_t lim) {
	uint4 q = w.q;
	uint32_t qa = w.q0, len = 0;
	for (;;) {
		const uint4 d = neq32(q, ld_subject(c, (int32_t)qa + dg));
		const uint32_t f = first_from(d, from);
		len += f - from;
		if (f < WNT || len >= lim) break;
		STAT(ST_EXT_LOOP);
		qa += WNT, from = 0;
		q = ld_query(c, qa);
	}
	return len < lim ? len : lim;
}

// model_count (src/model.c:309-337) of Q[q..q+len) against S[s..s+len) through the window.
__device__ __forceinline__ void lane_count_gap(LWin &w, const PairCtx &c, Tally &tally, uint32_t q, uint32_t s,
											   uint32_t len) {
	const int32_t dg = (int32_t)(s - q);
	if (KNOCK(c, 2)) return;
	while (len) {
		if (w.q0 == EMPTY || w.dg != dg || q < w.q0 || q >= w.q0 + WNT) {
			win_load(w, c, q & ~1u, dg);
			STAT(ST_GAP_RELOAD);
		}
		const uint32_t lo = q - w.q0, hi = lo + len < WNT ? lo + len : WNT;
		for (uint32_t j = lo >> 3; 8 * j < hi; ++j) {
			const uint32_t a = lo > 8 * j ? lo - 8 * j : 0u, b = hi - 8 * j < 8 ? hi - 8 * j : 8u;
			const uint32_t qw = pick(w.q, j), sw = pick(w.s, j), dw = neq8(qw, sw) >> 3;
			STAT(ST_GAP_WORDS);
			// both symbols are nucleotides (bit 2 clear), src/model.c:318-320
			const uint32_t ok = symbol_range(a, b) & ~(qw >> 2) & ~(sw >> 2);
			const uint32_t eq = ok & ~dw, b0 = qw, b1 = qw >> 1;
			// (equal pairs go to the diagonal cells in LDS like the substitutions, not to tally.same: four registers less)
			lds_add(&tally.hist[0], (uint32_t)__builtin_popcount(eq & ~(b0 | b1)));
			lds_add(&tally.hist[5 * tally.hs], (uint32_t)__builtin_popcount(eq & b0 & ~b1));
			lds_add(&tally.hist[10 * tally.hs], (uint32_t)__builtin_popcount(eq & b1 & ~b0));
			lds_add(&tally.hist[15 * tally.hs], (uint32_t)__builtin_popcount(eq & b0 & b1));
			for (uint32_t ne = ok & dw; ne; ne &= ne - 1) {
				const uint32_t k = (uint32_t)__builtin_ctz(ne);
				STAT(ST_SUBST);
				lds_add(&tally.hist[((((sw >> k) & 3u) << 2) | ((qw >> k) & 3u)) * tally.hs], 1u);
			}
		}
		const uint32_t done = hi - lo;
		q += done, s += done, len -= done;
	}
}

// model_count_equal (src/model.c:246-279) for the anchor Q[qpos..qpos+len)
// k_pair_estimate keeps pairs with unrelated stretches away from k_lane_quad only while their mean match is below this:
// with longer matches k_lane_cold's lane-by-lane following of matches that cover many segments costs more than
// k_lane_quad's slow probing (genomes with structure 1e-5 ... 1e-4 apart: pass A 23.4 ms without the bound, 11.8 with)
#ifndef ANDI_ISLAND_MEAN_MAX
#define ANDI_ISLAND_MEAN_MAX 256u
#endif
template <bool EXACT>
__device__ __forceinline__ void lane_count_anchor(const PairCtx &c, Tally &t, uint32_t qpos, uint32_t len) {
	if constexpr (!EXACT) {
		count_equal(t, len);
		return;
	}
	const uint32_t qa = qpos & ~1u, end = qpos - qa + len;
	uint32_t lo = qpos - qa;
	for (uint32_t base = 0; base < end; base += WNT, lo = 0) {
		const uint4 qv = ld_query(c, qa + base);
		const uint32_t hi = end - base < WNT ? end - base : WNT;
		for (uint32_t j = lo >> 3; 8 * j < hi; ++j) {
			const uint32_t a = lo > 8 * j ? lo - 8 * j : 0u, b = hi - 8 * j < 8 ? hi - 8 * j : 8u;
			const uint32_t qw = pick(qv, j);
			const uint32_t ok = symbol_range(a, b) & ~(qw >> 2), b0 = qw, b1 = qw >> 1;
			lds_add(&t.hist[0], (uint32_t)__builtin_popcount(ok & ~(b0 | b1)));
			lds_add(&t.hist[5 * t.hs], (uint32_t)__builtin_popcount(ok & b0 & ~b1));
			lds_add(&t.hist[10 * t.hs], (uint32_t)__builtin_popcount(ok & b1 & ~b0));
			lds_add(&t.hist[15 * t.hs], (uint32_t)__builtin_popcount(ok & b0 & b1));
		}
	}
}

// anchor() (src/process.c:113-123) through the probe table, as scan.hip's probe_step
// (`cap`: the caller does not care about match lengths beyond it)
// (`fetched`: the caller has made sure that the window holds the K-mer at p)
__device__ __forceinline__ Probe lane_probe(const PairCtx &c, uint32_t p, LWin &w, uint32_t cap = ~0u, bool fetched = false) {
	const EsaG &E = c.E;
	const uint32_t qrem = c.qlen - p < cap ? c.qlen - p : cap, K = (uint32_t)E.deepK;
	g_u8p q = c.Q + p;
	STAT(ST_PROBE);
	if (qrem <= K) return sa_range_match<1>(E, q, qrem, 0, E.n - 1, 0);
	if (KNOCK(c, 4)) {
		Probe r0;
		r0.len = K - 1, r0.unique = false, r0.pos = 0;
		return r0;
	}
	uint32_t o = p - w.q0;
	if (!fetched && (w.q0 == EMPTY || p < w.q0 || o + K > WNT)) {
		const uint32_t qa = p & ~1u;
		int32_t dg = w.dg; // stay on the diagonal the window was on while that is inside the text
		if (dg != NO_DIAG && (uint32_t)((int32_t)qa + dg) >= (uint32_t)E.n) dg = NO_DIAG;
		win_load(w, c, qa, dg);
		STAT(ST_PROBE_RELOAD);
		o = p & 1u;
	}
	uint32_t code;
	if (!lane_kmer(w, o, K, code)) return sa_range_match<1>(E, q, qrem, 0, E.n - 1, 0); // separator inside

	if (KNOCK(c, 3)) {
		Probe r0;
		r0.len = K - 1, r0.unique = false, r0.pos = 0;
		return r0;
	}
	const uint64_t raw = ld_u64_unaligned((g_u8p)(E.deep + code));
	STAT(ST_TABLE);
	const uint32_t x = (uint32_t)raw, y = (uint32_t)(raw >> 32), kind = y & 3u;
	Probe r;
	if (kind == DEEP_FINAL) {
		r.len = y >> 8, r.unique = (y >> 2) & 1u;
		r.pos = (r.unique && r.len >= (uint32_t)E.thr) ? (uint32_t)E.SA[x] : 0u;
		if (r.unique && r.len >= (uint32_t)E.thr) STAT(ST_FINAL_SA);
		return r;
	}
	if (kind != DEEP_SINGLE && kind != DEEP_MULTI) return sa_range_match<1>(E, q, qrem, 0, E.n - 1, 0);
	const uint32_t cnt = kind == DEEP_SINGLE ? 1u : (y >> 8) + 1;
	if (kind == DEEP_SINGLE) STAT(ST_SINGLE); else STAT(ST_MULTI);
	if (KNOCK(c, 0) && kind == DEEP_MULTI) {
		r.len = K, r.unique = false, r.pos = 0;
		return r;
	}
	if (cnt > MULTI_MAX) STAT(ST_SEARCH);
	if (cnt > MULTI_MAX) return sa_range_match<1>(E, q, qrem, (int32_t)x, (int32_t)(x + cnt - 1), K);
	// the longest match is the best of the occurrences' own common prefixes with the
	// query and it is unique iff exactly one attains it.  A K-mer that occurs once takes the
	// same loop as one that occurs several times (its position is in the entry itself): one
	// place where the lanes of a wavefront extend matches, not two in a row.
	// (Four occurrences per round trip -- one 16-byte load of positions, four windows in flight -- were measured:
	// 12 more registers, a wavefront less per SIMD, pass A 7.4 -> 7.9 ms.)
	// (The positions of a repeated K-mer are fetched two at a time: a trip of a wavefront's loop waits for each of these
	// loads in turn, as often as its lane with the most occurrences needs.)
	uint32_t bestLen = 0, bestCnt = 0, bestPos = 0, nextPos = 0;
	for (uint32_t i = 0; i < cnt; ++i) {
		uint32_t pos = x;
		if (kind != DEEP_SINGLE) {
			if (i & 1u) {
				pos = nextPos;
			} else {
				const uint64_t two = ld_u64_unaligned((g_u8p)(E.SA + x + i)); // (SA is padded by eight entries)
				pos = (uint32_t)two, nextPos = (uint32_t)(two >> 32);
			}
		}
		if (kind != DEEP_SINGLE) STAT(ST_MULTI_CAND);
		const uint32_t len = K + lane_extend(w, c, o + K, (int32_t)(pos - p), qrem - K);
		if (len > bestLen) {
			bestLen = len, bestCnt = 1, bestPos = pos;
		} else if (len == bestLen) {
			++bestCnt;
		}
	}
	r.len = bestLen, r.unique = bestCnt == 1, r.pos = bestPos;
	return r;
}

// What an anchor at subject offset curS found at query offset st.p does to the counts
// (src/process.c:157-190), apart from recording its own length.
template <bool EXACT>
__device__ __forceinline__ void lane_account(const PairCtx &c, ChainState &st, Tally &tally, LWin &w, uint32_t curS) {
	const uint32_t endS = st.lastS + st.lastLen;
	const uint32_t endQ = st.lastQ + st.lastLen;
	if (curS > endS && st.p - endQ == curS - endS && (curS < c.border) == (st.lastS < c.border)) {
		lane_count_anchor<EXACT>(c, tally, st.lastQ, st.lastLen);
		lane_count_gap(w, c, tally, endQ, endS, st.p - endQ);
		st.lwra = 1;
	} else {
		if (st.lwra || st.lastLen >= 2 * c.thr) lane_count_anchor<EXACT>(c, tally, st.lastQ, st.lastLen);
		st.lwra = 0;
	}
}

// One trip of the while loop, src/process.c:153-197.
template <bool EXACT>
__device__ __forceinline__ ChainState lane_step(const PairCtx &c, ChainState st, Tally &tally, LWin &w, bool &found) {
	const uint32_t n = (uint32_t)c.E.n;
	uint32_t curS = 0, curLen = 0;
	found = false;
	STAT(ST_STEP);

	// lucky_anchor, src/process.c:82-100.  A match that runs on past the window is followed in a window of its own
	// (t): w keeps the symbols around p -- the gap behind the last anchor, which is counted once the match is known
	// to be an anchor (one place where gaps are counted, not one before the slide and one after), and the K-mer at p,
	// should it fail -- and takes over t's last piece when the step is done.
	const uint32_t advance = st.p - st.lastQ;
	const uint32_t gap = advance - st.lastLen;
	const uint32_t tryS = st.lastS + advance;
	uint4 tq = make_uint4(0, 0, 0, 0), ts = tq;
	uint32_t tq0 = EMPTY;
	const bool lucky = tryS < n && gap <= c.thr;
	// ONE place where the window around p is fetched, whatever it is wanted for: the lucky attempt (on the last
	// anchor's diagonal, from the gap behind it on) or the K-mer of the probe -- a wavefront waits at every such
	// place once per trip.  It is fetched so that it also holds the K-mer at p, should the attempt fail (lcp_window
	// and lane_probe find what they need and fetch nothing).
	{
		const uint32_t K = (uint32_t)c.E.deepK, o = st.p - w.q0;
		int32_t dg = lucky ? (int32_t)(tryS - st.p) : w.dg;
		uint32_t back = lucky ? (gap < 16 ? gap : 16u) : 0u;
		if (w.q0 == EMPTY || st.p < w.q0 || o + K > WNT || (lucky && w.dg != dg)) {
			const uint32_t qa = (st.p - back) & ~1u;
			if (dg != NO_DIAG && (uint32_t)((int32_t)qa + dg) >= n) dg = NO_DIAG; // (only without a lucky attempt: tryS < n)
			win_load(w, c, qa, dg);
			STAT(ST_LCP_RELOAD);
		}
	}
	if (lucky) {
		STAT(ST_LUCKY_TRY);
		const uint32_t maxlen = c.qlen - st.p;
		bool open;
		curS = tryS;
		curLen = lcp_window(w, st.p, open);
		if (open && !KNOCK(c, 1)) {
			const int32_t dg = w.dg;
			uint32_t qa = w.q0;
			// (A wavefront runs this loop as often as its longest match needs, so a round is kept to the test "all 32
			// symbols equal?" -- seven instructions; where the match ends inside the last window is found once, behind it.)
			bool differ = false;
			while (curLen < maxlen) {
				qa += WNT;
				tq0 = qa, tq = ld_query(c, qa), ts = ld_subject(c, (int32_t)qa + dg);
				STAT(ST_LCP_SLIDE);
				differ = (((tq.x ^ ts.x) | (tq.y ^ ts.y)) | ((tq.z ^ ts.z) | (tq.w ^ ts.w))) != 0;
				if (differ) break;
				curLen += WNT;
			}
			if (differ) curLen += first_from(neq32(tq, ts), 0);
		}
		if (curLen > maxlen) curLen = maxlen;
		found = curLen >= c.thr;
	}
	// anchor, src/process.c:113-123
	if (!found) {
		Probe pr = lane_probe(c, st.p, w, ~0u, true);
		curS = pr.pos;
		curLen = pr.len;
		found = pr.unique && curLen >= c.thr;
	}

	if (found) {
		lane_account<EXACT>(c, st, tally, w, curS);
		st.lastS = curS;
		st.lastQ = st.p;
		st.lastLen = curLen;
	}
	if (tq0 != EMPTY) w.q0 = tq0, w.q = tq, w.s = ts; // (w.dg: the diagonal followed)
	st.p += curLen + 1;
	return st;
}

} // namespace
