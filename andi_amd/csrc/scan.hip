// scan.hip — the anchor scan (dist_anchor, src/process.c:141-214) on gfx950.
//
// dist_anchor is a sequential chain over one query: every iteration either
// continues on the last anchor's diagonal by direct comparison ("lucky",
// src/process.c:82-100) or asks the ESA for the longest match
// (src/process.c:113-123), pairs equidistant anchors and counts the
// substitutions between them (src/model.c:246-337).  The chain is made parallel
// by cutting each query into segments:
//
//   pass A  k_scan_cold    one wavefront per (subject, query segment) runs the
//                          chain from a cold state at the segment start and
//                          records the counts it adds and the state in which
//                          it leaves the segment.
//   pass B  k_scan_stitch  the true chain enters segment s in the state the
//                          chain of segment s-1 left in.  It is replayed next
//                          to the cold chain until both are in the same state;
//                          from there on the cold chain's counts are the true
//                          ones.  This pass assumes the predecessor's cold exit
//                          state IS the true entry state.
//   pass C  k_scan_reduce  one block per pair checks that assumption for
//                          every segment (true exit of s-1 == cold exit of
//                          s-1), sums the per-segment counts, re-stitches the
//                          rare segments whose assumption failed, and applies
//                          the loop epilogue (src/process.c:199-211).
//
// A chain is run by a group of G consecutive lanes (G = 8 by default, so one
// wavefront advances 8 segments at once): the chain's control flow is uniform
// within the group, the G lanes share the byte comparisons (common_prefix<G>,
// one 16*G-byte window per step) and the substitution counting (LDS histogram
// per group).  Every step of a chain is a handful of dependent, mostly random
// memory accesses, so throughput is set by the number of chains in flight; the
// result is bit-identical to the sequential loop.
#include "scan_dev.h"
#include "knobs.h"

#include <cstdlib>

// A chain spends most of its steps on one diagonal (query offset p against
// subject offset p + d), advancing ~1/divergence bytes per step.  Window keeps
// what the group fetched last -- 16*G bytes of the query as 2-bit codes plus an
// ACGT mask, and a bitmask of the positions where the subject's bytes differ --
// in registers: the following lucky comparisons, gap counts and K-mer codes on
// that diagonal are answered from it without touching memory, so each
// 16*G-byte line of Q and S is fetched once.
template <int G>
struct Window {
	uint32_t q0, s0; // offsets of byte 0; q0 == ~0u: empty
	uint32_t qc;   // 2-bit codes of this lane's 16 Q bytes, first byte in the top bits
	uint32_t mask; // bits 0..15: byte differs; bits 16..31: Q byte is ACGT
	uint4 sb;      // this lane's 16 S bytes (read where a substitution is counted)
};

__device__ __forceinline__ uint32_t diff_bits4(uint32_t x) { // one bit per non-zero byte
	x |= x >> 4;
	x |= x >> 2;
	x |= x >> 1;
	x &= 0x01010101u;
	return ((x * 0x01020408u) >> 24) & 0xfu;
}

// 2-bit codes (first byte in the top bits) and ACGT mask of 16 bytes
__device__ __forceinline__ void codes16(uint4 v, uint32_t &code, uint32_t &valid) {
	auto pack4 = [](uint32_t x) {
		x &= 0x06060606u;
		x ^= x >> 1;
		x = (x >> 1) & 0x03030303u;
		return ((x & 0xffu) << 6) | (((x >> 8) & 0xffu) << 4) | (((x >> 16) & 0xffu) << 2) | (x >> 24);
	};
	auto ok4 = [](uint32_t x) { // bit t set <=> byte t has bit 6 (ACGT; separators and NUL have not)
		x = (x >> 6) & 0x01010101u;
		return ((x * 0x01020408u) >> 24) & 0xfu;
	};
	code = (pack4(v.x) << 24) | (pack4(v.y) << 16) | (pack4(v.z) << 8) | pack4(v.w);
	valid = ok4(v.x) | (ok4(v.y) << 4) | (ok4(v.z) << 8) | (ok4(v.w) << 12);
}

// same[x] += number of selected bytes (bit t of sel16 = byte t) whose 2-bit code is x;
// byte t of a 16-byte chunk sits at bits 31-2t..30-2t of `code`
__device__ __forceinline__ void add_composition(uint32_t code, uint32_t sel16, uint32_t *same) {
	uint32_t m = __brev(sel16) >> 16; // byte t -> bit 15-t
	m = (m | (m << 8)) & 0x00ff00ffu;
	m = (m | (m << 4)) & 0x0f0f0f0fu;
	m = (m | (m << 2)) & 0x33333333u;
	m = (m | (m << 1)) & 0x55555555u; // byte t -> bit 30-2t
	uint32_t c0 = code & m, c1 = (code >> 1) & m; // low / high code bit of the selected bytes
	same[0] += (uint32_t)__builtin_popcount(m & ~(c0 | c1));
	same[1] += (uint32_t)__builtin_popcount(c0 & ~c1);
	same[2] += (uint32_t)__builtin_popcount(c1 & ~c0);
	same[3] += (uint32_t)__builtin_popcount(c0 & c1);
}

// model_count_equal (src/model.c:246-279) for the anchor Q[qpos..qpos+len).  RAW, JC and
// Kimura split the length evenly; LogDet and ANI count the anchor's nucleotides, which
// streams the anchor once more (16 bytes per lane and step).
template <int G, bool EXACT>
__device__ __forceinline__ void count_anchor(const PairCtx &c, Tally &t, uint32_t qpos, uint32_t len) {
	if constexpr (!EXACT) {
		count_equal(t, len);
		return;
	}
	for (uint32_t off = 16 * Group<G>::sub(); off < len; off += 16 * G) {
		uint32_t code, valid;
		codes16(ld_u128_unaligned(c.Q + qpos + off), code, valid);
		uint32_t left = len - off;
		if (left < 16) valid &= (1u << left) - 1u;
		add_composition(code, valid, t.same);
	}
}

template <int G>
__device__ __forceinline__ void window_load(Window<G> &w, g_u8p Q, g_u8p S, uint32_t q0, uint32_t s0) {
	const uint32_t sub = Group<G>::sub();
	w.q0 = q0, w.s0 = s0;
	uint4 qb = ld_u128_unaligned(Q + q0 + 16 * sub);
	uint4 sb = ld_u128_unaligned(S + s0 + 16 * sub);
	uint32_t qv;
	codes16(qb, w.qc, qv);
	w.sb = sb;
	w.mask = diff_bits4(qb.x ^ sb.x) | (diff_bits4(qb.y ^ sb.y) << 4) | (diff_bits4(qb.z ^ sb.z) << 8) |
			 (diff_bits4(qb.w ^ sb.w) << 12) | (qv << 16);
}

// lcp(Q + p, S + t, maxlen) (src/process.c:59-65) through the window.  A match that
// runs past the window it started in is followed BULK windows at a time (2*BULK
// independent loads in flight) instead of one dependent window after the other.
// A window that has to be fetched starts `back` (<= 16) bytes before p, where the
// gap since the last anchor begins: if this comparison yields a right anchor, the
// gap's substitutions are then counted from the window too.
template <int G>
__device__ __forceinline__ uint32_t window_lcp(Window<G> &w, g_u8p Q, g_u8p S, uint32_t p, uint32_t t,
											   uint32_t maxlen, uint32_t back) {
	constexpr uint32_t W = 16 * G;
	const uint32_t sub = Group<G>::sub();
	uint32_t len = 0;
	{
		uint32_t o = p - w.q0; // offset into the window, if it applies
		if (w.q0 == ~0u || t - p != w.s0 - w.q0 || p < w.q0 || o >= W) {
			if (back > 16) back = 16;
			window_load(w, Q, S, p - back, t - back);
			o = back;
		}
		// first differing byte at or after o
		int sh = (int)o - (int)(16 * sub);
		uint32_t d = w.mask & 0xffffu;
		uint32_t m = sh <= 0 ? d : (sh >= 16 ? 0u : (d >> sh) << sh);
		uint64_t hit = Group<G>::slice(__ballot(m != 0));
		if (hit) {
			uint32_t first = (uint32_t)__builtin_ctzll(hit);
			uint32_t bit = (uint32_t)__shfl((int)__builtin_ctz(m | 0x10000u), (int)(Group<G>::base() + first));
			len = 16 * first + bit - o;
			return len < maxlen ? len : maxlen;
		}
		len = W - o;
	}
	while (len < maxlen) {
		g_u8p a = Q + p + len + 16 * sub, b = S + t + len + 16 * sub;
		constexpr int BULK = 2;
		uint4 qa[BULK], sa[BULK];
#pragma unroll
		for (int u = 0; u < BULK; ++u) qa[u] = ld_u128_unaligned(a + u * W), sa[u] = ld_u128_unaligned(b + u * W);
		uint32_t key = ~0u; // offset of this lane's first differing byte within the BULK windows
#pragma unroll
		for (int u = BULK - 1; u >= 0; --u) {
			uint32_t f = first_diff_byte(make_uint4(qa[u].x ^ sa[u].x, qa[u].y ^ sa[u].y, qa[u].z ^ sa[u].z,
													 qa[u].w ^ sa[u].w));
			if (f < 16) key = u * W + 16 * sub + f;
		}
		for (int dlt = G / 2; dlt; dlt >>= 1) {
			uint32_t other = (uint32_t)__shfl_xor((int)key, dlt);
			key = other < key ? other : key;
		}
		if (key != ~0u) {
			len += key;
			break;
		}
		len += BULK * W;
	}
	return len < maxlen ? len : maxlen;
}

// model_count (src/model.c:309-337) of Q[q..q+len) against S[s..s+len), through the
// window: the part it covers is counted from its 2-bit codes, for the rest the
// window is moved along the gap.
template <int G>
__device__ __forceinline__ void window_count_gap(Window<G> &w, Tally &tally, g_u8p Q, g_u8p S, uint32_t q,
												 uint32_t s, uint32_t len) {
	constexpr uint32_t W = 16 * G;
	const uint32_t mine = 16 * Group<G>::sub();
	while (len) {
		if (w.q0 == ~0u || s - q != w.s0 - w.q0 || q < w.q0 || q >= w.q0 + W) window_load(w, Q, S, q, s);
		const uint32_t lo = q - w.q0, hi = lo + len < W ? lo + len : W;
		uint32_t a = lo > mine ? lo - mine : 0, b = hi > mine ? (hi - mine < 16 ? hi - mine : 16) : 0;
		// this lane's gap bytes whose query byte is a nucleotide (model.c:318-320)
		uint32_t range = b > a ? ((0xffffu >> (16 - (b - a))) << a) : 0u;
		uint32_t qok = (w.mask >> 16) & range;
		// equal pairs (the bulk of a gap): per nucleotide by population count; an equal
		// byte of the subject is a nucleotide too
		add_composition(w.qc, qok & ~w.mask & 0xffffu, tally.same);
		// substitutions: fetch the subject's byte, one LDS add each
		for (uint32_t d = qok & w.mask & 0xffffu; d; d &= d - 1) {
			uint32_t t = (uint32_t)__builtin_ctz(d);
			uint32_t word = t < 8 ? (t < 4 ? w.sb.x : w.sb.y) : (t < 12 ? w.sb.z : w.sb.w);
			uint8_t sb = (uint8_t)(word >> (8 * (t & 3u)));
			if ((int8_t)sb >= 'A') lds_add(&tally.hist[((nt_code(sb) << 2) | ((w.qc >> (30 - 2 * t)) & 3u)) * tally.hs], 1u);
		}
		const uint32_t done = hi - lo;
		q += done, s += done, len -= done;
	}
}

// Rare paths of the probe.  (Keeping them out of line shrinks the kernel 50x but
// measured ~10 % slower: the calls constrain register allocation on the hot path.)
template <int G>
__device__ __forceinline__ Probe reference_probe(const EsaG &E, g_u8p q, uint32_t qrem) {
	return esa_probe<G>(E, q, qrem); // the reference's own walk, ANDI_MODE_REFERENCE
}

template <int G>
__device__ __forceinline__ Probe root_search(const EsaG &E, g_u8p q, uint32_t qrem) {
	return sa_range_match<G>(E, q, qrem, 0, E.n - 1, 0);
}

// The probe of one chain step (anchor(), src/process.c:113-123) in
// ANDI_MODE_PROBE: the K-mer at Q[p] selects a probe-table entry that either is
// the answer, or names the one suffix to extend along, or names a few suffixes
// that the lanes of the group extend along in parallel.
template <int G, int MODE>
__device__ __forceinline__ Probe probe_step(const PairCtx &c, uint32_t p, const Window<G> &w) {
	const EsaG &E = c.E;
	const uint32_t qrem = c.qlen - p, K = (uint32_t)E.deepK;
	g_u8p q = c.Q + p;
	if constexpr (MODE == ANDI_MODE_REFERENCE) return reference_probe<G>(E, q, qrem);
	if (qrem <= K) return root_search<G>(E, q, qrem);

	uint32_t code, valid;
	const uint32_t o = p - w.q0;
	if (w.q0 != ~0u && p >= w.q0 && o + 16 <= 16 * G) {
		// the K-mer's 16 bytes sit in the window, across lanes o/16 and o/16 + 1
		uint32_t l0 = Group<G>::base() + (o >> 4), r = o & 15u;
		uint32_t c0 = (uint32_t)__shfl((int)w.qc, (int)l0), c1 = (uint32_t)__shfl((int)w.qc, (int)(l0 + 1));
		uint32_t v0 = (uint32_t)__shfl((int)w.mask, (int)l0) >> 16, v1 = (uint32_t)__shfl((int)w.mask, (int)(l0 + 1)) >> 16;
		code = r ? ((c0 << (2 * r)) | (c1 >> (32 - 2 * r))) : c0;
		valid = r ? ((v0 >> r) | (v1 << (16 - r))) : v0;
	} else {
		codes16(ld_u128_unaligned(q), code, valid);
	}
	if ((valid & ((1u << K) - 1u)) != ((1u << K) - 1u)) // separator inside the K-mer
		return root_search<G>(E, q, qrem);

	uint64_t raw = ld_u64_unaligned((g_u8p)(E.deep + (code >> (32 - 2 * K))));
	const uint32_t x = (uint32_t)raw, y = (uint32_t)(raw >> 32), kind = y & 3u;
	Probe r;
	if (kind == DEEP_FINAL) {
		r.len = y >> 8, r.unique = (y >> 2) & 1u;
		r.pos = (r.unique && r.len >= (uint32_t)E.thr) ? (uint32_t)E.SA[x] : 0u;
		return r;
	}
	if (kind == DEEP_SINGLE) {
		r.pos = x, r.unique = true;
		r.len = K + common_prefix<G>(q + K, E.S + x + K, qrem - K);
		return r;
	}
	if (kind != DEEP_MULTI) return root_search<G>(E, q, qrem);

	// Several occurrences: the longest match is the best of their own common
	// prefixes with the query and it is unique iff exactly one attains it.  One
	// occurrence per lane, G at a time.
	const uint32_t sub = Group<G>::sub(), cnt = (y >> 8) + 1;
	uint32_t bestLen = 0, bestCnt = 0, bestPos = 0;
	for (uint32_t off = 0; off < cnt; off += G) {
		const bool mine = off + sub < cnt;
		uint32_t pos = 0, len = 0;
		if (mine) {
			pos = (uint32_t)E.SA[x + off + sub];
			len = K + common_prefix<1>(q + K, E.S + pos + K, qrem - K);
		}
		uint32_t best = len;
		for (int d = G / 2; d; d >>= 1) {
			uint32_t other = (uint32_t)__shfl_xor((int)best, d);
			best = other > best ? other : best;
		}
		uint64_t who = Group<G>::slice(__ballot(mine && len == best));
		uint32_t n = (uint32_t)__builtin_popcountll(who);
		uint32_t bp = (uint32_t)__shfl((int)pos, (int)(Group<G>::base() + (uint32_t)__builtin_ctzll(who)));
		if (bestCnt == 0 || best > bestLen) {
			bestLen = best, bestCnt = n, bestPos = bp;
		} else if (best == bestLen) {
			bestCnt += n;
		}
	}
	r.len = bestLen, r.unique = bestCnt == 1, r.pos = bestPos;
	return r;
}

// One trip of the while loop, src/process.c:153-197.  Uniform within the group.
template <int G, int MODE, bool EXACT>
__device__ __forceinline__ ChainState chain_step(const PairCtx &c, ChainState st, Tally &tally,
												 Window<G> &w) {
	const uint32_t n = (uint32_t)c.E.n;
	uint32_t curS = 0, curLen = 0;
	bool found = false;

	// lucky_anchor, src/process.c:82-100
	uint32_t advance = st.p - st.lastQ;
	uint32_t gap = advance - st.lastLen;
	uint32_t tryS = st.lastS + advance;
	if (tryS < n && gap <= c.thr) {
		curS = tryS;
		curLen = window_lcp<G>(w, c.Q, c.E.S, st.p, tryS, c.qlen - st.p, gap);
		found = curLen >= c.thr;
	}
	// anchor, src/process.c:113-123
	if (!found) {
		Probe pr = probe_step<G, MODE>(c, st.p, w);
		curS = pr.pos;
		curLen = pr.len;
		found = pr.unique && curLen >= c.thr;
	}

	if (found) {
		uint32_t endS = st.lastS + st.lastLen;
		uint32_t endQ = st.lastQ + st.lastLen;
		if (curS > endS && st.p - endQ == curS - endS &&
			(curS < c.border) == (st.lastS < c.border)) {
			count_anchor<G, EXACT>(c, tally, st.lastQ, st.lastLen);
			window_count_gap<G>(w, tally, c.Q, c.E.S, endQ, endS, st.p - endQ);
			st.lwra = 1;
		} else {
			if (st.lwra || st.lastLen >= 2 * c.thr) count_anchor<G, EXACT>(c, tally, st.lastQ, st.lastLen);
			st.lwra = 0;
		}
		st.lastS = curS;
		st.lastQ = st.p;
		st.lastLen = curLen;
	}
	st.p += curLen + 1;
	return st;
}

// ------------------------------------------------------------------ pass A
// MODE is a template parameter so that the kernel for probe-table subjects does
// not carry the reference walk's code; blocks of the other mode's subjects exit.
template <int G, int MODE, bool EXACT>
__global__ __launch_bounds__(BLOCK, 8) void k_scan_cold(ScanArgs a) {
	__shared__ uint32_t s_hist[BLOCK / G * 16];
	if (a.subjects[blockIdx.y].mode != MODE) return;
	WorkItem it = decode_item<G>(a);
	if (!it.valid || it.is_self) return;
	Tally tally;
	tally_begin<G>(tally, G == 1 ? s_hist + threadIdx.x : s_hist + threadIdx.x / G * 16);

	PairCtx c = make_ctx(a, it.sub, it.qidx);
	ChainState st = it.seg_in_q == 0 ? initial_state() : cold_state(it.start, (uint32_t)c.E.n);
	Window<G> w;
	w.q0 = ~0u;
	while (st.p < it.end) st = chain_step<G, MODE, EXACT>(c, st, tally, w);

	size_t slot = (size_t)it.sub * a.total_segs + it.w;
	uint32_t lane = Group<G>::sub();
	st.pad[1] = ANDI_ANCHORS_UNKNOWN; // (this kernel does not count its anchors: no memory is inherited through its segments)
	if (lane == 0) a.cold_exit[slot] = st, a.exit_p[slot] = st.p;
	tally_finish<G>(tally);
	for (uint32_t t = lane; t < 16; t += G) a.cold_counts[slot * 16 + t] = tally.hist[t * tally.hs];
}

// Replays the true chain (entering in state T) through [start, end) next to the
// segment's cold chain.  On return T is the true chain's state on leaving the
// segment and histT[0..16) the counts it added inside the segment.
template <int G, int MODE, bool EXACT>
__device__ __forceinline__ void stitch_segment(const PairCtx &c, ChainState &T, uint32_t start,
											   uint32_t end, const ChainState &coldExit,
											   const uint32_t *coldCounts, uint32_t *histT,
											   uint32_t *histC) {
	Tally tT, tC;
	tally_begin<G>(tT, histT);
	tally_begin<G>(tC, histC);
	ChainState C = cold_state(start, (uint32_t)c.E.n);
	Window<G> w; // shared by both chains: they run next to each other
	w.q0 = ~0u;
	bool synced = false;
	for (;;) {
		if (same_state(T, C)) {
			synced = true;
			break;
		}
		if (T.p >= end) break;
		const bool stepT = C.p >= end || T.p <= C.p; // one call site keeps the code small
		Tally tx = stepT ? tT : tC;
		ChainState nx = chain_step<G, MODE, EXACT>(c, stepT ? T : C, tx, w);
		if (stepT) {
			T = nx, tT = tx;
		} else {
			C = nx, tC = tx;
		}
	}
	tally_finish<G>(tT);
	tally_finish<G>(tC);
	if (synced) { // from the meeting point on, the cold chain's trajectory is the true one
		for (uint32_t t = Group<G>::sub(); t < 16; t += G) histT[t * tT.hs] += coldCounts[t] - histC[t * tT.hs];
		T = coldExit;
	}
}

// ------------------------------------------------------------------ pass B
template <int G, int MODE, bool EXACT>
__global__ __launch_bounds__(BLOCK, 5) void k_scan_stitch(ScanArgs a) {
	__shared__ uint32_t s_hist[2][BLOCK / G * 16];
	if (a.subjects[blockIdx.y].mode != MODE) return;
	WorkItem it = decode_item<G>(a);
	if (!it.valid || it.is_self) return;
	size_t slot = (size_t)it.sub * a.total_segs + it.w;
	uint32_t lane = Group<G>::sub();

	if (it.seg_in_q == 0) { // the first segment's "cold" chain is the true chain
		if (lane == 0) a.true_exit[slot] = a.cold_exit[slot];
		for (uint32_t t = lane; t < 16; t += G) a.owned[slot * 16 + t] = a.cold_counts[slot * 16 + t];
		return;
	}
	const uint32_t cell0 = G == 1 ? threadIdx.x : threadIdx.x / G * 16;
	uint32_t *histT = s_hist[0] + cell0, *histC = s_hist[1] + cell0;
	PairCtx c = make_ctx(a, it.sub, it.qidx);
	// assumed entry; verified in pass C
	ChainState T = assumed_entry(a, slot - it.seg_in_q, it.seg_in_q, a.seg, c.qlen);
	if (lane == 0) a.used_entry[slot] = T;
	stitch_segment<G, MODE, EXACT>(c, T, it.start, it.end, a.cold_exit[slot], a.cold_counts + slot * 16, histT,
					  histC);
	if (lane == 0) a.true_exit[slot] = T;
	for (uint32_t t = lane; t < 16; t += G) a.owned[slot * 16 + t] = histT[t * hist_stride<G>()];
}

// ------------------------------------------------------------------ pass C
// One block per pair: all its wavefronts check the segments and add up their counts; the
// first wavefront then does what is sequential (fix-ups, epilogue).
__global__ __launch_bounds__(BLOCK) void k_scan_reduce(ScanArgs a) {
	__shared__ uint32_t s_hist[3][16];
	const uint32_t sub = blockIdx.y, qidx = blockIdx.x;
	const uint32_t lane = __lane_id();
	andi_hip_model *out = a.M + (size_t)sub * a.nq + qidx;

	if (a.self[sub] == (int64_t)qidx) { // src/dist_hack.h:61-64
		if (threadIdx.x < 17) {
			uint32_t *o = (uint32_t *)out;
			o[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == 16) ? 9u : 0u;
		}
		return;
	}

	if (a.route) { // a routed call: every layout reduces its own pairs
		const uint32_t cls = a.pair_class[sub * a.nq + qidx];
		const bool mine = a.route == ANDI_LAYOUT_COOP ? (cls & (ANDI_ROUTE_COOP | ANDI_ROUTE_LEFT)) == ANDI_ROUTE_COOP
						: a.route == ANDI_LAYOUT_LANES2 ? (cls & ANDI_ROUTE_L2) != 0
						: !(cls & (ANDI_ROUTE_COOP | ANDI_ROUTE_L2));
		if (!mine) return;
	}
	uint32_t *total = s_hist[0];
	uint32_t *histT = s_hist[1], *histC = s_hist[2];
	if (threadIdx.x < 16) total[threadIdx.x] = 0;
	__syncthreads();

	uint32_t nseg, seg;
	size_t row;
	if (a.adaptive) {
		const PairGeom g = pair_geometry(a, sub, qidx);
		nseg = g.nseg, seg = g.seg, row = g.slot0;
	} else {
		const uint32_t base = a.qseg_start[qidx];
		nseg = a.qseg_start[qidx + 1] - base, seg = a.seg;
		row = (size_t)sub * a.total_segs + base;
	}
	PairCtx c = make_ctx(a, sub, qidx);

	// every segment k >= 1 was stitched assuming it is entered in the state used_entry[k]: right iff the true chain
	// left segment k - 1 in that state
	bool ok = true;
	for (uint32_t k = 1 + threadIdx.x; k < nseg; k += blockDim.x)
		ok = ok && same_state(a.true_exit[row + k - 1], a.used_entry[row + k]);
	const bool all_ok = __syncthreads_and(ok);
	{ // the owned counts of all segments (a segment that turns out wrong below is taken out again)
		uint32_t sum[16];
#pragma unroll
		for (int t = 0; t < 16; ++t) sum[t] = 0;
		for (uint32_t k = threadIdx.x; k < nseg; k += blockDim.x) {
			const uint4 *o = (const uint4 *)(a.owned + (row + k) * 16);
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const uint4 v = o[t];
				sum[4 * t] += v.x, sum[4 * t + 1] += v.y, sum[4 * t + 2] += v.z, sum[4 * t + 3] += v.w;
			}
		}
#pragma unroll
		for (int t = 0; t < 16; ++t) {
			uint32_t v = sum[t];
			for (int d = 32; d; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
			if (lane == 0) atomicAdd(&total[t], v);
		}
	}
	__syncthreads();
	if (threadIdx.x >= 64) return; // what follows is sequential: the first wavefront alone
	ChainState fin = a.true_exit[row + nseg - 1];
	if (!all_ok) {
		// Fix-ups.  The segments are consistent up to the first k whose check fails; from there the true chain is
		// stitched again, segment after segment, until it enters a segment in the state that was assumed for it --
		// from that segment on what pass B recorded holds again, up to the next check that fails.  (The wavefront
		// looks for the failing checks 64 at a time: walking all segments of a pair one after the other, three
		// dependent loads each, took longer than the fix-ups themselves -- 3 ms per step on the realistic set.)
		uint32_t next = 1; // segments below this are settled
		for (uint32_t base = 1; base < nseg; base += 64) {
			const uint32_t kk = base + lane;
			uint64_t bad = __ballot(kk < nseg && !same_state(a.true_exit[row + kk - 1], a.used_entry[row + kk]));
			for (; bad; bad &= bad - 1) {
				uint32_t k = base + (uint32_t)__builtin_ctzll(bad);
				if (k < next) continue; // (inside, or right behind, a stretch that has been stitched again)
				ChainState st = a.true_exit[row + k - 1];
				while (k < nseg && !same_state(st, a.used_entry[row + k])) {
					if (lane == 0) atomicAdd(a.fixups, 1ull);
					const uint32_t start = k * seg, e = start + seg, end = e < c.qlen ? e : c.qlen;
					// rare; the general variant (runtime mode, per-nucleotide counting if asked for)
					if (c.E.mode == ANDI_MODE_REFERENCE) {
						if (c.exact)
							stitch_segment<64, ANDI_MODE_REFERENCE, true>(c, st, start, end, a.cold_exit[row + k],
																		  a.cold_counts + (row + k) * 16, histT, histC);
						else
							stitch_segment<64, ANDI_MODE_REFERENCE, false>(c, st, start, end, a.cold_exit[row + k],
																		   a.cold_counts + (row + k) * 16, histT, histC);
					} else {
						if (c.exact)
							stitch_segment<64, ANDI_MODE_PROBE, true>(c, st, start, end, a.cold_exit[row + k],
																	  a.cold_counts + (row + k) * 16, histT, histC);
						else
							stitch_segment<64, ANDI_MODE_PROBE, false>(c, st, start, end, a.cold_exit[row + k],
																	   a.cold_counts + (row + k) * 16, histT, histC);
					}
					if (lane < 16) total[lane] += histT[lane] - a.owned[(row + k) * 16 + lane];
					++k;
				}
				if (k == nseg) fin = st; // (the last segment itself was stitched again)
				next = k + 1;
			}
		}
	}

	// src/process.c:199-211
	Tally last;
	last.hist = (lds_u32 *)total, last.hs = 1, last.quarter = 0, last.rest = 0;
	last.same[0] = last.same[1] = last.same[2] = last.same[3] = 0;
	if (fin.lastLen >= c.qlen || fin.lwra || fin.lastLen >= 2 * c.thr) {
		const bool whole = fin.lastLen >= c.qlen;
		const uint32_t from = whole ? 0u : fin.lastQ, len = whole ? c.qlen : fin.lastLen;
		if (c.exact)
			count_anchor<64, true>(c, last, from, len);
		else
			count_anchor<64, false>(c, last, from, len);
	}
	tally_finish<64>(last);
	if (lane < 16) out->counts[lane] = total[lane];
	if (lane == 0) out->seq_len = c.qlen;
}

// ------------------------------------------------------------------ K5 hook
// get_match / get_match_cached for consecutive suffixes of one query, one
// thread per suffix (test hook and building block; src/esa.c:615-656).
__global__ __launch_bounds__(256) void k_match_positions(EsaDev Ed, const uint8_t *qd, uint32_t qlen,
														 uint32_t first, uint32_t count, int cached,
														 andi_hip_interval *out) {
	uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= count) return;
	const EsaG E = esa_global(Ed);
	g_u8p q = (g_u8p)qd;
	uint32_t pos = first + k;
	Ival r = cached ? esa_match_cached<1>(E, q + pos, qlen - pos)
					: esa_match<1>(E, q + pos, qlen - pos);
	andi_hip_interval o;
	o.l = r.l, o.i = r.i, o.j = r.j;
	o.m = r.i >= 0 ? E.SA[r.i] : -1;
	out[k] = o;
}

// ------------------------------------------------------------------ launchers
// Passes A and B of probe-table subjects are scan_lane.hip's / scan_coop.hip's (one lane or one wavefront per chain on the
// packed symbols); this file's lane groups on bytes run them for the subjects that need the reference's own walk
// (ANDI_MODE_REFERENCE: a 10-mer table entry may span a separator).
template <bool EXACT>
static hipError_t launch_reference(const ScanArgs &a, hipStream_t st, bool stitch) {
	const uint32_t per_block = BLOCK / SCAN_G;
	dim3 grid((a.total_segs + per_block - 1) / per_block, a.nsub);
	if (stitch)
		k_scan_stitch<SCAN_G, ANDI_MODE_REFERENCE, EXACT><<<grid, BLOCK, 0, st>>>(a);
	else
		k_scan_cold<SCAN_G, ANDI_MODE_REFERENCE, EXACT><<<grid, BLOCK, 0, st>>>(a);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_scan_cold(const ScanArgs &a, hipStream_t st) {
	hipError_t e = a.coop ? andi_launch_coop_cold(a, st) : andi_launch_lane_cold(a, st);
	if (e != hipSuccess || !a.any_reference) return e;
	return a.exact_equal ? launch_reference<true>(a, st, false) : launch_reference<false>(a, st, false);
}

hipError_t andi_launch_scan_stitch(const ScanArgs &a, hipStream_t st) {
	hipError_t e = andi_launch_lane_stitch(a, st);
	if (e != hipSuccess || !a.any_reference) return e;
	return a.exact_equal ? launch_reference<true>(a, st, true) : launch_reference<false>(a, st, true);
}

hipError_t andi_launch_scan_reduce(const ScanArgs &a, hipStream_t st) {
	dim3 grid(a.nq, a.nsub); // one block per pair (of one wavefront where no query has more than 64 segments: calls of
	// thousands of short queries -- 90 000 blocks of four wavefronts took 0.7 ms to launch)
	k_scan_reduce<<<grid, a.reduce_threads == 64 ? 64 : BLOCK, 0, st>>>(a);
	CHECK_LAUNCH();
	return hipSuccess;
}

hipError_t andi_launch_match_positions(const EsaDev &E, const uint8_t *q, uint32_t qlen,
									   uint32_t first, uint32_t count, int cached,
									   andi_hip_interval *out, hipStream_t st) {
	if (count == 0) return hipSuccess;
	k_match_positions<<<(count + 255) / 256, 256, 0, st>>>(E, q, qlen, first, count, cached, out);
	CHECK_LAUNCH();
	return hipSuccess;
}
