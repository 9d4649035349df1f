// scan_rounds.hip — pass A of the anchor scan (cold chains of dist_anchor,
// src/process.c:141-214) with one lane per chain, executed in ROUNDS.
//
// scan_lane.hip's pass A lets every lane run its chain as straight-line code with loads
// wherever the chain needs data.  The lanes of a wavefront are then at 10 different places
// of that code at any time, a load instruction is issued for 5 or 6 of the 64 lanes
// (profiles/r01n_pmc.txt: 86 M wave-level loads for 480 M lane-loads), and the wavefront
// waits out every one of those latencies in turn: the kernel's time is inversely
// proportional to the resident wavefronts (profiles/r02a_occupancy_sweep.txt).
//
// Here a chain is a small state machine and ALL global loads of a wavefront are issued at
// one place, the memory phase of a round: every lane posts what it needs next (a line of
// its query, a line of the subject on its diagonal, a probe-table entry, the suffix-array
// entries or first windows of a repeated K-mer's occurrences), the wavefront issues the
// loads of all 64 lanes back to back, waits once, and then every lane computes -- from
// registers and LDS only -- until it needs memory again.  One latency per round instead
// of one per lane and load.
//
// The two streams of a chain -- its query, and the subject along the diagonal it is on --
// are read through LINE BUFFERS of the lane's own in LDS: 16 * QP bytes of the query (a
// line of 32 * QP symbols, aligned in the sequence) and the 16 * SP subject bytes that lie
// against such a line on one diagonal, filled by LDS-DMA (piece k of lane l lands at
// [k][l], no data registers).  The memory system charges per request that misses the L2,
// not per byte (profiles/micro/r02_line_fetch.txt: 56 G/s for 16-byte pieces, 64- and
// 128-byte lines alike), and round 1's windows, fetched piece by piece, fetched every line
// about three times.
//
// Probes: one table entry decides most of them.  An entry of a K-mer that occurs once
// carries the 13 symbols that follow it in the text (esa_build.hip), so a chance match is
// settled without touching the text; a match that survives those is an anchor and is
// followed through the line buffers on its own diagonal.
//
// The chain logic is scan_lane.hip's lane_step, cut at the points where it needs memory;
// results are bit-identical (tests/test_scan_gpu.py runs both).  Models that count the
// nucleotides of every anchor (LogDet, ANI) keep scan_lane.hip's kernel.
#include "lane_dev.h"
#include "knobs.h"

namespace {

typedef __attribute__((address_space(3))) void *lds_vp;

// where a chain resumes
enum : uint32_t {
	PC_STEP,       // top of the loop, src/process.c:153
	PC_LCP,        // comparing along a diagonal: lucky_anchor's lcp(), or the one occurrence of a K-mer
	PC_PROBE,      // anchor(): needs the piece(s) of the query that hold the K-mer
	PC_PROBE_NX,   // ... the piece behind the window's arrives in ra
	PC_KMER,       // form the code, ask for the table entry
	PC_TABLE,      // the entry arrives in ra
	PC_MULTI_SA,   // the occurrences' positions arrive in ra, rb
	PC_MULTI_REQ,  // ask for the first windows of the next two occurrences
	PC_MULTI_EXT,  // they arrive in ra, rb
	PC_DONE_PROBE, // curS, curLen, found are set
	PC_ACCOUNT,    // src/process.c:157-190 for an anchor at curS
	PC_GAP,        // model_count over the gap behind it
	PC_FINISH,     // src/process.c:191-197
	PC_IDLE        // the chain has left its segment
};
enum : uint32_t { RQ = 1, RS = 2, RA = 4, RB = 8 }; // requests: query line, subject line, 16 bytes at addrA / addrB

// 2-bit code (first symbol most significant) of the K symbols at offset o of the 64
// symbols q, nx (o < 32, K <= 13); false if one of them is not a nucleotide.
__device__ __forceinline__ uint32_t word8(const uint4 &q, const uint4 &nx, uint32_t k) {
	return k < 4 ? pick(q, k) : pick(nx, k - 4);
}

__device__ __forceinline__ uint32_t squeeze8(uint32_t x) { // 8 nibbles -> 8 x 2 bits, first symbol in the low bits
	x &= 0x33333333u;
	x = (x | (x >> 2)) & 0x0f0f0f0fu;
	x = (x | (x >> 4)) & 0x00ff00ffu;
	x = (x | (x >> 8)) & 0x0000ffffu;
	return x;
}

// 16 symbols (nibbles) from offset o (< 64) of the 64 symbols q, nx
__device__ __forceinline__ uint64_t nibbles_at(const uint4 &q, const uint4 &nx, uint32_t o) {
	const uint32_t j = o >> 3, r = (o & 7u) * 4u;
	const uint32_t a = word8(q, nx, j), b = word8(q, nx, j + 1), e = word8(q, nx, j + 2);
	const uint32_t lo = __builtin_amdgcn_alignbit(b, a, r), hi = __builtin_amdgcn_alignbit(e, b, r);
	return lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ bool kmer2(const uint4 &q, const uint4 &nx, uint32_t o, uint32_t K, uint32_t &code) {
	const uint64_t v = nibbles_at(q, nx, o);
	const uint64_t inside = ~0ull >> (64 - 4 * K);
	uint32_t y = __brev(squeeze8((uint32_t)v) | (squeeze8((uint32_t)(v >> 32)) << 16)); // first symbol on top, bits of a pair swapped
	y = ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
	code = y >> (32 - 2 * K);
	return (v & inside & 0x4444444444444444ull) == 0;
}

// rare: an occurrence of a repeated K-mer that matches beyond its first window -- follow it with direct loads
__device__ __forceinline__ uint32_t extend_direct(const PairCtx &c, uint32_t qa, int32_t dg, uint32_t len, uint32_t lim) {
	while (len < lim) {
		const uint4 d = neq32(ld_query(c, qa), ld_subject(c, (int32_t)qa + dg));
		const uint32_t f = first_from(d, 0);
		len += f;
		if (f < WNT) break;
		qa += WNT;
	}
	return len < lim ? len : lim;
}

template <int QP, int SP>
__global__ __launch_bounds__(64) void k_rounds_cold(ScanArgs a) {
	__shared__ uint4 s_q[QP * 64];
	__shared__ uint4 s_s[SP * 64];
	__shared__ uint32_t s_hist[16 * 64];
	if (!a.adaptive && a.subjects[blockIdx.y].mode != ANDI_MODE_PROBE) return;
	const LaneItem it = lane_item<64>(a);
	if (!__ballot(it.valid)) return;

	constexpr uint32_t QSYM = WNT * QP, SSYM = WNT * SP; // symbols per line
	const uint32_t lane = threadIdx.x;
	uint4 *const lq = s_q + lane, *const ls = s_s + lane; // this lane's piece 0; piece k is 64 further

	Tally tally;
	tally.hist = (lds_u32 *)(s_hist + lane), tally.hs = 64, tally.quarter = 0, tally.rest = 0;
	tally.same[0] = tally.same[1] = tally.same[2] = tally.same[3] = 0;
#pragma unroll
	for (int t = 0; t < 16; ++t) tally.hist[t * 64] = 0;

	const PairCtx c = make_ctx(a, it.sub, it.qidx);
	const EsaG &E = c.E;
	const uint32_t n = (uint32_t)E.n, thr = c.thr, qlen = c.qlen, K = (uint32_t)E.deepK;
	ChainState st = it.seg_in_q == 0 ? initial_state() : cold_state(it.start, n);
	const size_t slot = it.slot;
	ColdMark *marks = a.marks + slot * ANDI_COLD_MARKS;
	uint32_t anchors = 0;

	uint32_t pc = it.valid ? PC_STEP : PC_IDLE;
	uint32_t req = 0;
	uint32_t qline = EMPTY, sline = EMPTY; // what the line buffers hold
	int32_t sdg = NO_DIAG;
	uint32_t rq_line = 0, rs_line = 0;
	int32_t rs_dg = 0;
	g_u8p addrA = c.Qn, addrB = c.Qn;
	uint4 ra = make_uint4(0, 0, 0, 0), rb = ra;
	// the step in progress
	uint32_t curS = 0, curLen = 0, ret = PC_FINISH;
	bool found = false, accounted = false, ext_mode = false;
	uint32_t gq = 0, glen = 0; // gap being counted
	int32_t gdg = 0;
	uint4 nxv = ra; // the 32 query symbols behind the piece that holds p (when needed and at hand)
	bool nx_have = false;
	uint4 msa0 = ra, msa1 = ra; // positions of the occurrences of a repeated K-mer
	uint32_t cand = 0, cnt = 0, bestLen = 0, bestCnt = 0, bestPos = 0;

	auto q_ok = [&](uint32_t x) { return x / QSYM == qline; };
	auto s_ok = [&](uint32_t x, int32_t dg) { return x / SSYM == sline && dg == sdg; };
	auto qp = [&](uint32_t x) { return lq[((x / WNT) & (QP - 1)) * 64]; };
	auto sp = [&](uint32_t x) { return ls[((x / WNT) & (SP - 1)) * 64]; };

	for (;;) {
		// ------------------------------------------------------------ memory phase
		if (__ballot(req & RQ)) {
			if (req & RQ) {
				g_u8p src = c.Qn + (size_t)rq_line * (16 * QP);
#pragma unroll
				for (int k = 0; k < QP; ++k) __builtin_amdgcn_global_load_lds(src + 16 * k, (lds_vp)(s_q + k * 64), 16, 0, 0);
				qline = rq_line;
				STAT(ST_LCP_RELOAD);
			}
		}
		if (__ballot(req & RS)) {
			if (req & RS) {
				const int32_t sa = (int32_t)(rs_line * SSYM) + rs_dg, odd = sa & 1; // sa > -SSYM: the text has that much in front
				g_u8p src = (odd ? E.N1 : E.N0) + ((sa + odd) >> 1);
#pragma unroll
				for (int k = 0; k < SP; ++k) __builtin_amdgcn_global_load_lds(src + 16 * k, (lds_vp)(s_s + k * 64), 16, 0, 0);
				sline = rs_line, sdg = rs_dg;
				STAT(ST_GAP_RELOAD);
			}
		}
		if (__ballot(req & RA)) {
			if (req & RA) ra = ld_u128_unaligned(addrA);
		}
		if (__ballot(req & RB)) {
			if (req & RB) rb = ld_u128_unaligned(addrB);
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		req = 0;
		if (lane == 0) STAT(ST_LCP_SLIDE); // (diagnostic builds: rounds)

		// ------------------------------------------------------------ compute phase
		bool run = pc != PC_IDLE;
		// A lane computes until it needs memory, but the wavefront does not wait for its slowest lane:
		// after max_passes trips it goes to memory with what has been posted (lanes still able to
		// run just continue in the next round).
		for (uint32_t pass = 0; pass < a.max_passes && __ballot(run); ++pass) {
			if (lane == 0) STAT(ST_PROBE_RELOAD); // (diagnostic builds: passes)
			if (run && pc == PC_STEP) {
				if (st.p >= it.end) {
					pc = PC_IDLE, run = false;
				} else {
					STAT(ST_STEP);
					// lucky_anchor, src/process.c:82-100
					const uint32_t advance = st.p - st.lastQ, gap = advance - st.lastLen, tryS = st.lastS + advance;
					accounted = false, found = false;
					if (tryS < n && gap <= thr) {
						STAT(ST_LUCKY_TRY);
						curS = tryS, curLen = 0, ext_mode = false, pc = PC_LCP;
					} else {
						pc = PC_PROBE;
					}
				}
			}

			// lcp(Q + p, S + curS, qlen - p) (src/process.c:59-65) from curLen on, through the line buffers
			if (run && pc == PC_LCP) {
				const int32_t dg = (int32_t)(curS - st.p);
				const uint32_t maxlen = qlen - st.p;
				bool done = false;
				for (;;) {
					if (curLen >= maxlen) {
						done = true;
						break;
					}
					const uint32_t pos = st.p + curLen, x = pos & ~(WNT - 1u);
					if (!q_ok(x) || !s_ok(x, dg)) {
						if (!accounted && curLen >= thr && maxlen >= thr) {
							// certainly an anchor already: count the gap behind it while the buffers still hold it
							ret = PC_LCP, pc = PC_ACCOUNT;
						} else {
							if (!q_ok(x)) req |= RQ, rq_line = x / QSYM;
							if (!s_ok(x, dg)) req |= RS, rs_line = x / SSYM, rs_dg = dg;
							run = false;
						}
						break;
					}
					const uint32_t o = pos - x, f = first_from(neq32(qp(x), sp(x)), o);
					curLen += f - o;
					if (f < WNT) {
						done = true;
						break;
					}
				}
				if (done) {
					if (curLen > maxlen) curLen = maxlen;
					found = curLen >= thr; // (the occurrence of a K-mer that occurs once is unique)
					if (found) {
						ret = PC_FINISH, pc = accounted ? PC_FINISH : PC_ACCOUNT;
					} else {
						pc = ext_mode ? PC_FINISH : PC_PROBE;
					}
				}
			}

			// anchor, src/process.c:113-123, through the probe table
			if (run && pc == PC_PROBE) {
				const uint32_t qrem = qlen - st.p;
				STAT(ST_PROBE);
				if (qrem <= K) {
					const Probe pr = sa_range_match<1>(E, c.Q + st.p, qrem, 0, E.n - 1, 0);
					curS = pr.pos, curLen = pr.len, found = pr.unique && pr.len >= thr, pc = PC_DONE_PROBE;
				} else {
					const uint32_t x = st.p & ~(WNT - 1u), o = st.p - x;
					if (!q_ok(x)) {
						req |= RQ, rq_line = x / QSYM, run = false;
					} else if (q_ok(x + WNT)) {
						nxv = qp(x + WNT), nx_have = true, pc = PC_KMER;
					} else if (o + K >= WNT) { // the K-mer or the symbol behind it lies in the next line: not worth moving the buffer
						addrA = c.Qn + ((x + WNT) >> 1), req |= RA, pc = PC_PROBE_NX, run = false;
					} else {
						nx_have = false, pc = PC_KMER;
					}
				}
			}
			if (run && pc == PC_PROBE_NX) nxv = ra, nx_have = true, pc = PC_KMER;
			if (run && pc == PC_KMER) {
				const uint32_t x = st.p & ~(WNT - 1u), o = st.p - x;
				const uint4 q0 = qp(x);
				const uint4 nx = nx_have ? nxv : make_uint4(0, 0, 0, 0);
				uint32_t code;
				if (!kmer2(q0, nx, o, K, code)) { // a separator inside
					STAT(ST_SEARCH);
					const Probe pr = sa_range_match<1>(E, c.Q + st.p, qlen - st.p, 0, E.n - 1, 0);
					curS = pr.pos, curLen = pr.len, found = pr.unique && pr.len >= thr, pc = PC_DONE_PROBE;
				} else {
					addrA = (g_u8p)(E.deep + code), req |= RA, pc = PC_TABLE, run = false;
					STAT(ST_TABLE);
				}
			}
			if (run && pc == PC_TABLE) {
				const uint32_t tx = ra.x, ty = ra.y, kind = ty & 3u, qrem = qlen - st.p;
				if (kind == DEEP_FINAL) {
					curLen = ty >> 8;
					found = ((ty >> 2) & 1u) && curLen >= thr;
					curS = found ? (uint32_t)E.SA[tx] : 0u; // (only texts so short that K >= thr)
					pc = PC_DONE_PROBE;
				} else if (kind == DEEP_SINGLE) {
					// the K-mer occurs once, at tx; the entry holds the (up to 13) nucleotides behind it
					const uint32_t x = st.p & ~(WNT - 1u), from = st.p - x + K;
					const uint32_t nval = (ty >> 2) & 15u, ext = ty >> 6;
					const uint4 nx = nx_have ? nxv : make_uint4(0, 0, 0, 0);
					const uint64_t v = nibbles_at(qp(x), nx, from);
					const uint32_t at_hand = (nx_have ? 2 * WNT : WNT) - from; // query symbols from p + K on that v holds (of 16)
					const uint64_t stops = (v & 0x4444444444444444ull) | (1ull << 62);
					const uint32_t qstop = (uint32_t)__builtin_ctzll(stops) >> 2; // leading nucleotides of the query there
					const uint32_t diff = (squeeze8((uint32_t)v) | (squeeze8((uint32_t)(v >> 32)) << 16)) ^ ext;
					const uint32_t m = diff ? (uint32_t)__builtin_ctz(diff) >> 1 : 16u; // first differing nucleotide
					uint32_t lim = nval < at_hand ? nval : at_hand;
					if (qstop < lim) lim = qstop;
					if (qrem - K < lim) lim = qrem - K;
					curS = tx;
					STAT(ST_SINGLE);
					if (m < lim) { // settled by the entry
						curLen = K + m, found = curLen >= thr, pc = PC_DONE_PROBE;
					} else if (K + lim >= qrem) { // the query ends inside the match
						curLen = qrem, found = curLen >= thr, pc = PC_DONE_PROBE;
					} else if (lim == nval && nval < 13 && qstop > lim && at_hand > lim) {
						// the text has a separator there, the query a nucleotide
						curLen = K + lim, found = curLen >= thr, pc = PC_DONE_PROBE;
					} else { // all that could be compared matches: follow the occurrence on its diagonal
						STAT(ST_EXT_LOOP);
						curLen = K + lim, ext_mode = true, pc = PC_LCP;
					}
				} else if (kind == DEEP_MULTI) {
					cnt = (ty >> 8) + 1;
					STAT(ST_MULTI);
					if (cnt > ROUNDS_MULTI_MAX) {
						STAT(ST_SEARCH);
						const Probe pr = sa_range_match<1>(E, c.Q + st.p, qrem, (int32_t)tx, (int32_t)(tx + cnt - 1), K);
						curS = pr.pos, curLen = pr.len, found = pr.unique && pr.len >= thr, pc = PC_DONE_PROBE;
					} else {
						addrA = (g_u8p)(E.SA + tx), addrB = addrA + 16;
						req |= cnt > 4 ? (RA | RB) : RA;
						pc = PC_MULTI_SA, run = false;
					}
				} else {
					STAT(ST_SEARCH);
					const Probe pr = sa_range_match<1>(E, c.Q + st.p, qrem, 0, E.n - 1, 0);
					curS = pr.pos, curLen = pr.len, found = pr.unique && pr.len >= thr, pc = PC_DONE_PROBE;
				}
			}
			// A repeated K-mer: the longest match is the best of the occurrences' own common prefixes
			// with the query, unique iff exactly one attains it.  Two occurrences per round.
			if (run && pc == PC_MULTI_SA) msa0 = ra, msa1 = rb, cand = 0, bestLen = 0, bestCnt = 0, bestPos = 0, pc = PC_MULTI_REQ;
			if (run && (pc == PC_MULTI_REQ || pc == PC_MULTI_EXT)) {
				const uint32_t e0 = st.p + K, xe = e0 & ~(WNT - 1u), from = e0 - xe, qrem = qlen - st.p;
				const uint4 &mv = cand < 4 ? msa0 : msa1;
				const uint32_t pos0 = (cand & 2u) ? mv.z : mv.x, pos1 = (cand & 2u) ? mv.w : mv.y;
				const bool two = cand + 1 < cnt;
				if (pc == PC_MULTI_REQ) {
					const int32_t s0 = (int32_t)(xe + pos0 - st.p), s1 = (int32_t)(xe + pos1 - st.p);
					addrA = ((s0 & 1) ? E.N1 : E.N0) + ((s0 + (s0 & 1)) >> 1);
					req |= RA;
					if (two) addrB = ((s1 & 1) ? E.N1 : E.N0) + ((s1 + (s1 & 1)) >> 1), req |= RB;
					pc = PC_MULTI_EXT, run = false;
				} else {
					const uint4 qv = xe == (st.p & ~(WNT - 1u)) ? qp(xe) : nxv; // (nxv is at hand whenever p + K is in the next piece)
					for (uint32_t k = 0; k < (two ? 2u : 1u); ++k) {
						const uint32_t pos = k ? pos1 : pos0;
						const uint32_t f = first_from(neq32(qv, k ? rb : ra), from);
						uint32_t len = K + f - from;
						STAT(ST_MULTI_CAND);
						if (f >= WNT) len = extend_direct(c, xe + WNT, (int32_t)(pos - st.p), len, qrem);
						if (len > qrem) len = qrem;
						if (len > bestLen) {
							bestLen = len, bestCnt = 1, bestPos = pos;
						} else if (len == bestLen) {
							++bestCnt;
						}
					}
					cand += 2;
					if (cand < cnt) {
						pc = PC_MULTI_REQ; // (next pass of the while loop)
					} else {
						curS = bestPos, curLen = bestLen, found = bestCnt == 1 && bestLen >= thr, pc = PC_DONE_PROBE;
					}
				}
			}
			if (run && pc == PC_DONE_PROBE) {
				ret = PC_FINISH;
				pc = (found && !accounted) ? PC_ACCOUNT : PC_FINISH;
			}

			// What an anchor at curS, found at p, does to the counts (src/process.c:157-190)
			if (run && pc == PC_ACCOUNT) {
				const uint32_t endS = st.lastS + st.lastLen, endQ = st.lastQ + st.lastLen;
				accounted = true;
				if (curS > endS && st.p - endQ == curS - endS && (curS < c.border) == (st.lastS < c.border)) {
					count_equal(tally, st.lastLen);
					st.lwra = 1;
					gq = endQ, gdg = (int32_t)(endS - endQ), glen = st.p - endQ, pc = PC_GAP;
				} else {
					if (st.lwra || st.lastLen >= 2 * thr) count_equal(tally, st.lastLen);
					st.lwra = 0;
					pc = ret;
				}
			}
			// model_count (src/model.c:309-337) of Q[gq .. gq + glen) against the subject on diagonal gdg
			if (run && pc == PC_GAP) {
				while (glen) {
					const uint32_t x = gq & ~(WNT - 1u);
					if (!q_ok(x) || !s_ok(x, gdg)) {
						if (!q_ok(x)) req |= RQ, rq_line = x / QSYM;
						if (!s_ok(x, gdg)) req |= RS, rs_line = x / SSYM, rs_dg = gdg;
						run = false;
						break;
					}
					const uint4 qv = qp(x), sv = sp(x), dv = neq32(qv, sv);
					const uint32_t lo = gq - x, hi = lo + glen < WNT ? lo + glen : WNT;
					for (uint32_t j = lo >> 3; 8 * j < hi; ++j) {
						const uint32_t wa = lo > 8 * j ? lo - 8 * j : 0u, wb = hi - 8 * j < 8 ? hi - 8 * j : 8u;
						const uint32_t qw = pick(qv, j), sw = pick(sv, j), dw = pick(dv, j) >> 3;
						STAT(ST_GAP_WORDS);
						// both symbols are nucleotides (bit 2 clear), src/model.c:318-320
						const uint32_t ok = symbol_range(wa, wb) & ~(qw >> 2) & ~(sw >> 2);
						const uint32_t eq = ok & ~dw, b0 = qw, b1 = qw >> 1;
						tally.same[0] += (uint32_t)__builtin_popcount(eq & ~(b0 | b1));
						tally.same[1] += (uint32_t)__builtin_popcount(eq & b0 & ~b1);
						tally.same[2] += (uint32_t)__builtin_popcount(eq & b1 & ~b0);
						tally.same[3] += (uint32_t)__builtin_popcount(eq & b0 & b1);
						for (uint32_t ne = ok & dw; ne; ne &= ne - 1) {
							const uint32_t k = (uint32_t)__builtin_ctz(ne);
							STAT(ST_SUBST);
							lds_add(&tally.hist[((((sw >> k) & 3u) << 2) | ((qw >> k) & 3u)) * 64], 1u);
						}
					}
					const uint32_t dn = hi - lo;
					gq += dn, glen -= dn;
				}
				if (run) pc = ret;
			}

			// src/process.c:191-197
			if (run && pc == PC_FINISH) {
				if (found) st.lastS = curS, st.lastQ = st.p, st.lastLen = curLen;
				st.p += curLen + 1;
				if (found && ++anchors == 1) *(uint4 *)marks[0].first = make_uint4(st.lastQ, st.lastS, st.lastLen, 0);
				if (found && anchors >= 2 && anchors < 2 + ANDI_COLD_MARKS) { // remember the state after anchors 2, 3, 4
					ColdMark *mk = marks + (anchors - 2);
					ChainState ms = st;
					ms.pad[0] = 1;
					mk->st = ms;
					uint32_t v[16];
#pragma unroll
					for (int t = 0; t < 16; ++t) v[t] = tally.hist[t * 64];
					v[0] += tally.quarter + tally.same[0], v[5] += tally.quarter + tally.same[1];
					v[10] += tally.quarter + tally.same[2], v[15] += tally.quarter + tally.rest + tally.same[3];
					uint4 *mc = (uint4 *)mk->counts;
#pragma unroll
					for (int t = 0; t < 4; ++t) mc[t] = make_uint4(v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]);
				}
				pc = PC_STEP;
			}
		}
		if (!__ballot(pc != PC_IDLE)) break;
	}

	if (!it.valid) return;
	for (uint32_t k = anchors < 2 ? 0 : anchors - 1; k < ANDI_COLD_MARKS; ++k) marks[k].st.pad[0] = 0; // unused marks
	st.pad[1] = anchors < 255 ? anchors : 255;
	a.cold_exit[slot] = st;
	a.exit_p[slot] = st.p;
	tally_finish<1>(tally);
	uint32_t out[16];
#pragma unroll
	for (int t = 0; t < 16; ++t) out[t] = tally.hist[t * 64];
	uint4 *dst = (uint4 *)(a.cold_counts + slot * 16);
#pragma unroll
	for (int t = 0; t < 4; ++t) dst[t] = make_uint4(out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
}

} // namespace

// pieces (16 bytes, 32 symbols) per line buffer: query * 16 + subject; 0 = pass A without rounds
// (scan_lane.hip's kernel, the default).  ANDI_ROUNDS=QP,SP switches the rounds on.
int andi_rounds_lines(void) {
	// measured slower than scan_lane.hip's pass A: an experiment, not the default
	const char *e = andi_knob(KNOB_ROUNDS);
	int qp = 0, sp = 0;
	if (e && sscanf(e, "%d,%d", &qp, &sp) == 2 && (qp == 2 || qp == 4 || qp == 8) && (sp == 2 || sp == 4 || sp == 8))
		return qp * 16 + sp;
	return 0;
}

hipError_t andi_launch_rounds_cold(const ScanArgs &a, hipStream_t st) {
	const dim3 grid = a.adaptive ? dim3(a.max_waves, 1) : dim3((a.total_segs + 63) / 64, a.nsub);
	switch (andi_rounds_lines()) {
		case 0x22: k_rounds_cold<2, 2><<<grid, 64, 0, st>>>(a); break;
		case 0x44: k_rounds_cold<4, 4><<<grid, 64, 0, st>>>(a); break;
		case 0x84: k_rounds_cold<8, 4><<<grid, 64, 0, st>>>(a); break;
		case 0x48: k_rounds_cold<4, 8><<<grid, 64, 0, st>>>(a); break;
		case 0x42: k_rounds_cold<4, 2><<<grid, 64, 0, st>>>(a); break;
		case 0x82: k_rounds_cold<8, 2><<<grid, 64, 0, st>>>(a); break;
		default: k_rounds_cold<8, 8><<<grid, 64, 0, st>>>(a); break;
	}
	hipError_t e = hipGetLastError();
#ifdef ANDI_LANE_STATS
	if (e == hipSuccess && andi_knob(KNOB_LANE_STATS)) {
		static const char *names[16] = {"steps", "q_line_fills", "rounds(waves)", "probes", "passes(waves)", "table", "-",
										"single", "single_undecided", "multi", "multi_cand", "search", "s_line_fills", "gap_words",
										"substitutions", "lucky_tries"};
		unsigned long long h[16];
		(void)hipStreamSynchronize(st);
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lane_stats), sizeof h);
		for (int k = 0; k < 16; ++k) fprintf(stderr, "round_stats %-16s %llu\n", names[k], h[k]);
		memset(h, 0, sizeof h);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), h, sizeof h);
	}
#endif
	return e;
}
