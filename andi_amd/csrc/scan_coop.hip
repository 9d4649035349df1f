// scan_coop.hip -- pass A of the anchor scan (dist_anchor, src/process.c:141-214) with ONE WAVEFRONT PER CHAIN.
//
// scan_lane.hip gives every lane a cold chain of its own: 64 chains per wavefront stand at ten different places
// of the step, a load instruction has 7 of 64 lanes active, and the wavefront executes the whole loop body once
// per trip (profiles/r03e_pmc.txt).  Here the 64 lanes work on ONE chain -- the cold chain of one segment,
// exactly the states and counts the sequential loop produces (what pass B stitches is unchanged) -- in two modes:
//
//   G  generic steps, one after the other: lucky_anchor (src/process.c:82-100) compares 2048 symbols per round
//      trip (64 lanes x 32 symbols), anchor() (113-123) probes the next 64 query positions at once (the chain
//      then hops through the answers: each step lands at most 64 positions on), the accounting of 157-190 counts
//      a gap with all lanes.  The chain is in this mode until it has found a LUCKY anchor: two anchors in a row on
//      one diagonal.
//   W  a window of W = 2048 NCH query symbols along that diagonal (NCH = 5 on segments of 32768 symbols and more), streamed
//      with fully coalesced loads -- both texts bit-sliced, 12 bytes per 32 symbols (LogDet / ANI: 4-bit symbols) -- and
//      turned into one bit per position (query symbol != subject symbol); every mismatch is counted at once as a
//      single-position gap, by kind (what the chain does not reach is taken back).  Behind a mismatch that is
//      followed by >= threshold equal symbols the next step is a lucky anchor whose outcome the bits alone decide
//      ("easy").  A mismatch followed by a shorter run but preceded by a long one is a HEAD: the chain arrives
//      there in a known (canonical) state, so the walks from ALL heads of the window -- probe, step, probe ... until
//      an anchor on the diagonal is found -- are independent and are taken by the lanes in parallel, speculatively
//      (a lane that is done takes the next head).  The chain then hops from head to head; counts follow from the
//      gap positions (all mismatches, and the stretch between a head and the anchor its walk lands on) and the
//      anchors between them.  tests/coop_model.py restates this decomposition on the CPU and checks it against the
//      plain loop; tests/test_coop_gpu.py checks this kernel against k_lane_cold slot by slot.
//
// Layout: one segment length for the call (slot = subject * total_segs + segment), wavefront w of subject
// blockIdx.y takes segment w.  LogDet and ANI count every anchor's nucleotides (EXACT: src/model.c:256-278).
#include "lane_chain.h"
#include "knobs.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

// (LDS bounds the window at eight wavefronts per SIMD: 5120 bytes each.  Five chunks with 112 heads are 5108 bytes; fewer heads per chunk
// cost more than the longer window gains -- at four chunks 16 per chunk: + 4 %, 20: + 1 %, 32: - 0.5 %; at five 100 heads drop 6 % of
// them (a dropped head ends its window), 112 with the list of stretches counted nowhere at 12 instead of 32: - 2 %; profiles/r07_coop/README.md)
#define COOP_HCAP (NCH <= 2 ? 64u : NCH == 5 ? 112u : 24u * NCH) /* heads of a window that are walked (more: the window ends at the first one dropped) */
constexpr int COOP_WAVES = 1; // wavefronts per block: single wavefronts find a place on a CU the moment one leaves (bench set 5.39 -> 5.24 ms against blocks of two, 5.61 with four); seven per SIMD (72 registers): 4.94
constexpr uint32_t COOP_KCAP = 12; // stretches of a window that are counted nowhere (more: the window ends before the next one)
// (routed calls hand a pair back when one of its segments needs more generic steps than ScanArgs.route_giveup -- 1024: long stretches without homology that the sampling missed (clean sets: <= 45 steps; an island of 20 kbp: some 1500).  Grinding through them instead -- the limit at 16384 -- took the structured set's pass A from 9.4 to 28 ms for 9 of its 812 pairs, and its passes B/C from 6.9 to 15 ms: their true chains cross the islands on long segments, one lane each)
constexpr uint32_t COOP_TRIAL_LCP = 16;  // routed calls: a match followed through more rounds of 2048 symbols than this is longer than its segment
constexpr uint32_t COOP_PARK = 16; // parked lanes (their probe needs lane_probe) that are served together
constexpr uint32_t COOP_MAX_X = 3;  // anchors off the window's diagonal a walk follows before it gives up
constexpr uint32_t NOPOS = 0xffffffffu;

// -DANDI_COOP_STATS: how the kernel's work splits (diagnostic builds only; ANDI_COOP_STATS=1 prints)
#ifdef ANDI_COOP_STATS
__device__ unsigned long long g_coop_stats[24];
#define CSTAT(k, v)                                                              \
	do {                                                                         \
		if (__lane_id() == 0) atomicAdd(&g_coop_stats[k], (unsigned long long)(v)); \
	} while (0)
// wave-cycles (s_memtime) spent per phase: TICK(t) starts the clock, TOCK(t, phase) charges the time since to `phase`
__device__ unsigned long long g_coop_cycles[8];
#define TICK(t) unsigned long long t = __builtin_readcyclecounter()
#define TOCK(t, ph)                                                                        \
	do {                                                                                   \
		const unsigned long long now_ = __builtin_readcyclecounter();                      \
		if (__lane_id() == 0) atomicAdd(&g_coop_cycles[ph], now_ - t);                     \
		t = now_;                                                                          \
	} while (0)
#else
#define CSTAT(k, v) ((void)0)
#define TICK(t) ((void)0)
#define TOCK(t, ph) ((void)0)
#endif
#ifdef ANDI_COOP_STATS
__device__ unsigned int g_coop_max[4];
__device__ unsigned long long g_coop_trip_lanes[2][8]; // trips by lanes at work: 0, 1, 2-3, 4-7, 8-15, 16-31, 32-63, 64 (plain trips, service trips)
#endif
enum { PH_G, PH_STREAM, PH_HEADS, PH_WALKS, PH_HOPS, PH_STRETCH, PH_FINAL };
enum { CS_SEGMENTS, CS_G_STEPS, CS_BLOCKS, CS_LCP, CS_WINDOWS, CS_MOVED, CS_HEADS, CS_TRIPS, CS_LANE_STEPS, CS_PROBES, CS_ONPATH,
	   CS_HOPS, CS_G_GAPS, CS_COVERED, CS_DROPPED, CS_X, CS_LCP_ROUNDS, CS_NODES, CS_SERVICE, CS_SERVICE_LANES, CS_WHY_PRE, CS_WHY_MULTI, CS_WHY_LONG, CS_WHY_OTHER };

// how a head's walk ended
enum : uint32_t { W_OK = 1, W_OPEN = 2, W_EXIT = 3, W_BREAK = 4, W_STATUS = 7u, W_HADX = 8u, W_LUCKY = 16u, W_NX_SHIFT = 8, W_ONPATH = 1u << 12 };

template <int NCH>
struct CoopLds {
	uint32_t mbits[64 * NCH + 4]; // mismatch bits of the window, bit (x - wbase); the words behind it stay 0: nothing known there
	union {
		uint32_t q2[128 * NCH + 4]; // the window's query symbols as 2-bit codes, 16 per word, symbol k of a word at bits 2k, 2k + 1: for the walks;
		uint32_t ebits[64 * NCH];   // once they are done: the stretches behind the heads the chain came by, [head, landing): counted as gaps -- except
	};
	uint32_t kpos[COOP_KCAP];     // those that start at one of these positions (the walk met anchors off the diagonal): counted nowhere
	uint32_t nhadx;               // walks of the window that ended that way (at most COOP_KCAP: the others give up)
	union {
		struct {
			uint32_t hend[COOP_HCAP]; // from the walk: length of the anchor it landed on (W_LUCKY: not known yet); then: where that anchor ends
			uint16_t ha[COOP_HCAP];   // landing position of the head's walk - wbase
			uint16_t hflag[COOP_HCAP];
			uint16_t hpos[COOP_HCAP]; // the head's position - wbase
		};
		struct {
			uint32_t pl[64], pp[64]; // mode G (no window open): the block of probes, length | unique << 31, position
		};
	};
	uint32_t hist[16];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { // a wave-uniform value: into a scalar register
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// the lanes of the wavefront exchange data through LDS: order this lane's accesses around the hand-over
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) {
	(void)__hip_atomic_fetch_or((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// inclusive prefix sums / prefix maxima over the 64 lanes in six DPP steps (shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then
// lane 15 of a row into the next row, lane 31 into the upper half; lanes without a source take 0: the identity of both for unsigned
// values) -- six vector instructions where six __shfl_up were six LDS-crossbar round trips with their address arithmetic
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
	return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
	uint32_t o;
	o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false), v = o > v ? o : v;
	o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false), v = o > v ? o : v;
	o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false), v = o > v ? o : v;
	o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false), v = o > v ? o : v;
	o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false), v = o > v ? o : v;
	o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false), v = o > v ? o : v;
	return v;
}

// Lane l's value for a wave-uniform l: v_readlane; the value of the lane below / above (lane 0 / 63: its own): one DPP move.
// (__shfl and its kin are ds_bpermute: a trip through the LDS crossbar each.)
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t l) {
	return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)uni(l));
}
__device__ __forceinline__ uint32_t lane_below(uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false); // wave_shr:1
}
__device__ __forceinline__ uint32_t lane_above(uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false); // wave_shl:1
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { // (all lanes active) the sum over the wavefront, wave-uniform
	return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(v), 63);
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
	for (int d = 32; d; d >>= 1) {
		const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
		v = o < v ? o : v;
	}
	return v;
}

// 8 nibble flags (bit 4k + 3 of a word: symbols k differ) -> 8 bits
__device__ __forceinline__ uint32_t squeeze8(uint32_t x) {
	x = (x >> 3) & ONES;
	x = (x | (x >> 3)) & 0x03030303u;
	x = (x | (x >> 6)) & 0x000f000fu;
	return (x | (x >> 12)) & 0xffu;
}

__device__ __forceinline__ uint32_t squeeze32(const uint4 &d) {
	return squeeze8(d.x) | (squeeze8(d.y) << 8) | (squeeze8(d.z) << 16) | (squeeze8(d.w) << 24);
}

// 32 symbols of the subject from offset sa on (any sa >= -32; at and beyond the text's end: NUL, the padding's symbol)
__device__ __forceinline__ uint4 ld_subject_guarded(const PairCtx &c, int64_t sa) {
	if (sa >= (int64_t)c.E.n) return make_uint4(0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u);
	return ld_subject(c, (int32_t)sa);
}

struct Chain { // the chain of the wavefront: everything wave-uniform
	ChainState st;
	uint32_t quarter, rest; // model_count_equal's split (src/model.c:247-253), folded into the histogram at the end
	uint32_t anchors;
	uint32_t marked; // the mark behind the 2nd anchor has been written
	uint32_t blk_base; // the block of probes in LDS answers positions blk_base ... blk_base + 63 (NOPOS: none)
	uint32_t stuck;    // the last window ended at a head whose walk gave up (W_BREAK: the chain leaves the diagonal there): the next step is mode G's --
					   // a window opened at that head again would stream, walk all its heads and not move (one such window per contig end, round 6)
};

// lcp(Q + p, S + s, maxlen) (src/process.c:59-65) with all lanes: 2048 symbols per round trip
// (max_rounds: give up after that many round trips and return NOPOS)
__device__ __forceinline__ uint32_t coop_lcp(const PairCtx &c, uint32_t p, uint32_t s, uint32_t maxlen, uint32_t max_rounds = ~0u) {
	const uint32_t lane = __lane_id(), pe = p & ~1u, skip = p & 1u;
	const int64_t dg = (int64_t)s - (int64_t)p;
	CSTAT(CS_LCP, 1);
	for (uint32_t base = 0;; base += 64 * WNT) {
		const uint32_t x0 = pe + base + WNT * lane;
		CSTAT(CS_LCP_ROUNDS, 1);
		uint32_t f = 0; // first differing symbol of the lane's piece (a piece at or beyond the query's end: at once)
		if (x0 < c.qlen) {
			const uint4 d = neq32(ld_query(c, x0), ld_subject_guarded(c, (int64_t)x0 + dg));
			f = first_from(d, base == 0 && lane == 0 ? skip : 0u);
		}
		const uint64_t hit = __ballot(f < WNT);
		if (hit) {
			const uint32_t l = (uint32_t)__builtin_ctzll(hit);
			const uint32_t fl = lane_read(f, l);
			const uint32_t len = uni(pe + base + WNT * l + fl - p);
			return len < maxlen ? len : maxlen;
		}
		if (base + 64 * WNT - skip >= maxlen) return maxlen;
		if (base / (64 * WNT) + 1 >= max_rounds) return NOPOS;
	}
}

// model_count (src/model.c:309-337) of Q[q..q+len) against S[s..s+len) with all lanes
__device__ __forceinline__ void coop_count_gap(const PairCtx &c, lds_u32 *hist, uint32_t q, uint32_t s, uint32_t len) {
	const uint32_t lane = __lane_id(), qe = q & ~1u, end = q + len;
	const int64_t dg = (int64_t)s - (int64_t)q;
	for (uint32_t base = qe; base < end; base += 64 * WNT) {
		const uint32_t x0 = base + WNT * lane;
		if (x0 >= end) continue;
		const uint4 qv = ld_query(c, x0), sv = ld_subject_guarded(c, (int64_t)x0 + dg);
		const uint32_t lo = q > x0 ? q - x0 : 0u, hi = end - x0 < WNT ? end - x0 : WNT;
		for (uint32_t j = lo >> 3; 8 * j < hi; ++j) {
			const uint32_t a = lo > 8 * j ? lo - 8 * j : 0u, b = hi - 8 * j < 8 ? hi - 8 * j : 8u;
			const uint32_t qw = pick(qv, j), sw = pick(sv, j), dw = neq8(qw, sw) >> 3;
			const uint32_t ok = symbol_range(a, b) & ~(qw >> 2) & ~(sw >> 2); // both nucleotides, src/model.c:318-320
			const uint32_t eq = ok & ~dw, b0 = qw, b1 = qw >> 1;
			if (eq) {
				lds_add(&hist[0], (uint32_t)__builtin_popcount(eq & ~(b0 | b1)));
				lds_add(&hist[5], (uint32_t)__builtin_popcount(eq & b0 & ~b1));
				lds_add(&hist[10], (uint32_t)__builtin_popcount(eq & b1 & ~b0));
				lds_add(&hist[15], (uint32_t)__builtin_popcount(eq & b0 & b1));
			}
			for (uint32_t ne = ok & dw; ne; ne &= ne - 1) {
				const uint32_t k = (uint32_t)__builtin_ctz(ne);
				lds_add(&hist[(((sw >> k) & 3u) << 2) | ((qw >> k) & 3u)], 1u);
			}
		}
	}
}

// model_count_equal for LogDet and ANI (src/model.c:256-278): the nucleotides of the anchor Q[q..q+len), with all lanes
// (add = false: an anchor that was counted and should not have been is taken back)
__device__ __forceinline__ void coop_count_anchor(const PairCtx &c, lds_u32 *hist, uint32_t q, uint32_t len, bool add = true) {
	const uint32_t lane = __lane_id(), qe = q & ~1u, end = q + len;
	uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
	for (uint32_t base = qe; base < end; base += 64 * WNT) {
		const uint32_t x0 = base + WNT * lane;
		if (x0 >= end) continue;
		const uint4 qv = ld_query(c, x0);
		const uint32_t lo = q > x0 ? q - x0 : 0u, hi = end - x0 < WNT ? end - x0 : WNT;
		for (uint32_t j = lo >> 3; 8 * j < hi; ++j) {
			const uint32_t a = lo > 8 * j ? lo - 8 * j : 0u, b = hi - 8 * j < 8 ? hi - 8 * j : 8u;
			const uint32_t qw = pick(qv, j), ok = symbol_range(a, b) & ~(qw >> 2), b0 = qw, b1 = qw >> 1;
			n0 += (uint32_t)__builtin_popcount(ok & ~(b0 | b1)), n1 += (uint32_t)__builtin_popcount(ok & b0 & ~b1);
			n2 += (uint32_t)__builtin_popcount(ok & b1 & ~b0), n3 += (uint32_t)__builtin_popcount(ok & b0 & b1);
		}
	}
	n0 = wave_sum(n0), n1 = wave_sum(n1), n2 = wave_sum(n2), n3 = wave_sum(n3);
	if (lane == 0) {
		if (add) lds_add(&hist[0], n0), lds_add(&hist[5], n1), lds_add(&hist[10], n2), lds_add(&hist[15], n3);
		else lds_add(&hist[0], 0u - n0), lds_add(&hist[5], 0u - n1), lds_add(&hist[10], 0u - n2), lds_add(&hist[15], 0u - n3);
	}
}

// first mismatch bit of the window at or after position x (x >= wbase), NOPOS if the window shows none
template <int NCH>
__device__ __forceinline__ uint32_t coop_next_mismatch(const CoopLds<NCH> &L, uint32_t wbase, uint32_t x) {
	const uint32_t lane = __lane_id(), o = x - wbase;
	for (uint32_t w0 = o >> 5; w0 < 64 * NCH; w0 += 64) {
		const uint32_t w = w0 + lane;
		uint32_t v = w < 64 * NCH ? L.mbits[w] : 0u;
		if (w == (o >> 5)) v &= ~0u << (o & 31u);
		const uint64_t hit = __ballot(v != 0);
		if (hit) {
			const uint32_t l = (uint32_t)__builtin_ctzll(hit);
			const uint32_t vl = lane_read(v, l);
			return uni(wbase + 32 * (w0 + l) + (uint32_t)__builtin_ctz(vl));
		}
	}
	return NOPOS;
}

// last mismatch bit of the window before position x (x > wbase), NOPOS if none
template <int NCH>
__device__ __forceinline__ uint32_t coop_prev_mismatch(const CoopLds<NCH> &L, uint32_t wbase, uint32_t x) {
	const uint32_t lane = __lane_id(), o = x - wbase; // bits 0 .. o - 1
	if (o == 0) return NOPOS;
	const uint32_t top = (o - 1) >> 5;
	for (int32_t w0 = (int32_t)top; w0 >= 0; w0 -= 64) {
		const int32_t w = w0 - (int32_t)lane;
		uint32_t v = w >= 0 && w < 64 * NCH ? L.mbits[w] : 0u;
		if (w == (int32_t)top && ((o - 1) & 31u) != 31u) v &= (2u << ((o - 1) & 31u)) - 1u;
		const uint64_t hit = __ballot(v != 0);
		if (hit) {
			const uint32_t l = (uint32_t)__builtin_ctzll(hit);
			const uint32_t vl = lane_read(v, l);
			return uni(wbase + 32 * (uint32_t)(w0 - (int32_t)l) + 31u - (uint32_t)__builtin_clz(vl));
		}
	}
	return NOPOS;
}

// 8 nibbles -> 8 x 2 bits, first symbol in the low bits (the symbols must be nucleotides)
__device__ __forceinline__ uint32_t squeeze_codes(uint32_t x) {
	x &= 0x33333333u;
	x = (x | (x >> 2)) & 0x0f0f0f0fu;
	x = (x | (x >> 4)) & 0x00ff00ffu;
	return (x | (x >> 8)) & 0x0000ffffu;
}

// anchor() (src/process.c:113-123) for a walk's position p inside the window, the part that ONE dependent load
// answers: the K-mer and the symbols behind it come from the window's 2-bit codes in LDS (no load of the query), the
// table's entry settles an absent K-mer, and one that occurs once unless the match is longer than the 13 nucleotides
// its extended entry carries.  Returns false for everything else -- K-mers with several occurrences, long matches,
// windows with separators: the lane then waits until enough lanes of its wavefront need lane_probe, and they take it
// together (a path one lane in ten takes is otherwise executed on nine trips in ten).
// (on_diag: the K-mer occurs once, at p + dg -- on the window's diagonal: the match is the run of equal symbols the bits show)
// (multi_x, multi_n, multi_q: the K-mer occurs multi_n <= 4 times, at SA[multi_x ...]; multi_q = the 16 symbols behind it)
// (the part behind the query symbols: lo = the 16 symbols from p on as 2-bit codes, first in the low bits; hi = the next 16)
__device__ __forceinline__ bool coop_probe_codes(const PairCtx &c, uint32_t p, uint32_t sd, uint32_t lo, uint32_t hi, Probe &r, bool &on_diag,
												 uint32_t &multi_x, uint32_t &multi_n, uint32_t &multi_q) {
	const EsaG &E = c.E;
	const uint32_t K = (uint32_t)E.deepK, qrem = c.qlen - p;
#ifdef ANDI_COOP_STATS
#define WHY(k) atomicAdd(&g_coop_stats[k], 1ull)
#else
#define WHY(k) ((void)0)
#endif
	uint32_t y = __brev(lo); // first symbol on top, the bits of a pair swapped
	y = ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
	const uint32_t code = y >> (32 - 2 * K);
	const uint32_t behind = (uint32_t)((((uint64_t)hi << 32) | lo) >> (2 * K)); // the 16 symbols behind the K-mer
	const uint64_t raw = ld_u64_unaligned((g_u8p)(E.deep + code));
	const uint32_t x = (uint32_t)raw, ty = (uint32_t)(raw >> 32), kind = ty & 3u;
	if (kind == DEEP_FINAL) {
		r.len = ty >> 8, r.unique = (ty >> 2) & 1u, r.pos = 0;
		return !(r.unique && r.len >= (uint32_t)E.thr); // (only texts so short that K >= thr: the position is one more load)
	}
	if (kind != DEEP_SINGLE) {
		if (kind == DEEP_MULTI) WHY(CS_WHY_MULTI); else WHY(CS_WHY_OTHER);
		if (kind == DEEP_MULTI && (ty >> 8) < 4) multi_x = x, multi_n = (ty >> 8) + 1, multi_q = behind;
		return false;
	}
	if (x == p + sd) {
		on_diag = true;
		return true;
	}
	if (!E.deep_ext) { // (plain entries: the occurrence has to be looked at -- with the parked lanes, its position is known)
		multi_x = x, multi_n = 0x80000001u, multi_q = behind;
		return false;
	}
	const uint32_t nval = (ty >> 2) & 15u, ext = ty >> 6;
	const uint32_t diff = (behind ^ ext) & 0x03ffffffu;
	const uint32_t m = diff ? (uint32_t)__builtin_ctz(diff) >> 1 : 16u; // first differing nucleotide
	const uint32_t lim = nval < qrem - K ? nval : qrem - K;
	// how many symbols an entry can hold at all: 13 in the long form, what the sorter's keys had behind the K-mer in the
	// short one (andi_dev.h); an entry that holds fewer says that the text has no nucleotide there
	const uint32_t room = E.deep_ext == 2 ? (16u - K < 4u ? 16u - K : 4u) : 13u;
	r.unique = true, r.pos = x;
	if (m < lim) {
		r.len = K + m; // settled by the entry
	} else if (K + lim >= qrem) {
		r.len = qrem; // the query ends inside the match
	} else if (lim == nval && nval < room) {
		r.len = K + lim; // the text has a separator (or its end) there, the query a nucleotide
	} else {
		WHY(CS_WHY_LONG);
		return false; // all that could be compared matches: the occurrence has to be followed
	}
	return true;
}

// lane_probe with a window of its own, for the walks' rare probes that nothing shorter answers (0.4 % of them: window edges,
// separators, matches deeper than 16 symbols off the diagonal)
// (as a real call -- noinline -- the walks' loop shrinks from 4400 to 1500 instructions, but the call's register convention costs more
// than the code's size: k_coop_cold 4.12 -> 4.32 ms, k_pool_cold 4.37 -> 5.07; profiles/r07_pool/readlane.txt)
__device__ __forceinline__ Probe coop_generic_probe(const PairCtx &c, uint32_t pp) {
	LWin w;
	w.q0 = EMPTY, w.dg = NO_DIAG;
	return lane_probe(c, pp, w);
}

template <int NCH>
__device__ __forceinline__ bool coop_probe_fast(const PairCtx &c, const CoopLds<NCH> &L, uint32_t wbase, bool clean, uint32_t p, uint32_t sd, Probe &r, bool &on_diag,
												uint32_t &multi_x, uint32_t &multi_n, uint32_t &multi_q) {
	on_diag = false, multi_n = 0;
	const uint32_t o = p - wbase;
	// (a window with a separator -- joined contigs: every probe of it goes the long way.  Per-word flags in LDS were built and measured
	// in round 6: 1 % faster on genomes of 100 contigs, 1.7 % slower on whole ones -- profiles/r07_pool/join_ab.txt; k_pool_cold, whose
	// windows are sixteen times as long, keeps a flag per head's record)
	if (!(clean && o + 32 <= 2048 * NCH && p + 32 <= c.qlen)) {
		WHY(CS_WHY_PRE);
		return false;
	}
	const uint32_t j = o >> 4, sh = 2 * (o & 15u);
	const uint32_t w0 = L.q2[j], w1 = L.q2[j + 1], w2 = L.q2[j + 2];
	return coop_probe_codes(c, p, sd, __builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh), r, on_diag, multi_x, multi_n, multi_q);
}

// The probe of a parked lane whose K-mer occurs n <= 4 times (at SA[x ...]): the longest match is the best of the
// occurrences' own matches, unique iff one attains it (as lane_probe).  The occurrence on the window's diagonal, if
// there is one, is the run of equal symbols the bits show (*diag_run; *diag_seen: a mismatch of the window ends it);
// the others are compared 16 symbols deep, two per round trip.  false: a match goes deeper than that -- lane_probe's.
// The probe of a K-mer that occurs n <= 4 times (at SA[x ...]) from what the device sorter left beside the suffix array
// (EsaDev.R2): the nucleotides behind every occurrence's K-mer, `room` of them -- ONE load of eight bytes where the suffix
// array's entries and the text behind each of them were three or more.  An occurrence whose symbols differ from the query's
// within them (or whose text ends there) is settled; the diagonal's occurrence (there is one iff the diagonal's run is at
// least K long) is the run the bits show.  false: something is left open -- an occurrence off the diagonal that matches all
// `room` symbols (one in 256) -- or the records are not there: coop_probe_multi's long way.
__device__ __forceinline__ bool coop_probe_r2(const PairCtx &c, uint32_t p, uint32_t sd, uint32_t x, uint32_t n, uint32_t behind,
											  uint32_t diag_run, bool diag_seen, Probe &r, bool &on_diag_long) {
	const EsaG &E = c.E;
	const uint32_t K = (uint32_t)E.deepK, qrem = c.qlen - p;
	on_diag_long = false;
	if (!E.R2 || (n & 0x80000000u) || K + 4 > qrem) return false;
	{
		const uint32_t room = 16u - K < 4u ? 16u - K : 4u;
		const uint64_t w = ld_u64_unaligned((g_u8p)(E.R2 + x));
		const bool diag_in = diag_run >= K; // the K-mer at p occurs on the diagonal: one of these occurrences
		uint32_t open_n = 0, bestLen = 0, bestCnt = 0, bestIdx = 0;
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t e16 = (uint32_t)(w >> (16 * i)) & 0xffffu, nval = e16 >> 8;
			const uint32_t diff = (behind ^ e16) & ((1u << (2 * nval)) - 1u);
			const uint32_t m = diff ? (uint32_t)__builtin_ctz(diff) >> 1 : nval;
			if (m == nval && nval == room) {
				++open_n; // matches as far as the record goes
				continue;
			}
			const uint32_t len = K + m; // a mismatch, or the text has no nucleotide there (m == nval < room)
			if (len > bestLen) bestLen = len, bestCnt = 1, bestIdx = i;
			else if (len == bestLen) ++bestCnt;
		}
		const bool diag_open = diag_in && (diag_run >= K + room || !diag_seen);
		if (open_n == (diag_open ? 1u : 0u)) { // nothing open but the diagonal's own occurrence
			if (diag_open) { // ... which beats every settled one
				r.unique = true, r.pos = p + sd;
				r.len = diag_seen ? diag_run : 0x40000000u;
				on_diag_long = !diag_seen;
				return true;
			}
			r.len = bestLen, r.unique = bestCnt == 1;
			if (diag_in && bestLen == diag_run && bestCnt == 1) {
				r.pos = p + sd; // the one occurrence of that length is the diagonal's
			} else {
				r.pos = (r.unique && bestLen >= (uint32_t)E.thr) ? (uint32_t)E.SA[x + bestIdx] : 0u; // (an anchor off the diagonal: rare)
			}
			return true;
		}
	}
	return false;
}

template <bool USE_R2 = true>
__device__ __forceinline__ bool coop_probe_multi(const PairCtx &c, uint32_t p, uint32_t sd, uint32_t x, uint32_t n, uint32_t behind,
												 uint32_t diag_run, bool diag_seen, Probe &r, bool &on_diag_long) {
	const EsaG &E = c.E;
	const uint32_t K = (uint32_t)E.deepK, qrem = c.qlen - p;
	on_diag_long = false;
	if (USE_R2 && coop_probe_r2(c, p, sd, x, n, behind, diag_run, diag_seen, r, on_diag_long)) return true;
	uint4 pos4 = make_uint4(x, 0, 0, 0); // (n & 0x80000000: x is the one occurrence's position itself)
	if (!(n & 0x80000000u)) pos4 = ld_u128_unaligned((g_u8p)(E.SA + x)); // (SA is padded by eight entries)
	n &= 0x7fffffffu;
	uint32_t bestLen = 0, bestCnt = 0, bestPos = 0;
	bool deep = false;
	for (uint32_t i = 0; i < n; i += 2) { // two occurrences per round trip
		const uint32_t pa = i == 0 ? pos4.x : pos4.z, pb = i == 0 ? pos4.y : pos4.w;
		const uint4 sa = ld_subject(c, (int32_t)(pa + K));
		uint4 sb = sa;
		if (i + 1 < n) sb = ld_subject(c, (int32_t)(pb + K));
		for (uint32_t h = 0; h < 2 && i + h < n; ++h) {
			const uint4 sv = h ? sb : sa;
			const uint32_t ps = h ? pb : pa;
			uint32_t len;
			if (ps == p + sd) { // the diagonal's occurrence
				len = diag_run;
				if (!diag_seen) len = 0x40000000u, on_diag_long = true; // (longer than anything 16 symbols can show)
			} else {
				const uint32_t bad_lo = sv.x & 0x44444444u, bad_hi = sv.y & 0x44444444u; // separators, the text's end
				const uint32_t stop = bad_lo ? (uint32_t)__builtin_ctz(bad_lo) >> 2 : (bad_hi ? 8u + ((uint32_t)__builtin_ctz(bad_hi) >> 2) : 16u);
				const uint32_t diff = behind ^ (squeeze_codes(sv.x) | (squeeze_codes(sv.y) << 16));
				uint32_t m = diff ? (uint32_t)__builtin_ctz(diff) >> 1 : 16u;
				if (stop < m) m = stop;
				if (m >= 16 && K + 16 < qrem) deep = true;
				len = K + m;
				if (len > qrem) len = qrem;
			}
			if (len > bestLen)
				bestLen = len, bestCnt = 1, bestPos = ps;
			else if (len == bestLen)
				++bestCnt;
		}
	}
	r.len = bestLen, r.unique = bestCnt == 1, r.pos = bestPos;
	return !deep;
}

// the record of an anchor (scan.h: the cold chain's first anchor, the state and counts right after its second)
template <class LDS>
__device__ __forceinline__ void coop_note_anchor(const ScanArgs &a, size_t slot, Chain &ch, const LDS &L) {
	const uint32_t lane = __lane_id();
	ColdMark *m = a.marks + slot * ANDI_COLD_MARKS;
	++ch.anchors;
	if (ch.anchors == 1 && lane == 0) *(uint4 *)m->first = make_uint4(ch.st.lastQ, ch.st.lastS, ch.st.lastLen, 0);
	if (ch.anchors == 2) {
		ch.marked = 1;
		if (lane == 0) {
			ChainState ms = ch.st;
			ms.pad[0] = 1, ms.pad[1] = ms.pad[2] = 0;
			m->st = ms;
		}
		if (lane < 16) {
			uint32_t v = L.hist[lane];
			if (lane == 0 || lane == 5 || lane == 10 || lane == 15) v += ch.quarter;
			if (lane == 15) v += ch.rest;
			m->counts[lane] = v;
		}
	}
}

// What an anchor at subject offset curS found at query offset st.p does to the counts (src/process.c:157-190)
template <bool EXACT, class LDS>
__device__ __forceinline__ void coop_account(const PairCtx &c, Chain &ch, LDS &L, uint32_t curS) {
	ChainState &st = ch.st;
	const uint32_t endS = st.lastS + st.lastLen, endQ = st.lastQ + st.lastLen;
	auto count_last = [&]() { // model_count_equal of the last anchor
		if constexpr (EXACT) coop_count_anchor(c, (lds_u32 *)L.hist, st.lastQ, st.lastLen);
		else ch.quarter += st.lastLen >> 2, ch.rest += st.lastLen & 3u;
	};
	if (curS > endS && st.p - endQ == curS - endS && (curS < c.border) == (st.lastS < c.border)) {
		count_last();
		coop_count_gap(c, (lds_u32 *)L.hist, endQ, endS, st.p - endQ);
		CSTAT(CS_G_GAPS, 1);
		st.lwra = 1;
	} else {
		if (st.lwra || st.lastLen >= 2 * c.thr) count_last();
		st.lwra = 0;
	}
}

#include "coop_pool.h" // (mode P, and what mode W takes from it: the texts bit-sliced, the substitutions counted without a loop)

// The twelve counters of all lanes into the 4 x 4 counts in LDS (cell = subject nucleotide << 2 | query nucleotide, src/model.c:309-337), added or
// taken back.  A window of mode W has at most 32 NCH <= 256 mismatches per lane and kind: two counters to a register, six sums over the lanes;
// the sums go to lanes 0 ... 11, which add them to their cells in ONE instruction (subst_flush's conditional add per counter by lane 0 cost as
// much as its scans; every lane adding its own counters to the twelve cells -- 64 adds to one address each -- took the kernel from 3.7 to 4.3 ms).
__device__ __forceinline__ void subst_flush_pairs(const SubstAcc &acc, lds_u32 *hist, bool add) {
	const uint32_t lane = __lane_id();
	uint32_t w = 0; // lane k < 6: the sums of counters k (low half) and k + 6 (high half)
#pragma unroll
	for (int k = 0; k < 6; ++k) {
		const uint32_t sum = wave_sum(acc.n[k] | (acc.n[k + 6] << 16));
		w = lane == (uint32_t)k ? sum : w;
	}
	const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x116, 0xf, 0xf, false); // row_shr:6: lane k + 6 sees lane k's
	const uint32_t v = lane < 6 ? w & 0xffffu : up >> 16;
	const uint32_t qn = lane / 3, r = lane - 3 * qn, sn = r + (r >= qn ? 1u : 0u);
	if (lane < 12 && v) lds_add(&hist[(sn << 2) | qn], add ? v : 0u - v);
}

// ------------------------------------------------------------------ mode W
// The chain stands at a canonical state of diagonal dg: st.p = e0 + 1 behind the anchor [lastQ, e0), e0 a mismatch
// of the diagonal.  Returns true if the chain moved; st is a genuine loop-top state either way.
template <int NCH, bool EXACT>
__device__ __forceinline__ bool coop_window(const ScanArgs &a, const PairCtx &c, Chain &ch, CoopLds<NCH> &L, uint32_t end) {
	static_assert(64 * 32 * NCH < 65536 && 2048 * NCH <= 65536, "two 16-bit sums to a register (subst_flush_pairs, the stretches' equal symbols); offsets into a window are 16 bits");
	const uint32_t lane = __lane_id(), thr = c.thr, n = (uint32_t)c.E.n;
	ChainState &st = ch.st;
	const int64_t dg = (int64_t)st.lastS - (int64_t)st.lastQ;
	const uint32_t sd = st.lastS - st.lastQ; // subject position of window position x: x + sd (x + dg >= lastS >= 0: 32 bits do)
	const uint32_t e0 = st.lastQ + st.lastLen;
	const uint32_t wbase = e0 & ~31u;
	constexpr uint32_t W = 2048 * NCH;

	TICK(tph);
	// ---- the bits: position x of the window against subject position x + dg; the query's symbols as 2-bit codes
	uint32_t dirty = 0; // a query symbol of the window that is no nucleotide ('!' of joined contigs): the walks then read the query itself
	lds_u32 *hist = (lds_u32 *)L.hist;
	// RAW / JC / Kimura (round 6): the texts are streamed BIT-SLICED (EsaDev.P, ScanArgs.qplanes: three words per 32 symbols), as k_pool_cold's
	// sweep S does -- the mismatch bits are three instructions per word, and every mismatch of the window is counted as a single-position gap
	// at once, by kind, without a loop (subst_count): what the chain turns out not to reach, and the stretches that are counted nowhere, are
	// taken back below.  Until then the counting pass fetched both texts a second time for every word with a mismatch and every stretch's lane
	// walked its text symbol by symbol: 22 % of the kernel (profiles/r07_coop/README.md).  LogDet / ANI keep the 4-bit stream.
	// the mismatches of [from, end of the window) out of the counts again
	auto take_back = [&](uint32_t from) {
		if constexpr (!EXACT) {
			if ((from - wbase) / 2048u >= (uint32_t)NCH) return;
			SubstAcc acc;
#pragma unroll
			for (int k = 0; k < 12; ++k) acc.n[k] = 0;
			for (uint32_t t = (from - wbase) / 2048u; t < (uint32_t)NCH; ++t) {
				const uint32_t x0 = wbase + 2048 * t + WNT * lane;
				if (x0 >= c.qlen || x0 + WNT <= from) continue;
				const Planes qv = ld_query_planes(c, x0), sv = ld_subject_planes(c, (int64_t)x0 + dg);
				uint32_t mc = ((qv.b0 ^ sv.b0) | (qv.b1 ^ sv.b1) | (qv.b2 ^ sv.b2)) & ~(qv.b2 | sv.b2);
				if (c.qlen - x0 < WNT) mc &= ~(~0u << (c.qlen - x0));
				if (from > x0) mc &= ~0u << (from - x0);
				subst_count(acc, mc, qv, sv);
			}
			subst_flush_pairs(acc, hist, false);
		}
	};
	if constexpr (!EXACT) {
		SubstAcc acc;
#pragma unroll
		for (int k = 0; k < 12; ++k) acc.n[k] = 0;
		const uint32_t sh = (uint32_t)((int64_t)(wbase + WNT * lane) + dg) & 31u; // (a lane's subject offset keeps its low five bits from chunk to chunk)
#pragma unroll 2
		for (int ck = 0; ck < NCH; ++ck) {
			const uint32_t x0 = wbase + 2048 * ck + WNT * lane;
			uint32_t m = ~0u, mc = 0; // positions at and beyond the query's end: lcp() stops there; mc: the mismatches that are counted
			uint2 codes = make_uint2(0, 0);
			if (x0 < c.qlen) {
				RawPlanes raw;
				ld_raw_planes(c, x0, (int64_t)x0 + dg, raw);
				Planes qv, sv;
				planes_of(raw, sh, qv, sv);
				m = (qv.b0 ^ sv.b0) | (qv.b1 ^ sv.b1) | (qv.b2 ^ sv.b2);
				mc = m & ~(qv.b2 | sv.b2); // both nucleotides (src/model.c:318-320)
				codes = make_uint2(spread16(qv.b0) | (spread16(qv.b1) << 1), spread16(qv.b0 >> 16) | (spread16(qv.b1 >> 16) << 1));
				uint32_t inq = ~0u;
				if (c.qlen - x0 < WNT) inq = ~(~0u << (c.qlen - x0)), m |= ~inq, mc &= inq;
				dirty |= qv.b2 & inq;
				if (x0 <= e0 && e0 - x0 < WNT) mc &= ~0u << (e0 - x0);
				if (x0 + WNT <= e0) mc = 0;
				subst_count(acc, mc, qv, sv);
			}
			if (x0 <= e0 && e0 - x0 < WNT) m &= ~0u << (e0 - x0); // (what lies before the anchor is none of the window's business)
			if (x0 + WNT <= e0) m = 0;
			L.mbits[64 * ck + lane] = m;
			*(uint2 *)&L.q2[2 * (64 * ck + lane)] = codes;
		}
		subst_flush_pairs(acc, hist, true);
	} else {
	// (-DCOOP_STREAM_PIPELINE: the loads of chunk ck + 1 issued before chunk ck is worked on -- the wait for a chunk's loads stands right
	// behind them otherwise, four memory latencies per window.  Measured no gain, same box: 4.10 against 4.08 ms, three more spilled
	// registers; profiles/r07_pool/coop_stream_pipeline_ab.txt.  k_pool_cold's sweep S, where the stream is most of the work, keeps its pipeline.)
#ifdef COOP_STREAM_PIPELINE
	auto fetch_chunk = [&](int ck, uint4 &q_, uint4 &s_) {
		const uint32_t x0 = wbase + 2048 * (uint32_t)ck + WNT * lane;
		if (x0 < c.qlen) q_ = ld_query(c, x0), s_ = ld_subject_guarded(c, (int64_t)x0 + dg);
	};
	uint4 qv = make_uint4(0, 0, 0, 0), sv = make_uint4(0, 0, 0, 0);
	fetch_chunk(0, qv, sv);
#pragma unroll
#else
#pragma unroll 2
#endif
	for (int ck = 0; ck < NCH; ++ck) {
#ifdef COOP_STREAM_PIPELINE
		uint4 qn = make_uint4(0, 0, 0, 0), sn = make_uint4(0, 0, 0, 0);
		if (ck + 1 < NCH) fetch_chunk(ck + 1, qn, sn);
#endif
		const uint32_t x0 = wbase + 2048 * ck + WNT * lane;
		uint32_t m = ~0u; // positions at and beyond the query's end: lcp() stops there
		uint2 codes = make_uint2(0, 0);
		if (x0 < c.qlen) {
#ifndef COOP_STREAM_PIPELINE
			const uint4 qv = ld_query(c, x0), sv = ld_subject_guarded(c, (int64_t)x0 + dg);
#endif
			m = squeeze32(neq32(qv, sv));
			codes = make_uint2(squeeze_codes(qv.x) | (squeeze_codes(qv.y) << 16), squeeze_codes(qv.z) | (squeeze_codes(qv.w) << 16));
			if (c.qlen - x0 < WNT) m |= ~0u << (c.qlen - x0);
			dirty |= (qv.x | qv.y | qv.z | qv.w) & 0x44444444u; // bit 2 of a symbol: no nucleotide (the padding behind the query's end too: its last window's walks read the query itself)
		}
		if (x0 <= e0 && e0 - x0 < WNT) m &= ~0u << (e0 - x0); // (what lies before the anchor is none of the window's business)
		if (x0 + WNT <= e0) m = 0;
		L.mbits[64 * ck + lane] = m;
		*(uint2 *)&L.q2[2 * (64 * ck + lane)] = codes;
#ifdef COOP_STREAM_PIPELINE
		qv = qn, sv = sn;
#endif
	}
	}
	const bool clean = !__any(dirty != 0);
	if (lane < 4) L.q2[128 * NCH + lane] = 0;
	if (lane == 0) L.nhadx = 0;
	if (lane < 4) L.mbits[64 * NCH + lane] = 0;
	ch.blk_base = NOPOS; // (mode G's block of probes lies where the heads are about to be listed)
	ch.stuck = 0;
	wave_sync();

	TOCK(tph, PH_STREAM);
	// ---- heads: a mismatch with fewer than thr equal symbols behind it and at least thr before it.  Lane l looks
	// at the NCH words of positions 32 NCH l ...; a word together with its neighbours, thr < 32
	uint32_t hmask[NCH], nh = 0;
#pragma unroll
	for (int j = 0; j < NCH; ++j) {
		const uint32_t w = NCH * lane + j;
		const uint32_t cur = L.mbits[w], nxt = L.mbits[w + 1], prv = w ? L.mbits[w - 1] : 0u;
		// a mismatch among the next thr positions / among the thr positions before: the bits smeared over thr - 1 more positions (doubling,
		// then the remainder) -- on 32-bit words with funnel shifts, as k_pool_cold's sweep S does it (64-bit shifts and ors: twice the instructions)
		uint32_t soon = __builtin_amdgcn_alignbit(nxt, cur, 1), sn = nxt >> 1;   // bit x: position x + 1
		uint32_t before = __builtin_amdgcn_alignbit(cur, prv, 31), bp = prv << 1; // bit x: position x - 1
		uint32_t have = 1;
		while (2 * have <= thr) {
			soon |= __builtin_amdgcn_alignbit(sn, soon, have), sn |= sn >> have;
			before |= __builtin_amdgcn_alignbit(before, bp, 32 - have), bp |= bp << have;
			have *= 2;
		}
		if (have < thr) {
			const uint32_t r = thr - have;
			soon |= __builtin_amdgcn_alignbit(sn, soon, r);
			before |= __builtin_amdgcn_alignbit(before, bp, 32 - r);
		}
		uint32_t live = ~0u; // a chain that stands behind position x >= end - 1 has left the segment
		const uint32_t x0 = wbase + 32 * w;
		if (x0 + 1 >= end) live = 0;
		else if (end - 1 - x0 < 32) live = (1u << (end - 1 - x0)) - 1u;
		hmask[j] = cur & soon & ~before & live;
		nh += (uint32_t)__builtin_popcount(hmask[j]);
	}
	uint32_t hbase = wave_scan_add(nh); // (inclusive; made exclusive below)
	const uint32_t nheads_all = lane_read(hbase, 63);
	hbase -= nh;
	uint32_t dropped = NOPOS; // the first head beyond the list's capacity: nothing is decided from there on
#pragma unroll
	for (int j = 0; j < NCH; ++j)
		for (uint32_t hm = hmask[j]; hm; hm &= hm - 1) {
			const uint32_t off = 32 * (NCH * lane + j) + (uint32_t)__builtin_ctz(hm);
			if (hbase < COOP_HCAP)
				L.hpos[hbase] = (uint16_t)off;
			else if (dropped == NOPOS)
				dropped = wbase + off;
			++hbase;
		}
	const uint32_t nheads = nheads_all < COOP_HCAP ? nheads_all : COOP_HCAP;
	CSTAT(CS_WINDOWS, 1);
	CSTAT(CS_HEADS, nheads);
	CSTAT(CS_DROPPED, nheads_all - nheads);
	const uint32_t f_cap = nheads_all > COOP_HCAP ? uni(wave_min(dropped)) : NOPOS;
	wave_sync();

	TOCK(tph, PH_HEADS);
	// ---- the walks, one lane each; a lane that is done takes the next head.  A lane whose probe needs more than the
	// table's entry is parked; the parked lanes take lane_probe together once there are enough of them (or nothing else
	// is left to do).
	{
		uint32_t hk = NOPOS, e = 0, p = 0, Xq = 0, Xs = 0, Xl = 0, nX = 0, next_head = 0;
		uint32_t mx = 0, mn = 0, mq = 0; // a parked lane's K-mer, if it occurs a few times: first suffix-array index, occurrences, the 16 symbols behind it
		bool parked = false;
		// equal symbols from p on along the diagonal as far as 32 bits of the window show; seen: a mismatch of the window ends them
		auto run_ahead = [&](uint32_t pp, bool &seen) {
			const uint32_t o = pp - wbase, wi = o >> 5;
			const uint32_t lo = L.mbits[wi], hi = L.mbits[wi + 1];
			const uint32_t v = (o & 31u) ? (lo >> (o & 31u)) | (hi << (32u - (o & 31u))) : lo;
			const uint32_t r = v ? (uint32_t)__builtin_ctz(v) : 32u;
			seen = v != 0 && o + r < W;
			return r;
		};
		auto generic_probe = [&](uint32_t pp) { return coop_generic_probe(c, pp); };
		// what a step's outcome does to its walk: on with the next position, or the head's result (res: how it ended, ra: where it landed, rlen: the
		// length of the anchor it landed on)
		auto settle = [&](uint32_t res, uint32_t ra, uint32_t rlen, const Probe &pr, bool have) {
			if (!res && have) {
				if (pr.unique && pr.len >= thr) {
					if (pr.pos == p + sd) { // on the diagonal
						// a right anchor of the anchor before the head only on the same strand (src/process.c:162)
						const bool same_side = (p + sd < c.border) == (e + sd <= c.border);
						if (!same_side || (Xl && Xl >= 2 * thr))
							res = W_BREAK;
						else
							res = W_OK | (rlen == NOPOS ? W_LUCKY : 0u) | (Xl ? W_HADX : 0u) | (nX << W_NX_SHIFT), ra = p, rlen = rlen == NOPOS ? 0u : pr.len;
					} else {
						const uint32_t endS = Xs + Xl, endQ = Xq + Xl;
						if (Xl && ((pr.pos > endS && p - endQ == pr.pos - endS && (pr.pos < c.border) == (Xs < c.border)) || Xl >= 2 * thr))
							res = W_BREAK; // a right anchor off the diagonal (its gap is not in the window), or an anchor that is counted by itself
						else if (++nX > COOP_MAX_X)
							res = W_BREAK;
						else
							Xq = p, Xs = pr.pos, Xl = pr.len;
					}
				}
				if (!res) p += pr.len + 1;
			}
			if (res) {
				if ((res & W_LUCKY) && (ra + sd < c.border) != (e + sd <= c.border)) res = W_BREAK;
				if ((res & W_STATUS) == W_OK && (res & W_HADX) && atomicAdd(&L.nhadx, 1u) >= COOP_KCAP) res = W_BREAK; // (the list of such stretches is full)
				L.ha[hk] = (uint16_t)(ra - wbase), L.hend[hk] = rlen, L.hflag[hk] = (uint16_t)res;
				hk = NOPOS;
			}
		};
		// The parked lanes' turn -- coop_probe_multi, lane_probe: most of this loop's code -- stands in FRONT of the loop of the ordinary trips, not
		// inside it (two loops, the inner one small): 3.52 -> 3.36 ms on the bench set.
		bool due = false;
		for (;;) {
			if (due) {
				due = false;
				if (hk != NOPOS && parked) {
					uint32_t res = 0, ra = 0, rlen = 0;
					Probe pr;
					pr.len = 0, pr.pos = 0, pr.unique = false;
					bool have = false; // pr is the answer for p
					bool long_diag = false;
					if (mn) {
						bool seen;
						const uint32_t r = run_ahead(p, seen);
						have = coop_probe_multi(c, p, sd, mx, mn, mq, r, seen, pr, long_diag);
					}
					if (!have) pr = generic_probe(p), long_diag = false;
					have = true, parked = false;
#ifdef ANDI_COOP_STATS
					atomicAdd(&g_coop_stats[CS_PROBES], 1ull);
#endif
					if (long_diag && pr.unique) { // the diagonal's occurrence is the longest, longer than the bits at hand show
						if (wbase + W - p >= 32) {
							const bool same_side = (p + sd < c.border) == (e + sd <= c.border);
							res = (!same_side || (Xl && Xl >= 2 * thr)) ? W_BREAK : (W_OK | W_LUCKY | (Xl ? W_HADX : 0u) | (nX << W_NX_SHIFT));
							ra = p;
						} else {
							pr = generic_probe(p); // (the window's end)
						}
					}
					settle(res, ra, rlen, pr, have);
				}
			}
			for (;;) {
				const uint64_t idle = __ballot(hk == NOPOS);
				if (idle && next_head < nheads) {
					const uint32_t my = next_head + (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
					if (hk == NOPOS && my < nheads) hk = my, e = wbase + L.hpos[my], p = e + 1, Xl = 0, nX = 0, parked = false;
					next_head += (uint32_t)__builtin_popcountll(idle);
				}
				const uint64_t busy = __ballot(hk != NOPOS), waiting = __ballot(hk != NOPOS && parked);
				if (!busy) break;
				// the parked lanes' turn?
				// (... or no head is left to take: served only when nothing else was left, a parked lane's remaining steps ran alone behind all the others)
				const bool service = waiting && ((uint32_t)__builtin_popcountll(waiting) >= COOP_PARK || waiting == busy || next_head >= nheads);
#ifdef ANDI_COOP_STATS
				if (lane == (uint32_t)__builtin_ctzll(__ballot(1))) {
					const uint32_t nb = (uint32_t)__builtin_popcountll(service ? waiting : busy & ~waiting); // lanes at work in this trip
					atomicAdd(&g_coop_trip_lanes[service ? 1 : 0][nb ? 32 - __builtin_clz(nb) : 0], 1ull);
					atomicAdd(&g_coop_stats[service ? CS_SERVICE : CS_TRIPS], 1ull);
					atomicAdd(&g_coop_stats[service ? CS_SERVICE_LANES : CS_LANE_STEPS], (unsigned long long)__builtin_popcountll(service ? waiting : busy & ~waiting));
				}
#endif
				if (service) {
					due = true;
					break;
				}
				if (hk == NOPOS || parked) continue;
				uint32_t res = 0, ra = 0, rlen = 0;
				Probe pr;
				pr.len = 0, pr.pos = 0, pr.unique = false;
				bool have = false; // pr is the answer for p
				// ---- a lane that is not parked: the step's conditions as values, few branches (a wavefront executes every
				// branch some lane takes, and keeps the books of the execution mask for each: the nested form of this cost
				// 230 vector and 260 scalar instructions per trip)
				const uint32_t o = p - wbase;
				const bool inwin = p < end && o < W;
				bool seen;
				const uint32_t r = run_ahead(inwin ? p : wbase, seen); // equal symbols from p on along the diagonal
				const uint32_t left = W - (inwin ? o : 0u);             // positions of the window from p on
				const bool near = Xl == 0 && p + sd < n && p - e <= thr; // lucky_anchor applies on the diagonal: the bits answer
				const bool run_ok = seen ? r >= thr : left >= thr;      // an anchor (not seen: thr equal symbols and more, its end is settled later)
				res = !inwin ? (Xl ? W_BREAK : (p >= end ? W_EXIT : W_OPEN)) // (the walk has left the segment / the window)
					  : (near && run_ok) ? (W_OK | W_LUCKY)
					  : (near && !seen && left < thr) ? W_OPEN : 0u;
				ra = p;
				if (inwin && !res && o + 32 > W) res = Xl ? W_BREAK : W_OPEN; // (a probe in the window's last 32 symbols would be lane_probe's, by one lane: the chain stops at this head instead, the next window opens there)
				if (inwin && Xl) { // lucky_anchor on the diagonal of the anchor off the window's: compare (rare)
					const uint32_t adv = p - Xq;
					if (Xs + adv < n && adv - Xl <= thr) {
						const uint32_t qa = p & ~1u;
						const uint4 d = neq32(ld_query(c, qa), ld_subject_guarded(c, (int64_t)qa + ((int64_t)Xs - (int64_t)Xq)));
						uint32_t l = first_from(d, p & 1u) - (p & 1u);
						if (l > c.qlen - p) l = c.qlen - p;
						if (l >= thr) res = W_BREAK; // the chain really changes its diagonal: not this window's business
					}
				}
				if (inwin && !res) { // the probe (the lucky attempt has failed or does not apply: it is not repeated)
					bool on_diag;
					have = coop_probe_fast<NCH>(c, L, wbase, clean, p, sd, pr, on_diag, mx, mn, mq);
					if (!have && mn) { // a K-mer with a few occurrences: the sorter's records settle it as a rule -- no wait for the parked lanes' turn
						bool long_diag;
						if (coop_probe_r2(c, p, sd, mx, mn, mq, r, seen, pr, long_diag)) {
							have = true;
							if (long_diag && pr.unique) { // the diagonal's occurrence is the longest, longer than the bits at hand show
								if (wbase + W - p >= 32) {
									const bool same_side = (p + sd < c.border) == (e + sd <= c.border);
									res = (!same_side || (Xl && Xl >= 2 * thr)) ? W_BREAK : (W_OK | W_LUCKY | (Xl ? W_HADX : 0u) | (nX << W_NX_SHIFT));
								} else {
									have = false; // (the window's end: lane_probe's)
								}
							}
						}
					}
					if (on_diag) { // the K-mer's one occurrence is the diagonal's: the bits know the match
						pr.unique = true, pr.pos = p + sd, pr.len = r;
						if (!seen && left < 32) have = false; // (the window's end: lane_probe follows the occurrence)
						else if (!seen) rlen = NOPOS;         // longer than the bits at hand show: an anchor (thr < 32) whose end is settled later
					}
					parked = !have;
#ifdef ANDI_COOP_STATS
					if (have) atomicAdd(&g_coop_stats[CS_PROBES], 1ull);
#endif
				}
				settle(res, ra, rlen, pr, have);
			}
			if (!due) break;
		}
	}
	wave_sync();

	TOCK(tph, PH_WALKS);
	if constexpr (EXACT)
		for (uint32_t jw = 0; jw < (uint32_t)NCH; ++jw) L.ebits[NCH * lane + jw] = 0; // (the walks are done with the query codes that lay there)
	// ---- the chain hops from head to head; between them every mismatch is followed by a lucky anchor.
	// Nearly every head is on the chain's path and is followed by the next one: the lanes work out, head by head,
	// where its walk's anchor ends and whether anything is unusual about it (the next head lies inside the walk's
	// span; the walk did not land; a position at which the window's knowledge ends comes first); the chain is then
	// followed from one unusual head to the next with ballots -- a few trips per window, not one per head.
	uint32_t F = coop_prev_mismatch(L, wbase, wbase + W); // behind the last mismatch nothing is known
	const uint32_t last_mm = F;
	if (f_cap < F) F = f_cap;
	{
		const int64_t b = (int64_t)c.border - dg; // '#': the next anchor lies on the other strand, no right anchor (src/process.c:162)
		if (b >= (int64_t)e0 && b < (int64_t)F) F = (uint32_t)b;
		if (end - 1 < F) { // a chain that stands behind a position >= end - 1 has left the segment
			const uint32_t fe = coop_next_mismatch(L, wbase, end - 1 > e0 ? end - 1 : e0);
			if (fe != NOPOS && fe < F) F = fe;
		}
	}
	if (e0 >= F) {
		take_back(e0);
		wave_sync();
		return false;
	}
	uint32_t cur = e0, kcur = 0, hop = NOPOS; // hop: the last head the chain hopped from
	bool done = false;
	for (uint32_t base = 0; base < nheads && !done; base += 64) {
		const uint32_t k = base + lane;
		const bool valid = k < nheads;
		const uint32_t pos = valid ? wbase + L.hpos[k] : NOPOS, fl = valid ? L.hflag[k] : 0u;
		uint32_t endk = NOPOS; // where the anchor the walk landed on ends: the chain's next stand
		if ((fl & W_STATUS) == W_OK) {
			const uint32_t la = wbase + L.ha[k];
			if (fl & W_LUCKY) {
				const uint32_t o = la - wbase;
				uint32_t wi = o >> 5, v = L.mbits[wi] & (~0u << (o & 31u));
				while (v == 0 && wi + 1 < 64 * NCH) v = L.mbits[++wi];
				if (v) endk = wbase + 32 * wi + (uint32_t)__builtin_ctz(v);
			} else {
				endk = la + L.hend[k];
			}
		}
		const bool hop_ok = endk != NOPOS;
		uint32_t nxtpos = lane_above(pos);
		if (lane == 63) nxtpos = k + 1 < nheads ? wbase + L.hpos[k + 1] : NOPOS;
		const bool unusual = valid && (!hop_ok || pos >= F || nxtpos < endk || endk >= F);
		bool onpath = false;
		if (kcur < base) kcur = base;
		{ // heads the chain has jumped over already
			const uint64_t at = __ballot(valid && pos >= cur);
			const uint32_t first = at ? (uint32_t)__builtin_ctzll(at) : 64u;
			if (base + first > kcur) kcur = base + first;
		}
		while (!done && kcur < base + 64 && kcur < nheads) {
			CSTAT(CS_HOPS, 1);
			const uint32_t lo = kcur - base;
			const uint64_t ev = __ballot(unusual) & (~0ull << lo);
			const uint32_t j = ev ? (uint32_t)__builtin_ctzll(ev) : 64u; // every head before it: hopped, on to the next
			if (lane >= lo && lane < j && valid) onpath = true;
			if (j > lo) {
				const uint32_t last = (j < 64 ? j : 64u) - 1u;
				const uint32_t lastv = nheads - base - 1u < last ? nheads - base - 1u : last;
				hop = base + lastv, cur = lane_read(endk, lastv);
			}
			if (j >= 64) {
				kcur = base + 64;
				break;
			}
			const uint32_t pj = lane_read(pos, j), ej = lane_read(endk, j);
			if (pj >= F) { // the chain reaches a position where the window's knowledge ends before this head
				cur = F, done = true;
			} else if (ej == NOPOS) { // its walk did not land (or where the anchor ends is not in the window): the chain stops at the head
				cur = pj, done = true;
				if ((lane_read(fl, j) & W_STATUS) == W_BREAK) ch.stuck = 1;
			} else { // hopped; the chain stands where the anchor ends
				if (lane == j) onpath = true;
				hop = base + j, cur = ej;
				if (ej >= F) {
					done = true;
				} else {
					const uint64_t at = __ballot(valid && pos >= ej) & (~0ull << j);
					kcur = base + (at ? (uint32_t)__builtin_ctzll(at) : 64u);
				}
			}
		}
		if (valid) {
			L.hend[k] = endk;
			if (onpath) L.hflag[k] = (uint16_t)(fl | W_ONPATH);
		}
#ifdef ANDI_COOP_STATS
		{
			const uint64_t on = __ballot(onpath), ox = __ballot(onpath && (fl & W_HADX));
			CSTAT(CS_ONPATH, __builtin_popcountll(on));
			CSTAT(CS_X, __builtin_popcountll(ox));
		}
#endif
	}
	if (!done) cur = F; // past the last head: lucky anchors up to where the window's knowledge ends
	if (cur == e0) {
		take_back(e0);
		wave_sync();
		return false;
	}
	// What the chain did not reach is taken back.  As a rule it stands at the window's LAST mismatch (cur: the next window's e0): that one
	// mismatch, by one lane -- its query symbol from the codes in LDS (still there), the subject's by one load; anything else: take_back, below
	bool taken_back = false;
	if constexpr (!EXACT) {
		if (cur == last_mm && cur < c.qlen && clean) {
			if (lane == 0) {
				const uint32_t o = cur - wbase, qn = (L.q2[o >> 4] >> (2 * (o & 15u))) & 3u;
				const uint32_t sn = ld_subject_guarded(c, (int64_t)cur + dg).x & 15u;
				if (!(sn & 4u)) lds_add(&hist[((sn & 3u) << 2) | qn], 0u - 1u); // (counted only if both are nucleotides)
			}
			taken_back = true;
		}
	}
	CSTAT(CS_MOVED, 1);
	CSTAT(CS_COVERED, cur - e0);
	wave_sync();
	// the anchor that ends at cur: the one the last hop landed on, or the one behind the mismatch before cur
	uint32_t aQ, lw = 1;
	if (hop != NOPOS && uni(L.hend[hop]) == cur) {
		aQ = wbase + uni(L.ha[hop]), lw = (uni(L.hflag[hop]) & W_HADX) ? 0u : 1u;
	} else {
		aQ = coop_prev_mismatch(L, wbase, cur) + 1;
	}
	// Heads whose walk met anchors off the diagonal (rare): nothing pairs up across such a stretch; the anchor
	// before the head is counted only if it was a right anchor itself or is long (src/process.c:176-186)
	uint32_t extra_anchors = 0, kn = 0; // kn: stretches counted nowhere, listed in L.kpos
	for (uint32_t base = 0; base < nheads; base += 64) {
		const uint32_t k = base + lane;
		const uint32_t fl = k < nheads ? L.hflag[k] : 0u;
		for (uint64_t bx = __ballot((fl & W_ONPATH) && (fl & W_HADX)); bx; bx &= bx - 1) {
			const uint32_t kk = base + (uint32_t)__builtin_ctzll(bx);
			const uint32_t pk = wbase + uni(L.hpos[kk]);
			uint32_t a_before = NOPOS, lw_before = 1;
			if (pk == e0) {
				a_before = st.lastQ, lw_before = st.lwra;
			} else {
				for (uint32_t b2 = 0; b2 < nheads && a_before == NOPOS; b2 += 64) { // a hop that landed right before this head?
					const uint32_t k2 = b2 + lane;
					const bool m = k2 < nheads && (L.hflag[k2] & W_ONPATH) && L.hend[k2] == pk;
					const uint64_t bm = __ballot(m);
					if (bm) {
						const uint32_t kp = b2 + (uint32_t)__builtin_ctzll(bm);
						a_before = wbase + uni(L.ha[kp]), lw_before = (uni(L.hflag[kp]) & W_HADX) ? 0u : 1u;
					}
				}
				if (a_before == NOPOS) a_before = coop_prev_mismatch(L, wbase, pk) + 1;
			}
			if constexpr (EXACT) { // (the counting pass below counts every anchor's nucleotides: this one is taken back if it does not count)
				if (!(lw_before || pk - a_before >= 2 * thr)) coop_count_anchor(c, (lds_u32 *)L.hist, a_before, pk - a_before, false);
			} else {
				if (lw_before || pk - a_before >= 2 * thr) ch.quarter += (pk - a_before) >> 2, ch.rest += (pk - a_before) & 3u;
			}
			extra_anchors += (uni(L.hflag[kk]) >> W_NX_SHIFT) & 7u;
			if (lane == 0) L.kpos[kn] = pk;
			++kn;
			if constexpr (!EXACT) // (the stretch is counted nowhere: its mismatches were, with all the window's)
				pool_uncount_coop(c, hist, pk, (uint32_t)((int64_t)pk + dg), wbase + uni(L.ha[kk]) - pk);
		}
	}
	wave_sync();

	TOCK(tph, PH_HOPS);
	// ---- the stretches behind the heads the chain came by: gap positions (ebits), unless listed in kpos (counted nowhere);
	// their symbols are counted here, by the lane of the head (model_count, src/model.c:309-337)
	if constexpr (!EXACT) {
		// the mismatches were counted with all the window's: what is left of a stretch (head, landing) are its EQUAL symbols, by nucleotide -- from
		// the window's bits and the query's codes, both still in LDS (the codes lie where the stretches' bits are about to be written); a window
		// with a separator ('!' = '!' is no pair of nucleotides) reads the texts instead
		uint32_t eq0 = 0, eq1 = 0, eq2 = 0, eq3 = 0;
		uint32_t tt = 0, t0 = 0, t1 = 0, t01 = 0; // equal symbols in all: whose code has bit 0, bit 1, both
		for (uint32_t b = 0; b < nheads; b += 64) {
			const uint32_t i = b + lane;
			const uint32_t fl = i < nheads ? L.hflag[i] : 0u;
			const bool ord = (fl & W_ONPATH) && !(fl & W_HADX);
			const uint32_t o0 = ord ? L.hpos[i] + 1u : 0u, o1 = ord ? L.ha[i] : 0u; // [o0, o1): behind the head's own mismatch
			if (clean) {
				for (uint32_t wd = o0 >> 5; 32 * wd < o1; ++wd) {
					uint32_t rm = ~0u;
					if (wd == (o0 >> 5)) rm &= ~0u << (o0 & 31u);
					if (32 * wd + 32 > o1) rm &= (1u << (o1 & 31u)) - 1u;
					const uint32_t eq = ~L.mbits[wd] & rm, c0 = L.q2[2 * wd], c1 = L.q2[2 * wd + 1];
					const uint32_t lo = spread16(eq), hi = spread16(eq >> 16); // (a position's bit where its code's low bit lies)
					const uint32_t l0 = lo & c0, l1 = lo & (c0 >> 1), h0 = hi & c1, h1 = hi & (c1 >> 1);
					tt += (uint32_t)__builtin_popcount(eq);
					t0 += (uint32_t)__builtin_popcount(l0) + (uint32_t)__builtin_popcount(h0);
					t1 += (uint32_t)__builtin_popcount(l1) + (uint32_t)__builtin_popcount(h1);
					t01 += (uint32_t)__builtin_popcount(l0 & l1) + (uint32_t)__builtin_popcount(h0 & h1);
				}
			} else {
				for (uint64_t sl = __ballot(ord && o1 > o0); sl; sl &= sl - 1) {
					const uint32_t l = (uint32_t)__builtin_ctzll(sl);
					const uint32_t q0 = wbase + lane_read(o0, l), ln = lane_read(o1, l) - lane_read(o0, l);
					pool_count_equal_coop(c, q0, (uint32_t)((int64_t)q0 + dg), ln, eq0, eq1, eq2, eq3);
				}
			}
		}
		eq0 += tt - t0 - t1 + t01, eq1 += t0 - t01, eq2 += t1 - t01, eq3 += t01; // (A: neither bit, C: bit 0 alone, G: bit 1 alone, T: both)
		{
			const uint32_t v0 = wave_sum(eq0 | (eq2 << 16)), v1 = wave_sum(eq1 | (eq3 << 16)); // (the stretches of a window are disjoint: fewer than 2048 NCH <= 16384 symbols in all -- two sums to a register)
			if (lane == 0) lds_add(&hist[0], v0 & 0xffffu), lds_add(&hist[5], v1 & 0xffffu), lds_add(&hist[10], v0 >> 16), lds_add(&hist[15], v1 >> 16);
		}
		wave_sync();
		for (uint32_t jw = 0; jw < (uint32_t)NCH; ++jw) L.ebits[NCH * lane + jw] = 0; // (the query's codes that lay there are done with)
		wave_sync();
		for (uint32_t b = 0; b < nheads; b += 64) {
			const uint32_t i = b + lane;
			const uint32_t fl = i < nheads ? L.hflag[i] : 0u;
			if (!(fl & W_ONPATH)) continue;
			const uint32_t o0 = L.hpos[i], o1 = L.ha[i]; // [o0, o1)
			for (uint32_t wd = o0 >> 5; 32 * wd < o1; ++wd) {
				uint32_t m = ~0u;
				if (wd == (o0 >> 5)) m &= ~0u << (o0 & 31u);
				if (32 * wd + 32 > o1) m &= (1u << (o1 & 31u)) - 1u;
				lds_or(&L.ebits[wd], m);
			}
		}
	} else
	{
		Tally tl;
		tl.hist = (lds_u32 *)L.hist, tl.hs = 1, tl.quarter = tl.rest = 0;
		tl.same[0] = tl.same[1] = tl.same[2] = tl.same[3] = 0;
		LWin w;
		w.q0 = EMPTY, w.dg = NO_DIAG;
		for (uint32_t b = 0; b < nheads; b += 64) {
			const uint32_t i = b + lane;
			const uint32_t fl = i < nheads ? L.hflag[i] : 0u;
			if (!(fl & W_ONPATH)) continue;
			const uint32_t o0 = L.hpos[i], o1 = L.ha[i]; // [o0, o1)
			uint32_t *dst = L.ebits;
			for (uint32_t wd = o0 >> 5; 32 * wd < o1; ++wd) {
				uint32_t m = ~0u;
				if (wd == (o0 >> 5)) m &= ~0u << (o0 & 31u);
				if (32 * wd + 32 > o1) m &= (1u << (o1 & 31u)) - 1u;
				lds_or(&dst[wd], m);
			}
			if (!(fl & W_HADX)) lane_count_gap(w, c, tl, wbase + o0, (uint32_t)((int64_t)(wbase + o0) + dg), o1 - o0);
		}
	}
	wave_sync();

	TOCK(tph, PH_STRETCH);
	// ---- the mismatches behind which a lucky anchor follows at once (all that are in no such stretch), and the
	// anchors: one ends at every position that starts a gap.  Positions e0 ... cur - 1; lane l takes the NCH words of
	// positions 32 NCH l ... (what lies before its first word comes from one scan over the lanes).
	uint32_t q_acc = 0, r_acc = 0, n_acc = 0, x_acc[4] = {0, 0, 0, 0};
	auto in_range = [&](uint32_t x0) { // the positions e0 ... cur - 1 of the word that starts at x0
		uint32_t rm = ~0u;
		if (x0 + WNT <= e0 || x0 >= cur) rm = 0;
		if (rm && e0 > x0) rm &= ~0u << (e0 - x0);
		if (rm && cur - x0 < WNT) rm &= (1u << (cur - x0)) - 1u;
		return rm;
	};
	uint32_t before = 0; // (the last position in no anchor before the lane's words) + 2; 0: none
	for (uint32_t jw = 0; jw < (uint32_t)NCH; ++jw) {
		const uint32_t w = NCH * lane + jw, x0 = wbase + 32 * w;
		const uint32_t u = (L.mbits[w] | L.ebits[w]) & in_range(x0);
		if (u) before = x0 + 31u - (uint32_t)__builtin_clz(u) + 2u;
	}
	{
		const uint32_t scan = wave_scan_max(before);
		before = lane_below(scan);
		if (lane == 0 || before < st.lastQ + 1u) before = st.lastQ + 1u; // (the anchor before e0 starts at lastQ: lastQ - 1 is the position before it)
	}
	uint32_t prev_top = 0; // the last position of the word before is in no anchor (what lies before e0 is the anchor)
	{
		const uint32_t w = NCH * lane;
		if (w && wbase + 32 * w > e0) prev_top = ((L.mbits[w - 1] | L.ebits[w - 1]) & in_range(wbase + 32 * (w - 1))) >> 31;
	}
	// LogDet / ANI: the nucleotides of every anchor that ends at a gap start of [e0, cur) -- the positions from the start of
	// the anchor before e0 up to aQ that are in no gap (the part of that anchor that lies before the window: at once)
	auto anchor_range = [&](uint32_t x0) { // the positions lastQ ... aQ - 1 of the word that starts at x0
		uint32_t rm = ~0u;
		if (x0 + WNT <= st.lastQ || x0 >= aQ) rm = 0;
		if (rm && st.lastQ > x0) rm &= ~0u << (st.lastQ - x0);
		if (rm && aQ - x0 < WNT) rm &= (1u << (aQ - x0)) - 1u;
		return rm;
	};
	if constexpr (EXACT)
		if (st.lastQ < wbase) coop_count_anchor(c, (lds_u32 *)L.hist, st.lastQ, (aQ < wbase ? aQ : wbase) - st.lastQ);
	auto fetch = [&](uint32_t jw, uint4 &qv, uint4 &sv) { // the symbols of a word that has single mismatches (or, LogDet / ANI, anchors) to count
		const uint32_t w = NCH * lane + jw, x0 = wbase + 32 * w;
		qv = sv = make_uint4(0, 0, 0, 0);
		if constexpr (!EXACT) return; // (every mismatch was counted when the window was streamed)
		if (jw >= (uint32_t)NCH) return;
		const bool singles = (L.mbits[w] & ~L.ebits[w] & in_range(x0)) != 0;
		if (singles || (EXACT && anchor_range(x0))) qv = ld_query(c, x0);
		if (singles) sv = ld_subject_guarded(c, (int64_t)x0 + dg);
	};
	uint4 qnext, snext;
	fetch(0, qnext, snext);
	for (uint32_t jw = 0; jw < (uint32_t)NCH; ++jw) {
		const uint32_t w = NCH * lane + jw, x0 = wbase + 32 * w;
		const uint4 qv = qnext, sv = snext;
		fetch(jw + 1, qnext, snext); // (in flight while this word is counted)
		const uint32_t rm = in_range(x0), m = L.mbits[w] & rm, eb = L.ebits[w] & rm, u = m | eb;
		uint32_t gs = u & ~((u << 1) | prev_top); // gap starts: a position in no anchor whose predecessor is in one
		n_acc += (uint32_t)__builtin_popcount(gs);
		for (; gs; gs &= gs - 1) {
			const uint32_t b = (uint32_t)__builtin_ctz(gs), below = u & ((1u << b) - 1u);
			const uint32_t pv = below ? x0 + 31u - (uint32_t)__builtin_clz(below) : before - 2u; // the last position before x0 + b in no anchor
			const uint32_t len = x0 + b - 1u - pv; // (pv may be lastQ - 1 = -1: unsigned wrap is fine)
			// (the anchor before a stretch that is counted nowhere: the hop above has dealt with it, src/process.c:176-186)
			bool nowhere = false;
			for (uint32_t t = 0; t < kn; ++t) nowhere |= L.kpos[t] == x0 + b;
			if (!EXACT && !nowhere) q_acc += len >> 2, r_acc += len & 3u;
		}
		if constexpr (EXACT) { // this word's positions inside anchors, by nucleotide
			const uint32_t am = anchor_range(x0) & ~u;
			for (uint32_t j = 0; j < 4 && am; ++j) {
				uint32_t t = (am >> (8 * j)) & 0xffu; // 8 positions -> bit 4k of a word
				if (!t) continue;
				t = (t | (t << 12)) & 0x000f000fu, t = (t | (t << 6)) & 0x03030303u, t = (t | (t << 3)) & ONES;
				const uint32_t qw = pick(qv, j), ok = t & ~(qw >> 2), b0 = qw, b1 = qw >> 1;
				x_acc[0] += (uint32_t)__builtin_popcount(ok & ~(b0 | b1)), x_acc[1] += (uint32_t)__builtin_popcount(ok & b0 & ~b1);
				x_acc[2] += (uint32_t)__builtin_popcount(ok & b1 & ~b0), x_acc[3] += (uint32_t)__builtin_popcount(ok & b0 & b1);
			}
		}
		if (u) before = x0 + 31u - (uint32_t)__builtin_clz(u) + 2u;
		prev_top = u >> 31;
		for (uint32_t singles = EXACT ? m & ~eb : 0u; singles; singles &= singles - 1) { // mismatches in no head's stretch: single-position gaps
			const uint32_t b = (uint32_t)__builtin_ctz(singles), sh = 4 * (b & 7u);
			const uint32_t qn = (pick(qv, b >> 3) >> sh) & 15u, sn = (pick(sv, b >> 3) >> sh) & 15u;
			if (!((qn | sn) & 4u)) lds_add((lds_u32 *)&L.hist[((sn & 3u) << 2) | (qn & 3u)], 1u);
		}
	}
	TOCK(tph, PH_FINAL);
	ch.quarter += wave_sum(q_acc), ch.rest += wave_sum(r_acc);
	if constexpr (EXACT) {
		for (int k = 0; k < 4; ++k) {
			const uint32_t v = wave_sum(x_acc[k]);
			if (lane == 0) lds_add((lds_u32 *)&L.hist[5 * k], v);
		}
	}
	ch.anchors += wave_sum(n_acc) + extra_anchors;
	{
		const uint32_t nodes = wave_sum(n_acc);
		(void)nodes;
		CSTAT(CS_NODES, nodes);
	}
	if (!taken_back) take_back(cur); // (the mismatches the chain did not reach)
	st.p = cur + 1, st.lastS = (uint32_t)((int64_t)aQ + dg), st.lastQ = aQ, st.lastLen = cur - aQ, st.lwra = lw;
	return true;
}


// ------------------------------------------------------------------ the kernel
template <int NCH, bool EXACT>
#ifndef COOP_OCC
#define COOP_OCC 8
#endif
__global__ __launch_bounds__(64 * COOP_WAVES, NCH <= 5 ? COOP_OCC : 4) void k_coop_cold(ScanArgs a) {
	__shared__ CoopLds<NCH> s_lds[COOP_WAVES];
	CoopLds<NCH> &L = s_lds[threadIdx.x >> 6];
	const uint32_t lane = __lane_id();
	const uint32_t sub = a.sub_order ? uni(a.sub_order[blockIdx.y]) : blockIdx.y; // (routed calls: the heaviest subjects first, scan.h)
	if (a.subjects[sub].mode != ANDI_MODE_PROBE) return;
	const uint32_t wseg = uni(blockIdx.x * COOP_WAVES + (threadIdx.x >> 6));
	if (wseg >= a.total_segs) return;
	const uint32_t qidx = uni(a.seg2query[wseg]);
	if (a.self[sub] == (int64_t)qidx) return;
	const uint32_t seg_in_q = wseg - uni(a.qseg_start[qidx]);
	PairCtx c = make_ctx(a, sub, qidx);
	const uint32_t start = seg_in_q * a.seg, end = start + a.seg < c.qlen ? start + a.seg : c.qlen;
	const size_t slot = (size_t)sub * a.total_segs + wseg;
	const uint32_t n = (uint32_t)c.E.n, thr = c.thr;

	// a routed call (scan.h): the pairs k_pair_estimate marked for this kernel; a wavefront that meets what the kernel is slow
	// at hands its PAIR back to the lane scan -- a mark all the pair's wavefronts look at
	uint8_t *route = a.route ? a.pair_class + (size_t)sub * a.nq + qidx : nullptr;
	auto give_up = [&]() {
		if (lane == 0) a.restitch_count[ANDI_ROUTE_ANY_LEFT] = 1;
		if (lane == 0) __hip_atomic_store(route, (uint8_t)(__hip_atomic_load(route, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | ANDI_ROUTE_LEFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	auto given_up = [&]() { return route && (uni(__hip_atomic_load(route, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & (ANDI_ROUTE_COOP | ANDI_ROUTE_LEFT)) != ANDI_ROUTE_COOP; };
	if (given_up()) return;
	if (lane < 16) L.hist[lane] = 0;
	Chain ch;
	ch.st = seg_in_q == 0 ? initial_state() : cold_state(start, n);
	ch.quarter = ch.rest = ch.anchors = ch.marked = 0, ch.blk_base = NOPOS, ch.stuck = 0;
	ChainState &st = ch.st;
	wave_sync();

	CSTAT(CS_SEGMENTS, 1);
	TICK(tall);
	uint32_t my_g = 0;
	while (st.p < end) {
		CSTAT(CS_G_STEPS, 1);
		++my_g;
		if (route) {
			if (my_g > a.route_giveup + (a.seg >> 12)) return give_up(); // (a few more per window the segment holds)
			if (given_up()) return;
		}
		// ---- one step of mode G (src/process.c:153-197)
		bool found = false, lucky = false;
		uint32_t curS = 0, curLen = 0;
		if (lucky_applies(st, n, thr)) {
			curS = st.lastS + (st.p - st.lastQ);
			curLen = coop_lcp(c, st.p, curS, c.qlen - st.p, route ? COOP_TRIAL_LCP : ~0u);
			if (curLen == NOPOS) return give_up();
			found = lucky = curLen >= thr;
		}
		if (!found) {
			if (ch.blk_base == NOPOS || st.p < ch.blk_base || st.p - ch.blk_base >= 64) { // the next 64 positions' answers at once
				const uint32_t p = st.p + lane;
				CSTAT(CS_BLOCKS, 1);
				Probe pr;
				pr.len = 0, pr.pos = 0, pr.unique = false;
				if (p < c.qlen) {
					LWin w;
					w.q0 = EMPTY, w.dg = NO_DIAG;
					pr = lane_probe(c, p, w);
				}
				wave_sync();
				L.pl[lane] = pr.len | (pr.unique ? 0x80000000u : 0u), L.pp[lane] = pr.pos;
				ch.blk_base = st.p;
				wave_sync();
			}
			const uint32_t v = uni(L.pl[st.p - ch.blk_base]);
			curLen = v & 0x7fffffffu, curS = uni(L.pp[st.p - ch.blk_base]);
			found = (v >> 31) && curLen >= thr;
		}
		if (found) {
			coop_account<EXACT>(c, ch, L, curS);
			st.lastS = curS, st.lastQ = st.p, st.lastLen = curLen;
		}
		st.p += curLen + 1;
		if (found) {
			wave_sync();
			coop_note_anchor(a, slot, ch, L);
		}
		// ---- windows one after the other while the chain stays canonical on the diagonal and moves
		if (found && lucky)
			while (st.p < end && st.lastQ + st.lastLen < c.qlen && coop_window<NCH, EXACT>(a, c, ch, L, end)) {
				if (given_up()) return;
				if (ch.stuck) break; // (the chain stands at a head whose walk gave up: mode G's step)
			}
	}

	TOCK(tall, 7);
#ifdef ANDI_COOP_STATS
	if (lane == 0) atomicMax(&g_coop_max[0], my_g);
#endif
	(void)my_g;
	// ---- what pass B reads (scan.h)
	wave_sync();
	if (lane == 0) {
		ColdMark *m = a.marks + slot * ANDI_COLD_MARKS;
		if (!ch.marked) m->st.pad[0] = 0; // unused mark
		ChainState out = st;
		out.pad[0] = 0, out.pad[1] = ch.anchors < 255 ? ch.anchors : 255, out.pad[2] = 0;
		a.cold_exit[slot] = out;
		a.exit_p[slot] = st.p;
	}
	if (lane < 16) {
		uint32_t v = L.hist[lane];
		if (lane == 0 || lane == 5 || lane == 10 || lane == 15) v += ch.quarter;
		if (lane == 15) v += ch.rest;
		a.cold_counts[slot * 16 + lane] = v;
	}
}

} // namespace

// ANDI_COOP=0: never; 2, 4, 5, 8: pass A with one wavefront per chain for every pair, windows of 2048 n symbols, whatever
// the call (any other value: 4, with a warning); unset (< 0): large calls are routed per pair (scan.h)
int andi_coop_enabled(void) {
	const char *e = andi_knob(KNOB_COOP);
	if (!e) return -4;
	const int v = atoi(e);
	if (v == 0 || v == 2 || v == 4 || v == 5 || v == 8) return v;
	static bool warned = false;
	if (!warned) fprintf(stderr, "andi-hip: ANDI_COOP=%s: windows of 2, 4, 5 or 8 chunks of 2048 symbols; taking 4\n", e), warned = true;
	return 4;
}

static bool pool_enabled() { // ANDI_POOL=0: the windows stay in LDS (coop_window), as up to round 4
	const char *e = andi_knob(KNOB_POOL);
	return !e || atoi(e) != 0;
}

// The pooled kernel's scratch: a window of POOL_MW positions per resident wavefront
constexpr uint32_t POOL_FUSED_CHUNKS = POOL_MW / 2048u, POOL_FUSED_HC = POOL_MW / 64u;
size_t andi_pool_scratch_bytes(int device, uint32_t *waves) {
	*waves = 0;
	if (!pool_enabled()) return 0;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
	*waves = (uint32_t)prop.multiProcessorCount * 4u * POOL_OCC;
	return 4096 + (size_t)*waves * pool_scratch_bytes(POOL_FUSED_CHUNKS, POOL_FUSED_HC);
}

// the pooled kernel would take this launch if it had its scratch (items and ticket are 32 bits: coop_pool.h)
int andi_coop_wants_pool(const ScanArgs &a) {
	const int nch = andi_coop_enabled();
	return a.coop && !a.exact_equal && pool_enabled() && (nch < 0 || nch == 4) && a.seg >= 32768 && (!a.route || a.pool_use) &&
		   (uint64_t)a.total_segs * a.nsub < (1ull << 32);
}

int andi_coop_will_pool(const ScanArgs &a) {
	return andi_coop_wants_pool(a) && a.pool_scratch && a.pool_waves;
}

hipError_t andi_launch_coop_cold(const ScanArgs &a, hipStream_t st) { // one segment length for the call, RAW/JC/Kimura
	const dim3 grid((a.total_segs + COOP_WAVES - 1) / COOP_WAVES, a.nsub);
	const int nch = andi_coop_enabled();
	// The windows' walks pooled through global memory (k_pool_cold: persistent wavefronts take the segments in order): segments long enough
	// to fill its windows; in a routed call where the pairs with long sampled matches hold most of the segments (ScanArgs.pool_use)
	// (both kernels stream the texts bit-sliced -- LogDet / ANI: the 4-bit symbols --: the subjects' planes are made with their scan indexes,
	// esa_build.hip: andi_launch_index_build)
	if (andi_coop_will_pool(a)) {
		const uint64_t items = (uint64_t)a.total_segs * a.nsub;
		ScanArgs b = a;
		b.pool_first = 64;
		if (const char *pf = andi_knob(KNOB_POOL_FIRST)) // (experiments)
			if (atoi(pf) >= 1 && atoi(pf) <= (int)POOL_FUSED_CHUNKS) b.pool_first = (uint32_t)atoi(pf);
		b.pool_maxchunks = POOL_FUSED_CHUNKS, b.pool_hc = POOL_FUSED_HC;
		hipError_t e = hipMemsetAsync(a.pool_ticket, 0, sizeof(uint32_t), st);
		if (e != hipSuccess) return e;
		k_pool_cold<<<(uint32_t)(items < a.pool_waves ? items : a.pool_waves), 64, 0, st>>>(b);
	} else
	// windows of five chunks where the segments are long enough to fill them (round 6: 4.10 -> 3.87 ms on the bench set against four -- a
	// window's walks take as many trips as its longest, whatever the number of heads); short segments of small calls keep four
	switch (nch < 0 ? (a.seg >= 32768 ? 5 : 4) : nch) {
		case 5: a.exact_equal ? k_coop_cold<5, true><<<grid, 64 * COOP_WAVES, 0, st>>>(a) : k_coop_cold<5, false><<<grid, 64 * COOP_WAVES, 0, st>>>(a); break;
		case 2: a.exact_equal ? k_coop_cold<2, true><<<grid, 64 * COOP_WAVES, 0, st>>>(a) : k_coop_cold<2, false><<<grid, 64 * COOP_WAVES, 0, st>>>(a); break;
		case 8: a.exact_equal ? k_coop_cold<8, true><<<grid, 64 * COOP_WAVES, 0, st>>>(a) : k_coop_cold<8, false><<<grid, 64 * COOP_WAVES, 0, st>>>(a); break;
		default: a.exact_equal ? k_coop_cold<4, true><<<grid, 64 * COOP_WAVES, 0, st>>>(a) : k_coop_cold<4, false><<<grid, 64 * COOP_WAVES, 0, st>>>(a); break;
	}
#ifdef ANDI_COOP_STATS
	if (andi_knob(KNOB_COOP_STATS)) {
		static const char *names[24] = {"segments", "G steps", "probe blocks", "coop_lcp calls", "windows", "windows that moved", "heads", "walk trips",
										"walk lane-steps", "walk probes", "heads on the path", "hops", "gaps counted in G", "positions covered by windows",
										"heads dropped", "walks with anchors off the diagonal", "coop_lcp rounds", "nodes", "service trips (lane_probe)", "lanes served", "parked: window edge / separators / plain table", "parked: K-mer occurs several times", "parked: long match off the diagonal", "parked: other"};
		unsigned long long h[24];
		(void)hipStreamSynchronize(st);
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_coop_stats), sizeof h);
		for (int k = 0; k < 24; ++k) fprintf(stderr, "coop_stats %-36s %llu\n", names[k], h[k]);
		memset(h, 0, sizeof h);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_coop_stats), h, sizeof h);
		unsigned long long tl[2][8];
		(void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_coop_trip_lanes), sizeof tl);
		for (int k = 0; k < 2; ++k)
			fprintf(stderr, "coop_stats %s trips by lanes at work (0, 1, 2-3, 4-7, 8-15, 16-31, 32-63, 64): %llu %llu %llu %llu %llu %llu %llu %llu\n", k ? "service" : "walk", tl[k][0], tl[k][1], tl[k][2], tl[k][3], tl[k][4], tl[k][5], tl[k][6], tl[k][7]);
		memset(tl, 0, sizeof tl);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_coop_trip_lanes), tl, sizeof tl);
		unsigned int mx[4];
		(void)hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_coop_max), sizeof mx);
		fprintf(stderr, "coop_stats most G steps of a segment          %u\n", mx[0]);
		memset(mx, 0, sizeof mx);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_coop_max), mx, sizeof mx);
		unsigned long long cy[8];
		(void)hipMemcpyFromSymbol(cy, HIP_SYMBOL(g_coop_cycles), sizeof cy);
		static const char *ph[7] = {"(mode G: the rest)", "window: bits", "window: heads", "window: walks", "window: hops", "window: stretches", "window: counting"};
		unsigned long long in_w = 0;
		for (int k = 1; k < 7; ++k) in_w += cy[k];
		for (int k = 1; k < 7; ++k) fprintf(stderr, "coop_cycles %-24s %6.2f %%\n", ph[k], 100.0 * (double)cy[k] / (double)cy[7]);
		fprintf(stderr, "coop_cycles %-24s %6.2f %%\n", ph[0], 100.0 * (double)(cy[7] - in_w) / (double)cy[7]);
		memset(cy, 0, sizeof cy);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_coop_cycles), cy, sizeof cy);
	}
#endif
	return hipGetLastError();
}
