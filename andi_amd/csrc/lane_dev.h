// lane_dev.h -- device helpers shared by the kernels that run ONE LANE PER CHAIN on the 4-bit
// symbols (scan_lane.hip: passes A and B with direct loads; scan_rounds.hip: pass A in rounds
// with line buffers): windows of 32 symbols, their comparison, K-mer codes, work items.
#pragma once
#include "scan_dev.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

// -DANDI_LANE_STATS: count the memory accesses of pass A by kind (diagnostic builds only)
#ifdef ANDI_LANE_STATS
static __device__ unsigned long long g_lane_stats[24];
#define STAT(k) atomicAdd(&g_lane_stats[k], 1ull)
#else
#define STAT(k) ((void)0)
#endif
// -DANDI_KNOCKOUT: parts of pass A can be switched off at run time (ANDI_KNOCK=bits) to time them; results are then wrong
#ifdef ANDI_KNOCKOUT
#define KNOCK(c, bit) (((c).knock >> (bit)) & 1u)
#else
#define KNOCK(c, bit) false
#endif
enum { ST_STEP, ST_LCP_RELOAD, ST_LCP_SLIDE, ST_PROBE, ST_PROBE_RELOAD, ST_TABLE, ST_FINAL_SA, ST_SINGLE, ST_EXT_LOOP,
	   ST_MULTI, ST_MULTI_CAND, ST_SEARCH, ST_GAP_RELOAD, ST_GAP_WORDS, ST_SUBST, ST_LUCKY_TRY,
	   ST_X0, ST_X1, ST_X2, ST_X3, ST_X4, ST_X5, ST_X6, ST_X7 };

constexpr uint32_t WNT = 32; // symbols per window
constexpr uint32_t EMPTY = ~0u;
constexpr int32_t NO_DIAG = INT32_MIN;
constexpr uint32_t ONES = 0x11111111u;
constexpr uint32_t TOPS = 0x88888888u;
#ifndef ANDI_MULTI_MAX
#define ANDI_MULTI_MAX 32
#endif
constexpr uint32_t MULTI_MAX = ANDI_MULTI_MAX; // occurrences a lane extends along one by one; more: binary search
constexpr uint32_t ROUNDS_MULTI_MAX = 8; // (scan_rounds.hip keeps the positions in registers)

// 32 symbols of the query from q0 (even) and, if dg != NO_DIAG, of the subject from q0 + dg
struct LWin {
	uint32_t q0;
	int32_t dg;
	uint4 q, s; // (where they differ is two instructions per word away, neq32: not worth four registers)
};

__device__ __forceinline__ uint32_t pick(const uint4 &v, uint32_t j) {
	return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : j == 3 ? v.w : 0u;
}

// One bit per differing symbol, the top bit of its nibble (bit 4k + 3 of a word: its symbols k differ).  Symbols are 0 ... 7 (k_pack_symbols; the paddings are
// filled with 7s), so a nibble of a ^ b is 0 ... 7 and adding 7 carries into its top bit iff it is not 0 -- never
// beyond: two instructions per word (v_xad_u32, v_and_b32) where or-ing the nibble's bits down took six, at ten
// places of pass A's loop.
__device__ __forceinline__ uint32_t neq8(uint32_t a, uint32_t b) {
	return ((a ^ b) + 0x77777777u) & TOPS;
}

__device__ __forceinline__ uint4 neq32(const uint4 &a, const uint4 &b) {
	return make_uint4(neq8(a.x, b.x), neq8(a.y, b.y), neq8(a.z, b.z), neq8(a.w, b.w));
}

// index (0..31) of the first marked symbol (any bit of its nibble) at or after symbol o, 32 if none
__device__ __forceinline__ uint32_t first_from(const uint4 &d, uint32_t o) {
	uint64_t lo = d.x | ((uint64_t)d.y << 32), hi = d.z | ((uint64_t)d.w << 32);
	const uint32_t sh = 4 * o;
	if (sh < 64) {
		lo = (lo >> sh) << sh;
	} else {
		lo = 0;
		hi = sh < 128 ? (hi >> (sh - 64)) << (sh - 64) : 0;
	}
	if (lo) return (uint32_t)__builtin_ctzll(lo) >> 2;
	if (hi) return 16 + ((uint32_t)__builtin_ctzll(hi) >> 2);
	return WNT;
}

// bits 4a, 4a+4, ... 4b-4 (0 <= a < b <= 8)
__device__ __forceinline__ uint32_t symbol_range(uint32_t a, uint32_t b) {
	return (0xffffffffu >> (32 - 4 * b)) & ~((1u << (4 * a)) - 1u) & ONES;
}

__device__ __forceinline__ uint4 ld_query(const PairCtx &c, uint32_t qa) { // qa even
	return ld_u128_unaligned(c.Qn + (qa >> 1));
}

// (One uniform base and a 32-bit lane offset -- N1 lies behind N0 in the same allocation, less than 4 GB away -- so
// that the load takes its base from scalar registers: no 64-bit address arithmetic in the vector ALUs.)
__device__ __forceinline__ uint4 ld_subject(const PairCtx &c, int32_t sa) { // sa >= -32
	const uint32_t odd = (uint32_t)sa & 1u;
	const uint32_t n1 = (uint32_t)(c.E.N1 - c.E.N0);
	const uint32_t off = ((uint32_t)(sa + 32 + (int32_t)odd) >> 1) + (odd ? n1 : 0u);
	return ld_u128_unaligned(c.E.N0 - 16 + off);
}

// 2-bit code (first symbol most significant) of the K symbols at offset o of the
// window (o + K <= 32); false if one of them is not a nucleotide.
__device__ __forceinline__ bool lane_kmer(const LWin &w, uint32_t o, uint32_t K, uint32_t &code) {
	const uint32_t j = o >> 3, r = (o & 7u) * 4u;
	const uint32_t a = pick(w.q, j), b = pick(w.q, j + 1), e = pick(w.q, j + 2);
	const uint32_t lo = __builtin_amdgcn_alignbit(b, a, r), hi = __builtin_amdgcn_alignbit(e, b, r);
	const uint64_t v = lo | ((uint64_t)hi << 32);
	const uint64_t inside = ~0ull >> (64 - 4 * K);
	auto squeeze = [](uint32_t x) { // 8 nibbles -> 8 x 2 bits, first symbol in the low bits
		x &= 0x33333333u;
		x = (x | (x >> 2)) & 0x0f0f0f0fu;
		x = (x | (x >> 4)) & 0x00ff00ffu;
		x = (x | (x >> 8)) & 0x0000ffffu;
		return x;
	};
	uint32_t y = __brev(squeeze(lo) | (squeeze(hi) << 16)); // first symbol on top, bits of a pair swapped
	y = ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
	code = y >> (32 - 2 * K);
	return (v & inside & 0x4444444444444444ull) == 0;
}

// ------------------------------------------------------------------ work items
// One lane = one segment.  Uniform mode: segment w of subject blockIdx.y (decode_item).
// Adaptive mode: wavefront W of the launch belongs to the pair whose slots contain
// 64 * W; all its lanes work on that pair.
struct LaneItem {
	uint32_t sub, qidx, seg_in_q, start, end, seg;
	uint32_t cls; // per-pair segment lengths: the pair's pair_class byte (bits 0-1 the class of its segment length, bit 7: long matches), else 0
	size_t slot;
	bool valid;
};

// the work item of a slot (see lane_item for the layouts)
__device__ __forceinline__ LaneItem lane_item_of_slot(const ScanArgs &a, unsigned long long slot);

// LAYOUT: 0 = whichever the call uses (a.adaptive), 1 = per-pair segments, 2 = one segment length.  A kernel compiled
// for layout 1 knows that all lanes of a wavefront work on one pair: the query's base address and length stay in
// scalar registers (five vector registers less per lane, query loads with a scalar base).
template <int NT = BLOCK, int LAYOUT = 0> // NT: threads per block; `wave`: the wavefront to work as (per-pair segments), or ~0; `seg_w`: the segment to take (one segment length), or ~0
__device__ __forceinline__ LaneItem lane_item(const ScanArgs &a, uint32_t wave = ~0u, uint32_t seg_w = ~0u) {
	LaneItem it;
	if (LAYOUT == 2 || (LAYOUT == 0 && !a.adaptive)) {
		it.sub = blockIdx.y;
		const uint32_t w = seg_w != ~0u ? seg_w : blockIdx.x * NT + threadIdx.x;
		it.valid = w < a.total_segs;
		it.qidx = it.seg_in_q = it.start = it.end = 0;
		it.seg = a.seg, it.cls = 0;
		it.slot = (size_t)it.sub * a.total_segs + w;
		if (it.valid) {
			it.qidx = a.seg2query[w];
			it.seg_in_q = w - a.qseg_start[it.qidx];
			const uint32_t qlen = a.qlen[it.qidx];
			it.start = it.seg_in_q * a.seg;
			it.end = it.start + a.seg < qlen ? it.start + a.seg : qlen;
			it.valid = a.self[it.sub] != (int64_t)it.qidx;
			if (a.route && it.valid) { // a routed call (scan.h): the pairs of this layout -- the wavefront kernel's, the lane scan's, the pairs handed back
				const uint32_t cls = a.pair_class[it.sub * a.nq + it.qidx];
				it.valid = a.route == ANDI_LAYOUT_COOP ? (cls & (ANDI_ROUTE_COOP | ANDI_ROUTE_LEFT)) == ANDI_ROUTE_COOP
						 : a.route == ANDI_LAYOUT_LANES2 ? (cls & ANDI_ROUTE_L2) != 0
														 : (cls & (ANDI_ROUTE_COOP | ANDI_ROUTE_L2)) == 0;
			}
		}
		return it;
	}
	const uint32_t P = a.nsub * a.nq;
	const uint32_t W = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave != ~0u ? wave : blockIdx.x * (NT / 64) + (threadIdx.x >> 6)));
	it.valid = false;
	it.sub = it.qidx = it.seg_in_q = it.start = it.end = it.seg = it.cls = 0, it.slot = 0;
	if (W >= a.pair_wave0[P]) return it;
	uint32_t lo = 0, hi = P; // the last pair whose first wavefront is <= W (pairs without work share their successor's)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (a.pair_wave0[mid] <= W) lo = mid; else hi = mid;
	}
	const uint32_t pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
	it.sub = pair / a.nq, it.qidx = pair % a.nq;
	const uint32_t seg = a.seg0 << (a.pair_class[pair] & 3u), qlen = a.qlen[it.qidx];
	it.seg_in_q = (W - a.pair_wave0[pair]) * 64 + (threadIdx.x & 63u);
	it.seg = seg, it.cls = a.pair_class[pair];
	it.start = it.seg_in_q * seg;
	it.valid = it.start < qlen;
	it.end = it.start + seg < qlen ? it.start + seg : qlen;
	it.slot = (size_t)64 * W + (threadIdx.x & 63u);
	return it;
}


__device__ __forceinline__ LaneItem lane_item_of_slot(const ScanArgs &a, unsigned long long slot) {
	LaneItem it;
	it.slot = (size_t)slot;
	it.valid = true;
	if (!a.adaptive) {
		it.sub = (uint32_t)(slot / a.total_segs);
		const uint32_t w = (uint32_t)(slot % a.total_segs);
		it.qidx = a.seg2query[w];
		it.seg_in_q = w - a.qseg_start[it.qidx];
		it.seg = a.seg, it.cls = 0;
		const uint32_t qlen = a.qlen[it.qidx];
		it.start = it.seg_in_q * a.seg;
		it.end = it.start + a.seg < qlen ? it.start + a.seg : qlen;
		return it;
	}
	const uint32_t P = a.nsub * a.nq, W = (uint32_t)(slot >> 6);
	uint32_t lo = 0, hi = P; // the last pair whose first wavefront is <= W
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (a.pair_wave0[mid] <= W) lo = mid; else hi = mid;
	}
	it.sub = lo / a.nq, it.qidx = lo % a.nq;
	it.seg = a.seg0 << (a.pair_class[lo] & 3u), it.cls = a.pair_class[lo];
	const uint32_t qlen = a.qlen[it.qidx];
	it.seg_in_q = (W - a.pair_wave0[lo]) * 64 + (uint32_t)(slot & 63u);
	it.start = it.seg_in_q * it.seg;
	it.end = it.start + it.seg < qlen ? it.start + it.seg : qlen;
	return it;
}

// Pairs with long matches (k_pair_estimate: mean sampled match length >= a.quad_min_match) go to k_lane_quad, the
// others to k_lane_cold; with one segment length for the call no pair is sampled and k_lane_cold takes everything.
__device__ __forceinline__ bool lane_is_mine(const ScanArgs &a, const LaneItem &it, bool quad_kernel) {
	const bool quad_pair = a.adaptive && (it.cls & 0x80u);
	return quad_pair == quad_kernel;
}
