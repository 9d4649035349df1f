// scan_dev.h — device-side definitions shared by the two implementations of the
// anchor scan: scan.hip (a group of G lanes per chain, byte sequences) and
// scan_lane.hip (one lane per chain, nibble-packed sequences).
#pragma once
#include "scan.h"

#ifndef WAVES_PER_BLOCK
#define WAVES_PER_BLOCK 4
#endif
#define BLOCK (64 * WAVES_PER_BLOCK)
#define SCAN_G 4 /* lanes per chain in passes A and B (measured best of 4/8/16 on MI355X) */

#define CHECK_LAUNCH()                                                                             \
	do {                                                                                           \
		hipError_t e_ = hipGetLastError();                                                         \
		if (e_ != hipSuccess) return e_;                                                           \
	} while (0)

struct PairCtx {
	EsaG E;
	g_u8p Q;
	g_u8p Qn; // packed symbols of the query
	g_u32p Qp; // ... bit-sliced (block 0 = its symbols 0 ... 31)
	uint32_t qlen;
	uint32_t thr;
	uint32_t border; // n / 2, src/process.c:149
	bool exact;      // LogDet/ANI: equal runs are counted per nucleotide (src/model.c:256-278)
	uint32_t knock;  // diagnostic builds only
};

__device__ __forceinline__ bool same_state(const ChainState &a, const ChainState &b) {
	return a.p == b.p && a.lastS == b.lastS && a.lastQ == b.lastQ && a.lastLen == b.lastLen &&
		   a.lwra == b.lwra;
}

__device__ __forceinline__ ChainState initial_state() {
	ChainState s;
	s.p = s.lastS = s.lastQ = s.lastLen = s.lwra = 0;
	s.pad[0] = s.pad[1] = s.pad[2] = 0;
	return s;
}

// A state no real chain can be in: the "last anchor" sits at RS offset n, so
// neither the lucky test (try_pos_S >= len, src/process.c:90) nor the
// right-anchor test (pos_S > end_S, src/process.c:160) can fire until a real
// anchor has replaced it, and its length 0 never gets counted.
__device__ __forceinline__ ChainState cold_state(uint32_t start, uint32_t n) {
	ChainState s = initial_state();
	s.p = start;
	s.lastS = n;
	return s;
}

// What a chain adds to the 4x4 matrix.  Substitutions between anchors go to a
// per-group histogram in LDS; the equal runs of model_count_equal (RAW/JC/Kimura:
// len/4 to A->A, C->C, G->G and len/4 + len%4 to T->T, src/model.c:247-253) are
// two running sums in registers, folded into the histogram at the end.
// (The histogram pointer carries its address space: pass B picks one of two tallies per lane, and through a generic
// pointer the compiler can no longer tell that the counts go to LDS -- flat atomics, the tally structs on the stack.)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void lds_add(lds_u32 *p, uint32_t v) {
	(void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Tally {
	lds_u32 *hist;         // LDS, 16 cells `hs` words apart: substitutions found in gaps
	uint32_t hs;
	uint32_t quarter, rest; // equal runs (uniform within the group)
	uint32_t same[4];      // A->A, C->C, G->G, T->T pairs this lane saw in gaps
};

__device__ __forceinline__ void count_equal(Tally &t, uint32_t len) {
	t.quarter += len >> 2;
	t.rest += len & 3u;
}

// One lane per chain: cell-major layout (cell c of chain k at [c][k]) keeps the lanes of a
// wavefront on different LDS banks; otherwise the 16 cells of a group are contiguous.
template <int G>
__device__ __forceinline__ constexpr uint32_t hist_stride() {
	return G == 1 ? BLOCK : 1;
}

template <int G>
__device__ __forceinline__ void tally_begin(Tally &t, uint32_t *hist) {
	t.hist = (lds_u32 *)hist, t.hs = hist_stride<G>(), t.quarter = 0, t.rest = 0;
	t.same[0] = t.same[1] = t.same[2] = t.same[3] = 0;
	for (uint32_t c = Group<G>::sub(); c < 16; c += G) hist[c * t.hs] = 0;
}

// Fold the register-held parts into the LDS histogram (once, when the chain is done).
template <int G>
__device__ __forceinline__ void tally_finish(Tally &t) {
#pragma unroll
	for (int x = 0; x < 4; ++x) {
		uint32_t v = t.same[x];
		for (int d = G / 2; d; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
		t.same[x] = v;
	}
	if (Group<G>::sub() == 0) {
		t.hist[0] += t.quarter + t.same[0];
		t.hist[5 * t.hs] += t.quarter + t.same[1];
		t.hist[10 * t.hs] += t.quarter + t.same[2];
		t.hist[15 * t.hs] += t.quarter + t.rest + t.same[3];
	}
}

struct WorkItem {
	uint32_t sub, w, qidx, seg_in_q, start, end;
	bool valid, is_self;
};

__device__ __forceinline__ PairCtx make_ctx(const ScanArgs &a, uint32_t sub, uint32_t qidx) {
	PairCtx c;
	c.E = esa_global(a.subjects[sub]);
	c.Q = (g_u8p)(a.qpool + a.qoff[qidx]);
	c.Qn = (g_u8p)(a.qnib + a.qoff[qidx] / 2);
	c.Qp = (g_u32p)(a.qplanes + 3 * (a.qoff[qidx] / 32));
	c.qlen = a.qlen[qidx];
	c.thr = (uint32_t)c.E.thr;
	c.border = (uint32_t)c.E.n / 2;
	c.exact = a.exact_equal != 0;
	c.knock = a.knock;
	return c;
}

// The segment whose cold exit pass B takes as the state in which the true chain enters
// segment k (k >= 1) of the pair whose segments start at slot `row`.  Normally k - 1.
// But a chain that leaves a segment at a position beyond the end of the next segment(s)
// -- a long anchor spans them -- never runs in those: segments that one of the
// ANDI_SPAN_WINDOW segments before them jumps over are passed over.  (Cold chains that
// start inside such an anchor end where it ends, so it does not matter whether the
// jumping segment was itself visited.)  Pass C applies the same rule when it verifies.
#define ANDI_SPAN_WINDOW 8
__device__ __forceinline__ uint32_t entry_source(const ScanArgs &a, size_t row, uint32_t k, uint32_t seg,
												 uint32_t qlen) {
	uint32_t j = k - 1;
	while (j >= 1) {
		const uint32_t e = (j + 1) * seg, end_j = e < qlen ? e : qlen;
		bool covered = false; // (the window's exit positions are fetched together, not one round trip after the other)
#pragma unroll
		for (uint32_t back = 1; back <= ANDI_SPAN_WINDOW; ++back)
			covered |= a.exit_p[row + j - (back <= j ? back : j)] >= end_j && back <= j;
		if (!covered) break;
		--j;
	}
	return j;
}

// The state in which pass B assumes the true chain enters segment k (k >= 1), verified by pass C.
// Its position is where the cold chain of the segment before it (entry_source) left.  What it
// remembers -- its last anchor and last_was_right_anchor -- is what that cold chain left with,
// unless that chain found no anchor at all: the true chain then passes through such a segment
// with the memory it came with (in a stretch without homology -- a genomic island, an unrelated
// contig -- no chain finds anything for many segments), so the memory is taken from the nearest
// segment before it whose cold chain did find one.
#define ANDI_MEMORY_WINDOW 4096
__device__ __forceinline__ ChainState assumed_entry(const ScanArgs &a, size_t row, uint32_t k, uint32_t seg, uint32_t qlen) {
	uint32_t j = entry_source(a, row, k, seg, qlen);
	ChainState T = a.cold_exit[row + j];
	if (T.pad[1] != 0 || j == 0) return T;
	for (uint32_t hops = 0; hops < ANDI_MEMORY_WINDOW; ++hops) {
		j = entry_source(a, row, j, seg, qlen); // the segment the chain was in before j
		const ChainState m = a.cold_exit[row + j];
		if (m.pad[1] != 0 || j == 0) {
			T.lastS = m.lastS, T.lastQ = m.lastQ, T.lastLen = m.lastLen, T.lwra = m.lwra;
			break;
		}
	}
	return T; // (window exhausted: the assumption fails pass C's check and the pair is fixed up sequentially)
}

// lucky_anchor's precondition (src/process.c:86-92): while it does not hold and no anchor is found, a
// chain's next position depends on its position alone
__device__ __forceinline__ bool lucky_applies(const ChainState &s, uint32_t n, uint32_t thr) {
	const uint32_t advance = s.p - s.lastQ;
	return s.lastS + advance < n && advance - s.lastLen <= thr;
}

// adaptive mode: first slot, segment length and segment count of a pair
struct PairGeom {
	size_t slot0;
	uint32_t seg, nseg;
};

__device__ __forceinline__ PairGeom pair_geometry(const ScanArgs &a, uint32_t sub, uint32_t qidx) {
	const uint32_t pair = sub * a.nq + qidx;
	PairGeom g;
	g.slot0 = (size_t)64 * a.pair_wave0[pair];
	g.seg = a.seg0 << (a.pair_class[pair] & 3u);
	g.nseg = (a.qlen[qidx] + g.seg - 1) / g.seg;
	return g;
}

// work item of this lane's group: segment w of subject blockIdx.y
template <int G>
__device__ __forceinline__ WorkItem decode_item(const ScanArgs &a) {
	WorkItem it;
	it.sub = blockIdx.y;
	it.w = (blockIdx.x * BLOCK + threadIdx.x) / G;
	it.valid = it.w < a.total_segs;
	it.qidx = it.seg_in_q = it.start = it.end = 0;
	it.is_self = false;
	if (it.valid) {
		it.qidx = a.seg2query[it.w];
		it.seg_in_q = it.w - a.qseg_start[it.qidx];
		uint32_t qlen = a.qlen[it.qidx];
		it.start = it.seg_in_q * a.seg;
		uint32_t e = it.start + a.seg;
		it.end = e < qlen ? e : qlen;
		it.is_self = a.self[it.sub] == (int64_t)it.qidx;
	}
	return it;
}

